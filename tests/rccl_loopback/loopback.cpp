// loopback.cpp — TEST INFRASTRUCTURE, never part of the product: a stand-in for the thirteen RCCL entry points
// libpfd_hip.so binds (nm -D libpfd_hip.so | grep ' U nccl'), interposed with LD_PRELOAD by tests/ and by
// `bench.py --gpus N` rehearsals so that SEVERAL ranks that share ONE GPU run the world > 1 branches of
// pyflwdir_amd/csrc/dist.hip (pfd_upstream_area_cell_dist, pfd_comm_exchange_rows, pfd_comm_allgather_host).
// Real RCCL refuses two ranks on one device; the test box has one device.
//
// What it keeps of the RCCL contract: every call is stream-ordered with respect to the caller's stream (the
// stand-in is stricter: it synchronises the stream, moves the bytes and returns when they are in place), ranks
// must issue collectives in the same order, send/recv inside ncclGroupStart/End complete together and cannot
// deadlock on each other.  What it does not model: asynchrony, xGMI, bandwidth.  Bytes travel through a
// file-backed shared mapping (hipMemcpy device -> mapping -> device): nothing here needs hipIpc, so it also runs
// where the dmabuf exporter is not available.
//
// Record / replay (measurement aid: what ONE rank of an N-rank job spends when its peers answer at once).  With
// PFD_LOOPBACK_RECORD=<dir> rank 0 of a normal N-rank run writes the result of every collective it takes part in, in call
// order, to <dir>/NNNNNN.bin.  With PFD_LOOPBACK_REPLAY=<dir> a SINGLE process creates a communicator of N ranks without
// peers: at ncclCommInitRank the recorded results are loaded into device memory, and every ncclAllGather / ncclAllReduce
// is served by ONE device-to-device copy enqueued on the caller's stream — no synchronisation, no host round trip, i.e.
// an ideal transport.  The caller must issue the collectives of the recorded run (same program, same arguments); the
// sequence restarts from PFD_LOOPBACK_REPLAY_LOOP (default 0) when it runs out, so a program may repeat its passes.
//
// Rendezvous: ncclGetUniqueId creates the backing file and writes its path into the 128-byte id; every rank maps it
// in ncclCommInitRank.  All waits are bounded (PFD_LOOPBACK_TIMEOUT_S, default 120 s): a rank that never arrives
// makes the others return ncclSystemError instead of hanging the box.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {

constexpr int MAXR = 8;
constexpr size_t COLL_CAP = 4u << 20;  // all-gather / all-reduce slot of a rank
constexpr size_t BOX_CAP = 1u << 20;   // mailbox of an ordered pair of ranks
constexpr char MAGIC[8] = {'p', 'f', 'd', 'l', 'b', '0', '1', 0};

struct Box {
  std::atomic<uint64_t> wr, rd;  // chunks published / consumed
  std::atomic<uint64_t> bytes;   // size of the published chunk
};
struct Shm {
  std::atomic<uint32_t> arrived, departed;
  std::atomic<uint32_t> bar_count, bar_gen;
  std::atomic<uint32_t> aborted;
  std::atomic<uint64_t> stats[8];  // [0] all-gathers, [1] all-reduces, [2] sends, [3] recvs, [4] payload bytes
  Box box[MAXR][MAXR];
  alignas(4096) char coll[MAXR][COLL_CAP];
  alignas(4096) char mail[MAXR][MAXR][BOX_CAP];
};

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
double timeout_s() {
  static double t = [] {
    const char *e = getenv("PFD_LOOPBACK_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
  }();
  return t;
}
struct Waiter {  // bounded spinning: yield first, sleep later
  double t0 = now_s();
  unsigned spins = 0;
  bool step(Shm *s) {
    if (s->aborted.load(std::memory_order_relaxed)) return false;
    if (++spins < 2000) {
      sched_yield();
    } else {
      timespec d{0, 50000};
      nanosleep(&d, nullptr);
      if ((spins & 1023u) == 0 && now_s() - t0 > timeout_s()) {
        s->aborted.store(1);
        return false;
      }
    }
    return true;
  }
};

struct P2P {
  bool send;
  char *dev;
  size_t bytes, off;
  int peer;
  uint64_t chunk;  // chunks of this op already moved
};

}  // namespace

struct ncclComm {
  Shm *shm = nullptr;
  int rank = 0, world = 1, device = 0;
  char path[120] = {0};
  // record (rank 0 of a real run) / replay (one process, no peers)
  char rec_dir[200] = {0};
  unsigned rec_next = 0;
  bool replay = false;
  std::vector<std::pair<void *, size_t>> replay_bufs;  // device copies of the recorded results, in call order
  size_t replay_next = 0, replay_loop = 0;
};

namespace {

thread_local int group_depth = 0;
thread_local std::vector<std::pair<ncclComm *, P2P>> group_ops;
thread_local std::vector<hipStream_t> group_streams;

bool barrier(ncclComm *c) {
  Shm *s = c->shm;
  const uint32_t gen = s->bar_gen.load(std::memory_order_acquire);
  if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
    s->bar_count.store(0, std::memory_order_relaxed);
    s->bar_gen.fetch_add(1, std::memory_order_release);
    return true;
  }
  Waiter w;
  while (s->bar_gen.load(std::memory_order_acquire) == gen)
    if (!w.step(s)) return false;
  return true;
}

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <class T>
void reduce_t(T *acc, const T *x, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; ++i) {
    switch (op) {
      case ncclSum: acc[i] = (T)(acc[i] + x[i]); break;
      case ncclProd: acc[i] = (T)(acc[i] * x[i]); break;
      case ncclMax: acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
      case ncclMin: acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
      default: break;
    }
  }
}
bool reduce_any(void *acc, const void *x, size_t n, ncclDataType_t t, ncclRedOp_t op) {
  switch (t) {
    case ncclInt8: reduce_t((int8_t *)acc, (const int8_t *)x, n, op); return true;
    case ncclUint8: reduce_t((uint8_t *)acc, (const uint8_t *)x, n, op); return true;
    case ncclInt32: reduce_t((int32_t *)acc, (const int32_t *)x, n, op); return true;
    case ncclUint32: reduce_t((uint32_t *)acc, (const uint32_t *)x, n, op); return true;
    case ncclInt64: reduce_t((int64_t *)acc, (const int64_t *)x, n, op); return true;
    case ncclUint64: reduce_t((uint64_t *)acc, (const uint64_t *)x, n, op); return true;
    case ncclFloat32: reduce_t((float *)acc, (const float *)x, n, op); return true;
    case ncclFloat64: reduce_t((double *)acc, (const double *)x, n, op); return true;
    default: return false;
  }
}

// one non-blocking step of a send / recv; returns true when something moved
bool p2p_step(ncclComm *c, P2P &o, bool *failed) {
  Shm *s = c->shm;
  if (o.off >= o.bytes && !(o.bytes == 0 && o.chunk == 0)) return false;
  const size_t len = o.bytes - o.off < BOX_CAP ? o.bytes - o.off : BOX_CAP;
  if (o.send) {
    Box &b = s->box[c->rank][o.peer];
    if (b.wr.load(std::memory_order_acquire) != b.rd.load(std::memory_order_acquire)) return false;  // not consumed yet
    if (len && hipMemcpy(s->mail[c->rank][o.peer], o.dev + o.off, len, hipMemcpyDeviceToHost) != hipSuccess) *failed = true;
    b.bytes.store(len, std::memory_order_relaxed);
    b.wr.fetch_add(1, std::memory_order_release);
  } else {
    Box &b = s->box[o.peer][c->rank];
    if (b.wr.load(std::memory_order_acquire) == b.rd.load(std::memory_order_acquire)) return false;  // nothing there
    if (b.bytes.load(std::memory_order_relaxed) != len) *failed = true;  // (the two sides disagree on the size)
    if (len && hipMemcpy(o.dev + o.off, s->mail[o.peer][c->rank], len, hipMemcpyHostToDevice) != hipSuccess) *failed = true;
    b.rd.fetch_add(1, std::memory_order_release);
  }
  o.off += len;
  ++o.chunk;
  return true;
}
bool p2p_done(const P2P &o) { return o.off >= o.bytes && o.chunk > 0; }

ncclResult_t run_p2p(std::vector<std::pair<ncclComm *, P2P>> &ops, std::vector<hipStream_t> &streams) {
  for (hipStream_t st : streams)
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (ops.empty()) return ncclSuccess;
  Waiter w;
  bool failed = false;
  for (;;) {
    bool all = true, moved = false;
    for (auto &e : ops) {
      if (p2p_done(e.second)) continue;
      moved |= p2p_step(e.first, e.second, &failed);
      all &= p2p_done(e.second);
    }
    if (failed) {
      ops[0].first->shm->aborted.store(1);
      return ncclInternalError;
    }
    if (all) return ncclSuccess;
    if (moved) {
      w = Waiter();
    } else if (!w.step(ops[0].first->shm)) {
      return ncclSystemError;
    }
  }
}

ncclResult_t enqueue_p2p(bool send, const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm *c, hipStream_t st) {
  if (c && c->replay) return ncclInvalidUsage;  // (replay serves the collectives of pfd_upstream_area_cell_dist only)
  if (!c || !c->shm || peer < 0 || peer >= c->world || peer == c->rank || type_size(t) == 0) return ncclInvalidArgument;
  P2P o{send, (char *)buf, count * type_size(t), 0, peer, 0};
  c->shm->stats[send ? 2 : 3].fetch_add(1);
  if (send) c->shm->stats[4].fetch_add(o.bytes);
  if (group_depth > 0) {
    group_ops.emplace_back(c, o);
    group_streams.push_back(st);
    return ncclSuccess;
  }
  std::vector<std::pair<ncclComm *, P2P>> one{{c, o}};
  std::vector<hipStream_t> sts{st};
  return run_p2p(one, sts);
}

void record_result(ncclComm *c, const void *dev, size_t bytes) {
  if (!c->rec_dir[0] || c->rank != 0) return;
  std::vector<char> host(bytes);
  if (bytes && hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return;
  char path[260];
  snprintf(path, sizeof(path), "%s/%06u.bin", c->rec_dir, c->rec_next++);
  if (FILE *f = fopen(path, "wb")) {
    fwrite(host.data(), 1, bytes, f);
    fclose(f);
  }
}
bool replay_load(ncclComm *c, const char *dir) {
  for (unsigned i = 0;; ++i) {
    char path[260];
    snprintf(path, sizeof(path), "%s/%06u.bin", dir, i);
    FILE *f = fopen(path, "rb");
    if (!f) break;
    fseek(f, 0, SEEK_END);
    const size_t bytes = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> host(bytes);
    const size_t got = fread(host.data(), 1, bytes, f);
    fclose(f);
    void *dev = nullptr;
    if (got != bytes || hipMalloc(&dev, bytes ? bytes : 4) != hipSuccess) return false;
    if (bytes && hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    c->replay_bufs.emplace_back(dev, bytes);
  }
  return !c->replay_bufs.empty();
}
ncclResult_t replay_serve(ncclComm *c, void *recvbuf, size_t bytes, hipStream_t st) {
  if (c->replay_next >= c->replay_bufs.size()) c->replay_next = c->replay_loop;
  if (c->replay_next >= c->replay_bufs.size()) return ncclInvalidUsage;
  const auto &b = c->replay_bufs[c->replay_next++];
  if (b.second != bytes) {
    fprintf(stderr, "[rccl loopback] replay: call %zu was recorded with %zu bytes, asked for %zu\n", c->replay_next - 1, b.second, bytes);
    return ncclInvalidUsage;
  }
  if (bytes && hipMemcpyAsync(recvbuf, b.first, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // namespace

extern "C" {

// lets a caller (tests, bench.py) see that the stand-in — not RCCL — is bound: ctypes.CDLL(None).pfd_rccl_loopback_active
int pfd_rccl_loopback_active(void) { return 1; }
// counters of the communicator's shared segment: [0] all-gathers, [1] all-reduces, [2] sends, [3] recvs, [4] bytes sent
int pfd_rccl_loopback_stats(ncclComm_t c, unsigned long long out[5]) {
  if (!c || !c->shm) return 1;
  for (int i = 0; i < 5; ++i) out[i] = c->shm->stats[i].load();
  return 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  const char *dir = getenv("PFD_LOOPBACK_DIR");
  if (!dir || !*dir) dir = "/tmp";
  char path[112];
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf(path, sizeof(path), "%s/pfd_rccl_loopback_%d_%lld%09ld", dir, (int)getpid(), (long long)t.tv_sec, t.tv_nsec);
  const int fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600);
  if (fd < 0) return ncclSystemError;
  const int rc = ftruncate(fd, (off_t)sizeof(Shm));  // (sparse: pages exist once touched)
  close(fd);
  if (rc != 0) {
    unlink(path);
    return ncclSystemError;
  }
  memset(id->internal, 0, sizeof(id->internal));
  memcpy(id->internal, MAGIC, sizeof(MAGIC));
  snprintf(id->internal + sizeof(MAGIC), sizeof(id->internal) - sizeof(MAGIC), "%s", path);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (memcmp(id.internal, MAGIC, sizeof(MAGIC)) != 0) return ncclInvalidArgument;  // (an id of the real library)
  ncclComm *c = new ncclComm();
  c->rank = rank, c->world = nranks;
  (void)hipGetDevice(&c->device);
  if (const char *dir = getenv("PFD_LOOPBACK_REPLAY")) {  // one process stands for rank `rank` of `nranks`: no peers, no mapping
    c->replay = true;
    if (const char *e = getenv("PFD_LOOPBACK_REPLAY_LOOP")) c->replay_loop = (size_t)atol(e);
    if (!replay_load(c, dir)) {
      delete c;
      return ncclSystemError;
    }
    unlink(id.internal + sizeof(MAGIC));  // (the id's backing file is not needed)
    *comm = c;
    return ncclSuccess;
  }
  if (const char *dir = getenv("PFD_LOOPBACK_RECORD")) snprintf(c->rec_dir, sizeof(c->rec_dir), "%s", dir);
  snprintf(c->path, sizeof(c->path), "%s", id.internal + sizeof(MAGIC));
  const int fd = open(c->path, O_RDWR);
  if (fd < 0) {
    delete c;
    return ncclSystemError;
  }
  void *m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->shm = (Shm *)m;
  c->shm->arrived.fetch_add(1, std::memory_order_acq_rel);
  Waiter w;
  while (c->shm->arrived.load(std::memory_order_acquire) < (uint32_t)nranks)
    if (!w.step(c->shm)) {
      munmap(m, sizeof(Shm));
      delete c;
      return ncclSystemError;
    }
  if (!barrier(c)) {  // (nobody unlinks before everybody has mapped)
    munmap(m, sizeof(Shm));
    delete c;
    return ncclSystemError;
  }
  if (rank == 0) unlink(c->path);
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->shm) munmap((void *)c->shm, sizeof(Shm));
  for (auto &b : c->replay_bufs) (void)hipFree(b.first);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) {
  if (!c || !n) return ncclInvalidArgument;
  *n = c->world;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) {
  if (!c || !r) return ncclInvalidArgument;
  *r = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int *d) {
  if (!c || !d) return ncclInvalidArgument;
  *d = c->device;
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (loopback stand-in)";
    case ncclUnhandledCudaError: return "unhandled HIP error (loopback stand-in)";
    case ncclSystemError: return "system error or peer timeout (loopback stand-in)";
    case ncclInternalError: return "internal error (loopback stand-in)";
    case ncclInvalidArgument: return "invalid argument (loopback stand-in)";
    case ncclInvalidUsage: return "invalid usage (loopback stand-in)";
    default: return "error (loopback stand-in)";
  }
}

ncclResult_t ncclGroupStart(void) {
  ++group_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd(void) {
  if (group_depth <= 0) return ncclInvalidUsage;
  if (--group_depth > 0) return ncclSuccess;
  const ncclResult_t r = run_p2p(group_ops, group_streams);
  group_ops.clear();
  group_streams.clear();
  return r;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  return enqueue_p2p(true, buf, count, t, peer, c, st);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  return enqueue_p2p(false, buf, count, t, peer, c, st);
}

ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
  if (c && c->replay) return replay_serve(c, recvbuf, count * type_size(t) * (size_t)c->world, st);
  if (!c || !c->shm || type_size(t) == 0 || group_depth > 0) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  Shm *s = c->shm;
  const size_t bytes = count * type_size(t);
  if (c->rank == 0) s->stats[0].fetch_add(1);
  s->stats[4].fetch_add(bytes);
  for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += COLL_CAP) {
    const size_t len = bytes - off < COLL_CAP ? bytes - off : COLL_CAP;
    bool ok = true;
    if (len) ok = hipMemcpy(s->coll[c->rank], (const char *)sendbuf + off, len, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) s->aborted.store(1);
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world && ok; ++r)
      if (len) ok = hipMemcpy((char *)recvbuf + (size_t)r * bytes + off, s->coll[r], len, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) s->aborted.store(1);
    if (!barrier(c)) return ncclSystemError;  // (the slots are free again)
    if (!ok) return ncclUnhandledCudaError;
    if (bytes == 0) break;
  }
  record_result(c, recvbuf, bytes * (size_t)c->world);
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t st) {
  if (c && c->replay) return replay_serve(c, recvbuf, count * type_size(t), st);
  if (!c || !c->shm || type_size(t) == 0 || group_depth > 0) return ncclInvalidArgument;
  const size_t bytes = count * type_size(t);
  if (bytes > COLL_CAP) return ncclInvalidArgument;  // (the library reduces a handful of counters)
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  Shm *s = c->shm;
  if (c->rank == 0) s->stats[1].fetch_add(1);
  bool ok = bytes == 0 || hipMemcpy(s->coll[c->rank], sendbuf, bytes, hipMemcpyDeviceToHost) == hipSuccess;
  if (!ok) s->aborted.store(1);
  if (!barrier(c)) return ncclSystemError;
  std::vector<char> acc(s->coll[0], s->coll[0] + bytes);
  for (int r = 1; r < c->world && ok; ++r) ok = reduce_any(acc.data(), s->coll[r], count, t, op);
  if (ok && bytes) ok = hipMemcpy(recvbuf, acc.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) s->aborted.store(1);
  if (!barrier(c)) return ncclSystemError;
  if (ok) record_result(c, recvbuf, bytes);
  return ok ? ncclSuccess : ncclUnhandledCudaError;
}

}  // extern "C"
