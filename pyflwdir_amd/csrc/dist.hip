// dist.hip — multi-GPU driver of the tiled accumulation: one row block (+1 halo row per inner
// edge) per GPU, one process per GPU, ONE small all-gather over RCCL/xGMI per pass.
//
// A D8 edge changes the row by at most one, so the only cross-GPU edges connect the last row
// of block k with the first row of block k+1.  Per pass:
//
//   phase A (local, all GPUs in parallel)   tile pass + exit-graph solve with zero outside flow.
//       Each halo cell (a weightless sink standing for the neighbour's boundary cell) ends up
//       with L = flow that leaves the block through it; each own boundary-row cell gets
//       sink = the halo cell where flow entering THERE would leave the block again.
//   exchange                                 all-gather of the 4*ncol-word boundary record
//       {L_top, L_bottom, sink_first_row, sink_last_row} of every block (<= 1.4 MB per GPU for
//       90000 columns): ncclAllGather, nothing else crosses xGMI.
//   interface solve (redundant on every GPU) the interface cells form a forest
//       F(h) = L(h) + sum of F over the interface cells whose flow leaves through h
//       (2*N*ncol nodes); same pointer doubling as everywhere else.
//   phase B (local, parallel)                F of the own boundary rows is pushed down the local
//       exit paths, then the final tile pass writes the block's result.
//
// The in-process variant (pfd_upstream_area_cell_blocks) runs the identical phases for several
// blocks that live on the GPUs of ONE process and "gathers" with device copies; it is what the
// single-GPU test box exercises, and it differs from the RCCL path only in the transport.
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "tiled.h"

struct pfd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  int *flag_dev = nullptr;  // two agreement words, allocated with the communicator (no allocation on the collective path)
  // neighbour exchange (pfd_comm_exchange_rows): receive buffer for the two halo rows, counters; grown on demand, which
  // every rank does in the same call (the row size is the same everywhere)
  void *xbuf = nullptr;
  size_t xcap = 0;
  long long *cnt_dev = nullptr;  // 8 words: [0..3] in, [4..7] out
  // boundary records of pfd_upstream_area_cell_dist, kept between passes: once they exist (every rank allocates them
  // in the same pass, behind a set-up agreement) a pass needs no set-up agreement — a rank whose local set-up fails
  // still owns what it needs to reach every collective
  u32 *rec_dev = nullptr, *allrec_dev = nullptr;
  size_t rec_words = 0;
};

#define NCCLCHK(expr)                                                                       \
  do {                                                                                      \
    ncclResult_t r_ = (expr);                                                               \
    if (r_ != ncclSuccess) {                                                                \
      pfd_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return PFD_ECOMM;                                                                     \
    }                                                                                       \
  } while (0)

extern "C" int pfd_comm_unique_id(void *id_out, size_t len) {
  if (!id_out || len < sizeof(ncclUniqueId)) {
    pfd_set_error("pfd_comm_unique_id: buffer must hold %zu bytes", sizeof(ncclUniqueId));
    return PFD_EINVAL;
  }
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return PFD_OK;
}

extern "C" int pfd_comm_create(const void *id, size_t len, int rank, int world, int device, pfd_comm **out) {
  if (!id || len < sizeof(ncclUniqueId) || !out || world < 1 || rank < 0 || rank >= world) {
    pfd_set_error("pfd_comm_create: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  pfd_comm *c = new pfd_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    pfd_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
    delete c;
    return PFD_ECOMM;
  }
  if (hipMalloc((void **)&c->flag_dev, 64) != hipSuccess) {
    pfd_set_error("pfd_comm_create: device allocation failed");
    (void)ncclCommDestroy(c->comm);
    delete c;
    return PFD_ENOMEM;
  }
  *out = c;
  return PFD_OK;
}

extern "C" int pfd_comm_destroy(pfd_comm *c) {
  if (c) {
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->flag_dev) (void)hipFree(c->flag_dev);
    if (c->xbuf) (void)hipFree(c->xbuf);
    if (c->cnt_dev) (void)hipFree(c->cnt_dev);
    if (c->rec_dev) (void)hipFree(c->rec_dev);
    if (c->allrec_dev) (void)hipFree(c->allrec_dev);
    delete c;
  }
  return PFD_OK;
}

extern "C" int pfd_comm_info(pfd_comm *c, int *nranks, int *rank, int *device) {
  if (!c || !c->comm) {
    pfd_set_error("pfd_comm_info: no communicator");
    return PFD_EINVAL;
  }
  int n = 0, r = 0, d = 0;
  NCCLCHK(ncclCommCount(c->comm, &n));
  NCCLCHK(ncclCommUserRank(c->comm, &r));
  NCCLCHK(ncclCommCuDevice(c->comm, &d));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// neighbour exchange of the iterated row-block collectives: boundary rows device to device
// ---------------------------------------------------------------------------------------------
template <class W>
__global__ void __launch_bounds__(256) k_seed_update(const W *__restrict__ recv, W *__restrict__ seed, size_t n,
                                                     long long *__restrict__ changed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool diff = false;
  if (i < n) {
    const W v = recv[i];
    diff = v != seed[i];  // (bit patterns: W is an unsigned integer type)
    if (diff) seed[i] = v;
  }
  if (__any((int)diff) && (threadIdx.x & 63u) == 0u) *changed = 1;  // (racing stores of the same value)
}
__global__ void k_cnt_pack(long long *c) {  // the local "changed" flag joins the counts that are summed
  c[3] = c[2];
}

static int comm_reserve(pfd_comm *c, size_t bytes) {
  if (!c->cnt_dev) HIPCHK(hipMalloc((void **)&c->cnt_dev, 8 * sizeof(long long)));
  if (bytes > c->xcap) {
    if (c->xbuf) (void)hipFree(c->xbuf);
    c->xbuf = nullptr, c->xcap = 0;
    HIPCHK(hipMalloc(&c->xbuf, bytes));
    c->xcap = bytes;
  }
  return PFD_OK;
}

extern "C" int pfd_comm_exchange_rows(pfd_comm *c, pfd_raster *h, const void *result_dev, int elem_bytes, void *seed_dev,
                                      int64_t counters[4]) {
  if (!c || !c->comm || !h || !result_dev || !seed_dev || !counters || (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16)) {
    pfd_set_error("pfd_comm_exchange_rows: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(h->device));
  const int rank = c->rank, world = c->world;
  // A rank that returned before the collectives would leave its neighbours waiting in RCCL for ever: whatever can fail
  // locally (a handle whose halo rows do not fit the rank, a HIP call) is remembered, the rank still takes part in the
  // send / recv group and in the all-reduce — with rows of zeros when its own are unusable — and its failure is added to
  // counters[1], which every caller sums as "ranks that failed"; the local error is returned afterwards.
  int lrc = PFD_OK;
  if (h->halo_top != (rank > 0) || h->halo_bot != (rank + 1 < world)) {
    pfd_set_error("pfd_comm_exchange_rows: rank %d of %d must hold %d top / %d bottom halo rows", rank, world, rank > 0,
                  rank + 1 < world);
    lrc = PFD_EINVAL;
  }
  const size_t rowb = (size_t)h->ncol * (size_t)elem_bytes;
  // (an allocation failure here does leave the other ranks in the collective: the buffers are a few hundred KB, kept with
  //  the communicator, so this can only happen in the very first exchange)
  PFDCHK(comm_reserve(c, 4 * rowb));
  hipStream_t st = h->stream;
  const char *res = (const char *)result_dev;
  char *rtop = (char *)c->xbuf, *rbot = (char *)c->xbuf + rowb, *zeros = (char *)c->xbuf + 2 * rowb;
  const char *first = res + (size_t)h->halo_top * rowb, *last = res + (size_t)(h->halo_top + h->own_rows - 1) * rowb;
  if (lrc != PFD_OK) {  // (rows of zeros stand in for rows this rank cannot name)
    if (hipMemsetAsync(zeros, 0, 2 * rowb, st) != hipSuccess) (void)hipGetLastError();
    first = zeros, last = zeros + rowb;
  }
  long long in[4] = {(long long)counters[0], (long long)counters[1] + (lrc != PFD_OK ? 1 : 0), 0, 0};
  if (hipMemcpyAsync(c->cnt_dev, in, sizeof(in), hipMemcpyHostToDevice, st) != hipSuccess && lrc == PFD_OK) {
    pfd_set_error("pfd_comm_exchange_rows: upload of the counters failed");
    lrc = PFD_EHIP;
  }
  if (world > 1) {
    ncclResult_t r = ncclGroupStart();
    if (rank > 0) {
      if (r == ncclSuccess) r = ncclSend(first, rowb, ncclUint8, rank - 1, c->comm, st);
      if (r == ncclSuccess) r = ncclRecv(rtop, rowb, ncclUint8, rank - 1, c->comm, st);
    }
    if (rank + 1 < world) {
      if (r == ncclSuccess) r = ncclSend(last, rowb, ncclUint8, rank + 1, c->comm, st);
      if (r == ncclSuccess) r = ncclRecv(rbot, rowb, ncclUint8, rank + 1, c->comm, st);
    }
    const ncclResult_t r2 = ncclGroupEnd();
    if ((r != ncclSuccess || r2 != ncclSuccess) && lrc == PFD_OK) {
      pfd_set_error("pfd_comm_exchange_rows: ncclSend/ncclRecv failed: %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
      lrc = PFD_ECOMM;
    }
    auto update = [&](const char *recv, char *seed) {
      if (elem_bytes == 1)
        k_seed_update<u8><<<cdiv_u32(rowb, 256), 256, 0, st>>>((const u8 *)recv, (u8 *)seed, rowb, c->cnt_dev + 2);
      else
        k_seed_update<u32><<<cdiv_u32(rowb / 4, 256), 256, 0, st>>>((const u32 *)recv, (u32 *)seed, rowb / 4, c->cnt_dev + 2);
    };
    if (lrc == PFD_OK) {
      if (rank > 0) update(rtop, (char *)seed_dev);
      if (rank + 1 < world) update(rbot, (char *)seed_dev + rowb);
      if (hipGetLastError() != hipSuccess) {
        pfd_set_error("pfd_comm_exchange_rows: the seed update failed");
        lrc = PFD_EHIP;
      }
    }
  }
  k_cnt_pack<<<1, 1, 0, st>>>(c->cnt_dev);
  const ncclResult_t r3 = ncclAllReduce(c->cnt_dev, c->cnt_dev + 4, 4, ncclInt64, ncclSum, c->comm, st);
  if (r3 != ncclSuccess && lrc == PFD_OK) {
    pfd_set_error("pfd_comm_exchange_rows: ncclAllReduce failed: %s", ncclGetErrorString(r3));
    lrc = PFD_ECOMM;
  }
  long long out[4] = {0, 0, 0, 0}, mine = 0;
  if ((hipMemcpyAsync(out, c->cnt_dev + 4, sizeof(out), hipMemcpyDeviceToHost, st) != hipSuccess ||
       hipMemcpyAsync(&mine, c->cnt_dev + 2, sizeof(mine), hipMemcpyDeviceToHost, st) != hipSuccess ||
       hipStreamSynchronize(st) != hipSuccess) &&
      lrc == PFD_OK) {
    pfd_set_error("pfd_comm_exchange_rows: download of the counters failed");
    lrc = PFD_EHIP;
  }
  counters[0] = out[0], counters[1] = out[1], counters[2] = mine, counters[3] = out[3];
  return lrc;
}

extern "C" int pfd_comm_allgather_host(pfd_comm *c, pfd_raster *h, const void *in_host, size_t nbytes, void *out_host) {
  if (!c || !c->comm || !h || !in_host || !out_host || nbytes == 0) {
    pfd_set_error("pfd_comm_allgather_host: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(h->device));
  DevBuf in, all;
  PFDCHK(in.alloc(nbytes));
  PFDCHK(all.alloc(nbytes * (size_t)c->world));
  HIPCHK(hipMemcpyAsync(in.p, in_host, nbytes, hipMemcpyHostToDevice, h->stream));
  NCCLCHK(ncclAllGather(in.p, all.p, nbytes, ncclUint8, c->comm, h->stream));
  HIPCHK(hipMemcpyAsync(out_host, all.p, nbytes * (size_t)c->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// interface graph.  Node id = (block*2 + side)*ncol + col stands for the halo cell `col` on
// `side` (0 top, 1 bottom) of `block`; rec = the gathered records, 4*ncol words per block.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_iface_build(const u32 *__restrict__ rec, u32 nblocks, u32 ncol,
                                                     u32 *__restrict__ T, u32 *__restrict__ J, u64 *ctrl) {
  const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nblocks * 2 * ncol) return;
  const u32 col = id % ncol, bs = id / ncol, side = bs & 1, blk = bs >> 1;
  T[id] = rec[(size_t)blk * 4 * ncol + side * ncol + col];  // L: flow leaving `blk` through this cell
  u32 j = id | XDONE;
  // the cell belongs to the neighbouring block, where it is a boundary-row cell on the other side
  const int owner = (int)blk + (side ? 1 : -1);
  if (owner >= 0 && owner < (int)nblocks) {
    const u32 s = rec[(size_t)owner * 4 * ncol + (2 + (1 - side)) * ncol + col];  // where that flow leaves `owner`
    if (s != NONE32) j = ((u32)owner * 2 + ((s & ENC_SIDE1) ? 1u : 0u)) * ncol + (s & ENC_COL);
  }
  J[id] = j;
}
// flow entering the own boundary rows: first row <- bottom halo of block-1, last row <- top halo of block+1
__global__ void __launch_bounds__(256) k_iface_inflow(const u32 *__restrict__ F, u32 nblocks, u32 ncol, u32 blk,
                                                      u32 *__restrict__ brow_inflow) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  const u32 side = t / ncol, col = t % ncol;
  const int src = (int)blk + (side ? 1 : -1);
  u32 v = 0;
  if (src >= 0 && src < (int)nblocks) v = F[((u32)src * 2 + (1 - side)) * ncol + col];
  brow_inflow[t] = v;
}
__global__ void k_pack_record(const u32 *__restrict__ haloL, const u32 *__restrict__ brow_sink, u32 ncol,
                              u32 *__restrict__ rec) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  rec[t] = haloL[t];
  rec[2 * ncol + t] = brow_sink[t];
}

// One doubling round of the interface forest (see k_coarse_round, tiled.hip, for the rotation of the three T buffers).
// The forest is thin — a path meets a handful of interface cells, a target collects a few pushes — so the pushes go
// to memory directly; the workgroup-wide combine of the level-4 rounds costs more here than it saves.
__global__ void __launch_bounds__(256) k_iface_round(const u32 *__restrict__ Told, u32 *__restrict__ Tnew, u32 *__restrict__ Tzero,
                                                     const u32 *__restrict__ Jold, u32 *__restrict__ Jnew, u32 n) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) {
    const u32 j = Jold[e], t = Told[e];
    Tzero[e] = 0;
    if (t) atomicAdd(&Tnew[e], t);
    u32 q = j;
    if (!(j & XDONE)) {
      q = Jold[j];
      if (t) atomicAdd(&Tnew[j], t);
    }
    Jnew[e] = q;
  }
}
// "is any pointer still unsaturated?" — asked once per batch of rounds (a flag raised by every wave of every round
// makes thousands of waves touch ONE address: ~27 us per launch, see k_check_saturated in tiled.hip)
__global__ void __launch_bounds__(256) k_iface_check(const u32 *__restrict__ J, u32 n, u64 *ctrl) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n && !(J[e] & XDONE)) ctrl[T_XACTIVE] = 1;  // (rare: only while the batch was too short)
}

// The interface forest by CHASING (the default): a path through the blocks meets a handful of interface cells, and only
// the cells that carry flow (L > 0) start one — each walks its path (the successor of a node is one lookup in the
// gathered records, no pointer array) and adds its L to the nodes this rank needs: the halo cells next to its own two
// boundary rows.  One launch instead of build + fill + log2(2 N) doubling rounds + check + host look + inflow
// (0.27 ms + two host round trips for 8 blocks of 90000 columns).  A path longer than `maxhops` (a river meandering
// along a block edge; a cycle through several blocks) raises MISS_IFACE: the pass is redone with the doubling rounds.
__device__ __forceinline__ u32 iface_next(const u32 *__restrict__ rec, u32 nblocks, u32 ncol, u32 node) {
  const u32 col = node % ncol, bs = node / ncol, side = bs & 1u;
  const int owner = (int)(bs >> 1) + (side ? 1 : -1);  // the block this halo cell belongs to
  if (owner < 0 || owner >= (int)nblocks) return NONE32;
  const u32 s = rec[(size_t)owner * 4 * ncol + (2 + (1 - side)) * ncol + col];  // where flow entering there leaves `owner`
  if (s == NONE32) return NONE32;
  return ((u32)owner * 2 + ((s & ENC_SIDE1) ? 1u : 0u)) * ncol + (s & ENC_COL);
}
// one atomic per DISTINCT target among the lanes of a wave: neighbouring boundary cells drain into the same river, so
// after a hop or two most lanes of a wave walk the same path, and the river's cell would take one same-address atomic
// per source (they serialise at ~12 ns: 0.19 ms on the block at the bottom of the benchmark raster).  All 64 lanes call.
__device__ __forceinline__ void wave_combined_add(u32 *__restrict__ base, u32 idx, u32 v, bool active) {
  u64 todo = __ballot(active);
  const u32 lane = threadIdx.x & 63u;
  while (todo) {  // (wave-uniform)
    const int leader = __ffsll((long long)todo) - 1;
    const u32 lidx = __shfl(idx, leader);
    const bool mine = active && idx == lidx;
    u32 x = mine ? v : 0u;
#pragma unroll
    for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off);
    if ((int)lane == leader) atomicAdd(&base[lidx], x);
    todo &= ~__ballot(mine);
  }
}
__global__ void __launch_bounds__(256) k_iface_chase(const u32 *__restrict__ rec, u32 nblocks, u32 ncol, u32 blk, u32 maxhops,
                                                     u32 *__restrict__ brow_inflow, u64 *__restrict__ ctrl) {
  const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inside = id < nblocks * 2 * ncol;
  const u32 bs0 = inside ? id / ncol : 0u;
  // flow leaving the block through this cell
  const u32 L = inside ? rec[(size_t)(bs0 >> 1) * 4 * ncol + (bs0 & 1u) * ncol + id % ncol] : 0u;
  // F of the bottom halo of block blk-1 enters the first own row, F of the top halo of block blk+1 the last
  const u32 want0 = blk ? (blk - 1) * 2 + 1 : NONE32, want1 = blk + 1 < nblocks ? (blk + 1) * 2 : NONE32;
  u32 node = id;
  bool live = L != 0u;  // (lanes stay in the loop until the whole wave is done: the combine needs all of them)
  for (u32 hop = 0; hop < maxhops && __ballot(live); ++hop) {
    const u32 bs = live ? node / ncol : NONE32 - 1u;
    const bool hit = live && (bs == want0 || bs == want1);
    wave_combined_add(brow_inflow, (bs == want1 ? ncol : 0u) + node % ncol, L, hit);
    if (live) {
      node = iface_next(rec, nblocks, ncol, node);
      live = node != NONE32;
    }
  }
  if (live) atomicOr((unsigned long long *)&ctrl[T_MISS], (unsigned long long)MISS_IFACE);
}

// the same forest by doubling rounds (after a chase ran out of hops); synchronises to see whether the rounds sufficed
static int interface_solve_doubling(TiledRun &run, const u32 *allrec_dev, u32 nblocks, u32 blk) {
  pfd_raster *h = run.h;
  const u32 ncol = (u32)h->ncol;
  const u32 nn = nblocks * 2 * ncol;
  DevBuf &buf = run.iface_buf;  // (lives as long as the run: no synchronisation needed before returning)
  PFDCHK(buf.alloc(5 * (size_t)nn * sizeof(u32)));
  u32 *T[3] = {buf.as<u32>(), buf.as<u32>() + nn, buf.as<u32>() + 2 * (size_t)nn};
  u32 *J[2] = {buf.as<u32>() + 3 * (size_t)nn, buf.as<u32>() + 4 * (size_t)nn};
  u32 *Tc = T[0], *Jc = J[0];
  pfd_seg_begin(h, "interface_solve");
  k_iface_build<<<cdiv_u32(nn, 256), 256, 0, h->stream>>>(allrec_dev, nblocks, ncol, Tc, Jc, h->ctrl);
  KCHK();
  bool done = false;
  i64 launches = 1;
  // (a path crosses at most 2 * nblocks interface cells when it runs straight through the blocks: that many hops
  //  saturate in log2 rounds, one more shows that nothing moved — sized so that the first host look is the last)
  int batch = 1, rounds = 0;
  for (u32 span = 1; span < 2u * nblocks; span <<= 1) ++batch;
  HIPCHK(hipMemsetAsync(T[1], 0, (size_t)nn * sizeof(u32), h->stream));
  while (rounds < 64 && !done) {
    for (int b = 0; b < batch; ++b) {
      ++rounds;
      k_iface_round<<<cdiv_u32(nn, 256), 256, 0, h->stream>>>(T[0], T[1], T[2], J[0], J[1], nn);
      ++launches;
      u32 *t0 = T[0];
      T[0] = T[1], T[1] = T[2], T[2] = t0;
      std::swap(J[0], J[1]);
    }
    HIPCHK(hipMemsetAsync(h->ctrl + T_XACTIVE, 0, sizeof(u64), h->stream));
    k_iface_check<<<cdiv_u32(nn, 256), 256, 0, h->stream>>>(J[0], nn, h->ctrl);
    ++launches;
    KCHK();
    u64 last = 0;
    HIPCHK(hipMemcpyAsync(&last, h->ctrl + T_XACTIVE, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    done = last == 0;  // every pointer is saturated
    batch = 2;
  }
  k_iface_inflow<<<cdiv_u32(2 * ncol, 256), 256, 0, h->stream>>>(T[0], nblocks, ncol, blk, run.brow_inflow);
  KCHK();
  pfd_seg_end(h, launches + 1);
  if (!done) run.coarse_done = false;  // a cycle through several blocks
  return PFD_OK;
}

// flow entering the own boundary rows from the gathered records (on the handle's device / stream), left in
// run.brow_inflow (zero since the pass began: k_pass_clear).  No host round trip unless the doubling form is asked for.
static int interface_solve(TiledRun &run, const u32 *allrec_dev, u32 nblocks, u32 blk) {
  if (run.iface_doubling) return interface_solve_doubling(run, allrec_dev, nblocks, blk);
  pfd_raster *h = run.h;
  const u32 ncol = (u32)h->ncol;
  u32 maxhops = 4u * nblocks + 32u;
  if (const char *e = pfd_knob("PFD_TEST_IFACE_HOPS")) maxhops = (u32)atoi(e);  // (tests: force the doubling form)
  pfd_seg_begin(h, "interface_solve");
  k_iface_chase<<<cdiv_u32(nblocks * 2 * ncol, 256), 256, 0, h->stream>>>(allrec_dev, nblocks, ncol, blk, maxhops, run.brow_inflow,
                                                                         h->ctrl);
  KCHK();
  pfd_seg_end(h, 1);
  return PFD_OK;
}
// after a pass's final synchronisation: did the chase run out of hops?  (then the pass is redone with doubling rounds)
static inline bool iface_missed(const u64 *c0) { return (c0[T_MISS] & MISS_IFACE) != 0; }

// interface solve + phase B of a block whose phase A has run and whose records everybody holds.  When the chase ran
// out of hops the whole pass of THIS block is redone with the doubling rounds (phase A is deterministic: the records
// the other blocks hold stay valid).
static int finish_block(TiledRun &run, const u32 *allrec_dev, u32 nblocks, u32 blk, int *complete) {
  if (nblocks > 1) PFDCHK(interface_solve(run, allrec_dev, nblocks, blk));
  PFDCHK(run.phase_b(complete));
  if (nblocks > 1 && (run.last_miss & MISS_IFACE) && !run.iface_doubling) {
    run.iface_doubling = true;
    PFDCHK(run.phase_a_checked());
    PFDCHK(interface_solve(run, allrec_dev, nblocks, blk));
    PFDCHK(run.phase_b(complete));
  }
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// in-process: all blocks of the raster are handles of THIS process (any devices)
// ---------------------------------------------------------------------------------------------
extern "C" int pfd_upstream_area_cell_blocks(pfd_raster **hs, int nblocks, int32_t **outs, int memspace) {
  if (!hs || !outs || nblocks < 1) {
    pfd_set_error("pfd_upstream_area_cell_blocks: bad arguments");
    return PFD_EINVAL;
  }
  const i64 ncol = hs[0]->ncol;
  for (int b = 0; b < nblocks; ++b) {
    if (!hs[b] || !outs[b] || hs[b]->ncol != ncol || hs[b]->halo_top != (b > 0) ||
        hs[b]->halo_bot != (b + 1 < nblocks)) {
      pfd_set_error("pfd_upstream_area_cell_blocks: block %d must have %d top / %d bottom halo rows and %lld columns",
                    b, b > 0, b + 1 < nblocks, (long long)ncol);
      return PFD_EINVAL;
    }
  }
  std::vector<TiledRun> runs(nblocks);
  std::vector<OutArg> o(nblocks);
  const size_t recw = 4 * (size_t)ncol;
  std::vector<u32> allrec_host((size_t)nblocks * recw);
  // phase A of every block is issued before the first synchronisation: blocks on different GPUs (and, as
  // far as the hardware allows, on different streams of one GPU) run concurrently
  for (int b = 0; b < nblocks; ++b) {
    pfd_raster *h = hs[b];
    PFDCHK(pfd_check_handle_lazy(h));
    pfd_seg_clear(h);
    PFDCHK(o[b].bind(outs[b], (size_t)h->own_rows * ncol * sizeof(i32), memspace));
    PFDCHK(runs[b].init(h, (i32 *)o[b].dev));
    if (!runs[b].supported) {
      pfd_set_error("pfd_upstream_area_cell_blocks: block %d is too large for the tiled engine", b);
      return PFD_EUNSUPPORTED;
    }
    PFDCHK(runs[b].phase_a());
  }
  std::vector<DevBuf> recs(nblocks);
  for (int b = 0; b < nblocks; ++b) {
    pfd_raster *h = hs[b];
    PFDCHK(pfd_check_handle_lazy(h));
    PFDCHK(runs[b].phase_a_check());
    PFDCHK(recs[b].alloc(recw * sizeof(u32)));
    k_pack_record<<<cdiv_u32(2 * (u32)ncol, 256), 256, 0, h->stream>>>(runs[b].haloL, runs[b].brow_sink, (u32)ncol,
                                                                     recs[b].as<u32>());
    KCHK();
    HIPCHK(hipMemcpyAsync(allrec_host.data() + (size_t)b * recw, recs[b].p, recw * sizeof(u32), hipMemcpyDeviceToHost,
                          h->stream));
  }
  for (int b = 0; b < nblocks; ++b) {
    PFDCHK(pfd_check_handle_lazy(hs[b]));
    HIPCHK(hipStreamSynchronize(hs[b]->stream));
  }
  int all_complete = 1;
  for (int b = 0; b < nblocks; ++b) {  // "all-gather" = every block gets all records; then phase B
    pfd_raster *h = hs[b];
    PFDCHK(pfd_check_handle_lazy(h));
    DevBuf allrec;
    if (nblocks > 1) {
      PFDCHK(allrec.alloc(allrec_host.size() * sizeof(u32)));
      HIPCHK(hipMemcpyAsync(allrec.p, allrec_host.data(), allrec_host.size() * sizeof(u32), hipMemcpyHostToDevice,
                            h->stream));
    }
    int complete = 0;
    PFDCHK(finish_block(runs[b], allrec.as<u32>(), (u32)nblocks, (u32)b, &complete));
    all_complete &= complete;
    PFDCHK(o[b].finish(h->stream));
  }
  if (!all_complete) {
    pfd_set_error("the raster holds cells that never reach a pit (cycles); the multi-block path requires a "
                  "valid flow direction raster (FlwdirRaster.isvalid)");
    return PFD_EUNSUPPORTED;
  }
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// split-phase form: the caller moves the boundary records itself (any transport — MPI, gloo,
// shared memory).  begin() runs phase A and returns this block's record (4*ncol words, host);
// finish() takes the records of ALL blocks (nblocks*4*ncol words, host, block order) and runs
// the interface solve + phase B.  Same kernels as the RCCL path.
// ---------------------------------------------------------------------------------------------
struct DistPending {
  TiledRun run;
  OutArg out;
};

void pfd_free_pending(pfd_raster *h) {
  delete (DistPending *)h->pending;
  h->pending = nullptr;
}

extern "C" int pfd_upstream_area_cell_begin(pfd_raster *h, int32_t *out, int memspace, uint32_t *record_host) {
  PFDCHK(pfd_check_handle_lazy(h));
  if (!out || !record_host) {
    pfd_set_error("pfd_upstream_area_cell_begin: bad arguments");
    return PFD_EINVAL;
  }
  delete (DistPending *)h->pending;
  h->pending = nullptr;
  DistPending *p = new DistPending();
  const u32 ncol = (u32)h->ncol;
  const size_t recw = 4 * (size_t)ncol;
  pfd_seg_clear(h);
  int rc = p->out.bind(out, (size_t)h->own_rows * ncol * sizeof(i32), memspace);
  if (rc == PFD_OK) rc = p->run.init(h, (i32 *)p->out.dev);
  if (rc == PFD_OK && !p->run.supported) {
    pfd_set_error("pfd_upstream_area_cell_begin: the block is too large for the tiled engine");
    rc = PFD_EUNSUPPORTED;
  }
  DevBuf rec;
  if (rc == PFD_OK) rc = rec.alloc(recw * sizeof(u32));
  // (the record goes through pinned staging: a pageable copy into a host array HIP has not seen costs 10 - 50 ms)
  size_t pcap = 0;
  void *pin = rc == PFD_OK ? pfd_pinned_take(recw * sizeof(u32), &pcap) : nullptr;
  void *land = pin ? pin : (void *)record_host;
  // phase A, the record and its control words leave in ONE synchronisation; phase A is redone when it fell short
  // (flat level 3 after an overflow of a hypertile's id range, more level-4 rounds)
  for (int tries = 0; rc == PFD_OK; ++tries) {
    rc = p->run.phase_a();
    if (rc != PFD_OK) break;
    k_pack_record<<<cdiv_u32(2 * ncol, 256), 256, 0, h->stream>>>(p->run.haloL, p->run.brow_sink, ncol, rec.as<u32>());
    u64 c8[8];
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(land, rec.p, recw * sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(c8, h->ctrl + 8, sizeof(c8), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
      pfd_set_error("pfd_upstream_area_cell_begin: phase A or the download of the boundary record failed");
      rc = PFD_EHIP;
      break;
    }
    if (!p->run.phase_a_needs_redo(c8, tries)) break;
  }
  if (rc != PFD_OK) {
    (void)hipStreamSynchronize(h->stream);
    pfd_pinned_give(pin, pcap);
    delete p;
    return rc;
  }
  if (pin) memcpy(record_host, pin, recw * sizeof(u32));
  pfd_pinned_give(pin, pcap);
  h->pending = p;
  return PFD_OK;
}

extern "C" int pfd_upstream_area_cell_finish(pfd_raster *h, const uint32_t *all_records_host, int nblocks, int block,
                                             int *complete) {
  PFDCHK(pfd_check_handle_lazy(h));
  DistPending *p = (DistPending *)h->pending;
  if (!p || !all_records_host || nblocks < 1 || block < 0 || block >= nblocks || !complete) {
    pfd_set_error("pfd_upstream_area_cell_finish: no pass in flight on this handle, or bad arguments");
    return PFD_EINVAL;
  }
  int rc = PFD_OK;
  const size_t recw = 4 * (size_t)h->ncol;
  DevBuf allrec;
  size_t pcap = 0;
  void *pin = nullptr;
  if (nblocks > 1) {
    const size_t nbytes = (size_t)nblocks * recw * sizeof(u32);
    rc = allrec.alloc(nbytes);
    const void *src = all_records_host;
    if (rc == PFD_OK && (pin = pfd_pinned_take(nbytes, &pcap)) != nullptr) {  // (pinned staging: see _begin)
      memcpy(pin, all_records_host, nbytes);
      src = pin;
    }
    if (rc == PFD_OK && hipMemcpyAsync(allrec.p, src, nbytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
      pfd_set_error("pfd_upstream_area_cell_finish: upload of the boundary records failed");
      rc = PFD_EHIP;
    }
  }
  if (rc == PFD_OK) rc = finish_block(p->run, allrec.as<u32>(), (u32)nblocks, (u32)block, complete);
  if (rc != PFD_OK) (void)hipStreamSynchronize(h->stream);  // (nothing may still read the staging when it is handed back)
  pfd_pinned_give(pin, pcap);
  if (rc == PFD_OK) rc = p->out.finish(h->stream);
  delete p;
  h->pending = nullptr;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// one process per GPU: collective call, every rank passes its own block
// ---------------------------------------------------------------------------------------------
extern "C" int pfd_upstream_area_cell_dist(pfd_raster *h, pfd_comm *comm, int32_t *out, int memspace) {
  PFDCHK(pfd_check_handle_lazy(h));
  if (!comm || !out) {
    pfd_set_error("pfd_upstream_area_cell_dist: bad arguments");
    return PFD_EINVAL;
  }
  const int rank = comm->rank, world = comm->world;
  if (h->halo_top != (rank > 0) || h->halo_bot != (rank + 1 < world)) {
    pfd_set_error("pfd_upstream_area_cell_dist: rank %d of %d must hold %d top / %d bottom halo rows", rank, world,
                  rank > 0, rank + 1 < world);
    return PFD_EINVAL;
  }
  const u32 ncol = (u32)h->ncol;
  const size_t recw = 4 * (size_t)ncol;
  pfd_seg_clear(h);
  // Every rank reaches every collective whatever happens locally (a rank that returned early would leave the
  // others waiting in RCCL forever).  The buffers the collectives need belong to the communicator and are
  // allocated once, behind an agreement; after that a pass costs ONE all-gather and ONE final agreement — a local
  // failure, set-up included, sends a zero record and is carried to the final agreement.
  OutArg o;
  TiledRun run;
  int rc = o.bind(out, (size_t)h->own_rows * ncol * sizeof(i32), memspace);
  if (rc == PFD_OK) rc = run.init(h, (i32 *)o.dev);
  if (rc == PFD_OK && !run.supported) {
    pfd_set_error("pfd_upstream_area_cell_dist: the block is too large for the tiled engine");
    rc = PFD_EUNSUPPORTED;
  }
  int *flag = comm->flag_dev;
  if (world > 1 && comm->rec_words < recw) {
    // first pass with this raster width: the communicator's record buffers are allocated behind an agreement (if an
    // allocation failed anywhere, all ranks return before any data collective starts)
    int arc = PFD_OK;
    if (comm->rec_dev) (void)hipFree(comm->rec_dev);
    if (comm->allrec_dev) (void)hipFree(comm->allrec_dev);
    comm->rec_dev = comm->allrec_dev = nullptr;
    comm->rec_words = 0;
    if (hipMalloc((void **)&comm->rec_dev, recw * sizeof(u32)) != hipSuccess ||
        hipMalloc((void **)&comm->allrec_dev, (size_t)world * recw * sizeof(u32)) != hipSuccess) {
      (void)hipGetLastError();
      pfd_set_error("pfd_upstream_area_cell_dist: the record buffers could not be allocated");
      arc = PFD_ENOMEM;
    }
    int ready = arc == PFD_OK ? 1 : 0, ready_all = 0;
    bool comm_ok = hipMemcpyAsync(flag, &ready, sizeof(int), hipMemcpyHostToDevice, h->stream) == hipSuccess;
    comm_ok = ncclAllReduce(flag, flag + 8, 1, ncclInt32, ncclMin, comm->comm, h->stream) == ncclSuccess && comm_ok;
    comm_ok = hipMemcpyAsync(&ready_all, flag + 8, sizeof(int), hipMemcpyDeviceToHost, h->stream) == hipSuccess && comm_ok;
    comm_ok = hipStreamSynchronize(h->stream) == hipSuccess && comm_ok;
    if (arc != PFD_OK) return arc;  // (every rank leaves here: ready_all is 0 everywhere)
    if (!comm_ok) {
      pfd_set_error("pfd_upstream_area_cell_dist: the set-up agreement (ncclAllReduce) failed");
      return PFD_ECOMM;
    }
    if (!ready_all) {
      pfd_set_error("pfd_upstream_area_cell_dist: another rank could not allocate its record buffers");
      return PFD_ENOMEM;
    }
    comm->rec_words = recw;
  } else if (world == 1 && rc != PFD_OK) {
    return rc;
  }
  // From here on every rank reaches every collective whatever happens locally: a local failure (set-up included)
  // sends a zero record and travels with the final agreement.
  //
  // One pass = phase A, all-gather, interface solve, phase B, agreement — issued on the stream WITHOUT a host round
  // trip in between (round 5; before: one after phase A, one inside the interface solve, one after phase B, one for
  // the agreement).  What a host look used to decide is decided on the device: a stage that fell short (a hypertile
  // overflowed its id range, level 4 or the interface chase ran out of their budgets) raises a sticky bit, the pass
  // runs on with whatever it has, and its verdict (2 fine / 1 redo / 0 failed) goes through the agreement all-reduce
  // (min): on "redo" EVERY rank repeats the pass — the collectives stay aligned — with the remedy applied where the
  // miss happened.  Misses are rare (none on the benchmark rasters); the price of one is a second pass.
  u32 *rec = comm->rec_dev, *allrec = comm->allrec_dev;
  int ok_all = 0, complete = 0;
  for (int tries = 0;; ++tries) {
    if (rc == PFD_OK) rc = run.phase_a();
    if (world > 1) {
      if (rc == PFD_OK) rc = run.stage_verdict_a();
      pfd_seg_begin(h, "allgather");
      if (rc == PFD_OK) {
        k_pack_record<<<cdiv_u32(2 * ncol, 256), 256, 0, h->stream>>>(run.haloL, run.brow_sink, ncol, rec);
      } else {
        (void)hipMemsetAsync(rec, 0, recw * sizeof(u32), h->stream);
      }
      const ncclResult_t r1 = ncclAllGather(rec, allrec, recw, ncclUint32, comm->comm, h->stream);
      pfd_seg_end(h, 2);
      if (r1 != ncclSuccess && rc == PFD_OK) {
        pfd_set_error("ncclAllGather failed: %s", ncclGetErrorString(r1));
        rc = PFD_ECOMM;
      }
      if (rc == PFD_OK) rc = interface_solve(run, allrec, (u32)world, (u32)rank);
    }
    if (rc == PFD_OK) rc = run.phase_b_issue();
    u64 c0[48] = {0};
    int verdict = 0;
    const char *what = nullptr;
    if (world > 1) {
      // every rank must agree: a cycle anywhere invalidates downstream blocks as well, a redo anywhere is a redo everywhere
      if (rc == PFD_OK) rc = run.block_verdict(flag, true);
      if (rc != PFD_OK && hipMemsetAsync(flag, 0, sizeof(int), h->stream) != hipSuccess) what = "clearing the agreement flag";
      const ncclResult_t r2 = ncclAllReduce(flag, flag + 8, 1, ncclInt32, ncclMin, comm->comm, h->stream);
      if (r2 != ncclSuccess) what = "ncclAllReduce";
      if (hipMemcpyAsync(&verdict, flag + 8, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess) what = "download of the agreement flag";
    }
    if (hipMemcpyAsync(c0, h->ctrl, sizeof(c0), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)  // the pass's ONE synchronisation
      what = "download of the control words";
    if (what && rc == PFD_OK) {
      pfd_set_error("pfd_upstream_area_cell_dist: %s failed", what);
      rc = PFD_ECOMM;
    }
    if (rc == PFD_OK) rc = run.phase_b_collect(c0, &complete);  // (adopts a deferred handle's counts; bad D8 codes surface here)
    bool redo;
    if (world > 1) {
      redo = !what && verdict == 1 && tries < 3;  // (the same on every rank: the all-reduce's result and the try count)
      ok_all = !what && verdict == 2;
    } else {
      redo = rc == PFD_OK && (run.overflowed || run.short_of_rounds) && tries < 3;
      ok_all = rc == PFD_OK && complete;
    }
    if (!redo) break;
    if (rc == PFD_OK) {  // the remedy, where the miss happened (a rank without a miss repeats its pass unchanged)
      if (run.overflowed || (c0[T_MISS] & MISS_OVERFLOW)) run.force_flat = true;
      else if (run.short_of_rounds || (c0[T_MISS] & MISS_ROUNDS4)) run.extra_rounds += 8;
      if (c0[T_MISS] & MISS_IFACE) run.iface_doubling = true;
    }
  }
  if (rc == PFD_OK) rc = o.finish(h->stream);
  if (rc != PFD_OK) return rc;  // the local failure (its message is already set)
  if (!ok_all) {
    pfd_set_error("a row block failed or the raster holds cells that never reach a pit (cycles); the multi-GPU "
                  "path requires a valid flow direction raster (FlwdirRaster.isvalid)");
    return PFD_EUNSUPPORTED;
  }
  return PFD_OK;
}
