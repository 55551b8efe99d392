// api.hip — C-ABI plumbing of libpfd_hip: errors, device memory helpers, handle life cycle,
// index exports.  See include/pfd.h for the contract of every entry point.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "common.h"

static thread_local char g_err[1024] = "";

void pfd_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *pfd_last_error(void) { return g_err; }
extern "C" int pfd_abi_version(void) { return PFD_ABI_VERSION; }

extern "C" int pfd_device_count(int *count) {
  if (!count) {
    pfd_set_error("pfd_device_count: NULL argument");
    return PFD_EINVAL;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    pfd_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return PFD_ENODEVICE;
  }
  *count = n;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// caching device allocator
// ---------------------------------------------------------------------------------------------
namespace {
// A reserved arena (pfd_reserve): ONE hipMalloc, carved into blocks of any size class by a first-fit free list with
// coalescing.  Measured on MI355X (tools/alloc_probe_big.py): hipMalloc of a block of tens of GiB returns in 0.2 ms
// or in 0.5 - 7 s, unpredictably (memory released shortly before, by this or another process, seems to be scrubbed
// first) — every stall the round-4 records show (a "cached" sweep of 591 ms, first calls of 1.2 - 3.2 s) was a
// hipMalloc of a size class the cache had not seen.  With an arena a steady state never calls hipMalloc at all.
struct Arena {
  int device = 0;
  char *base = nullptr;
  size_t bytes = 0;
  std::map<size_t, size_t> free_at;  // offset -> length of the free chunks (address order: neighbours coalesce)
};
struct DevCache {
  std::mutex mu;
  std::unordered_map<void *, std::pair<int, size_t>> live;       // ptr -> (device, class bytes)
  std::map<std::pair<int, size_t>, std::vector<void *>> idle;    // (device, class bytes) -> blocks
  size_t idle_bytes = 0;
  std::vector<Arena> arenas;
  std::unordered_map<void *, int> live_arena;                    // ptr -> arena index (blocks carved from an arena)
  size_t arena_hits = 0, malloc_calls = 0, idle_hits = 0, near_hits = 0;
};
DevCache &cache() {
  static DevCache c;
  return c;
}
size_t size_class(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes >= (1u << 20)) return (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);  // 2 MiB steps
  size_t c = 256;
  while (c < bytes) c <<= 1;
  return c;
}
// idle blocks kept per process: 96 GiB unless PFD_IDLE_CAP_GB says otherwise (0 = return every block at once)
size_t idle_cap() {
  static const size_t v = [] {
    const char *e = getenv("PFD_IDLE_CAP_GB");
    return (size_t)(e ? std::max(0L, strtol(e, nullptr, 10)) : 96L) << 30;
  }();
  return v;
}
}  // namespace

// Test-only switches (forcing a fallback engine, shrinking a capacity): inert unless PFD_ENABLE_KNOBS=1 is in the
// environment when the library is first used (tests/conftest.py sets it), so that a stray variable cannot change
// what a production process runs.
const char *pfd_knob(const char *name) {
  static const bool on = [] {
    const char *e = getenv("PFD_ENABLE_KNOBS");
    return e && e[0] == '1';
  }();
  return on ? getenv(name) : nullptr;
}

static int pfd_dmalloc_raw(void **p, size_t bytes);
#ifdef PFD_DEVTOOLS
// debugging aid (builds with DEVTOOLS=1 only): PFD_POISON=<byte> fills every block handed out with that byte
// (synchronously), which turns a read of memory the caller never wrote into a reproducible failure
static int poison_byte() {
  static const int v = [] {
    const char *e = getenv("PFD_POISON");
    return e ? (int)strtol(e, nullptr, 0) & 0xff : -1;
  }();
  return v;
}
int pfd_dmalloc(void **p, size_t bytes) {
  const int rc = pfd_dmalloc_raw(p, bytes);
  if (rc == PFD_OK && poison_byte() >= 0) {
    (void)hipDeviceSynchronize();
    (void)hipMemset(*p, poison_byte(), size_class(bytes));
    (void)hipDeviceSynchronize();
  }
  return rc;
}
#else
int pfd_dmalloc(void **p, size_t bytes) { return pfd_dmalloc_raw(p, bytes); }
#endif
#define ARENA_MIN (1u << 20)  // smaller blocks stay with hipMalloc + the class cache (they are cheap and many)
// carve `cls` bytes from an arena of the device: best fit over the free chunks (a handful)
static void *arena_take(DevCache &c, int dev, size_t cls) {
  for (size_t ai = 0; ai < c.arenas.size(); ++ai) {
    Arena &a = c.arenas[ai];
    if (a.device != dev) continue;
    auto best = a.free_at.end();
    for (auto it = a.free_at.begin(); it != a.free_at.end(); ++it)
      if (it->second >= cls && (best == a.free_at.end() || it->second < best->second)) best = it;
    if (best == a.free_at.end()) continue;
    const size_t off = best->first, len = best->second;
    a.free_at.erase(best);
    if (len > cls) a.free_at[off + cls] = len - cls;
    void *p = a.base + off;
    c.live_arena[p] = (int)ai;
    return p;
  }
  return nullptr;
}
static void arena_give(DevCache &c, int ai, void *p, size_t cls) {
  Arena &a = c.arenas[ai];
  size_t off = (size_t)((char *)p - a.base), len = cls;
  auto nx = a.free_at.lower_bound(off);
  if (nx != a.free_at.begin()) {
    auto pv = std::prev(nx);
    if (pv->first + pv->second == off) {
      off = pv->first, len += pv->second;
      a.free_at.erase(pv);
    }
  }
  if (nx != a.free_at.end() && off + len == nx->first) {
    len += nx->second;
    a.free_at.erase(nx);
  }
  a.free_at[off] = len;
}
// arenas of the device that hold no live block go back to the driver (pfd_reserve(device, 0); also what an allocation
// failure tries before it gives up: an arena too small for the request is only in its way)
static void release_empty_arenas(int device) {
  DevCache &c = cache();
  std::lock_guard<std::mutex> g(c.mu);
  for (Arena &a : c.arenas) {
    if (a.device != device || !a.base) continue;
    if (a.free_at.size() == 1 && a.free_at.begin()->second == a.bytes) {
      (void)hipFree(a.base);  // (hipFree needs no current device: the caller's stays as it is)
      a.base = nullptr, a.bytes = 0;
      a.free_at.clear();
    }
  }
  // dead entries at the END of the list can go (live_arena refers to arenas by index: the others keep their place)
  while (!c.arenas.empty() && !c.arenas.back().base) c.arenas.pop_back();
}
static int pfd_dmalloc_raw(void **p, size_t bytes) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  const size_t cls = size_class(bytes);
  DevCache &c = cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.idle.find({dev, cls});
    if (it != c.idle.end() && !it->second.empty()) {
      *p = it->second.back();
      it->second.pop_back();
      c.idle_bytes -= cls;
      c.live[*p] = {dev, cls};
      ++c.idle_hits;
      return PFD_OK;
    }
    if (cls >= ARENA_MIN) {
      if (void *q = arena_take(c, dev, cls)) {
        *p = q;
        c.live[q] = {dev, cls};
        ++c.arena_hits;
        return PFD_OK;
      }
      // no exact class idle: an idle block up to an eighth larger serves as well (row blocks of one raster differ by a
      // few MiB per array — every miss here is a hipMalloc that may stall for seconds)
      auto nb = c.idle.lower_bound({dev, cls});
      for (; nb != c.idle.end() && nb->first.first == dev && nb->first.second <= cls + cls / 8; ++nb) {
        if (nb->second.empty()) continue;
        *p = nb->second.back();
        nb->second.pop_back();
        c.idle_bytes -= nb->first.second;
        c.live[*p] = {dev, nb->first.second};
        ++c.near_hits;
        return PFD_OK;
      }
    }
    ++c.malloc_calls;
  }
  hipError_t e = hipMalloc(p, cls);
  if (e != hipSuccess) {  // give the cached blocks (and arenas nobody uses) back and retry once
    (void)hipGetLastError();
    pfd_trim(dev);
    release_empty_arenas(dev);
    e = hipMalloc(p, cls);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();  // (the failure is reported here; left sticky, the next launch check of ANY call would report it again)
    *p = nullptr;
    pfd_set_error("hipMalloc(%zu bytes) failed: %s", cls, hipGetErrorString(e));
    return PFD_ENOMEM;
  }
  std::lock_guard<std::mutex> g(c.mu);
  c.live[*p] = {dev, cls};
  return PFD_OK;
}

// The stream of the API call in progress on this thread (set by pfd_check_handle_lazy).  A block is only
// recycled (or returned to the driver) once that stream is idle: work that still reads or writes it may be in
// flight when a temporary is released in the middle of an asynchronous sequence (a re-used rocprim scratch
// buffer, a DevBuf re-allocated for the next step), and the next owner may be another handle's stream, which is
// not ordered behind this one.  The wait costs a few microseconds on an idle stream.
static thread_local hipStream_t g_cur_stream = nullptr;
static thread_local bool g_have_stream = false;

void pfd_dfree(void *p) {
  if (!p) return;
  if (g_have_stream) (void)hipStreamSynchronize(g_cur_stream);
  else (void)hipDeviceSynchronize();
  DevCache &c = cache();
  std::lock_guard<std::mutex> g(c.mu);
  auto it = c.live.find(p);
  if (it == c.live.end()) {
    (void)hipFree(p);
    return;
  }
  const auto key = it->second;
  c.live.erase(it);
  auto ar = c.live_arena.find(p);
  if (ar != c.live_arena.end()) {  // carved from an arena: back into its free list (any class can have it next)
    const int ai = ar->second;
    c.live_arena.erase(ar);
    arena_give(c, ai, p, key.second);
    return;
  }
  if (c.idle_bytes + key.second > idle_cap()) {
    (void)hipFree(p);
    return;
  }
  c.idle[key].push_back(p);
  c.idle_bytes += key.second;
}

extern "C" int pfd_trim(int device) {
  DevCache &c = cache();
  std::lock_guard<std::mutex> g(c.mu);
  for (auto &kv : c.idle) {
    if (device >= 0 && kv.first.first != device) continue;
    for (void *p : kv.second) {
      (void)hipFree(p);
      c.idle_bytes -= kv.first.second;
    }
    kv.second.clear();
  }
  return PFD_OK;
}

__global__ void __launch_bounds__(256) k_arena_touch(uint4 *__restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
extern "C" int pfd_reserve(int device, size_t bytes) {
  if (device < 0) {
    pfd_set_error("pfd_reserve: bad device %d", device);
    return PFD_EINVAL;
  }
  DevCache &c = cache();
  if (bytes == 0) {  // release the arenas of the device that hold no live block
    release_empty_arenas(device);
    return PFD_OK;
  }
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  HIPCHK(hipSetDevice(device));
  const size_t sz = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
  void *base = nullptr;
  hipError_t e = hipMalloc(&base, sz);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    pfd_trim(device);
    e = hipMalloc(&base, sz);
  }
  // (first use of freshly allocated HBM is slower than the second — measured: a row block's phase A takes 6.5 ms on
  //  fresh blocks against 2.9 ms on recycled ones — so the arena is written once here, where nobody is timing)
  if (e == hipSuccess && !pfd_knob("PFD_RESERVE_NO_TOUCH")) {
    // (a kernel of its own, not hipMemset: profiles of a pass count the runtime's fill kernel among the pass's clears)
    k_arena_touch<<<4096, 256>>>((uint4 *)base, sz / 16);
    (void)hipDeviceSynchronize();
  }
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // (see pfd_dmalloc)
    pfd_set_error("pfd_reserve(%zu bytes) failed: %s", sz, hipGetErrorString(e));
    return PFD_ENOMEM;
  }
  std::lock_guard<std::mutex> g(c.mu);
  Arena a;
  a.device = device, a.base = (char *)base, a.bytes = sz;
  a.free_at[0] = sz;
  c.arenas.push_back(std::move(a));
  return PFD_OK;
}
extern "C" int pfd_alloc_stats(int64_t out[8]) {
  if (!out) {
    pfd_set_error("pfd_alloc_stats: NULL argument");
    return PFD_EINVAL;
  }
  DevCache &c = cache();
  std::lock_guard<std::mutex> g(c.mu);
  size_t reserved = 0, reserved_free = 0;
  for (const Arena &a : c.arenas) {
    reserved += a.bytes;
    for (const auto &kv : a.free_at) reserved_free += kv.second;
  }
  out[0] = (int64_t)c.malloc_calls, out[1] = (int64_t)c.idle_hits, out[2] = (int64_t)c.near_hits, out[3] = (int64_t)c.arena_hits;
  out[4] = (int64_t)c.idle_bytes, out[5] = (int64_t)reserved, out[6] = (int64_t)reserved_free, out[7] = (int64_t)c.live.size();
  return PFD_OK;
}

extern "C" int pfd_mem_info(int device, int64_t out[2]) {
  if (!out || device < 0) {
    pfd_set_error("pfd_mem_info: bad arguments");
    return PFD_EINVAL;
  }
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  HIPCHK(hipSetDevice(device));
  size_t fr = 0, tot = 0;
  const hipError_t e = hipMemGetInfo(&fr, &tot);
  (void)hipSetDevice(prev);
  HIPCHK(e);
  out[0] = (int64_t)fr, out[1] = (int64_t)tot;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// host <-> device traffic of this thread's calls; pre-faulting of host results
// ---------------------------------------------------------------------------------------------
PfdTransfer &pfd_transfer() {
  static thread_local PfdTransfer t;
  return t;
}
double pfd_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
extern "C" int pfd_transfer_stats(double out[6], int reset) {
  if (!out) {
    pfd_set_error("pfd_transfer_stats: NULL argument");
    return PFD_EINVAL;
  }
  PfdTransfer &t = pfd_transfer();
  out[0] = t.h2d_bytes, out[1] = t.h2d_ms, out[2] = t.d2h_bytes, out[3] = t.d2h_ms, out[4] = t.prefault_ms, out[5] = t.host_results;
  if (reset) t = PfdTransfer();
  return PFD_OK;
}
namespace {
struct PrefaultJob {
  std::vector<std::thread> th;
  std::atomic<bool> stop{false};
  double t0 = 0;
  std::vector<double> end_ms;  // per thread: when it was done
};
constexpr size_t PREFAULT_MIN = 64u << 20;
int prefault_threads() {
  static const int v = [] {
    const char *e = getenv("PFD_PREFAULT_THREADS");  // 0 switches the pre-faulting off
    if (e) return (int)std::max(0L, std::min(64L, strtol(e, nullptr, 10)));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc / 2));
  }();
  return v;
}
}  // namespace
void *pfd_prefault_begin(void *dst, size_t bytes) {
  const int T = prefault_threads();
  if (!dst || bytes < PREFAULT_MIN || T <= 0) return nullptr;
  PrefaultJob *job = new PrefaultJob();
  job->t0 = pfd_now_ms();
  // transparent huge pages where the host offers them on request (numpy asks for them itself; a plain malloc does not):
  // 2 MiB pages fault in at ~20 GB/s per thread, 4 KiB pages at ~12 GB/s for the whole process
  const size_t HP = 2u << 20;
  char *lo = (char *)(((size_t)dst + HP - 1) & ~(HP - 1)), *hi = (char *)(((size_t)dst + bytes) & ~(HP - 1));
  if (hi > lo) (void)madvise(lo, (size_t)(hi - lo), MADV_HUGEPAGE);
  const size_t per = ((bytes / (size_t)T) + HP - 1) & ~(HP - 1);
  job->end_ms.assign((size_t)T, job->t0);
  for (int t = 0; t < T; ++t) {
    const size_t o = per * (size_t)t;
    if (o >= bytes) break;
    const size_t m = std::min(per, bytes - o);
    volatile char *p = (volatile char *)dst + o;
    std::atomic<bool> *stop = &job->stop;
    double *done = &job->end_ms[(size_t)t];
    job->th.emplace_back([p, m, stop, done] {
      struct Stamp {
        double *d;
        ~Stamp() { *d = pfd_now_ms(); }
      } stamp{done};
      // one byte per 4 KiB page, read and written back: a fresh page becomes resident, a page that holds data keeps it.
      // Without huge pages the faults are served at ~0.8 GB/s per thread — slower than the copy's own faulting: give up.
      const size_t PROBE = 32u << 20;
      const double t0 = pfd_now_ms();
      for (size_t i = 0; i < m; i += 4096) {
        if ((i & (PROBE - 1)) == 0 && i) {
          if (stop->load(std::memory_order_relaxed)) return;
          if (i == PROBE && (double)PROBE / ((pfd_now_ms() - t0) * 1e6) < 3.0) {  // GB/s
            stop->store(true, std::memory_order_relaxed);
            return;
          }
        }
        p[i] = p[i];
      }
    });
  }
  return job;
}
double pfd_prefault_join(void *state) {
  if (!state) return 0.0;
  PrefaultJob *job = (PrefaultJob *)state;
  for (auto &t : job->th) t.join();
  double ms = 0.0;
  for (double e : job->end_ms) ms = std::max(ms, e - job->t0);
  delete job;
  return ms;
}

// Pinned staging for the small host <-> device transfers of the split-phase row-block protocol: a pageable copy out of a
// host array HIP has not seen costs 10 - 50 ms the first time (measured: tools/alloc_probe.py), whatever its size.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::vector<std::pair<void *, size_t>> idle;
};
PinnedPool &pinned() {
  static PinnedPool p;
  return p;
}
}  // namespace
void *pfd_pinned_take(size_t bytes, size_t *cap) {
  PinnedPool &pp = pinned();
  {
    std::lock_guard<std::mutex> g(pp.mu);
    for (size_t i = 0; i < pp.idle.size(); ++i)
      if (pp.idle[i].second >= bytes) {
        auto b = pp.idle[i];
        pp.idle.erase(pp.idle.begin() + (long)i);
        *cap = b.second;
        return b.first;
      }
  }
  void *q = nullptr;
  const size_t sz = std::max<size_t>((bytes + 4095) & ~(size_t)4095, 1u << 16);
  if (hipHostMalloc(&q, sz, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  *cap = sz;
  return q;
}
void pfd_pinned_give(void *p, size_t cap) {
  if (!p) return;
  PinnedPool &pp = pinned();
  std::lock_guard<std::mutex> g(pp.mu);
  if (pp.idle.size() >= 8) {
    (void)hipHostFree(p);
    return;
  }
  pp.idle.emplace_back(p, cap);
}

// hipStreamCreate/Destroy cost ~1 ms each: streams are pooled per device and reused
namespace {
struct StreamPool {
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> idle;
};
StreamPool &streams() {
  static StreamPool p;
  return p;
}
}  // namespace
// (low: a stream of the lowest priority — the handle's SECOND stream, whose bandwidth pass runs beside latency-bound rounds
//  of the first: the rounds' few waves must not queue behind the pass's workgroups.  Pool key: device, or -1 - device)
static int acquire_stream(int device, hipStream_t *out, bool low = false) {
  StreamPool &p = streams();
  {
    std::lock_guard<std::mutex> g(p.mu);
    auto &v = p.idle[low ? -1 - device : device];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return PFD_OK;
    }
  }
  if (low) {
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, least));
    return PFD_OK;
  }
  HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return PFD_OK;
}
static void release_stream(int device, hipStream_t s, bool low = false) {
  StreamPool &p = streams();
  std::lock_guard<std::mutex> g(p.mu);
  p.idle[low ? -1 - device : device].push_back(s);
}

int pfd_aux_stream(pfd_raster *h) {
  if (!h->stream2) {
    h->stream2_low = !pfd_knob("PFD_AUX_PRIORITY_SAME");
    PFDCHK(acquire_stream(h->device, &h->stream2, h->stream2_low));
  }
  if (!h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  if (!h->ev_join) HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  return PFD_OK;
}

static int select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    pfd_set_error("no HIP device available (%s); the MI355X path has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return PFD_ENODEVICE;
  }
  if (device < 0 || device >= n) {
    pfd_set_error("device index %d out of range [0,%d)", device, n);
    return PFD_ENODEVICE;
  }
  HIPCHK(hipSetDevice(device));
  return PFD_OK;
}

extern "C" int pfd_malloc(int device, size_t bytes, void **ptr) {
  if (!ptr) {
    pfd_set_error("pfd_malloc: NULL ptr");
    return PFD_EINVAL;
  }
  PFDCHK(select_device(device));
  hipError_t e = hipMalloc(ptr, bytes ? bytes : 16);
  if (e != hipSuccess) {  // the library's own idle blocks / unused arenas may be what is in the way: give them back, retry once
    (void)hipGetLastError();
    pfd_trim(device);
    release_empty_arenas(device);
    e = hipMalloc(ptr, bytes ? bytes : 16);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();  // (see pfd_dmalloc)
    *ptr = nullptr;
    pfd_set_error("pfd_malloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return PFD_ENOMEM;
  }
  return PFD_OK;
}
extern "C" int pfd_free(int device, void *ptr) {
  PFDCHK(select_device(device));
  if (ptr) HIPCHK(hipFree(ptr));
  return PFD_OK;
}
extern "C" int pfd_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes) {
  PFDCHK(select_device(device));
  HIPCHK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return PFD_OK;
}
extern "C" int pfd_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes) {
  PFDCHK(select_device(device));
  // a large download into fresh pages is bound by the copy's own first-touch faults (15-20 GB/s): a few host threads touch
  // the pages first (pfd_prefault_begin: nothing for small buffers), then the copy runs at the PCIe rate
  (void)pfd_prefault_join(pfd_prefault_begin(dst_host, bytes));
  HIPCHK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return PFD_OK;
}
extern "C" int pfd_device_synchronize(int device) {
  PFDCHK(select_device(device));
  HIPCHK(hipDeviceSynchronize());
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
int pfd_check_handle_lazy(pfd_raster *h) {
  if (!h) {
    pfd_set_error("NULL raster handle");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(h->device));
  g_cur_stream = h->stream;
  g_have_stream = true;
  return PFD_OK;
}
int pfd_check_handle(pfd_raster *h) {
  PFDCHK(pfd_check_handle_lazy(h));
  return pfd_ensure_normalised(h);
}

static void free_handle(pfd_raster *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (auto &s : h->segs) {
    (void)hipEventDestroy(s.e0);
    (void)hipEventDestroy(s.e1);
  }
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  pfd_free_pending(h);
  pfd_free_pending_basins(h);
  pfd_free_hand_block(h);
  pfd_dfree(h->ncode);
  pfd_dfree(h->raw_owned);
  pfd_dfree(h->seq);
  pfd_dfree(h->seq_kids2);  // (seq_kids / seq_own / cell_kids live in the same allocation)
  pfd_dfree(h->halo_raw);
  pfd_free_xplan(h);
  pfd_free_general(h);
  pfd_dfree(h->pits);
  pfd_dfree(h->ctrl);
  if (h->stream2) {
    (void)hipStreamSynchronize(h->stream2);
    release_stream(h->device, h->stream2, h->stream2_low);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->stream) {
    if (g_have_stream && g_cur_stream == h->stream) g_have_stream = false;
    release_stream(h->device, h->stream);
  }
  delete h;
}

int pfd_handle_alloc(i64 nrow, i64 ncol, int device, pfd_raster **out) {
  PFDCHK(select_device(device));
  pfd_raster *h = new pfd_raster();
  h->device = device;
  h->nrow = nrow;
  h->ncol = ncol;
  h->n = nrow * ncol;
  h->own_rows = nrow;
  h->geo = make_geo(nrow, ncol);
  int rc = acquire_stream(device, &h->stream);
  if (rc == PFD_OK) g_cur_stream = h->stream, g_have_stream = true;
  if (rc == PFD_OK) rc = pfd_dmalloc((void **)&h->ncode, (size_t)h->n + 64);
  if (rc == PFD_OK) rc = pfd_dmalloc((void **)&h->ctrl, 64 * sizeof(u64));
  if (rc != PFD_OK) {
    free_handle(h);
    return rc;
  }
  h->bytes_held = (size_t)h->n + 64 * sizeof(u64);
  *out = h;
  return PFD_OK;
}
int pfd_reject_general(pfd_raster *h, const char *what) {
  if (!h->gen) return PFD_OK;
  pfd_set_error("%s is not available on a general idxs_ds graph (links outside the 8 neighbours)", what);
  return PFD_EUNSUPPORTED;
}

static int raster_create_impl(const uint8_t *d8, int64_t own_rows, int64_t ncol, int halo_top, int halo_bot,
                              int memspace, int device, pfd_raster **out, bool deferred = false) {
  if (!out) {
    pfd_set_error("pfd_raster_create: NULL out");
    return PFD_EINVAL;
  }
  *out = nullptr;
  if (!d8 || own_rows <= 0 || ncol <= 0 || halo_top < 0 || halo_top > 1 || halo_bot < 0 || halo_bot > 1) {
    pfd_set_error("pfd_raster_create: invalid raster (ptr=%p, shape=%lld x %lld, halo %d/%d)", (const void *)d8,
                  (long long)own_rows, (long long)ncol, halo_top, halo_bot);
    return PFD_EINVAL;
  }
  const int64_t nrow = own_rows + halo_top + halo_bot;
  // Rasters beyond 4294967294 cells cannot use the level engine (32-bit cell indices) but the
  // LDS-tiled path addresses cells as (tile, local index): accept them as long as the tile
  // bookkeeping fits its 30-bit slot ids (e.g. 90000 x 90000 = 8.1e9 cells on one 288 GB GPU).
  const unsigned __int128 n128 = (unsigned __int128)nrow * (unsigned __int128)ncol;
  const uint64_t ntr = ((uint64_t)nrow + 63) / 64, ntc = ((uint64_t)ncol + 63) / 64;
  const uint64_t nslots = ((ntr + 7) / 8) * ((ntc + 7) / 8) * 16384ull;
  if (n128 > 4294967294ull && (nslots >= 0x3FFFFFFFull || ntr > 65535ull || (uint64_t)ncol >= 0x40000000ull)) {
    pfd_set_error("pfd_raster_create: %lld x %lld cells are more than one handle can address; split the "
                  "raster into row blocks (one handle per GPU)", (long long)nrow, (long long)ncol);
    return PFD_EUNSUPPORTED;
  }
  if (memspace != PFD_HOST && memspace != PFD_DEVICE) {
    pfd_set_error("pfd_raster_create: bad memspace %d", memspace);
    return PFD_EINVAL;
  }
  PFDCHK(select_device(device));
  pfd_raster *h = new pfd_raster();
  h->device = device;
  h->nrow = nrow;
  h->ncol = ncol;
  h->n = nrow * ncol;
  h->halo_top = halo_top;
  h->halo_bot = halo_bot;
  h->own_rows = own_rows;
  h->geo = make_geo(nrow, ncol);
  int rc = PFD_OK;
  do {
    if ((rc = acquire_stream(device, &h->stream)) != PFD_OK) break;
    g_cur_stream = h->stream, g_have_stream = true;
    if ((rc = pfd_dmalloc((void **)&h->ncode, (size_t)h->n + 64)) != PFD_OK) break;  // +slack: dword halo loads
    if ((rc = pfd_dmalloc((void **)&h->ctrl, 64 * sizeof(u64))) != PFD_OK) break;
    h->bytes_held = (size_t)h->n + 64 * sizeof(u64);
    if (deferred && h->n >= 4) {
      // no kernel and no synchronisation here: the codes are normalised, validated and counted by
      // the first operation.  Device input is referenced, not copied (see include/pfd.h).
      if (memspace == PFD_HOST) {
        if ((rc = pfd_dmalloc((void **)&h->raw_owned, (size_t)h->n)) != PFD_OK) break;
        const double t0 = pfd_now_ms();
        if (hipMemcpyAsync(h->raw_owned, d8, (size_t)h->n, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
          pfd_set_error("pfd_raster_create_deferred: upload failed");
          rc = PFD_EHIP;
          break;
        }
        (void)hipStreamSynchronize(h->stream);  // the caller may reuse its host buffer at once
        pfd_transfer().h2d_bytes += (double)h->n, pfd_transfer().h2d_ms += pfd_now_ms() - t0;
        h->raw = h->raw_owned;
      } else {
        h->raw = d8;
      }
      h->normalised = false;
      h->n_valid = h->n_pits = -1;
      break;
    }
    InArg in;
    if ((rc = in.bind(d8, (size_t)h->n, memspace, h->stream)) != PFD_OK) break;
    if ((rc = pfd_normalise_and_count(h, (const u8 *)in.dev)) != PFD_OK) break;
  } while (0);
  if (rc != PFD_OK) {
    free_handle(h);
    return rc;
  }
  *out = h;
  return PFD_OK;
}

extern "C" int pfd_raster_create(const uint8_t *d8, int64_t nrow, int64_t ncol, int memspace, int device,
                                 pfd_raster **out) {
  return raster_create_impl(d8, nrow, ncol, 0, 0, memspace, device, out);
}

extern "C" int pfd_raster_create_block(const uint8_t *d8, int64_t own_rows, int64_t ncol, int halo_top, int halo_bot,
                                       int memspace, int device, pfd_raster **out) {
  return raster_create_impl(d8, own_rows, ncol, halo_top, halo_bot, memspace, device, out);
}

extern "C" int pfd_raster_create_deferred(const uint8_t *d8, int64_t own_rows, int64_t ncol, int halo_top, int halo_bot,
                                          int memspace, int device, pfd_raster **out) {
  return raster_create_impl(d8, own_rows, ncol, halo_top, halo_bot, memspace, device, out, true);
}

extern "C" int pfd_raster_validate(pfd_raster *h) { return pfd_check_handle(h); }

extern "C" int pfd_raster_destroy(pfd_raster *h) {
  free_handle(h);
  return PFD_OK;
}

extern "C" int pfd_raster_info(pfd_raster *h, int64_t info[8]) {
  PFDCHK(pfd_check_handle_lazy(h));  // (a deferred handle reports n_valid = n_pits = -1 until its first operation)
  if (!info) {
    pfd_set_error("pfd_raster_info: NULL info");
    return PFD_EINVAL;
  }
  info[0] = h->nrow;
  info[1] = h->ncol;
  info[2] = h->n_valid;
  info[3] = h->n_pits;
  info[4] = (h->ordered || (pfd_wide_cells(h) && h->n_seq >= 0)) ? h->n_seq : -1;  // (64-bit forms: set by rank / idxs_seq, order64.hip)
  info[5] = h->ordered ? h->n_levels : -1;
  info[6] = h->device;
  info[7] = (int64_t)h->bytes_held;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling segments
// ---------------------------------------------------------------------------------------------
void pfd_seg_clear(pfd_raster *h) {
  for (auto &s : h->segs) {
    (void)hipEventDestroy(s.e0);
    (void)hipEventDestroy(s.e1);
  }
  h->segs.clear();
}
void pfd_seg_begin(pfd_raster *h, const char *name) {
  if (!h->profiling) return;
  PfdSegment s;
  s.name = name;
  s.launches = 0;
  (void)hipEventCreate(&s.e0);
  (void)hipEventCreate(&s.e1);
  (void)hipEventRecord(s.e0, h->stream);
  h->segs.push_back(s);
}
void pfd_seg_end(pfd_raster *h, i64 launches) {
  if (!h->profiling || h->segs.empty()) return;
  PfdSegment &s = h->segs.back();
  s.launches = launches;
  (void)hipEventRecord(s.e1, h->stream);
}

extern "C" int pfd_set_block_io(pfd_raster *h, int seed_memspace) {
  if (!h || (seed_memspace != PFD_HOST && seed_memspace != PFD_DEVICE)) {
    pfd_set_error("pfd_set_block_io: bad arguments");
    return PFD_EINVAL;
  }
  h->block_seed_space = seed_memspace;
  return PFD_OK;
}

extern "C" int pfd_set_block_update(pfd_raster *h, int mode) {
  if (!h || mode < 0 || mode > 2) {
    pfd_set_error("pfd_set_block_update: bad arguments");
    return PFD_EINVAL;
  }
  h->block_update = mode;
  if (mode == 0 && h->xplan) pfd_xinc_drop(h);
  return PFD_OK;
}

extern "C" int pfd_set_profiling(pfd_raster *h, int enable) {
  PFDCHK(pfd_check_handle_lazy(h));
  h->profiling = enable != 0;
  h->count_rounds = enable >= 2;
  if (!enable) pfd_seg_clear(h);
  return PFD_OK;
}

extern "C" int pfd_last_timing(pfd_raster *h, int max_seg, double *ms, int64_t *launches, char *names,
                               size_t names_len, int *nseg) {
  PFDCHK(pfd_check_handle_lazy(h));
  if (!nseg) {
    pfd_set_error("pfd_last_timing: NULL nseg");
    return PFD_EINVAL;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  std::string joined;
  int k = 0;
  for (auto &s : h->segs) {
    if (k >= max_seg) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.e0, s.e1) != hipSuccess) t = -1.f;
    if (ms) ms[k] = (double)t;
    if (launches) launches[k] = s.launches;
    if (k) joined += ";";
    joined += s.name;
    ++k;
  }
  *nseg = k;
  if (names && names_len) {
    strncpy(names, joined.c_str(), names_len - 1);
    names[names_len - 1] = 0;
  }
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// index exports
// ---------------------------------------------------------------------------------------------
template <class IDX>
__global__ void k_export_idxs_ds(const u8 *__restrict__ ncode, Geo g, IDX *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const u32 code = ncode[i];
  IDX v;
  if (code == D8_MV)
    v = (IDX)-1;
  else
    v = (IDX)d8_down(g, i, code);
  out[i] = v;
}

template <class IDX>
__global__ void k_export_u32(const u32 *__restrict__ src, u64 m, IDX *__restrict__ out) {
  // (grid-stride: a launch of more than 2^32 threads per dimension does not run whole — m may exceed that)
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) out[i] = (IDX)src[i];
}

static size_t idx_size(int idx_dtype) {
  switch (idx_dtype) {
    case PFD_I32:
    case PFD_U32:
      return 4;
    case PFD_I64:
      return 8;
    default:
      return 0;
  }
}

extern "C" int pfd_idxs_ds(pfd_raster *h, int idx_dtype, void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (h->gen) return pfd_gen_idxs_ds(h, idx_dtype, out, memspace);
  PFDCHK(pfd_require_whole(h, "idxs_ds"));
  const size_t es = idx_size(idx_dtype);
  if (!es || !out) {
    pfd_set_error("pfd_idxs_ds: bad index dtype %d or NULL out", idx_dtype);
    return PFD_EINVAL;
  }
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * es, memspace));
  const u32 grid = cdiv_u32((u64)h->n, 256);
  if (idx_dtype == PFD_I32)
    k_export_idxs_ds<i32><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (i32 *)o.dev);
  else if (idx_dtype == PFD_U32)
    k_export_idxs_ds<u32><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (u32 *)o.dev);
  else
    k_export_idxs_ds<i64><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (i64 *)o.dev);
  KCHK();
  return o.finish(h->stream);
}

int pfd_export_u32(pfd_raster *h, const u32 *src, i64 m, int idx_dtype, void *out, int memspace) {
  const size_t es = idx_size(idx_dtype);
  if (!es || !out) {
    pfd_set_error("index export: bad index dtype %d or NULL out", idx_dtype);
    return PFD_EINVAL;
  }
  OutArg o;
  PFDCHK(o.bind(out, (size_t)m * es, memspace));
  if (m > 0) {
    const u32 grid = (u32)std::min<u64>(cdiv_u32((u64)m, 256), 1u << 22);
    if (idx_dtype == PFD_I32)
      k_export_u32<i32><<<grid, 256, 0, h->stream>>>(src, (u64)m, (i32 *)o.dev);
    else if (idx_dtype == PFD_U32)
      k_export_u32<u32><<<grid, 256, 0, h->stream>>>(src, (u64)m, (u32 *)o.dev);
    else
      k_export_u32<i64><<<grid, 256, 0, h->stream>>>(src, (u64)m, (i64 *)o.dev);
    KCHK();
  }
  return o.finish(h->stream);
}
