// common.h — shared declarations of libpfd_hip (gfx950 / MI355X only).
//
// Device-side representation of a D8 raster:
//   ncode  u8[n]   "normalised" D8 code per cell: one of the eight direction codes for a cell
//                  whose downstream neighbour is a valid in-raster cell, 0 for EVERY pit (pit
//                  code 0/255, target outside, or target nodata — reference
//                  pyflwdir/core_d8.py:57-63), 247 for nodata.  1 byte/cell is the whole graph:
//                  the downstream index and the upstream cells are decoded on the fly from
//                  the cell's own byte and its 8 neighbours' bytes.
//   seq    u32[n_seq]  cells grouped by rank (level l = cells l steps away from their pit),
//                  level 0 = pits ascending; lvl_off[l] .. lvl_off[l+1] delimits level l.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/pfd.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define D8_MV 247u
#define D8_HALO 254u  // valid cell of a halo row (owned by the neighbouring row block): a weightless sink

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void pfd_set_error(const char *fmt, ...);
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      pfd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
      return (e_ == hipErrorOutOfMemory) ? PFD_ENOMEM : PFD_EHIP;                                 \
    }                                                                                             \
  } while (0)
#define PFDCHK(expr)                 \
  do {                               \
    int rc_ = (expr);                \
    if (rc_ != PFD_OK) return rc_;   \
  } while (0)
#define KCHK() HIPCHK(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// raster geometry + exact u32 division by ncol (Granlund-Montgomery round-up method), so that
// kernels working from a linear cell index get (row, col) without a hardware divide.
// ---------------------------------------------------------------------------------------------
struct Geo {
  u32 nrow, ncol;
  u32 n;           // nrow*ncol (<= 4294967294)
  u32 dm, ds1, ds2;  // magic for / ncol
};
static inline Geo make_geo(i64 nrow, i64 ncol) {
  Geo g;
  g.nrow = (u32)nrow;
  g.ncol = (u32)ncol;
  g.n = (u32)(nrow * ncol);
  u32 d = g.ncol, l = 0;
  while ((1ull << l) < d) ++l;
  g.dm = (u32)((((1ull << l) - d) << 32) / d + 1);
  g.ds1 = l < 1 ? l : 1;
  g.ds2 = l > 0 ? l - 1 : 0;
  return g;
}
__device__ __forceinline__ u32 geo_row(const Geo &g, u32 i) {
  const u32 t = __umulhi(g.dm, i);
  return (t + ((i - t) >> g.ds1)) >> g.ds2;
}

// ---------------------------------------------------------------------------------------------
// D8 code helpers.  Direction slot k = log2(code): 0 E, 1 SE, 2 S, 3 SW, 4 W, 5 NW, 6 N, 7 NE
// (reference pyflwdir/core_d8.py:15 `_ds`).  The neighbour located in slot k of a cell drains
// INTO that cell iff its code is the opposite slot, 1 << ((k+4)&7) (core_d8.py:16 `_us`).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int d8_dr(int k) { return (k >= 1 && k <= 3) ? 1 : (k >= 5 ? -1 : 0); }
__device__ __forceinline__ int d8_dc(int k) { return (k == 0 || k == 1 || k == 7) ? 1 : ((k >= 3 && k <= 5) ? -1 : 0); }
__device__ __forceinline__ bool d8_is_dir(u32 code) { return code != 0 && (code & (code - 1)) == 0; }  // power of two
__device__ __forceinline__ int d8_slot(u32 code) { return __ffs((int)code) - 1; }

// children visiting orders: slots of the neighbours in DESCENDING linear index (the order in
// which the reference's serial loop adds children into a parent: SE,S,SW,E,W,NE,N,NW) and in
// ASCENDING linear index (order of a cell's upstream cells in core.idxs_seq).
__device__ __constant__ const int PFD_SLOT_DESC[8] = {1, 2, 3, 0, 4, 7, 6, 5};
__device__ __constant__ const int PFD_SLOT_ASC[8] = {5, 6, 7, 4, 0, 3, 2, 1};

// true if the neighbour in slot k of (r,c) exists and drains into (r,c); *nb = its index
__device__ __forceinline__ bool d8_child(const u8 *__restrict__ ncode, const Geo &g, u32 i, u32 r, u32 c,
                                         int k, u32 *nb) {
  const int dr = d8_dr(k), dc = d8_dc(k);
  const u32 rr = r + (u32)dr, cc = c + (u32)dc;  // wraps to >= nrow/ncol when negative
  if (rr >= g.nrow || cc >= g.ncol) return false;
  const u32 j = (u32)((i64)i + (i64)dr * (i64)g.ncol + dc);
  *nb = j;
  return ncode[j] == (1u << ((k + 4) & 7));
}
// downstream index of a cell with normalised code `code` (self for pits)
__device__ __forceinline__ u32 d8_down(const Geo &g, u32 i, u32 code) {
  if (!d8_is_dir(code)) return i;
  const int k = d8_slot(code);
  return (u32)((i64)i + (i64)d8_dr(k) * (i64)g.ncol + d8_dc(k));
}

// XCD-aware order of the tiles of a 2-D grid (one workgroup per 64 x 64 tile).  The hardware hands consecutive workgroup ids
// to the 8 XCDs of an MI355X in turn, each with its own L2.  With the identity map the left / right neighbours of a tile run
// on other XCDs, and every 128-byte line of a 1-byte-per-cell raster (two tiles wide) and every halo ring is fetched from
// HBM by two or three of them: measured on the count's tile passes at 90000^2, FETCH_SIZE 13.3 -> 4.1 GB (local) and
// 16.6 -> 7.0 GB (final) raw with this map (profiles/r05_xcd_order.txt).  Workgroup id L takes tile
// (L mod 8) * (n / 8) + L div 8 of the row-major sequence: every XCD sweeps one contiguous band of tile rows.
#ifndef PFD_XCD_ORDER
#define PFD_XCD_ORDER 1
#endif
#ifdef __HIPCC__
__device__ __forceinline__ void pfd_tile_of_block(u32 *bx, u32 *by) {
  *bx = blockIdx.x, *by = blockIdx.y;
  if (PFD_XCD_ORDER) {
    const u32 gx = gridDim.x, n = gx * gridDim.y, L = blockIdx.y * gx + blockIdx.x;
    const u32 q = n >> 3, r = n & 7u, k = L & 7u;
    const u32 t = k * q + min(k, r) + (L >> 3);
    *by = t / gx;
    *bx = t - *by * gx;
  }
}
// the same for a 1-D grid over a chain-order array: XCD k takes one contiguous range of blocks.  Since the chains are laid
// out tile by tile (k_plan_tail_list<true>), blocks that follow each other touch the same tiles of the raster — with the
// identity map their scattered sectors would be fetched by eight different L2s.
__device__ __forceinline__ u32 pfd_block_1d() {
  u32 L = blockIdx.x;
  if (PFD_XCD_ORDER) {
    const u32 n = gridDim.x, q = n >> 3, r = n & 7u, k = L & 7u;
    L = k * q + min(k, r) + (L >> 3);
  }
  return L;
}
#endif

// ---------------------------------------------------------------------------------------------
// profiling segments (HIP events on the handle's stream)
// ---------------------------------------------------------------------------------------------
struct PfdSegment {
  std::string name;
  hipEvent_t e0, e1;
  i64 launches;
};

// ---------------------------------------------------------------------------------------------
// the handle
// ---------------------------------------------------------------------------------------------
struct pfd_raster {
  int device = 0;
  hipStream_t stream = nullptr;
  // a second stream + two events (pfd_aux_stream, created on first use): work that may run beside the latency-bound last
  // rounds of an exact-order up-sweep (run_exact_up)
  hipStream_t stream2 = nullptr;
  bool stream2_low = false;  // (it came from the pool of lowest-priority streams)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  i64 nrow = 0, ncol = 0, n = 0;  // device raster incl. halo rows (row blocks of a multi-GPU job)
  i64 halo_top = 0, halo_bot = 0, own_rows = 0;  // owned rows = [halo_top, halo_top + own_rows)
  Geo geo{};
  u8 *ncode = nullptr;  // device
  i64 n_valid = 0, n_pits = 0;
  // deferred handles (pfd_raster_create_deferred): the raw codes wait here until the first
  // operation normalises them — fused into the first tile pass of upstream_area(cell), or by
  // k_normalise for every other entry point
  bool normalised = true;
  const u8 *raw = nullptr;   // device; the caller's buffer (PFD_DEVICE) or raw_owned (PFD_HOST)
  u8 *raw_owned = nullptr;
  // ordering
  bool ordered = false;
  u32 *pits = nullptr;  // device, n_pits entries, ascending (compacted on first use)
  bool pits_ready = false;
  u32 *seq = nullptr;   // device, capacity n_valid, allocated by the first ordering
  u8 *seq_kids = nullptr, *seq_own = nullptr;  // per ordered cell: mask of draining neighbours / own code
  u8 *cell_kids = nullptr;                     // per CELL: mask of draining neighbours (same allocation)
  u64 *seq_kids2 = nullptr;                    // per ordered cell: the child masks of its upstream cells (owns the allocation)
  int acyclic = 0;               // 0 unknown, 1 every valid cell reaches a pit, -1 the raster holds cycles
  // plan of the exact-order engine (exact.h): 0 not built, 1 ready, -1 not available
  void *gen = nullptr;    // general idxs_ds graph (general.hip): links outside the 8 neighbours; the D8 kernels stand down
  void *xplan = nullptr;
  int xplan_state = 0;
  bool aux_ready = false;
  i64 n_seq = -1, n_levels = -1;
  std::vector<i64> lvl_off;  // host copy, n_levels+1 entries
  // small device control block (counters), 64 x u64
  u64 *ctrl = nullptr;
  size_t bytes_held = 0;
  void *pending = nullptr;  // split-phase multi-block pass in flight (dist.hip)
  void *pending_basins = nullptr;  // split-phase multi-block basins query in flight (paths.hip)
  // row blocks, exact-order up-sweeps: the given values of the halo rows (2 * ncol elements, device) are put back into
  // the result right after the tile pass has written every cell (run_exact_up); set by the caller around the sweep
  const void *xseed = nullptr;
  void *xseed_out = nullptr;
  size_t xseed_elem = 0;
  int block_update = 0;  // pfd_set_block_update: 0 = block up-sweeps keep nothing, 1 = keep the sweep for updates, 2 = update
  int block_seed_space = PFD_HOST;  // where the pfd_*_block entry points read their halo seeds (pfd_set_block_io)
  u8 *halo_raw = nullptr;  // row blocks: the D8 codes of the two halo rows as given (2 * ncol; the normalised codes hold sinks there)
  void *hand_block_state = nullptr;  // cells of a row block whose HAND is still unknown, between pfd_hand_block calls (sweeps.hip)
  // profiling
  bool profiling = false;
  bool count_rounds = false;  // pfd_set_profiling(h, 2): the tile passes also count their doubling rounds (two atomics per tile)
  i64 tile_rounds[4] = {0, 0, 0, 0};  // last tiled pass with count_rounds on: doubling rounds max / sum over tiles, local and final pass
  std::vector<PfdSegment> segs;
};

// Caching device allocator (api.hip): hipMalloc/hipFree cost milliseconds for the 0.1-30 GB
// buffers of this path, so freed blocks are kept per device and size class and handed out
// again.  Every API call synchronises its stream before it returns, hence a block is idle
// when it is released.  pfd_trim() returns the cache to the driver.
int pfd_dmalloc(void **p, size_t bytes);
void *pfd_pinned_take(size_t bytes, size_t *cap);  // pinned host staging from a small process-wide pool (api.hip); null: none
void pfd_pinned_give(void *p, size_t cap);
const char *pfd_knob(const char *name);  // test-only switch: getenv(name) if PFD_ENABLE_KNOBS=1, else null (api.hip)
void pfd_dfree(void *p);

// scoped device selection + temp buffers ---------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  int rc = PFD_OK;
  DevBuf() {}
  ~DevBuf() {
    if (p) pfd_dfree(p);
  }
  int alloc(size_t bytes) {
    if (p) {
      pfd_dfree(p);
      p = nullptr;
    }
    return pfd_dmalloc(&p, bytes);
  }
  template <class T>
  T *as() { return (T *)p; }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};

// Host <-> device traffic of the calling thread's API calls (pfd_transfer_stats, api.hip).
struct PfdTransfer {
  double h2d_bytes = 0, h2d_ms = 0, d2h_bytes = 0, d2h_ms = 0, prefault_ms = 0, host_results = 0;
};
PfdTransfer &pfd_transfer();
double pfd_now_ms();
// Pre-faulting of a host result buffer while the kernels run (api.hip): a copy into fresh pages is bound by first-touch
// page faults (~15 GB/s on the test host, 50 GB/s into touched pages; tools/probes/pcie_probe.cpp, thp_probe.cpp).
void *pfd_prefault_begin(void *dst, size_t bytes);  // null: nothing started (small buffer)
double pfd_prefault_join(void *state);              // ms the touching took (0 for null)

// An input that may live on the host (staged through a temp buffer) or on the device (used as is).
struct InArg {
  DevBuf tmp;
  const void *dev = nullptr;
  int bind(const void *src, size_t bytes, int memspace, hipStream_t s) {
    if (src == nullptr) {
      dev = nullptr;
      return PFD_OK;
    }
    if (memspace == PFD_DEVICE) {
      dev = src;
      return PFD_OK;
    }
    PFDCHK(tmp.alloc(bytes));
    const double t0 = pfd_now_ms();
    HIPCHK(hipMemcpyAsync(tmp.p, src, bytes, hipMemcpyHostToDevice, s));
    if (bytes >= (1u << 20)) HIPCHK(hipStreamSynchronize(s));  // (a pageable upload has all but finished when the call returns)
    PfdTransfer &t = pfd_transfer();
    t.h2d_bytes += (double)bytes, t.h2d_ms += pfd_now_ms() - t0;
    dev = tmp.p;
    return PFD_OK;
  }
};
// An output that is produced in HBM and, for host callers, copied back at the end.  The pages of a large host result are
// touched by a few host threads while the kernels run (pfd_prefault_begin).
struct OutArg {
  DevBuf tmp;
  void *dev = nullptr;
  void *host = nullptr;
  size_t bytes = 0;
  void *prefault = nullptr;
  ~OutArg() { (void)pfd_prefault_join(prefault); }
  int bind(void *dst, size_t nbytes, int memspace) {
    bytes = nbytes;
    if (memspace == PFD_DEVICE) {
      dev = dst;
      return PFD_OK;
    }
    host = dst;
    PFDCHK(tmp.alloc(nbytes));
    dev = tmp.p;
    prefault = pfd_prefault_begin(dst, nbytes);
    return PFD_OK;
  }
  int finish(hipStream_t s) {
    if (host) {
      HIPCHK(hipStreamSynchronize(s));  // (the kernels are done: what follows is the download alone)
      PfdTransfer &t = pfd_transfer();
      t.prefault_ms += pfd_prefault_join(prefault);
      prefault = nullptr;
      const double t0 = pfd_now_ms();
      HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      t.d2h_bytes += (double)bytes, t.d2h_ms += pfd_now_ms() - t0, t.host_results += 1;
      return PFD_OK;
    }
    HIPCHK(hipStreamSynchronize(s));
    return PFD_OK;
  }
};

// internal entry points shared between translation units -----------------------------------------
int pfd_check_handle(pfd_raster *h);       // + normalises a deferred handle
int pfd_check_handle_lazy(pfd_raster *h);  // leaves a deferred handle raw (tiled count path)
int pfd_ensure_normalised(pfd_raster *h);
int pfd_adopt_counts(pfd_raster *h, const u64 *c);  // c = ctrl[0..48) after a normalising pass
void pfd_seg_clear(pfd_raster *h);
void pfd_seg_begin(pfd_raster *h, const char *name);
void pfd_seg_end(pfd_raster *h, i64 launches);
int pfd_normalise_and_count(pfd_raster *h, const u8 *d8_dev);  // order.hip
int pfd_order_cells_impl(pfd_raster *h, bool allow_block = false);                     // order.hip
int pfd_ensure_pits(pfd_raster *h);                             // order.hip
int pfd_exact_seq_dev(pfd_raster *h, DevBuf &oseq);              // order.hip: core.idxs_seq order in HBM
int pfd_basins_dev(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 k, int id_size, void *out_dev);  // sweeps.hip
int pfd_require_whole(pfd_raster *h, const char *what);          // order.hip: no row block, 32-bit cell indices
int pfd_require_unblocked(pfd_raster *h, const char *what);      // order.hip: no row block
void pfd_free_pending(pfd_raster *h);                            // dist.hip
void pfd_free_pending_basins(pfd_raster *h);                     // paths.hip
void pfd_free_hand_block(pfd_raster *h);                         // sweeps.hip
int pfd_order_cells_by_rank(pfd_raster *h, int *ok);            // paths.hip
bool pfd_wide_cells(const pfd_raster *h);                       // order64.hip: 64-bit cell indices (beyond 2^32 - 2 cells)
int pfd_rank_wide(pfd_raster *h, i32 *out, int memspace);        // order64.hip
int pfd_idxs_seq_wide(pfd_raster *h, int idx_dtype, void *out, int memspace);  // order64.hip
int pfd_wide_seq_dev(pfd_raster *h, struct DevBuf &q, u64 *nseq);  // order64.hip: the 64-bit sequence on the device
int pfd_aux_stream(pfd_raster *h);                                // api.hip: h->stream2 / ev_fork / ev_join exist afterwards
void pfd_free_xplan(pfd_raster *h);                             // exact.hip
void pfd_xinc_drop(pfd_raster *h);                               // exact.hip: releases a kept block sweep
void pfd_free_general(pfd_raster *h);                           // general.hip
int pfd_handle_alloc(i64 nrow, i64 ncol, int device, pfd_raster **out);  // api.hip: empty handle (stream, code raster, ctrl)
// general.hip: the entry points of a general idxs_ds graph (h->gen != nullptr)
int pfd_gen_idxs_ds(pfd_raster *h, int idx_dtype, void *out, int memspace);
int pfd_gen_order(pfd_raster *h);
int pfd_gen_idxs_seq(pfd_raster *h, int idx_dtype, void *out, int memspace);
int pfd_gen_rank(pfd_raster *h, i32 *out, int memspace);
int pfd_gen_upstream_count(pfd_raster *h, const u8 *mask, int8_t *out, int memspace);
int pfd_gen_accuflux(pfd_raster *h, int dtype, const void *data, bool by_row, int64_t nodata_i, double nodata_f,
                     int has_nodata, int direction, int mask_invalid, void *out, int memspace);
int pfd_gen_upstream_area_cell(pfd_raster *h, i32 *out, int memspace);
int pfd_gen_strahler(pfd_raster *h, const u8 *mask, u8 *out, int memspace);
int pfd_gen_basins(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 k, int id_size, void *out_dev);
int pfd_gen_hand(pfd_raster *h, const u8 *drain, int elev_dtype, const void *elevtn, double *out, int memspace);
int pfd_gen_stream_distance(pfd_raster *h, const u8 *mask, int real_length, const float *step_lengths, void *out, int memspace);
int pfd_gen_main_upstream(pfd_raster *h, int dtype, const void *uparea, double upa_min, int idx_dtype, void *out, int memspace);
int pfd_gen_classic(pfd_raster *h, int idx_dtype, const void *idxs_us_main, const u8 *mask, u8 *out, int memspace);
int pfd_gen_add_pits(pfd_raster *h, const i64 *idxs, i64 k);
int pfd_reject_general(pfd_raster *h, const char *what);        // api.hip: PFD_EUNSUPPORTED on a general graph
int pfd_basins_tiled(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 k, int id_size, void *out_dev,
                     int *ok);                                   // paths.hip
int pfd_upstream_area_cell_tiled(pfd_raster *h, i32 *out_dev, int *complete, const i32 *weights = nullptr);  // tiled.hip

static inline u32 cdiv_u32(u64 a, u32 b) { return (u32)((a + b - 1) / b); }
