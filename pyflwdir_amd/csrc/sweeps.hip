// sweeps.hip — level-by-level sweeps over the ordered cells, all in PULL form.
//
// The reference walks `seq` serially (up- to downstream for accumulations, down- to upstream
// for labels/HAND) and read-modify-writes the downstream cell.  Here every cell of one level
// (= one rank) is final before the next level is launched, and each thread computes the value
// of ITS OWN cell from its upstream cells (decoded from the 8 neighbour codes) or from its
// downstream cell.  No atomics, no write conflicts, and the children are combined in the
// exact order of the serial loop (descending linear index), so float32/float64 results are
// bit-identical to the reference, not merely within tolerance.
//
// Reference functions replaced:
//   streams.accuflux          pyflwdir/streams.py:15-41    AccuUp
//   streams.accuflux_ds       pyflwdir/streams.py:44-70    AccuDown
//   streams.strahler_order    pyflwdir/streams.py:228-269  Strahler
//   core.fillnodata_upstream  pyflwdir/core.py:120-146     Labels (basins.basins, basins.py:12-18)
//   dem.height_above_nearest_drain  pyflwdir/dem.py:299-330  Hand
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <unordered_set>

#include "common.h"
#include "exact.h"

// Every ordered cell carries one byte of pre-decoded graph next to its index (built once per
// ordering by k_seq_aux): the mask of the neighbour slots that drain into it (up-sweeps) and its
// own normalised code (down-sweeps).  A level kernel is then two dependent global round trips
// (seq + aux -> neighbour values) instead of three (seq -> 8 neighbour codes -> values), and needs
// neither bounds checks nor the row/column of the cell.
//
// MULTI-HOP LAUNCHES.  On river-like rasters a level holds ~one raster row of cells and the sweep
// is bound by launches x (kernel boundary + host launch cost, ~3.6 us), not by bandwidth.  The
// value of a cell depends only on FINAL values a few hops away, so one launch can cover several
// consecutive levels if a thread recomputes the not-yet-final values on its own dependency path:
//   up-sweeps   2 (optionally 3) levels per launch: a cell of a lower level evaluates each of its upstream
//               cells from THEIR upstream cells (recursively, down to final values), in the same
//               order and with the same arithmetic as the thread that owns that cell —
//               bit-identical, ~2x the work on a network with ~1 upstream cell per cell;
//   down-sweeps up to 16 levels per launch: a cell climbs to the ancestor whose downstream cell is
//               final and applies the per-cell update back down the chain (no fan-out at all).
// Wide levels (rough rasters: few levels, bandwidth-bound) are launched one by one as before.
__device__ __forceinline__ u32 kids_of(const u8 *__restrict__ ncode, const Geo &g, u32 x) {
  const u32 r = geo_row(g, x), c = x - r * g.ncol;
  u32 m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    u32 nb;
    if (d8_child(ncode, g, x, r, c, k, &nb)) m |= 1u << k;
  }
  return m;
}
// kids2: byte k = child mask of the upstream cell in slot k (0 if none) — a 2-hop launch then needs
// no lookup between the cell and its grandchildren
// (the per-cell masks are built first — k_cell_kids — and only gathered here: 8 neighbour tests per
//  cell instead of 72)
__global__ void __launch_bounds__(256) k_seq_aux(const u8 *__restrict__ ncode, Geo g, const u32 *__restrict__ seq,
                                                 u32 nseq, const u8 *__restrict__ cell_kids, u8 *__restrict__ kids,
                                                 u8 *__restrict__ own, u64 *__restrict__ kids2) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nseq) return;
  const u32 x = seq[j];
  const u32 m = cell_kids[x];
  u64 m2 = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // unconditional loads from clamped addresses, masked afterwards
    const i64 nb = (i64)x + (i64)d8_dr(k) * (i64)g.ncol + d8_dc(k);
    const u64 c = cell_kids[nb < 0 ? 0 : (nb >= (i64)g.n ? (i64)g.n - 1 : nb)];
    if (m & (1u << k)) m2 |= c << (8 * k);
  }
  kids[j] = (u8)m;
  own[j] = ncode[x];
  if (kids2) kids2[j] = m2;
}
// the same child mask per CELL (2-hop up-sweeps look up the children of a child)
__global__ void __launch_bounds__(256) k_cell_kids(const u8 *__restrict__ ncode, Geo g, u8 *__restrict__ kids) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.n) return;
  kids[x] = (u8)kids_of(ncode, g, x);
}

// row blocks: the cells of the halo rows that drain INTO the block.  The normalised codes hold sinks there, so the
// masks above miss them; the codes of the halo rows as given (halo_raw: top row, bottom row) tell.  One thread per
// cell of the first / last own row.
__global__ void __launch_bounds__(256) k_halo_kids(const u8 *__restrict__ ncode, const u8 *__restrict__ halo_raw, Geo g,
                                                   u32 row_first, u32 row_last, u8 *__restrict__ kids) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * g.ncol) return;
  const u32 side = t / g.ncol, c = t - side * g.ncol;
  if ((side == 0 && row_first == 0) || (side == 1 && row_last + 1 >= g.nrow)) return;  // no halo row on this side
  const u32 x = (side ? row_last : row_first) * g.ncol + c;
  if (ncode[x] == D8_MV) return;
  u32 m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (d8_dr(k) != (side ? 1 : -1)) continue;
    const u32 cc = c + (u32)d8_dc(k);  // wraps to >= ncol when negative
    if (cc < g.ncol && halo_raw[side * g.ncol + cc] == (1u << ((k + 4) & 7))) m |= 1u << k;
  }
  if (m) kids[x] |= (u8)m;
}

int pfd_ensure_seq_aux(pfd_raster *h) {
  if (h->aux_ready) return PFD_OK;
  if (h->seq_kids2) pfd_dfree(h->seq_kids2);
  h->seq_kids2 = nullptr;
  h->seq_kids = nullptr;
  // layout: kids2 u64[n_seq] | kids u8[n_seq] | own u8[n_seq] | cell_kids u8[n]
  // (row blocks sweep one level per launch and never read kids2: 8 of the 11 bytes per cell stay unallocated)
  const bool block = h->halo_top || h->halo_bot;
  const size_t ns = (size_t)std::max<i64>(h->n_seq, 1), ns2 = block ? 1 : ns;
  PFDCHK(pfd_dmalloc((void **)&h->seq_kids2, ns2 * 8 + ns * 2 + (size_t)h->n));
  h->seq_kids = (u8 *)(h->seq_kids2 + ns2);
  h->seq_own = h->seq_kids + ns;
  h->cell_kids = h->seq_own + ns;
  k_cell_kids<<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo, h->cell_kids);
  KCHK();
  if (block && h->halo_raw) {
    k_halo_kids<<<cdiv_u32(2 * (u64)h->ncol, 256), 256, 0, h->stream>>>(h->ncode, h->halo_raw, h->geo, (u32)h->halo_top,
                                                                      (u32)(h->halo_top + h->own_rows - 1), h->cell_kids);
    KCHK();
  }
  if (h->n_seq) {
    k_seq_aux<<<cdiv_u32((u64)h->n_seq, 256), 256, 0, h->stream>>>(h->ncode, h->geo, h->seq, (u32)h->n_seq, h->cell_kids,
                                                                   h->seq_kids, h->seq_own, block ? nullptr : h->seq_kids2);
    KCHK();
  }
  h->aux_ready = true;
  return PFD_OK;
}

// neighbour slot k of cell x (valid by construction of the mask / code)
__device__ __forceinline__ u32 nb_of(const Geo &g, u32 x, int k) {
  return (u32)((i64)x + (i64)d8_dr(k) * (i64)g.ncol + d8_dc(k));
}

// 2-hop evaluation from REGISTERS.  A load inside a data-dependent branch is waited for on the spot, so
// the nested "for each child / for each grandchild" loops serialise up to children x grandchildren
// global round trips (~6 us per launch).  The grandchildren of x lie in the 5x5 window around it:
// load the whole window unconditionally (clamped indices; unused values are never looked at), then
// evaluate from registers.  With unrolled loops all window indices are compile-time constants.
template <class A>
__device__ __forceinline__ void load_window5(const A *__restrict__ arr, const Geo &g, u32 x, A (&w)[25]) {
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    const i64 j = (i64)x + (i64)(i / 5 - 2) * (i64)g.ncol + (i % 5 - 2);
    w[i] = arr[j < 0 ? 0 : (j >= (i64)g.n ? (i64)g.n - 1 : j)];
  }
}
__device__ __forceinline__ constexpr int slot_dr(int k) { return k == 1 || k == 2 || k == 3 ? 1 : (k == 5 || k == 6 || k == 7 ? -1 : 0); }
__device__ __forceinline__ constexpr int slot_dc(int k) { return k == 0 || k == 1 || k == 7 ? 1 : (k == 3 || k == 4 || k == 5 ? -1 : 0); }
// window index of the cell reached from x by slot k, then slot k2 (k2 < 0: the child itself)
__device__ __forceinline__ constexpr int win_idx(int k, int k2) {
  return (2 + slot_dr(k) + (k2 < 0 ? 0 : slot_dr(k2))) * 5 + 2 + slot_dc(k) + (k2 < 0 ? 0 : slot_dc(k2));
}
__device__ __forceinline__ constexpr int slot_desc(int q) {  // slots in descending linear index of the neighbour
  return q == 0 ? 1 : q == 1 ? 2 : q == 2 ? 3 : q == 3 ? 0 : q == 4 ? 4 : q == 5 ? 7 : q == 6 ? 6 : 5;
}

// launches covering several levels are only used while they stay small (latency-bound regime)
static const u32 MULTIHOP_MAX_CELLS = 1u << 20;     // down-sweeps (a chain of single loads per thread)
static const u32 MULTIHOP_MAX_CELLS_UP = 1u << 18;  // up-sweeps (the window form loads ~50 values per thread)

// ---- up-sweeps --------------------------------------------------------------------------------
// Op: V leaf(nb) = final value of an upstream cell; V combine(x, kids, child) = value of x given
// child(nb) for its upstream cells (called in the order the op needs); store(x, v).
// seq positions [begin, s1) = lowest level (3 hops), [s1, s2) = middle level (2 hops), [s2, end) =
// upper level (children final); s1 = begin / s2 = s1 when fewer levels are taken.
// one level per launch (wide levels: bandwidth-bound, kept lean — the multi-hop kernel's register
// footprint would cost occupancy)
template <class Op>
__global__ void __launch_bounds__(256) k_sweep_up1(Op op, const u32 *__restrict__ seq, const u8 *__restrict__ kids_seq,
                                                   u32 begin, u32 end) {
  const u32 pos = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= end) return;
  const u32 x = seq[pos];
  op.store(x, op.combine(x, (u32)kids_seq[pos], [&](u32 nb, int) { return op.leaf(nb); }));
}
template <class Op>
__global__ void __launch_bounds__(256) k_sweep_up(Op op, const u32 *__restrict__ seq, const u8 *__restrict__ kids_seq,
                                                  const u64 *__restrict__ kids2_seq, const u8 *__restrict__ kids_cell,
                                                  u32 begin, u32 s1, u32 s2, u32 end) {
  const u32 pos = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= end) return;
  const u32 x = seq[pos];
  const u32 kids = kids_seq[pos];
  auto leaf = [&](u32 nb, int) { return op.leaf(nb); };
  if (pos >= s2) {
    op.store(x, op.combine(x, kids, leaf));
    return;
  }
  const u64 kids2 = kids2_seq[pos];  // child masks of the children: no lookup on the way to the grandchildren
  if (pos >= s1) {
    op.store(x, op.eval2(x, kids, kids2));
  } else {
    auto hop2 = [&](u32 nb, int) { return op.combine(nb, (u32)kids_cell[nb], leaf); };
    op.store(x, op.combine(x, kids, [&](u32 nb, int k) { return op.combine(nb, (u32)(kids2 >> (8 * k)) & 0xFFu, hop2); }));
  }
}

// up- to downstream: deepest level first (children final before their parent's level runs)
template <class Op>
static int run_up(pfd_raster *h, const Op &op, const char *name) {
  PFDCHK(pfd_ensure_seq_aux(h));
  pfd_seg_begin(h, name);
  i64 launches = 0;
  // (3 levels per launch are implemented but measured slower than 2: 35 vs 28 ms for 10003 levels with
  //  nested lookups — the third hop squares the divergent fan-out — and 58 vs 25 ms with a 7x7 register
  //  window, which spills: 100 loads per thread are too many registers)
  int maxk = pfd_knob("PFD_SINGLE_HOP") ? 1 : 2;
  if (const char *e = pfd_knob("PFD_UP_K")) maxk = std::max(1, std::min(3, atoi(e)));
  if (h->halo_top || h->halo_bot) maxk = 1;  // (a row block: the value of a halo cell is given, never recomputed)
  for (i64 l = h->n_levels - 1; l >= 0;) {
    const u32 end = (u32)h->lvl_off[l + 1];
    int k = 1;  // levels l, l-1, .. l-k+1
    while (k < maxk && l - k >= 0 && end - (u32)h->lvl_off[l - k] <= MULTIHOP_MAX_CELLS_UP) ++k;
    const u32 begin = (u32)h->lvl_off[l - k + 1];
    const u32 s2 = (u32)h->lvl_off[l];                      // start of the upper level
    const u32 s1 = k == 3 ? (u32)h->lvl_off[l - 1] : begin;  // start of the middle level
    if (end > begin) {
      if (k == 1)
        k_sweep_up1<Op><<<cdiv_u32(end - begin, 256), 256, 0, h->stream>>>(op, h->seq, h->seq_kids, begin, end);
      else
        k_sweep_up<Op><<<cdiv_u32(end - begin, 256), 256, 0, h->stream>>>(op, h->seq, h->seq_kids, h->seq_kids2,
                                                                          h->cell_kids, begin, s1, s2, end);
      ++launches;
    }
    l -= k;
  }
  KCHK();
  pfd_seg_end(h, launches);
  return PFD_OK;
}

// ---- down-sweeps ------------------------------------------------------------------------------
// Op: V top(p) = final value of a cell below the launch's levels; V apply(x, code, root, pv) =
// value of x from the value pv of its downstream cell (root: x is a pit, pv unused); store(x, v).
// The launch covers seq positions [begin, end) = up to DOWN_K consecutive levels; off.o[i] = start of
// the (i+1)-th of them (unused ones = end); `roots`: the first level is level 0 (the pits).
enum { DOWN_K = 16 };
struct DownOffsets {
  u32 o[DOWN_K - 1];
};
// one level per launch (wide levels: bandwidth-bound, kept lean)
template <class Op>
__global__ void __launch_bounds__(256) k_sweep_down1(Op op, const u32 *__restrict__ seq, const u8 *__restrict__ own_seq,
                                                     Geo g, u32 begin, u32 end, int roots) {
  const u32 pos = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= end) return;
  const u32 x = seq[pos], code = own_seq[pos];
  if (roots)
    op.store(x, op.apply(x, code, true, typename Op::V()));
  else
    op.store(x, op.apply(x, code, false, op.top(d8_down(g, x, code))));
}
template <class Op>
__global__ void __launch_bounds__(256) k_sweep_down(Op op, const u32 *__restrict__ seq, const u8 *__restrict__ own_seq,
                                                    const u8 *__restrict__ ncode, Geo g, u32 begin, DownOffsets off,
                                                    u32 end, int roots) {
  const u32 pos = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= end) return;
  int hops = 0;  // ancestors inside the launch
#pragma unroll
  for (int i = 0; i < DOWN_K - 1; ++i) hops += (int)(pos >= off.o[i]);
  u32 cx[DOWN_K], cc[DOWN_K];
  cx[0] = seq[pos];
  cc[0] = own_seq[pos];
  u32 tx = cx[0], tc = cc[0];  // top of the chain
#pragma unroll
  for (int i = 1; i < DOWN_K; ++i) {
    if (i <= hops) {
      tx = d8_down(g, tx, tc);
      tc = ncode[tx];
      cx[i] = tx;
      cc[i] = tc;
    }
  }
  typename Op::V v;
  if (roots)
    v = op.apply(tx, tc, true, typename Op::V());
  else
    v = op.apply(tx, tc, false, op.top(d8_down(g, tx, tc)));
#pragma unroll
  for (int i = DOWN_K - 2; i >= 0; --i)
    if (i < hops) v = op.apply(cx[i], cc[i], false, v);
  op.store(cx[0], v);
}

// down- to upstream: level 0 (the pits) first
template <class Op>
static int run_down(pfd_raster *h, const Op &op, const char *name) {
  PFDCHK(pfd_ensure_seq_aux(h));
  pfd_seg_begin(h, name);
  i64 launches = 0;
  int maxk = pfd_knob("PFD_SINGLE_HOP") ? 1 : DOWN_K;
  if (const char *e = pfd_knob("PFD_DOWN_K")) maxk = std::max(1, std::min((int)DOWN_K, atoi(e)));
  for (i64 l = 0; l < h->n_levels;) {
    const u32 begin = (u32)h->lvl_off[l];
    int k = 1;
    while (k < maxk && l + k < h->n_levels && (u32)h->lvl_off[l + k + 1] - begin <= MULTIHOP_MAX_CELLS) ++k;
    const u32 end = (u32)h->lvl_off[l + k];
    DownOffsets off;
    for (int i = 0; i < DOWN_K - 1; ++i) off.o[i] = i + 1 < k ? (u32)h->lvl_off[l + i + 1] : end;
    if (end > begin) {
      if (k == 1)
        k_sweep_down1<Op><<<cdiv_u32(end - begin, 256), 256, 0, h->stream>>>(op, h->seq, h->seq_own, h->geo, begin, end,
                                                                             l == 0 ? 1 : 0);
      else
        k_sweep_down<Op><<<cdiv_u32(end - begin, 256), 256, 0, h->stream>>>(op, h->seq, h->seq_own, h->ncode, h->geo,
                                                                            begin, off, end, l == 0 ? 1 : 0);
      ++launches;
    }
    l += k;
  }
  KCHK();
  pfd_seg_end(h, launches);
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// payload arithmetic: integers wrap like numba's fixed-width ints, floats are plain IEEE adds
// ---------------------------------------------------------------------------------------------
template <class T> struct Num;
// neutral(): x with add(x, y) == y bit for bit, for every y (-0.0 for floats: +0.0 would turn a -0.0 into +0.0)
template <> struct Num<i32> {
  static __device__ __forceinline__ i32 neutral() { return 0; }
  static __device__ __forceinline__ i32 add(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
};
template <> struct Num<i64> {
  static __device__ __forceinline__ i64 neutral() { return 0; }
  static __device__ __forceinline__ i64 add(i64 a, i64 b) { return (i64)((u64)a + (u64)b); }
};
template <> struct Num<float> {
  static __device__ __forceinline__ float neutral() { return -0.0f; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
};
template <> struct Num<double> {
  static __device__ __forceinline__ double neutral() { return -0.0; }
  static __device__ __forceinline__ double add(double a, double b) { return a + b; }
};

// payload accessors: a full per-cell array, or one value per raster ROW (cell areas of a regular
// grid depend on the row only — upstream_area(unit != "cell") then needs no n-element input at all)
template <class T>
struct CellData {
  const T *p;
  Geo g;
  __device__ __forceinline__ T at(u32 x) const { return p[x]; }
  __device__ __forceinline__ void load4(u32 x0, T (&v)[4]) const { __builtin_memcpy(v, p + x0, 4 * sizeof(T)); }
};
template <class T>
struct RowData {
  const T *row;
  Geo g;
  __device__ __forceinline__ T at(u32 x) const { return row[geo_row(g, x)]; }
  __device__ __forceinline__ void load4(u32 x0, T (&v)[4]) const {  // (a quad never crosses a row)
    const T r = row[geo_row(g, x0)];
    v[0] = v[1] = v[2] = v[3] = r;
  }
};
template <class D, class A>
__device__ __forceinline__ void load_window5_of(const D &d, const Geo &g, u32 x, A (&w)[25]) {
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    const i64 j = (i64)x + (i64)(i / 5 - 2) * (i64)g.ncol + (i % 5 - 2);
    w[i] = d.at((u32)(j < 0 ? 0 : (j >= (i64)g.n ? (i64)g.n - 1 : j)));
  }
}

// MASKINV (exact-order engine only): the tile pass, which writes every cell of the raster, leaves `nodata` on the raster's
// nodata cells instead of their payload — FlwdirRaster.upstream_area's `uparea[~mask] = -9999` (pyflwdir.py:800) without a
// pass of its own over the result (k_mask_invalid: 0.95 ms of 17 at 30000^2 for float64)
template <class T, class D = CellData<T>, bool MASKINV = false>
struct AccuUp {
  typedef T V;
  const u8 *ncode;
  Geo g;
  D data;
  T *out;
  T nodata;
  int has_nodata;
  __device__ __forceinline__ T leaf(u32 nb) const { return out[nb]; }
  template <class F>
  __device__ __forceinline__ T combine(u32 x, u32 kids, F child) const {
    T acc = data.at(x);
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // children in descending linear index: the serial loop's order
      const int k = PFD_SLOT_DESC[q];
      if (kids & (1u << k)) {
        const T a = child(nb_of(g, x, k), k);
        if (!has_nodata || (acc != nodata && a != nodata)) acc = Num<T>::add(acc, a);
      }
    }
    return acc;
  }
  // value of x from its grandchildren's final values (window form, see load_window5)
  __device__ __forceinline__ T eval2(u32 x, u32 kids, u64 kids2) const {
    T W[25], Dw[25];
    load_window5(out, g, x, W);
    load_window5_of(data, g, x, Dw);
    T acc = Dw[12];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = slot_desc(q);
      if (kids & (1u << k)) {
        const u32 kk = (u32)(kids2 >> (8 * k)) & 0xFFu;
        T a = Dw[win_idx(k, -1)];
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) {
          const int k2 = slot_desc(q2);
          if (kk & (1u << k2)) {
            const T b = W[win_idx(k, k2)];
            if (!has_nodata || (a != nodata && b != nodata)) a = Num<T>::add(a, b);
          }
        }
        if (!has_nodata || (acc != nodata && a != nodata)) acc = Num<T>::add(acc, a);
      }
    }
    return acc;
  }
  __device__ __forceinline__ void store(u32 x, T v) const { out[x] = v; }
  // ---- exact-order engine (exact_sweep.h) ----
  typedef T LV;
  typedef T Elem;
  __device__ __forceinline__ T join(T acc, T a) const {  // (branch-free: runs on the serial critical path)
    const T sum = Num<T>::add(acc, a);
    const bool ok = !has_nodata || (acc != nodata && a != nodata);
    return ok ? sum : acc;
  }
  __device__ __forceinline__ T tile_init(u32 x, bool nd) const { return (MASKINV && nd) ? nodata : data.at(x); }
  __device__ __forceinline__ T tile_combine(u32 l, u32 kids, const T *val) const {
    T acc = val[l];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = slot_desc(q);
      if (kids & (1u << k)) acc = join(acc, val[(int)l + slot_dr(k) * XT + slot_dc(k)]);
    }
    return acc;
  }
  __device__ __forceinline__ void tile_store(u32 x, T v) const { out[x] = v; }
  static constexpr bool NEEDS_NODATA = MASKINV;
  __device__ __forceinline__ void tile_init4(u32 x0, u32 nd, T (&v)[4]) const {
    data.load4(x0, v);
    if (MASKINV) {
#pragma unroll
      for (int b = 0; b < 4; ++b) v[b] = (nd & (1u << b)) ? nodata : v[b];
    }
  }
  __device__ __forceinline__ void tile_store4(u32 x0, const T (&v)[4]) const { __builtin_memcpy(out + x0, v, 4 * sizeof(T)); }
  // own payload + the light upstream cells that precede the heavy one (slot hs) in the serial loop's order
  __device__ __forceinline__ T pre_real(u32 x, u32 kids, u32 hs) const {
    u32 m = 0;
    bool before = true;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = slot_desc(q);
      if ((u32)k == hs) before = false;
      else if (before && (kids & (1u << k))) m |= 1u << k;
    }
    T v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // all needed loads in flight together
      const int k = slot_desc(q);
      v[q] = (m & (1u << k)) ? out[nb_of(g, x, k)] : T();
    }
    T acc = data.at(x);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (m & (1u << slot_desc(q))) acc = join(acc, v[q]);
    return acc;
  }
  __device__ __forceinline__ T pre_post(u32 child) const { return out[child]; }
  __device__ __forceinline__ T first(T e) const { return e; }
  // real slot: accumulator = the cell's own part, operand = the heavy upstream cell's value (the running
  // value); post slot: accumulator = the running value, operand = the light upstream cell
  __device__ __forceinline__ T fold(T t, T e, bool post) const { return join(post ? t : e, post ? e : t); }
  // speculative form (exact_sweep.h): a plain add is the exact result whenever neither operand is the nodata value
  static constexpr bool FAST = true;
  static constexpr bool FAST_CONST = false;  // (fold_fast leaves the running value unchanged whatever the element)
  // gather + fold of the short chains in one kernel (k_xtrunk_prescan): the gather is the bound, the fold hides under it
  static constexpr bool FUSE_UP = true;
  static constexpr bool FAST_SHORT = true;  // (the lane-per-chain fold of the short chains speculates too)
  __device__ __forceinline__ bool special(T t, T e) const { return has_nodata && ((t == nodata) | (e == nodata)); }
  __device__ __forceinline__ T fold_fast(T t, T e) const { return Num<T>::add(e, t); }
};

template <class T, class D = CellData<T>>
struct AccuDown {
  typedef T V;
  const u8 *ncode;
  Geo g;
  D data;
  T *out;
  T nodata;
  int has_nodata;
  __device__ __forceinline__ T top(u32 p) const { return out[p]; }
  __device__ __forceinline__ T apply(u32 x, u32, bool root, T pv) const {
    T a = data.at(x);  // a pit keeps its own value
    if (!root && (!has_nodata || (pv != nodata && a != nodata))) a = Num<T>::add(a, pv);
    return a;
  }
  __device__ __forceinline__ void store(u32 x, T v) const { out[x] = v; }
  // ---- exact-order engine ----
  typedef T DElem;
  __device__ __forceinline__ T dnodata(u32 x) const { return data.at(x); }  // nodata cells keep their input value
  __device__ __forceinline__ void dstore4(u32 x0, const T (&v)[4]) const { __builtin_memcpy(out + x0, v, 4 * sizeof(T)); }
  __device__ __forceinline__ T dpre(u32 x, u32) const { return data.at(x); }
  __device__ __forceinline__ T droot(T e) const { return e; }
  __device__ __forceinline__ T dfold(T e, T pv) const {
    if (!has_nodata || (pv != nodata && e != nodata)) e = Num<T>::add(e, pv);
    return e;
  }
  // tile image of the element (exact_sweep.h, k_xtile_down): the element itself, no flag
  typedef DElem DTile;
  static constexpr bool DTILE_FLAG = false;
  static constexpr bool DTILE4 = true;  // (quad form, loads only: see Hand)
  static constexpr bool DSCAN_LDS = false;
  struct DQuad {
    T d[4];
  };
  __device__ __forceinline__ void dtile4_load(u32 x0, u32, DQuad &q) const { data.load4(x0, q.d); }
  __device__ __forceinline__ T dtile4_get(const DQuad &q, int b, bool &) const { return q.d[b]; }
  __device__ __forceinline__ DElem dtile(u32 x, u32 code, bool &) const { return dpre(x, code); }
  __device__ __forceinline__ T dtroot(DElem e, bool) const { return droot(e); }
  __device__ __forceinline__ T dtfold(DElem e, bool, T pv) const { return dfold(e, pv); }
  __device__ __forceinline__ void top4(u32 x0, T (&v)[4]) const {
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = top(x0 + b);
  }
  static constexpr bool FAST = true;
  static constexpr bool FAST_CONST = false;  // (fold_fast leaves the running value unchanged whatever the element)
  __device__ __forceinline__ bool dspecial(T e, T pv) const { return has_nodata && ((pv == nodata) | (e == nodata)); }
  __device__ __forceinline__ T dfold_fast(T e, T pv) const { return Num<T>::add(e, pv); }
  __device__ __forceinline__ T dneutral() const { return Num<T>::neutral(); }  // dfold_fast(dneutral(), pv) == pv
};

// FlwdirRaster.upstream_area(unit="cell"): unit weights, nothing read but the codes
struct CountUp {
  typedef u32 V;
  const u8 *ncode;
  Geo g;
  u32 *out;
  __device__ __forceinline__ u32 leaf(u32 nb) const { return out[nb]; }
  template <class F>
  __device__ __forceinline__ u32 combine(u32 x, u32 kids, F child) const {
    u32 acc = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (kids & (1u << k)) acc += child(nb_of(g, x, k), k);
    return acc;
  }
  __device__ __forceinline__ u32 eval2(u32 x, u32 kids, u64 kids2) const {
    u32 W[25];
    load_window5(out, g, x, W);
    u32 acc = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (kids & (1u << k)) {
        const u32 kk = (u32)(kids2 >> (8 * k)) & 0xFFu;
        u32 a = 1;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2)
          if (kk & (1u << k2)) a += W[win_idx(k, k2)];
        acc += a;
      }
    }
    return acc;
  }
  __device__ __forceinline__ void store(u32 x, u32 v) const { out[x] = v; }
};

// Strahler order, order-independent closed form of the reference's two-array update
// (pyflwdir/streams.py:252-268): among the upstream cells that are inside the mask, let m be
// the largest order; the cell gets m+1 if at least two of them have order m, else m; with no
// such upstream cell it is a headwater: 1 if the cell itself is inside the mask, else 0.
struct Strahler {
  typedef u32 V;
  const u8 *ncode;
  Geo g;
  const u8 *mask;  // may be null
  u8 *out;
  __device__ __forceinline__ u32 leaf(u32 nb) const { return out[nb]; }
  template <class F>
  __device__ __forceinline__ u32 combine(u32 x, u32 kids, F child) const {
    u32 m = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 nb = nb_of(g, x, k);
      if ((kids & (1u << k)) && (mask == nullptr || mask[nb])) {
        const u32 v = child(nb, k) & 0xFFu;  // uint8 like the stored values
        if (v > m) {
          m = v;
          cnt = 1;
        } else if (v == m) {
          ++cnt;
        }
      }
    }
    if (cnt == 0) return (mask == nullptr || mask[x]) ? 1u : 0u;
    return cnt >= 2 ? m + 1 : m;
  }
  static __device__ __forceinline__ void join(u32 v, u32 &m, u32 &cnt) {
    if (v > m) {
      m = v;
      cnt = 1;
    } else if (v == m) {
      ++cnt;
    }
  }
  __device__ __forceinline__ u32 eval2(u32 x, u32 kids, u64 kids2) const {
    u8 W[25], M[25];
    load_window5(out, g, x, W);
    if (mask != nullptr) {
      load_window5(mask, g, x, M);
    } else {
#pragma unroll
      for (int i = 0; i < 25; ++i) M[i] = 1;
    }
    u32 m = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((kids & (1u << k)) && M[win_idx(k, -1)]) {
        const u32 kk = (u32)(kids2 >> (8 * k)) & 0xFFu;
        u32 m2 = 0, cnt2 = 0;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2)
          if ((kk & (1u << k2)) && M[win_idx(k, k2)]) join(W[win_idx(k, k2)], m2, cnt2);
        const u32 v = (cnt2 == 0 ? 1u : (cnt2 >= 2 ? m2 + 1 : m2)) & 0xFFu;  // the child is inside the mask
        join(v, m, cnt);
      }
    }
    if (cnt == 0) return M[12] ? 1u : 0u;
    return cnt >= 2 ? m + 1 : m;
  }
  __device__ __forceinline__ void store(u32 x, u32 v) const { out[x] = (u8)v; }
  // ---- exact-order engine: the closed form is order-independent, so the light upstream cells of a
  // trunk cell are folded into ONE element (max order, how many hold it) and post slots pass through.
  // LDS image of a tile: bit 7 = the cell is inside the mask, bits 0-6 = its order (an order of 128 would
  // need 2^127 cells)
  typedef u8 LV;
  typedef u32 Elem;
  __device__ __forceinline__ u8 tile_init(u32 x, bool nodata_cell) const {
    if (nodata_cell) return 0;
    const u32 m = (mask == nullptr || mask[x]) ? 1u : 0u;
    return (u8)((m << 7) | m);  // a headwater: 1 inside the mask, 0 outside
  }
  __device__ __forceinline__ u8 tile_combine(u32 l, u32 kids, const u8 *val) const {
    const u32 own = val[l] >> 7;
    u32 m = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (kids & (1u << k)) {
        const u32 c = val[(int)l + slot_dr(k) * XT + slot_dc(k)];
        if (c & 0x80u) join(c & 0x7Fu, m, cnt);
      }
    }
    const u32 r = cnt == 0 ? own : (cnt >= 2 ? m + 1 : m);
    return (u8)((own << 7) | (r & 0x7Fu));
  }
  __device__ __forceinline__ void tile_store(u32 x, u8 v) const { out[x] = v & 0x7Fu; }
  static constexpr bool NEEDS_NODATA = true;
  __device__ __forceinline__ void tile_init4(u32 x0, u32 nd, u8 (&v)[4]) const {
    u32 m4 = 0x01010101u;
    if (mask != nullptr) __builtin_memcpy(&m4, mask + x0, 4);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u32 m = ((m4 >> (8 * b)) & 0xFFu) ? 1u : 0u;
      v[b] = (nd & (1u << b)) ? (u8)0 : (u8)((m << 7) | m);
    }
  }
  __device__ __forceinline__ void tile_store4(u32 x0, const u8 (&v)[4]) const {
    const u32 o = (u32)(v[0] & 0x7Fu) | ((u32)(v[1] & 0x7Fu) << 8) | ((u32)(v[2] & 0x7Fu) << 16) | ((u32)(v[3] & 0x7Fu) << 24);
    __builtin_memcpy(out + x0, &o, 4);
  }
  // element (round 6: the fold as a three-way select).  With m / cnt = the highest order among the light upstream cells inside
  // the mask and how many hold it, the order of a cell whose heavy upstream cell (inside the mask) has order t is
  //   t > m: t     t == m: m + 1 (cnt >= 1; m else)     t < m: cnt >= 2 ? m + 1 : m
  // — three values the gather can compute: bits 0-7 m, 8-15 the value for t == m, 16-23 the value for t < m.  A cell whose
  // heavy upstream cell lies outside the mask, or that has none (the head of a chain), ignores t: m = 255 and bits 16-23
  // hold its order (own cell inside the mask and no upstream cell in it: 1; outside: 0).  A post slot passes t through:
  // the element 0 does that (t > 0: t; t == 0: bits 8-15 = 0).  The fold is two compares and two selects — no branch, a
  // dependent chain of three instead of a dozen — and the exact fold IS the fast one for the short chains (FAST_SHORT).
  __device__ __forceinline__ u32 pre_real(u32 x, u32 kids, u32 hs) const {
    u32 m = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 nb = nb_of(g, x, k);
      if ((kids & (1u << k)) && (u32)k != hs && (mask == nullptr || mask[nb])) join((u32)out[nb], m, cnt);
    }
    u32 hin = 0;
    if (hs < 8) hin = (mask == nullptr || mask[nb_of(g, x, (int)hs)]) ? 1u : 0u;
    const u32 own = (mask == nullptr || mask[x]) ? 1u : 0u;
    const u32 lt = cnt == 0 ? own : (cnt >= 2 ? m + 1 : m);  // the order without the heavy cell / with a lower one
    if (!hin) return 255u | (lt << 8) | (lt << 16);
    const u32 eq = cnt >= 1 ? m + 1 : m;
    return m | (eq << 8) | ((cnt == 0 ? m : lt) << 16);  // (cnt == 0: m = 0 and t < 0 never happens)
  }
  __device__ __forceinline__ u32 pre_post(u32) const { return 0u; }
  __device__ __forceinline__ u32 first(u32 e) const { return (e >> 16) & 0xFFu; }
  // speculative block fold (the long chains): along a main stem nearly every slot leaves the order unchanged (the heavy
  // cell's order exceeds that of the light cells); anything else redoes the block exactly
  static constexpr bool FAST = true;
  static constexpr bool FAST_CONST = true;   // (fold_fast leaves the running value unchanged whatever the element)
#ifndef STRAHLER_FUSE_UP
#define STRAHLER_FUSE_UP true
#endif
  // the fused gather + fold (k_xtrunk_prescan) LOST with the old, branchy fold (7.95 -> 8.3 ms at 30000^2: one lane per chain
  // doing a dozen dependent VALU steps per slot while its workgroup waits) and WINS with the select form: 7.65 -> 7.1 ms,
  // C5 shape 16.3 -> 13.7 ms (profiles/r06_ab_strahler_fold.txt)
  static constexpr bool FUSE_UP = STRAHLER_FUSE_UP;
  static constexpr bool FAST_SHORT = false;  // the short chains fold exactly right away: nothing to speculate on
  __device__ __forceinline__ u32 fold(u32 t, u32 e, bool) const {
    const u32 m = e & 0xFFu;
    const u32 le = t == m ? (e >> 8) & 0xFFu : (e >> 16) & 0xFFu;
    return t > m ? t : le;
  }
  __device__ __forceinline__ bool special(u32 t, u32 e) const { return fold(t, e, false) != t; }
  __device__ __forceinline__ u32 fold_fast(u32 t, u32) const { return t; }
};

template <class L>
struct Labels {
  typedef L V;
  const u8 *ncode;
  Geo g;
  L *out;
  __device__ __forceinline__ L top(u32 p) const { return out[p]; }
  // a seeded cell keeps its seed.  (A cell of this launch may be read here while its owner stores
  // its final label: both values lead to the same result, the label is a pure function of the path.)
  __device__ __forceinline__ L apply(u32 x, u32, bool root, L pv) const {
    const L own = out[x];
    if (own != 0) return own;
    return root ? (L)0 : pv;
  }
  __device__ __forceinline__ void store(u32 x, L v) const {
    if (v != 0) out[x] = v;
  }
};

template <class E>
struct Hand {
  typedef double V;
  const u8 *ncode;
  Geo g;
  const u8 *drain;
  const E *elev;
  double *out;
  // (exact_sweep.h XWatch) row blocks: the cells whose height is still unknown — their path ends in a halo cell whose
  // height has not arrived yet (-inf) — are counted and listed by the tile pass that stores them; null: nobody watches
  unsigned long long *watch_cnt = nullptr;
  u32 *watch_list = nullptr;
  u32 watch_cap = 0;
  __device__ __forceinline__ bool watched(u32 code, double v) const { return code != D8_HALO && v == -HUGE_VAL; }
  __device__ __forceinline__ double top(u32 p) const { return out[p]; }
  __device__ __forceinline__ double apply(u32 x, u32 code, bool root, double pv) const {
    if (drain[x] == 1) return 0.0;
    const u32 p = d8_down(g, x, code);
    const E dz = elev[x] - elev[p];  // difference in the elevation dtype (dem.py:328)
    return (root ? 0.0 : pv) + (double)dz;
  }
  __device__ __forceinline__ void store(u32 x, double v) const { out[x] = v; }
  // ---- exact-order engine ----
  struct DElem {
    E dz;  // the difference in the elevation dtype (dem.py:328); widened when it is added
    u32 is_drain;
  };
  __device__ __forceinline__ double dnodata(u32) const { return -9999.0; }
  __device__ __forceinline__ void dstore4(u32 x0, const double (&v)[4]) const { __builtin_memcpy(out + x0, v, 32); }
  __device__ __forceinline__ DElem dpre(u32 x, u32 code) const {
    DElem e;
    e.is_drain = drain[x] == 1 ? 1u : 0u;
    e.dz = elev[x] - elev[d8_down(g, x, code)];
    return e;
  }
  // tile image (k_xtile_down): the difference alone, the drain flag goes to the tile's flag bitmap
  // (already widened: the tile image of a leaf then has the type of the result and lives in the result's LDS word)
  typedef double DTile;
  static constexpr bool DTILE_FLAG = true;
  // the same for a whole quad inside the raster, loads only — unconditional, so that k_xtile_down has the loads of all
  // its 16 cells in flight at once (a load inside a per-cell branch is waited for on the spot: 32 round trips)
  static constexpr bool DTILE4 = true;
  // the short chains of a round folded through LDS (k_xtrunk_dscan_lds): the bulk rounds 3.60 -> 3.33 ms at 30000^2 here
  // (16 bytes per slot; the lane-per-chain kernel holds 203 VGPRs: 2 waves per SIMD); no gain for the 4-byte sweeps
  static constexpr bool DSCAN_LDS = true;
  struct DQuad {
    u32 d4;
    E ev[4], dn[4];
  };
  __device__ __forceinline__ void dtile4_load(u32 x0, u32 c4, DQuad &q) const {
    __builtin_memcpy(&q.d4, drain + x0, 4);
    __builtin_memcpy(q.ev, elev + x0, 4 * sizeof(E));
#pragma unroll
    for (int b = 0; b < 4; ++b) q.dn[b] = elev[d8_down(g, x0 + b, (c4 >> (8 * b)) & 0xFFu)];  // (itself unless a direction)
  }
  __device__ __forceinline__ double dtile4_get(const DQuad &q, int b, bool &is_drain) const {
    is_drain = ((q.d4 >> (8 * b)) & 0xFFu) == 1u;
    return (double)(E)(q.ev[b] - q.dn[b]);
  }
  __device__ __forceinline__ double dtile(u32 x, u32 code, bool &is_drain) const {
    is_drain = drain[x] == 1;
    return (double)(E)(elev[x] - elev[d8_down(g, x, code)]);
  }
  __device__ __forceinline__ double dtroot(double dz, bool is_drain) const { return is_drain ? 0.0 : 0.0 + dz; }
  __device__ __forceinline__ double dtfold(double dz, bool is_drain, double pv) const { return is_drain ? 0.0 : pv + dz; }
  __device__ __forceinline__ void top4(u32 x0, double (&v)[4]) const { __builtin_memcpy(v, out + x0, 32); }
  // speculative block fold (exact_sweep.h): a drain cell on a chain is rare; everything else is one add
  static constexpr bool FAST = true;
  static constexpr bool FAST_CONST = false;  // (fold_fast leaves the running value unchanged whatever the element)
  __device__ __forceinline__ bool dspecial(const DElem &e, double) const { return e.is_drain != 0u; }
  __device__ __forceinline__ double dfold_fast(const DElem &e, double pv) const { return pv + (double)e.dz; }
  __device__ __forceinline__ DElem dneutral() const { return DElem{(E)-0.0, 0u}; }  // pv + (-0.0) == pv for every pv
  __device__ __forceinline__ double droot(const DElem &e) const { return e.is_drain ? 0.0 : 0.0 + (double)e.dz; }
  __device__ __forceinline__ double dfold(const DElem &e, double pv) const { return e.is_drain ? 0.0 : pv + (double)e.dz; }
};

// dem.floodplains (reference pyflwdir/dem.py:333-379): down- to upstream.  A stream cell (upstream area >=
// upa_min) starts a floodplain with its own elevation z and height threshold h = uparea ** b (evaluated by the
// host in the reference's dtype: a pow() is not bit-reproducible across math libraries); any other cell joins
// the floodplain of its downstream cell if that cell is in one and elev - z <= h, and inherits (z, h).
struct FloodV {
  float z, h;
  i32 flag;  // 1 floodplain, 0 not (value of the result raster)
  i32 pad;
};
template <class E>
struct Flood {
  typedef FloodV V;
  const u8 *ncode;
  Geo g;
  const u8 *stream;  // 1 where uparea >= upa_min
  const float *h_in; // uparea ** b as float32 (meaningful on stream cells)
  const E *elev;
  FloodV *state;     // [n] running state (z, h, flag)
  struct DElem {
    E elev;
    float h;
    u32 is_stream;
  };
  typedef DElem DTile;
  static constexpr bool DTILE_FLAG = false;
  static constexpr bool DTILE4 = false;
  static constexpr bool DSCAN_LDS = false;
  static constexpr bool FAST = false;
  static constexpr bool FAST_CONST = false;  // (fold_fast leaves the running value unchanged whatever the element)
  __device__ __forceinline__ FloodV top(u32 p) const { return state[p]; }
  __device__ __forceinline__ void top4(u32 x0, FloodV (&v)[4]) const {
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = state[x0 + b];
  }
  __device__ __forceinline__ DElem dpre(u32 x, u32) const { return DElem{elev[x], h_in[x], (u32)(stream[x] == 1)}; }
  __device__ __forceinline__ DElem dtile(u32 x, u32 code, bool &) const { return dpre(x, code); }
  __device__ __forceinline__ FloodV droot(const DElem &e) const {
    // (a pit that is no stream cell looks at itself: its own flag is still 0)
    return e.is_stream ? FloodV{(float)e.elev, e.h, 1, 0} : FloodV{-9999.f, -9999.f, 0, 0};
  }
  __device__ __forceinline__ FloodV dfold(const DElem &e, const FloodV &pv) const {
    if (e.is_stream) return FloodV{(float)e.elev, e.h, 1, 0};
    if (pv.flag == 1) {
      const E dh = e.elev - (E)pv.z;  // float32 elevation: float32 arithmetic; float64: float64
      if (dh <= (E)pv.h) return FloodV{pv.z, pv.h, 1, 0};
    }
    return FloodV{-9999.f, -9999.f, 0, 0};
  }
  __device__ __forceinline__ FloodV dtroot(const DElem &e, bool) const { return droot(e); }
  __device__ __forceinline__ FloodV dtfold(const DElem &e, bool, const FloodV &pv) const { return dfold(e, pv); }
  __device__ __forceinline__ bool dspecial(const DElem &, const FloodV &) const { return false; }
  __device__ __forceinline__ FloodV dfold_fast(const DElem &, const FloodV &pv) const { return pv; }
  __device__ __forceinline__ FloodV apply(u32 x, u32 code, bool root, const FloodV &pv) const {
    const DElem e = dpre(x, code);
    return root ? droot(e) : dfold(e, pv);
  }
  __device__ __forceinline__ FloodV dnodata(u32) const { return FloodV{-9999.f, -9999.f, -1, 0}; }
  __device__ __forceinline__ void store(u32 x, const FloodV &v) const { state[x] = v; }
  __device__ __forceinline__ void dstore4(u32 x0, const FloodV (&v)[4]) const {
#pragma unroll
    for (int b = 0; b < 4; ++b) state[x0 + b] = v[b];
  }
};

#include "exact_sweep.h"

// up-/down-sweep of an operation: the exact-order engine when the raster has a plan (no cycles, whole
// raster), else the level engine
template <class Op>
static int sweep_up(pfd_raster *h, const Op &op, const char *name, const char *xname) {
  if (h->xplan_state == 1) return run_exact_up(h, op, xname);
  return run_up(h, op, name);
}
template <class Op>
static int sweep_down(pfd_raster *h, const Op &op, const char *name, const char *xname) {
  if (h->xplan_state == 1) return run_exact_down(h, op, xname);
  return run_down(h, op, name);
}
// the structure the sweeps of a handle run on: the exact plan, or the level structure
// allow_block: set by the pfd_*_block entry points only — every whole-raster operation refuses a row-block handle here
static int ensure_sweep_structure(pfd_raster *h, bool allow_block = false) {
  PFDCHK(pfd_ensure_xplan(h, allow_block));
  if (h->xplan_state == 1) return PFD_OK;
  return pfd_order_cells_impl(h, allow_block);
}

// ---------------------------------------------------------------------------------------------
// small streaming helpers
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void k_fill(T *__restrict__ out, u32 n, T v) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}
template <class T>
__global__ void k_mask_invalid(const u8 *__restrict__ ncode, u32 n, T *__restrict__ out, T v) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ncode[i] == D8_MV) out[i] = v;
}
__global__ void k_init_cell(const u8 *__restrict__ ncode, u32 n, i32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ncode[i] == D8_MV ? -9999 : 1;
}
template <class L>
__global__ void k_seed_labels(const i64 *__restrict__ idx, const L *__restrict__ ids, u32 k, L *__restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) out[idx[t]] = ids[t];
}

// ---------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------
static int upstream_area_cell_levels_dev(pfd_raster *h, i32 *out_dev) {
  PFDCHK(pfd_order_cells_impl(h));
  pfd_seg_begin(h, "init");
  k_init_cell<<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo.n, out_dev);
  KCHK();
  pfd_seg_end(h, 1);
  CountUp op{h->ncode, h->geo, (u32 *)out_dev};
  return run_up(h, op, "sweep_count_up");
}

extern "C" int pfd_upstream_area_cell_levels(pfd_raster *h, int32_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!out) {
    pfd_set_error("pfd_upstream_area_cell: NULL out");
    return PFD_EINVAL;
  }
  if (h->gen) return pfd_gen_upstream_area_cell(h, out, memspace);
  pfd_seg_clear(h);
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(i32), memspace));
  PFDCHK(upstream_area_cell_levels_dev(h, (i32 *)o.dev));
  return o.finish(h->stream);
}

// Fast path: LDS-tiled sweep (tiled.hip).  A raster with cycles (cells that never reach a pit)
// cannot be finished by it; those rasters are recomputed by the level engine, whose semantics
// for such cells are the reference's (they keep their own weight).
extern "C" int pfd_upstream_area_cell(pfd_raster *h, int32_t *out, int memspace) {
  PFDCHK(pfd_check_handle_lazy(h));  // a deferred handle is normalised inside the first tile pass
  if (!out) {
    pfd_set_error("pfd_upstream_area_cell: NULL out");
    return PFD_EINVAL;
  }
  if (h->gen) return pfd_gen_upstream_area_cell(h, out, memspace);
  if (h->halo_top || h->halo_bot) return pfd_require_whole(h, "pfd_upstream_area_cell");  // (blocks: _blocks / _begin / _dist)
  pfd_seg_clear(h);
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(i32), memspace));
  int complete = 0;
  PFDCHK(pfd_upstream_area_cell_tiled(h, (i32 *)o.dev, &complete));
  if (!complete) {
    PFDCHK(pfd_ensure_normalised(h));
    PFDCHK(upstream_area_cell_levels_dev(h, (i32 *)o.dev));
  }
  return o.finish(h->stream);
}

template <class T>
__global__ void k_fill_rows(const T *__restrict__ row, Geo g, T *__restrict__ out) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < g.n) out[x] = row[geo_row(g, x)];
}
// ---- int32 accuflux on the tiled engine ----------------------------------------------------------
// Integer accumulation is associative, so the LDS-tiled pointer-doubling engine (tiled.hip) gives the
// reference's result whenever the nodata rule of streams.accuflux (streams.py:38-40) cannot interfere:
// no valid cell holds the nodata value and no running sum can become it.  Checked in one pass: every
// valid cell's value >= 0, nodata < 0 (or no nodata test at all), and the total < 2^31 (no wrap, so
// every partial sum is >= 0).  Anything else goes through the level engine.
__global__ void __launch_bounds__(256) k_payload_check(const u8 *__restrict__ ncode, const i32 *__restrict__ data, u32 n,
                                                       i32 nodata, unsigned long long *__restrict__ res) {
  // res[0] = sum of the valid cells' values, res[1] = number of valid cells that are negative or nodata
  __shared__ unsigned long long s_sum;
  __shared__ u32 s_bad;
  if (threadIdx.x == 0) s_sum = 0, s_bad = 0;
  __syncthreads();
  unsigned long long sum = 0;
  u32 bad = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (ncode[i] == D8_MV) continue;
    const i32 v = data[i];
    if (v < 0 || v == nodata) ++bad; else sum += (unsigned long long)v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_down(sum, o);
    bad += __shfl_down(bad, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (sum) atomicAdd(&s_sum, sum);
    if (bad) atomicAdd(&s_bad, bad);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_sum) atomicAdd(&res[0], s_sum);
    if (s_bad) atomicAdd(&res[1], (unsigned long long)s_bad);
  }
}
__global__ void __launch_bounds__(256) k_restore_invalid(const u8 *__restrict__ ncode, const i32 *__restrict__ data, u32 n,
                                                         i32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ncode[i] == D8_MV) out[i] = data[i];  // (nodata cells of the raster keep their input value)
}
// *used = 1 if the tiled engine produced the result
static int accuflux_i32_tiled(pfd_raster *h, const i32 *data_dev, i32 nodata, int has_nodata, i32 *out_dev,
                              int *used) {
  *used = 0;
  if (pfd_knob("PFD_ACCUFLUX_LEVELS") || h->halo_top || h->halo_bot) return PFD_OK;
  if (has_nodata && nodata >= 0) return PFD_OK;
  DevBuf res;
  PFDCHK(res.alloc(2 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(res.p, 0, 2 * sizeof(unsigned long long), h->stream));
  pfd_seg_begin(h, "payload_check");
  k_payload_check<<<2048, 256, 0, h->stream>>>(h->ncode, data_dev, h->geo.n, has_nodata ? nodata : (i32)-1,
                                               res.as<unsigned long long>());
  KCHK();
  unsigned long long r[2];
  HIPCHK(hipMemcpyAsync(r, res.p, sizeof(r), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  pfd_seg_end(h, 1);
  if (r[1] != 0 || r[0] >= (1ull << 31)) return PFD_OK;
  int complete = 0;
  PFDCHK(pfd_upstream_area_cell_tiled(h, out_dev, &complete, data_dev));
  if (!complete) return PFD_OK;  // cycles: the level engine keeps the reference's semantics for them
  k_restore_invalid<<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, data_dev, h->geo.n, out_dev);
  KCHK();
  *used = 1;
  return PFD_OK;
}

// by_row: `data` holds one value per raster row (always a HOST pointer: nrow elements)
template <class T>
static int accuflux_t(pfd_raster *h, const void *data, bool by_row, T nodata, int has_nodata, int direction,
                      int mask_invalid, void *out, int memspace) {
  InArg d;
  if (by_row)
    PFDCHK(d.bind(data, (size_t)h->nrow * sizeof(T), PFD_HOST, h->stream));
  else
    PFDCHK(d.bind(data, (size_t)h->n * sizeof(T), memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(T), memspace));
  if (std::is_same<T, i32>::value && direction == PFD_UP && !by_row) {  // tiled engine where provably equivalent
    int used = 0;
    PFDCHK(accuflux_i32_tiled(h, (const i32 *)d.dev, (i32)nodata, has_nodata, (i32 *)o.dev, &used));
    if (used) {
      if (mask_invalid) {
        k_mask_invalid<T><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo.n, (T *)o.dev, nodata);
        KCHK();
      }
      return o.finish(h->stream);
    }
  }
  PFDCHK(ensure_sweep_structure(h));
  // (the exact engine's tile pass writes every cell, nodata cells from the payload: no initial copy)
  if (h->xplan_state != 1) {
    pfd_seg_begin(h, "init");
    if (by_row) {
      k_fill_rows<T><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((const T *)d.dev, h->geo, (T *)o.dev);
      KCHK();
    } else {
      HIPCHK(hipMemcpyAsync(o.dev, d.dev, (size_t)h->n * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    }
    pfd_seg_end(h, 1);
  }
  if (direction == PFD_UP && by_row && mask_invalid && h->xplan_state == 1) {
    // upstream_area in area units: the tile pass of the exact-order engine masks the nodata cells itself
    AccuUp<T, RowData<T>, true> op{h->ncode, h->geo, RowData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(run_exact_up(h, op, "exact_accuflux_up"));
    return o.finish(h->stream);
  } else if (direction == PFD_UP && by_row) {
    AccuUp<T, RowData<T>> op{h->ncode, h->geo, RowData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(sweep_up(h, op, "sweep_accuflux_up", "exact_accuflux_up"));
  } else if (direction == PFD_UP) {
    AccuUp<T> op{h->ncode, h->geo, CellData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(sweep_up(h, op, "sweep_accuflux_up", "exact_accuflux_up"));
  } else if (by_row) {
    AccuDown<T, RowData<T>> op{h->ncode, h->geo, RowData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(sweep_down(h, op, "sweep_accuflux_down", "exact_accuflux_down"));
  } else {
    AccuDown<T> op{h->ncode, h->geo, CellData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(sweep_down(h, op, "sweep_accuflux_down", "exact_accuflux_down"));
  }
  if (mask_invalid) {
    k_mask_invalid<T><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo.n, (T *)o.dev, nodata);
    KCHK();
  }
  return o.finish(h->stream);
}

static int accuflux_impl(pfd_raster *h, int dtype, const void *data, bool by_row, int64_t nodata_i, double nodata_f,
                         int has_nodata, int direction, int mask_invalid, void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!data || !out || (direction != PFD_UP && direction != PFD_DOWN)) {
    pfd_set_error("pfd_accuflux: bad arguments");
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  if (h->gen) return pfd_gen_accuflux(h, dtype, data, by_row, nodata_i, nodata_f, has_nodata, direction, mask_invalid, out, memspace);
  switch (dtype) {
    case PFD_I32:
      return accuflux_t<i32>(h, data, by_row, (i32)nodata_i, has_nodata, direction, mask_invalid, out, memspace);
    case PFD_I64:
      return accuflux_t<i64>(h, data, by_row, (i64)nodata_i, has_nodata, direction, mask_invalid, out, memspace);
    case PFD_F32:
      return accuflux_t<float>(h, data, by_row, (float)nodata_f, has_nodata, direction, mask_invalid, out, memspace);
    case PFD_F64:
      return accuflux_t<double>(h, data, by_row, nodata_f, has_nodata, direction, mask_invalid, out, memspace);
    default:
      pfd_set_error("pfd_accuflux: unsupported payload dtype code %d", dtype);
      return PFD_EUNSUPPORTED;
  }
}

extern "C" int pfd_accuflux(pfd_raster *h, int dtype, const void *data, int64_t nodata_i, double nodata_f,
                            int has_nodata, int direction, int mask_invalid, void *out, int memspace) {
  return accuflux_impl(h, dtype, data, false, nodata_i, nodata_f, has_nodata, direction, mask_invalid, out, memspace);
}

extern "C" int pfd_accuflux_rows(pfd_raster *h, int dtype, const void *row_values, int64_t nodata_i, double nodata_f,
                                 int has_nodata, int direction, int mask_invalid, void *out, int memspace) {
  return accuflux_impl(h, dtype, row_values, true, nodata_i, nodata_f, has_nodata, direction, mask_invalid, out,
                       memspace);
}

extern "C" int pfd_strahler(pfd_raster *h, const uint8_t *mask, uint8_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!out) {
    pfd_set_error("pfd_strahler: NULL out");
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  if (h->gen) return pfd_gen_strahler(h, mask, out, memspace);
  PFDCHK(ensure_sweep_structure(h));
  InArg m;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  if (h->xplan_state != 1) {  // (the exact engine's tile pass writes every cell)
    pfd_seg_begin(h, "init");
    HIPCHK(hipMemsetAsync(o.dev, 0, (size_t)h->n, h->stream));
    pfd_seg_end(h, 1);
  }
  Strahler op{h->ncode, h->geo, (const u8 *)m.dev, (u8 *)o.dev};
  PFDCHK(sweep_up(h, op, "sweep_strahler", "exact_strahler"));
  return o.finish(h->stream);
}

template <class L>
static int basins_t(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 k, void *out_dev) {
  pfd_seg_begin(h, "init");
  HIPCHK(hipMemsetAsync(out_dev, 0, (size_t)h->n * sizeof(L), h->stream));
  if (k) {
    k_seed_labels<L><<<cdiv_u32(k, 256), 256, 0, h->stream>>>(idx_dev, (const L *)ids_dev, k, (L *)out_dev);
    KCHK();
  }
  pfd_seg_end(h, 2);
  Labels<L> op{h->ncode, h->geo, (L *)out_dev};
  return run_down(h, op, "sweep_labels");
}

// labels from k distinct outlets (device arrays); fast path: LDS-tiled "first outlet downstream" query
// (paths.hip; needs no cell ordering).  Rasters with cycles (and PFD_BASINS_LEVELS=1) go through the level engine.
int pfd_basins_dev(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 ku, int id_size, void *out_dev) {
  if (h->gen) return pfd_gen_basins(h, idx_dev, ids_dev, ku, id_size, out_dev);
  int tiled_ok = 0;
  if (!pfd_knob("PFD_BASINS_LEVELS")) PFDCHK(pfd_basins_tiled(h, idx_dev, ids_dev, ku, id_size, out_dev, &tiled_ok));
  if (!tiled_ok) {
    if (pfd_wide_cells(h)) {  // (the level engine addresses cells with 32 bits; the caller has the row-block protocol)
      pfd_set_error("basins / ucat_area of a raster with cycles need the level engine's 32-bit cell order and are not available "
                    "for a raster of %lld cells on one handle (pfd_basins_begin / _finish run it in row blocks)", (long long)h->n);
      return PFD_EUNSUPPORTED;
    }
    PFDCHK(pfd_order_cells_impl(h));
    int rc;
    switch (id_size) {
      case 1: rc = basins_t<u8>(h, idx_dev, ids_dev, ku, out_dev); break;
      case 2: rc = basins_t<uint16_t>(h, idx_dev, ids_dev, ku, out_dev); break;
      case 4: rc = basins_t<u32>(h, idx_dev, ids_dev, ku, out_dev); break;
      default: rc = basins_t<u64>(h, idx_dev, ids_dev, ku, out_dev); break;
    }
    PFDCHK(rc);
  }
  return PFD_OK;
}

extern "C" int pfd_basins(pfd_raster *h, const int64_t *outlets, const void *ids, int64_t k, int id_size,
                          void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!out || k < 0 || (k > 0 && (!outlets || !ids)) ||
      (id_size != 1 && id_size != 2 && id_size != 4 && id_size != 8)) {
    pfd_set_error("pfd_basins: bad arguments (k=%lld, id_size=%d)", (long long)k, id_size);
    return PFD_EINVAL;
  }
  if (h->halo_top || h->halo_bot) return pfd_require_whole(h, "pfd_basins");  // (blocks: pfd_basins_begin / _finish)
  pfd_seg_clear(h);
  // numpy's `basins[idxs] = ids` keeps the LAST id of a repeated index: dedupe on the host so
  // that the device scatter has no write conflict.
  std::vector<i64> uidx;
  std::vector<unsigned char> uids;
  {
    std::unordered_set<i64> seen;
    uidx.reserve((size_t)k);
    uids.reserve((size_t)k * id_size);
    for (i64 j = k - 1; j >= 0; --j) {
      const i64 i = outlets[j];
      if (i < 0 || i >= h->n) {
        pfd_set_error("pfd_basins: outlet index %lld outside the raster", (long long)i);
        return PFD_EINVAL;
      }
      if (!seen.insert(i).second) continue;
      uidx.push_back(i);
      const unsigned char *src = (const unsigned char *)ids + (size_t)j * id_size;
      uids.insert(uids.end(), src, src + id_size);
    }
  }
  const u32 ku = (u32)uidx.size();
  InArg di, dl;
  PFDCHK(di.bind(ku ? uidx.data() : nullptr, (size_t)ku * sizeof(i64), PFD_HOST, h->stream));
  PFDCHK(dl.bind(ku ? uids.data() : nullptr, (size_t)ku * id_size, PFD_HOST, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * id_size, memspace));
  PFDCHK(pfd_basins_dev(h, (const i64 *)di.dev, dl.dev, ku, id_size, o.dev));
  return o.finish(h->stream);
}

template <class E>
static int hand_t(pfd_raster *h, const u8 *drain_dev, const void *elev_dev, double *out_dev) {
  if (h->xplan_state != 1) {  // (the exact engine's tile pass writes every cell, -9999 on nodata)
    pfd_seg_begin(h, "init");
    k_fill<double><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(out_dev, h->geo.n, -9999.0);
    KCHK();
    pfd_seg_end(h, 1);
  }
  Hand<E> op{h->ncode, h->geo, drain_dev, (const E *)elev_dev, out_dev};
  return sweep_down(h, op, "sweep_hand", "exact_hand");
}

// HAND of a row block (multi-GPU, DESIGN.md: sharded HAND): the halo cells stand for the neighbouring block's
// boundary cells and carry THEIR height (`seed`, 2 * ncol doubles on the device: top halo row, bottom halo row);
// every other cell is computed from its downstream cell exactly as on a whole raster, so a path that leaves the
// block continues the neighbour's sum with the same operands in the same order.
template <class E>
struct HandSeeded : Hand<E> {
  const double *seed;
  u32 row_first, row_last;  // owned rows of the device raster
  __device__ __forceinline__ double apply(u32 x, u32 code, bool root, double pv) const {
    if (code == D8_HALO) {
      const u32 r = geo_row(this->g, x);
      return seed[(r > row_last ? this->g.ncol : 0u) + (x - r * this->g.ncol)];
    }
    return Hand<E>::apply(x, code, root, pv);
  }
};

// incremental form (update != 0): `out` holds the result of an earlier call for other seeds.  Only the cells that
// are still unknown (-inf) can change — typically a few cells whose path weaves along the block edge.  The first call
// collects them once (one scan of the block); every later call seeds the halo cells, relaxes the listed cells
// (x <- hand[ds] + dz once hand[ds] is known: the very addition the sweep would do) until nothing moves, and drops the
// resolved ones from the list: O(unknown cells), not O(block).  More than n / 16 unknown cells: no list, the next
// call sweeps the block again.
struct HandBlockState {
  DevBuf list[2];
  u32 m = 0, cap = 0;
  unsigned long long unknown = 0;  // exact, also when the cells do not fit the list
  int cur = 0;
  bool valid = false;
};
void pfd_free_hand_block(pfd_raster *h) {
  delete (HandBlockState *)h->hand_block_state;
  h->hand_block_state = nullptr;
}
__global__ void __launch_bounds__(256) k_hb_seed(const u8 *__restrict__ ncode, u32 ncol, u32 nrow, u32 row_first,
                                                 u32 row_last, const double *__restrict__ seed, double *__restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  const u32 side = t / ncol, col = t - side * ncol;
  if ((side == 0 && row_first == 0) || (side == 1 && row_last + 1 >= nrow)) return;  // no halo row on this side
  const size_t x = (size_t)(side ? row_last + 1 : row_first - 1) * ncol + col;
  if (ncode[x] == D8_HALO) out[x] = seed[t];
}
// append the cells of src (or, src == nullptr, of the whole block) that are still unknown; halo cells are never listed
__global__ void __launch_bounds__(256) k_hb_collect(const u8 *__restrict__ ncode, const double *__restrict__ out,
                                                    const u32 *__restrict__ src, u32 n, u32 cap, u32 *__restrict__ list,
                                                    unsigned long long *__restrict__ cnt) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 x = i < n ? (src ? src[i] : i) : 0u;
  const bool u = i < n && ncode[x] != D8_HALO && out[x] == -HUGE_VAL;
  const u64 m = __ballot(u);
  if (!m) return;
  const u32 lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  u32 base = 0;
  if ((int)lane == leader) base = (u32)atomicAdd(cnt, (unsigned long long)__popcll(m));
  base = __shfl(base, leader);
  if (u) {
    const u32 pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
    if (pos < cap) list[pos] = x;
  }
}
template <class E>
__global__ void __launch_bounds__(256) k_hb_relax(const u8 *__restrict__ ncode, Geo g, const u8 *__restrict__ drain,
                                                  const E *__restrict__ elev, const u32 *__restrict__ list, u32 m,
                                                  double *__restrict__ out, u32 *__restrict__ changed) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const u32 x = list[i];
  if (out[x] != -HUGE_VAL) return;
  const u32 code = ncode[x];
  const u32 p = d8_down(g, x, code);
  const double pv = __hip_atomic_load(&out[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p == x || pv == -HUGE_VAL) return;  // (an unknown cell is never a root: roots are pits, drains or halo cells)
  const E dz = elev[x] - elev[p];
  __hip_atomic_store(&out[x], drain[x] == 1 ? 0.0 : pv + (double)dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (*changed == 0u) *changed = 1u;
}

// the same relaxation to its fixpoint in ONE launch, for a list one workgroup can sweep (the lists of the later exchanges
// hold a few thousand cells): rounds are separated by a workgroup barrier instead of a kernel boundary (a launch is ~6.5 us
// here, 64 of them were 0.42 ms of every exchange of a block), and the loop ends when a round moved nothing — no host look
template <class E>
__global__ void __launch_bounds__(1024) k_hb_relax_wg(const u8 *__restrict__ ncode, Geo g, const u8 *__restrict__ drain,
                                                      const E *__restrict__ elev, const u32 *__restrict__ list, u32 m,
                                                      double *__restrict__ out) {
  // a thread keeps its (at most 8) cells in registers: a round is ONE load per unresolved cell — the downstream cell's
  // height, all of a thread's loads in flight together — and a barrier
  u32 xs[8], ps[8];
  double dz[8];
  u32 open = 0, isdrain = 0;  // bit k: cell k is still unknown / is a drain cell (height 0 whatever arrives)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 i = threadIdx.x + 1024u * (u32)k;
    xs[k] = ps[k] = 0u;
    dz[k] = 0.0;
    if (i < m) {
      const u32 x = list[i];
      const u32 p = d8_down(g, x, (u32)ncode[x]);
      xs[k] = x, ps[k] = p;
      if (p != x && __hip_atomic_load(&out[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == -HUGE_VAL) {
        open |= 1u << k;
        const E d = elev[x] - elev[p];
        dz[k] = (double)d;
        isdrain |= drain[x] == 1 ? 1u << k : 0u;
      }
    }
  }
  for (u32 round = 0; round < (1u << 22); ++round) {
    double pv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      pv[k] = (open >> k) & 1u ? __hip_atomic_load(&out[ps[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -HUGE_VAL;
    int moved = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (((open >> k) & 1u) && pv[k] != -HUGE_VAL) {
        __hip_atomic_store(&out[xs[k]], (isdrain >> k) & 1u ? 0.0 : pv[k] + dz[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        open &= ~(1u << k);
        moved = 1;
      }
    }
    if (!__syncthreads_or(moved)) break;
  }
}
#define HB_WG_MAX 8192u  // cells one workgroup relaxes (8 per thread, in registers)

__global__ void __launch_bounds__(256) k_hb_count(const u8 *__restrict__ ncode, const double *__restrict__ out, u32 n,
                                                  unsigned long long *__restrict__ cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    c += ncode[i] != D8_HALO && out[i] == -HUGE_VAL;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(cnt, c);
}
// (re)build the list of unknown cells: from the whole block after a sweep (count first, so that the list is sized by
// the unknown cells — a few thousand — and not by the block), from the previous list after an update
static int hand_block_collect(pfd_raster *h, HandBlockState *st, const double *out, bool from_list) {
  const u32 n = h->geo.n;
  DevBuf ctr;
  PFDCHK(ctr.alloc(sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(ctr.p, 0, sizeof(unsigned long long), h->stream));
  unsigned long long m = 0;
  if (!from_list) {
    k_hb_count<<<4096, 256, 0, h->stream>>>(h->ncode, out, n, ctr.as<unsigned long long>());
    KCHK();
    HIPCHK(hipMemcpyAsync(&m, ctr.p, sizeof(m), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    st->unknown = m;
    st->valid = m <= (unsigned long long)std::max<u32>(n / 16u, 1024u);
    st->m = 0;
    if (!st->valid || m == 0) return PFD_OK;
    if (m > st->cap) {
      st->cap = (u32)m;
      PFDCHK(st->list[0].alloc((size_t)st->cap * sizeof(u32)));
      PFDCHK(st->list[1].alloc((size_t)st->cap * sizeof(u32)));
    }
    HIPCHK(hipMemsetAsync(ctr.p, 0, sizeof(unsigned long long), h->stream));
  }
  const int dst = from_list ? 1 - st->cur : st->cur;
  const u32 cnt = from_list ? st->m : n;
  if (cnt)
    k_hb_collect<<<cdiv_u32(cnt, 256), 256, 0, h->stream>>>(h->ncode, out, from_list ? st->list[st->cur].as<u32>() : nullptr, cnt,
                                                           st->cap, st->list[dst].as<u32>(), ctr.as<unsigned long long>());
  KCHK();
  HIPCHK(hipMemcpyAsync(&m, ctr.p, sizeof(m), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  st->cur = dst;
  st->unknown = m;
  st->m = (u32)std::min<unsigned long long>(m, st->cap);
  return PFD_OK;
}

template <class E>
static int hand_block_update(pfd_raster *h, HandBlockState *st, const u8 *drain, const E *elev, const double *seed_dev,
                             double *out) {
  const u32 ncol = (u32)h->ncol;
  const u32 rf = (u32)h->halo_top, rl = (u32)(h->halo_top + h->own_rows - 1);
  k_hb_seed<<<cdiv_u32(2 * ncol, 256), 256, 0, h->stream>>>(h->ncode, ncol, (u32)h->nrow, rf, rl, seed_dev, out);
  KCHK();
  if (st->m == 0) return PFD_OK;
  if (st->m <= HB_WG_MAX && !pfd_knob("PFD_HAND_RELAX_LAUNCHES")) {
    k_hb_relax_wg<E><<<1, 1024, 0, h->stream>>>(h->ncode, h->geo, drain, elev, st->list[st->cur].as<u32>(), st->m, out);
    KCHK();
    return hand_block_collect(h, st, out, true);  // drop what is known now
  }
  DevBuf fl;
  const int BATCH = 64;
  PFDCHK(fl.alloc(BATCH * sizeof(u32)));
  // rounds are issued in batches (a launch is ~3 us, a host round trip ~30; a first batch of 16 was tried: the unknown
  // stretches are longer than that more often than not, and the extra round trip costs more than 48 idle launches)
  const int batch = BATCH;
  for (int guard = 0; guard < (1 << 22); guard += batch) {
    HIPCHK(hipMemsetAsync(fl.p, 0, BATCH * sizeof(u32), h->stream));
    for (int r = 0; r < batch; ++r)
      k_hb_relax<E><<<cdiv_u32(st->m, 256), 256, 0, h->stream>>>(h->ncode, h->geo, drain, elev, st->list[st->cur].as<u32>(), st->m, out,
                                                                fl.as<u32>() + r);
    KCHK();
    u32 last = 0;
    HIPCHK(hipMemcpyAsync(&last, fl.as<u32>() + batch - 1, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!last) break;  // the last round of the batch moved nothing: fixpoint
  }
  return hand_block_collect(h, st, out, true);  // drop what is known now
}

extern "C" int pfd_hand_block(pfd_raster *h, const uint8_t *drain, int elev_dtype, const void *elevtn,
                              const double *halo_seed_host, int update, double *out, int memspace,
                              double *boundary_rows_host, int64_t *n_unknown) {
  PFDCHK(pfd_check_handle(h));
  if (!drain || !elevtn || !out || !halo_seed_host || (elev_dtype != PFD_F32 && elev_dtype != PFD_F64)) {
    pfd_set_error("pfd_hand_block: bad arguments (elevation dtype code %d)", elev_dtype);
    return PFD_EINVAL;
  }
  PFDCHK(pfd_reject_general(h, "hand_block"));
  pfd_seg_clear(h);
  // the structure the block sweeps on: the exact-order plan (its halo cells are cells with given values), or the level
  // structure (its halo cells are roots like its pits)
  PFDCHK(ensure_sweep_structure(h, true));
  InArg dr, el, sd;
  PFDCHK(dr.bind(drain, (size_t)h->n, memspace, h->stream));
  PFDCHK(el.bind(elevtn, (size_t)h->n * (elev_dtype == PFD_F32 ? 4 : 8), memspace, h->stream));
  PFDCHK(sd.bind(halo_seed_host, 2 * (size_t)h->ncol * sizeof(double), h->block_seed_space, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(double), memspace));
  HandBlockState *st = (HandBlockState *)h->hand_block_state;
  if (!st) h->hand_block_state = st = new HandBlockState();
  // (the list describes the device copy of `out`: a host caller's earlier result travels to the device, its list is
  //  rebuilt by a scan — host callers pay O(block) per call anyway)
  const bool incremental = update && st->valid && memspace == PFD_DEVICE;
  if (update && memspace == PFD_HOST) {
    HIPCHK(hipMemcpyAsync(o.dev, out, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    PFDCHK(hand_block_collect(h, st, (const double *)o.dev, false));
  }
  if (update && st->valid) {
    pfd_seg_begin(h, "hand_block_update");
    if (elev_dtype == PFD_F32)
      PFDCHK(hand_block_update<float>(h, st, (const u8 *)dr.dev, (const float *)el.dev, (const double *)sd.dev, (double *)o.dev));
    else
      PFDCHK(hand_block_update<double>(h, st, (const u8 *)dr.dev, (const double *)el.dev, (const double *)sd.dev, (double *)o.dev));
    pfd_seg_end(h, 3);
  } else {
    (void)incremental;
    bool counted = false;
    pfd_seg_begin(h, "init");
    k_fill<double><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((double *)o.dev, h->geo.n, -9999.0);
    KCHK();
    pfd_seg_end(h, 1);
    const u32 rf = (u32)h->halo_top, rl = (u32)(h->halo_top + h->own_rows - 1);
    if (h->xplan_state == 1) {  // exact-order engine: the halo cells hold their given heights from the start
      k_hb_seed<<<cdiv_u32(2 * (u64)h->ncol, 256), 256, 0, h->stream>>>(h->ncode, (u32)h->ncol, (u32)h->nrow, rf, rl,
                                                                      (const double *)sd.dev, (double *)o.dev);
      KCHK();
      // (the tile pass that stores the heights counts and lists the cells that are still unknown: no scan afterwards)
      const bool watch = (h->halo_top || h->halo_bot) && !pfd_knob("PFD_HAND_SCAN_UNKNOWN");
      DevBuf ctr;
      const u32 wcap = std::max<u32>(h->geo.n / 16u, 1024u);
      if (watch) {
        PFDCHK(ctr.alloc(sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(ctr.p, 0, sizeof(unsigned long long), h->stream));
        if (st->cap < wcap) {
          st->cap = wcap;
          PFDCHK(st->list[0].alloc((size_t)wcap * sizeof(u32)));
          PFDCHK(st->list[1].alloc((size_t)wcap * sizeof(u32)));
        }
        st->cur = 0;
      }
      if (elev_dtype == PFD_F32) {
        Hand<float> op{h->ncode, h->geo, (const u8 *)dr.dev, (const float *)el.dev, (double *)o.dev};
        if (watch) op.watch_cnt = ctr.as<unsigned long long>(), op.watch_list = st->list[0].as<u32>(), op.watch_cap = st->cap;
        PFDCHK(run_exact_down(h, op, "exact_hand_block"));
      } else {
        Hand<double> op{h->ncode, h->geo, (const u8 *)dr.dev, (const double *)el.dev, (double *)o.dev};
        if (watch) op.watch_cnt = ctr.as<unsigned long long>(), op.watch_list = st->list[0].as<u32>(), op.watch_cap = st->cap;
        PFDCHK(run_exact_down(h, op, "exact_hand_block"));
      }
      if (watch) {
        unsigned long long m = 0;
        HIPCHK(hipMemcpyAsync(&m, ctr.p, sizeof(m), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        st->unknown = m;
        st->valid = m <= (unsigned long long)wcap;
        st->m = st->valid ? (u32)m : 0u;
        counted = true;
      }
    } else if (elev_dtype == PFD_F32) {
      HandSeeded<float> op{{h->ncode, h->geo, (const u8 *)dr.dev, (const float *)el.dev, (double *)o.dev}, (const double *)sd.dev, rf, rl};
      PFDCHK(run_down(h, op, "sweep_hand_block"));
    } else {
      HandSeeded<double> op{{h->ncode, h->geo, (const u8 *)dr.dev, (const double *)el.dev, (double *)o.dev}, (const double *)sd.dev, rf, rl};
      PFDCHK(run_down(h, op, "sweep_hand_block"));
    }
    if (counted) {
      // (the tile pass listed them)
    } else if (h->halo_top || h->halo_bot) {
      PFDCHK(hand_block_collect(h, st, (const double *)o.dev, false));  // one scan: the unknown cells (own rows: never halo cells)
    } else {  // a block without halo rows (one rank): no height can be unknown — the scan of 9 bytes per cell is skipped
      st->unknown = 0, st->m = 0, st->valid = true;
    }
  }
  const size_t ncol = (size_t)h->ncol, own0 = (size_t)h->halo_top * ncol, own1 = own0 + (size_t)h->own_rows * ncol;
  if (n_unknown) *n_unknown = (int64_t)st->unknown;  // (the listed cells are own cells: halo cells are never listed)
  if (boundary_rows_host) {  // first and last OWN row: what the neighbouring blocks need as their halo heights
    HIPCHK(hipMemcpyAsync(boundary_rows_host, (const double *)o.dev + own0, ncol * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(boundary_rows_host + ncol, (const double *)o.dev + own1 - ncol, ncol * sizeof(double), hipMemcpyDeviceToHost,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return o.finish(h->stream);
}

// ---------------------------------------------------------------------------------------------
// Up-sweeps of a ROW BLOCK (accuflux, Strahler): the mirror image of pfd_hand_block.  A halo cell that drains into the
// block is one of the upstream cells of a boundary cell (k_halo_kids) and its value is GIVEN — the neighbouring
// block's boundary row, `halo_seed_host` — so the boundary cell adds it where the serial loop would (upstream cells
// in descending linear index): the same operands in the same order as on the whole raster, bit-identical for floats
// once the seeds are the neighbour's final values.  The caller iterates to the fixpoint (pyflwdir_amd/dist.py
// up_blocks): the blocks exchange their boundary rows until none of them changes; a value is final after as many
// exchanges as its longest upstream path crosses block edges.
// ---------------------------------------------------------------------------------------------
template <class Op>
struct OwnRows : Op {  // the op, storing into the own rows only: halo cells keep their seeds
  u32 lo, n_own;
  template <class VV>
  __device__ __forceinline__ void store(u32 x, VV v) const {
    if (x - lo < n_own) Op::store(x, v);
  }
};
__device__ __forceinline__ u64 bits_of(double v) { return (u64)__double_as_longlong(v); }
__device__ __forceinline__ u64 bits_of(float v) { return (u64)__float_as_uint(v); }
__device__ __forceinline__ u64 bits_of(i64 v) { return (u64)v; }
__device__ __forceinline__ u64 bits_of(i32 v) { return (u64)(u32)v; }
__device__ __forceinline__ u64 bits_of(u8 v) { return (u64)v; }
template <class T>
__device__ __forceinline__ bool bits_equal(const T &a, const T &b) { return bits_of(a) == bits_of(b); }
__device__ __forceinline__ bool bits_equal(const FloodV &a, const FloodV &b) {  // (the padding word carries nothing)
  return __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.h) == __float_as_uint(b.h) && a.flag == b.flag;
}
// local equations of every own cell against the values in place (halo rows = the neighbours' final rows): the
// all-cell check of a result that was computed in blocks, and the fixpoint the iteration ends in
template <class Op, class T>
__global__ void __launch_bounds__(256) k_verify_up(Op op, const u8 *__restrict__ ncode, const u8 *__restrict__ kids,
                                                   const T *__restrict__ out, u32 lo, u32 n_own,
                                                   unsigned long long *__restrict__ bad) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  bool b = false;
  if (i < n_own) {
    const u32 x = lo + i;
    if (ncode[x] != D8_MV) {
      const T v = (T)op.combine(x, (u32)kids[x], [&](u32 nb, int) { return op.leaf(nb); });
      b = bits_of(v) != bits_of(out[x]);  // (bitwise: -0.0 vs 0.0 and NaN payloads count)
    }
  }
  const u64 m = __ballot(b);
  if (m && (threadIdx.x & 63u) == 0u) atomicAdd(bad, (unsigned long long)__popcll(m));
}
template <class Op, class T>
static int up_block_run(pfd_raster *h, const Op &op0, T *out_dev, const T *seed_dev, int verify, T *brows_host,
                        int64_t *n_bad, const char *name) {
  const size_t ncol = (size_t)h->ncol, own0 = (size_t)h->halo_top * ncol, nown = (size_t)h->own_rows * ncol;
  if (h->halo_top)
    HIPCHK(hipMemcpyAsync(out_dev + own0 - ncol, seed_dev, ncol * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
  if (h->halo_bot)
    HIPCHK(hipMemcpyAsync(out_dev + own0 + nown, seed_dev + ncol, ncol * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
  OwnRows<Op> op{op0, (u32)own0, (u32)nown};
  if (verify) {
    const u8 *kids = nullptr;  // per cell: the neighbours draining into it, halo cells included
    if (h->xplan_state == 1) {
      kids = ((ExactPlan *)h->xplan)->kids;
    } else {
      PFDCHK(pfd_ensure_seq_aux(h));
      kids = h->cell_kids;
    }
    pfd_seg_begin(h, "verify_up_block");
    HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(u64), h->stream));
    k_verify_up<OwnRows<Op>, T><<<cdiv_u32((u64)nown, 256), 256, 0, h->stream>>>(op, h->ncode, kids, out_dev, (u32)own0,
                                                                                (u32)nown, (unsigned long long *)h->ctrl);
    KCHK();
    pfd_seg_end(h, 1);
    u64 bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, h->ctrl, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_bad) *n_bad = (int64_t)bad;
  } else if (h->xplan_state == 1) {
    // exact-order engine: the tile pass writes every cell, the halo cells among them; their given values go back in
    // before the trunk rounds read them (run_exact_up)
    h->xseed = seed_dev, h->xseed_out = out_dev, h->xseed_elem = sizeof(T);
    const int rc = run_exact_up(h, op0, "exact_up_block", h->block_update);
    h->xseed = nullptr, h->xseed_out = nullptr, h->xseed_elem = 0;
    PFDCHK(rc);
  } else {
    PFDCHK(run_up(h, op, name));
  }
  if (brows_host) {  // first and last OWN row: the neighbouring blocks' halo values
    HIPCHK(hipMemcpyAsync(brows_host, out_dev + own0, ncol * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(brows_host + ncol, out_dev + own0 + nown - ncol, ncol * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return PFD_OK;
}
static int up_block_prepare(pfd_raster *h, const char *what) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, what));
  if (!(h->halo_top || h->halo_bot) && h->own_rows != h->nrow) {
    pfd_set_error("%s: not a row-block handle", what);
    return PFD_EINVAL;
  }
  if ((h->halo_top || h->halo_bot) && !h->halo_raw) {
    pfd_set_error("%s: the D8 codes of the halo rows are not available on this handle", what);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  return ensure_sweep_structure(h, true);  // (the exact-order plan of the block, or its level structure)
}
// ---- down-sweeps of a row block (accuflux direction "down"): the halo cells a block drains into hold given values ----
template <class Op, class T = typename Op::V>
struct HaloSeeded : Op {  // level engine: a halo cell is a root of the block's ordering; its value is its seed
  const T *seed;  // (T: the stored type of the result, which the operation's V may be wider than)
  u32 row_last;
  __device__ __forceinline__ typename Op::V apply(u32 x, u32 code, bool root, typename Op::V pv) const {
    if (code == D8_HALO) {
      const u32 r = geo_row(this->g, x);
      return (typename Op::V)seed[(r > row_last ? this->g.ncol : 0u) + (x - r * this->g.ncol)];
    }
    return Op::apply(x, code, root, pv);
  }
};
template <class Op, class T>
__global__ void __launch_bounds__(256) k_verify_down(Op op, const u8 *__restrict__ ncode, Geo g, const T *__restrict__ out, u32 lo,
                                                     u32 n_own, unsigned long long *__restrict__ bad) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  bool b = false;
  if (i < n_own) {
    const u32 x = lo + i;
    const u32 code = ncode[x];
    if (code != D8_MV) {
      const bool root = !d8_is_dir(code);
      const T v = (T)op.apply(x, code, root, root ? T() : out[d8_down(g, x, code)]);
      b = !bits_equal(v, out[x]);
    }
  }
  const u64 m = __ballot(b);
  if (m && (threadIdx.x & 63u) == 0u) atomicAdd(bad, (unsigned long long)__popcll(m));
}
template <class Op, class T>
static int down_block_run(pfd_raster *h, const Op &op0, T *out_dev, const T *seed_dev, int verify, T *brows_host,
                          int64_t *n_bad, const char *name) {
  const size_t ncol = (size_t)h->ncol, own0 = (size_t)h->halo_top * ncol, nown = (size_t)h->own_rows * ncol;
  auto seed_rows = [&]() -> int {
    if (h->halo_top)
      HIPCHK(hipMemcpyAsync(out_dev + own0 - ncol, seed_dev, ncol * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    if (h->halo_bot)
      HIPCHK(hipMemcpyAsync(out_dev + own0 + nown, seed_dev + ncol, ncol * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    return PFD_OK;
  };
  PFDCHK(seed_rows());
  if (verify) {
    pfd_seg_begin(h, "verify_down_block");
    HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(u64), h->stream));
    k_verify_down<Op, T><<<cdiv_u32((u64)nown, 256), 256, 0, h->stream>>>(op0, h->ncode, h->geo, out_dev, (u32)own0, (u32)nown,
                                                                        (unsigned long long *)h->ctrl);
    KCHK();
    pfd_seg_end(h, 1);
    u64 bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, h->ctrl, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_bad) *n_bad = (int64_t)bad;
  } else if (h->xplan_state == 1) {
    PFDCHK(run_exact_down(h, op0, "exact_down_block"));  // (halo cells: given values, loaded and written back unchanged)
  } else {
    HaloSeeded<Op, T> op{op0, seed_dev, (u32)(h->halo_top + h->own_rows - 1)};
    PFDCHK(run_down(h, op, name));
    PFDCHK(seed_rows());  // (the level engine stored the seeds of the VALID halo cells only)
  }
  if (brows_host) {
    HIPCHK(hipMemcpyAsync(brows_host, out_dev + own0, ncol * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(brows_host + ncol, out_dev + own0 + nown - ncol, ncol * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return PFD_OK;
}

// pfd_set_block_update modes 1 / 2 keep POINTERS into the result buffer between calls: with a host `out` that buffer is
// OutArg's temporary, freed on return (and usually handed out again at the same address) — refused, not guessed
static int block_update_needs_device(pfd_raster *h, int memspace, const char *what) {
  if (h->block_update == 0 || memspace == PFD_DEVICE) return PFD_OK;
  pfd_xinc_drop(h);
  pfd_set_error("%s: pfd_set_block_update(%d) needs payload and result in DEVICE memory (memspace PFD_DEVICE)", what,
                h->block_update);
  return PFD_EINVAL;
}

template <class T>
static int accuflux_block_t(pfd_raster *h, const void *data, bool by_row, T nodata, int has_nodata, int direction,
                            const void *seed_host, int verify, void *out, int memspace, void *brows_host, int64_t *n_bad) {
  if (!verify) PFDCHK(block_update_needs_device(h, memspace, "pfd_accuflux_block"));
  InArg d, sd;
  if (by_row)
    PFDCHK(d.bind(data, (size_t)h->nrow * sizeof(T), PFD_HOST, h->stream));
  else
    PFDCHK(d.bind(data, (size_t)h->n * sizeof(T), memspace, h->stream));
  PFDCHK(sd.bind(seed_host, 2 * (size_t)h->ncol * sizeof(T), h->block_seed_space, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(T), memspace));
  if (verify && memspace == PFD_HOST)
    HIPCHK(hipMemcpyAsync(o.dev, out, (size_t)h->n * sizeof(T), hipMemcpyHostToDevice, h->stream));
  // (an update — pfd_set_block_update(h, 2) after a kept sweep of the same operation into the same buffer — starts
  //  from the result in `out`)
  const bool upd = direction == PFD_UP && xinc_applies(h, o.dev, by_row ? typeid(AccuUp<T, RowData<T>>).hash_code()
                                                                        : typeid(AccuUp<T>).hash_code());
  if (!verify && !upd) {
    pfd_seg_begin(h, "init");
    if (by_row) {
      k_fill_rows<T><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((const T *)d.dev, h->geo, (T *)o.dev);
      KCHK();
    } else {
      HIPCHK(hipMemcpyAsync(o.dev, d.dev, (size_t)h->n * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    }
    pfd_seg_end(h, 1);
  }
  if (direction == PFD_DOWN && by_row) {
    AccuDown<T, RowData<T>> op{h->ncode, h->geo, RowData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(down_block_run(h, op, (T *)o.dev, (const T *)sd.dev, verify, (T *)brows_host, n_bad, "sweep_accuflux_down_block"));
  } else if (direction == PFD_DOWN) {
    AccuDown<T> op{h->ncode, h->geo, CellData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(down_block_run(h, op, (T *)o.dev, (const T *)sd.dev, verify, (T *)brows_host, n_bad, "sweep_accuflux_down_block"));
  } else if (by_row) {
    AccuUp<T, RowData<T>> op{h->ncode, h->geo, RowData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(up_block_run(h, op, (T *)o.dev, (const T *)sd.dev, verify, (T *)brows_host, n_bad, "sweep_accuflux_block"));
  } else {
    AccuUp<T> op{h->ncode, h->geo, CellData<T>{(const T *)d.dev, h->geo}, (T *)o.dev, nodata, has_nodata};
    PFDCHK(up_block_run(h, op, (T *)o.dev, (const T *)sd.dev, verify, (T *)brows_host, n_bad, "sweep_accuflux_block"));
  }
  return verify ? PFD_OK : o.finish(h->stream);
}
extern "C" int pfd_accuflux_block(pfd_raster *h, int dtype, const void *data, int by_row, int64_t nodata_i, double nodata_f,
                                  int has_nodata, int direction, const void *halo_seed_host, int verify, void *out,
                                  int memspace, void *boundary_rows_host, int64_t *n_bad) {
  PFDCHK(up_block_prepare(h, "pfd_accuflux_block"));
  if (!data || !out || !halo_seed_host || (direction != PFD_UP && direction != PFD_DOWN)) {
    pfd_set_error("pfd_accuflux_block: bad arguments");
    return PFD_EINVAL;
  }
  switch (dtype) {
    case PFD_I32:
      return accuflux_block_t<i32>(h, data, by_row != 0, (i32)nodata_i, has_nodata, direction, halo_seed_host, verify, out, memspace, boundary_rows_host, n_bad);
    case PFD_I64:
      return accuflux_block_t<i64>(h, data, by_row != 0, (i64)nodata_i, has_nodata, direction, halo_seed_host, verify, out, memspace, boundary_rows_host, n_bad);
    case PFD_F32:
      return accuflux_block_t<float>(h, data, by_row != 0, (float)nodata_f, has_nodata, direction, halo_seed_host, verify, out, memspace, boundary_rows_host, n_bad);
    case PFD_F64:
      return accuflux_block_t<double>(h, data, by_row != 0, nodata_f, has_nodata, direction, halo_seed_host, verify, out, memspace, boundary_rows_host, n_bad);
    default:
      pfd_set_error("pfd_accuflux_block: unsupported payload dtype code %d", dtype);
      return PFD_EUNSUPPORTED;
  }
}
extern "C" int pfd_strahler_block(pfd_raster *h, const uint8_t *mask, const uint8_t *halo_seed_host, int verify, uint8_t *out,
                                  int memspace, uint8_t *boundary_rows_host, int64_t *n_bad) {
  PFDCHK(up_block_prepare(h, "pfd_strahler_block"));
  if (!out || !halo_seed_host) {
    pfd_set_error("pfd_strahler_block: bad arguments");
    return PFD_EINVAL;
  }
  if (!verify) PFDCHK(block_update_needs_device(h, memspace, "pfd_strahler_block"));
  InArg m, sd;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  PFDCHK(sd.bind(halo_seed_host, 2 * (size_t)h->ncol, h->block_seed_space, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  if (verify && memspace == PFD_HOST) HIPCHK(hipMemcpyAsync(o.dev, out, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  if (!verify && !xinc_applies(h, o.dev, typeid(Strahler).hash_code())) {
    pfd_seg_begin(h, "init");
    HIPCHK(hipMemsetAsync(o.dev, 0, (size_t)h->n, h->stream));
    pfd_seg_end(h, 1);
  }
  Strahler op{h->ncode, h->geo, (const u8 *)m.dev, (u8 *)o.dev};
  PFDCHK(up_block_run(h, op, (u8 *)o.dev, (const u8 *)sd.dev, verify, boundary_rows_host, n_bad, "sweep_strahler_block"));
  return verify ? PFD_OK : o.finish(h->stream);
}

extern "C" int pfd_hand(pfd_raster *h, const uint8_t *drain, int elev_dtype, const void *elevtn, double *out,
                        int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!drain || !elevtn || !out || (elev_dtype != PFD_F32 && elev_dtype != PFD_F64)) {
    pfd_set_error("pfd_hand: bad arguments (elevation dtype code %d)", elev_dtype);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  if (h->gen) return pfd_gen_hand(h, drain, elev_dtype, elevtn, out, memspace);
  PFDCHK(ensure_sweep_structure(h));
  InArg dr, el;
  PFDCHK(dr.bind(drain, (size_t)h->n, memspace, h->stream));
  PFDCHK(el.bind(elevtn, (size_t)h->n * (elev_dtype == PFD_F32 ? 4 : 8), memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(double), memspace));
  if (elev_dtype == PFD_F32)
    PFDCHK(hand_t<float>(h, (const u8 *)dr.dev, el.dev, (double *)o.dev));
  else
    PFDCHK(hand_t<double>(h, (const u8 *)dr.dev, el.dev, (double *)o.dev));
  return o.finish(h->stream);
}

// ---------------------------------------------------------------------------------------------
// main upstream cell, classic stream order, stream distance
//   core.main_upstream        pyflwdir/core.py:191-219      k_main_upstream (no ordering needed)
//   streams.stream_order      pyflwdir/streams.py:191-225   k_trib_flag + Classic (down- to upstream)
//   streams.stream_distance   pyflwdir/streams.py:272-315   Dist<T>            (down- to upstream)
// ---------------------------------------------------------------------------------------------
template <class I> struct IdxMv;
template <> struct IdxMv<i32> { static __device__ __forceinline__ i32 mv() { return -1; } };
template <> struct IdxMv<u32> { static __device__ __forceinline__ u32 mv() { return 0xFFFFFFFFu; } };
template <> struct IdxMv<i64> { static __device__ __forceinline__ i64 mv() { return -1; } };

// the reference scans the cells in ascending index and keeps a child only if its area is
// strictly larger than the best so far (initially upa_min): first maximum in ascending order
template <class T, class I>
__global__ void __launch_bounds__(256) k_main_upstream(const u8 *__restrict__ ncode, Geo g, const T *__restrict__ uparea,
                                                       T upa_min, I *__restrict__ out) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.n) return;
  I best_i = IdxMv<I>::mv();
  if (ncode[x] != D8_MV) {
    const u32 r = geo_row(g, x), c = x - r * g.ncol;
    T best = upa_min;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = PFD_SLOT_ASC[q];
      u32 nb;
      if (d8_child(ncode, g, x, r, c, k, &nb)) {
        const T a = uparea[nb];
        if (a > best) {
          best = a;
          best_i = (I)nb;
        }
      }
    }
  }
  out[x] = best_i;
}

// flag[x] = 1 when x is a tributary at a confluence: its downstream cell has more than one
// upstream cell inside the mask (core.upstream_count with mask, core.py:50-61) and x is not that
// cell's main upstream cell
template <class I>
__global__ void __launch_bounds__(256) k_trib_flag(const u8 *__restrict__ ncode, Geo g, const I *__restrict__ main_us,
                                                   const u8 *__restrict__ mask, u8 *__restrict__ flag) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.n) return;
  u8 f = 0;
  const u32 code = ncode[x];
  if (d8_is_dir(code)) {
    const u32 p = d8_down(g, x, code);
    const u32 r = geo_row(g, p), c = p - r * g.ncol;
    u32 nup = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      u32 nb;
      if (d8_child(ncode, g, p, r, c, k, &nb) && (mask == nullptr || mask[nb])) ++nup;
    }
    f = (nup > 1 && main_us[p] != (I)x) ? 1 : 0;
  }
  flag[x] = f;
}

struct Classic {
  typedef u32 V;
  const u8 *ncode;
  Geo g;
  const u8 *flag;
  const u8 *mask;  // may be null
  u8 *out;
  __device__ __forceinline__ u32 top(u32 p) const { return out[p]; }
  __device__ __forceinline__ u32 apply(u32 x, u32, bool root, u32 pv) const {
    if (mask != nullptr && !mask[x]) return 0;  // outside the mask: stays 0
    return root ? 1u : ((pv + flag[x]) & 0xFFu);  // uint8 arithmetic like the reference
  }
  __device__ __forceinline__ void store(u32 x, u32 v) const { out[x] = (u8)v; }
  // ---- exact-order engine: bit 0 = tributary flag, bit 1 = outside the mask ----
  typedef u32 DElem;
  __device__ __forceinline__ u32 dnodata(u32) const { return 0u; }
  __device__ __forceinline__ void dstore4(u32 x0, const u32 (&v)[4]) const {
    const u32 o = (v[0] & 0xFFu) | ((v[1] & 0xFFu) << 8) | ((v[2] & 0xFFu) << 16) | ((v[3] & 0xFFu) << 24);
    __builtin_memcpy(out + x0, &o, 4);
  }
  __device__ __forceinline__ u32 dpre(u32 x, u32) const {
    return (u32)flag[x] | ((mask != nullptr && !mask[x]) ? 2u : 0u);
  }
  // tile image of the element (exact_sweep.h, k_xtile_down): the element itself, no flag
  typedef DElem DTile;
  static constexpr bool DTILE_FLAG = false;
  static constexpr bool DTILE4 = false;
  static constexpr bool DSCAN_LDS = false;
  __device__ __forceinline__ DElem dtile(u32 x, u32 code, bool &) const { return dpre(x, code); }
  __device__ __forceinline__ u32 dtroot(DElem e, bool) const { return droot(e); }
  __device__ __forceinline__ u32 dtfold(DElem e, bool, u32 pv) const { return dfold(e, pv); }
  __device__ __forceinline__ void top4(u32 x0, u32 (&v)[4]) const {
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = top(x0 + b);
  }
  static constexpr bool FAST = false;
  static constexpr bool FAST_CONST = false;  // (fold_fast leaves the running value unchanged whatever the element)
  __device__ __forceinline__ bool dspecial(u32, u32) const { return false; }
  __device__ __forceinline__ u32 dfold_fast(u32, u32 pv) const { return pv; }
  __device__ __forceinline__ u32 droot(u32 e) const { return (e & 2u) ? 0u : 1u; }
  __device__ __forceinline__ u32 dfold(u32 e, u32 pv) const { return (e & 2u) ? 0u : ((pv + (e & 1u)) & 0xFFu); }
};

template <class T>
struct Dist {
  typedef T V;
  const u8 *ncode;
  Geo g;
  const u8 *mask;     // may be null
  const float *dtab;  // T == float: [3 * (2*nrow-1)] step lengths by row sum and kind of step
  T *out;
  __device__ __forceinline__ T top(u32 p) const { return out[p]; }
  __device__ __forceinline__ T apply(u32 x, u32 code, bool root, T pv) const {
    if (root || (mask != nullptr && mask[x])) return (T)0;
    if (dtab != nullptr) {
      const int k = d8_slot(code);
      const int dr = d8_dr(k), dc = d8_dc(k);
      const u32 s = 2u * geo_row(g, x) + (u32)dr;  // r0 + r1
      const int kind = (dr != 0 && dc != 0) ? 2 : (dr != 0 ? 0 : 1);
      return (T)((float)pv + dtab[3u * s + (u32)kind]);
    }
    return (T)((u32)pv + 1u);
  }
  __device__ __forceinline__ void store(u32 x, T v) const { out[x] = v; }
  // ---- exact-order engine: the step length of the cell (1 in cell units), negative = the distance restarts ----
  typedef T DElem;
  __device__ __forceinline__ T dnodata(u32) const { return (T)-9999; }
  __device__ __forceinline__ void dstore4(u32 x0, const T (&v)[4]) const { __builtin_memcpy(out + x0, v, 4 * sizeof(T)); }
  __device__ __forceinline__ T dpre(u32 x, u32 code) const {
    if (!d8_is_dir(code) || (mask != nullptr && mask[x])) return (T)-1;
    if (dtab != nullptr) {
      const int k = d8_slot(code);
      const int dr = d8_dr(k), dc = d8_dc(k);
      const u32 s = 2u * geo_row(g, x) + (u32)dr;
      const int kind = (dr != 0 && dc != 0) ? 2 : (dr != 0 ? 0 : 1);
      return (T)dtab[3u * s + (u32)kind];
    }
    return (T)1;
  }
  // tile image of the element (exact_sweep.h, k_xtile_down): the element itself, no flag
  typedef DElem DTile;
  static constexpr bool DTILE_FLAG = false;
  static constexpr bool DTILE4 = false;
  static constexpr bool DSCAN_LDS = false;
  __device__ __forceinline__ DElem dtile(u32 x, u32 code, bool &) const { return dpre(x, code); }
  __device__ __forceinline__ T dtroot(DElem e, bool) const { return droot(e); }
  __device__ __forceinline__ T dtfold(DElem e, bool, T pv) const { return dfold(e, pv); }
  __device__ __forceinline__ void top4(u32 x0, T (&v)[4]) const {
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = top(x0 + b);
  }
  static constexpr bool FAST = false;
  static constexpr bool FAST_CONST = false;  // (fold_fast leaves the running value unchanged whatever the element)
  __device__ __forceinline__ bool dspecial(T, T) const { return false; }
  __device__ __forceinline__ T dfold_fast(T, T pv) const { return pv; }
  __device__ __forceinline__ T droot(T) const { return (T)0; }
  __device__ __forceinline__ T dfold(T e, T pv) const {
    if (e < (T)0) return (T)0;
    if (dtab != nullptr) return (T)((float)pv + (float)e);
    return (T)((u32)pv + 1u);
  }
};

template <class T, class I>
static void launch_main_upstream(pfd_raster *h, const void *upa, double upa_min, void *out) {
  k_main_upstream<T, I><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo, (const T *)upa, (T)upa_min,
                                                                        (I *)out);
}
template <class T>
static int main_upstream_t(pfd_raster *h, const void *upa, double upa_min, int idx_dtype, void *out) {
  if (idx_dtype == PFD_I32)
    launch_main_upstream<T, i32>(h, upa, upa_min, out);
  else if (idx_dtype == PFD_U32)
    launch_main_upstream<T, u32>(h, upa, upa_min, out);
  else
    launch_main_upstream<T, i64>(h, upa, upa_min, out);
  KCHK();
  return PFD_OK;
}
static size_t idx_bytes(int idx_dtype) {
  return idx_dtype == PFD_I32 || idx_dtype == PFD_U32 ? 4 : (idx_dtype == PFD_I64 ? 8 : 0);
}
static size_t payload_bytes(int dtype) {
  return dtype == PFD_I32 || dtype == PFD_F32 ? 4 : (dtype == PFD_I64 || dtype == PFD_F64 ? 8 : 0);
}

extern "C" int pfd_main_upstream(pfd_raster *h, int dtype, const void *uparea, double upa_min, int idx_dtype,
                                 void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (h->gen) return pfd_gen_main_upstream(h, dtype, uparea, upa_min, idx_dtype, out, memspace);
  PFDCHK(pfd_require_whole(h, "pfd_main_upstream"));
  const size_t es = idx_bytes(idx_dtype), ps = payload_bytes(dtype);
  if (!uparea || !out || !es || !ps) {
    pfd_set_error("pfd_main_upstream: bad arguments (dtype %d, index dtype %d)", dtype, idx_dtype);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  InArg a;
  PFDCHK(a.bind(uparea, (size_t)h->n * ps, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * es, memspace));
  pfd_seg_begin(h, "main_upstream");
  int rc;
  switch (dtype) {
    case PFD_I32: rc = main_upstream_t<i32>(h, a.dev, upa_min, idx_dtype, o.dev); break;
    case PFD_I64: rc = main_upstream_t<i64>(h, a.dev, upa_min, idx_dtype, o.dev); break;
    case PFD_F32: rc = main_upstream_t<float>(h, a.dev, upa_min, idx_dtype, o.dev); break;
    default: rc = main_upstream_t<double>(h, a.dev, upa_min, idx_dtype, o.dev); break;
  }
  PFDCHK(rc);
  pfd_seg_end(h, 1);
  return o.finish(h->stream);
}

// arithmetics.upstream_sum (reference pyflwdir/arithmetics.py:147-169; Flwdir.upstream_sum flwdir.py:412-433): the sum of
// the values of the cells directly upstream.  The reference is a serial loop over ascending index
//     if ds != mv and ds != i:  if data[i] == nodata or data[ds] == nodata: out[i] = nodata  else: out[ds] += data[i]
// so what a cell x ends up with is decided by the events that touch out[x] in ascending index order: every upstream
// cell c adds data[c] (unless data[c] or data[x] is nodata), and x itself, in its place among them, OVERWRITES out[x]
// with nodata if data[x] or the value of its downstream cell is nodata.  Pull form: one thread per cell, neighbours
// in ascending index with the cell itself between its W and E neighbour — the operand order of the serial loop.
template <class T>
__global__ void __launch_bounds__(256) k_upstream_sum(const u8 *__restrict__ ncode, Geo g, const T *__restrict__ data,
                                                      T nodata, int has_nodata, T *__restrict__ out) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.n) return;
  const u32 cx = ncode[x];
  const T dx = data[x];
  T acc = T(0);
  const u32 row = geo_row(g, x), col = x - row * g.ncol;
#pragma unroll
  for (int q = 0; q < 9; ++q) {  // (-1,-1) (-1,0) (-1,1) (0,-1) SELF (0,1) (1,-1) (1,0) (1,1)
    if (q == 4) {
      if (d8_is_dir(cx)) {  // a pit points at itself, a nodata cell at nothing: no event
        const T dd = data[d8_down(g, x, cx)];
        if (has_nodata && (dx == nodata || dd == nodata)) acc = nodata;
      }
      continue;
    }
    const int dr = q / 3 - 1, dc = q % 3 - 1;
    const i64 r = (i64)row + dr, c = (i64)col + dc;
    if (r < 0 || r >= (i64)g.nrow || c < 0 || c >= (i64)g.ncol) continue;
    const u32 nb = (u32)(r * (i64)g.ncol + c);
    const u32 cn = ncode[nb];
    if (!d8_is_dir(cn) || d8_down(g, nb, cn) != x) continue;
    const T dn = data[nb];
    if (!has_nodata || (dn != nodata && dx != nodata)) acc = Num<T>::add(acc, dn);
  }
  out[x] = acc;
}

extern "C" int pfd_upstream_sum(pfd_raster *h, int dtype, const void *data, int64_t nodata_i, double nodata_f,
                                int has_nodata, void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "pfd_upstream_sum"));
  PFDCHK(pfd_require_whole(h, "pfd_upstream_sum"));
  const size_t ps = payload_bytes(dtype);
  if (!data || !out || !ps) {
    pfd_set_error("pfd_upstream_sum: bad arguments (dtype %d)", dtype);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  InArg a;
  PFDCHK(a.bind(data, (size_t)h->n * ps, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * ps, memspace));
  pfd_seg_begin(h, "upstream_sum");
  const u32 grid = cdiv_u32(h->geo.n, 256);
  switch (dtype) {
    case PFD_I32: k_upstream_sum<i32><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const i32 *)a.dev, (i32)nodata_i, has_nodata, (i32 *)o.dev); break;
    case PFD_I64: k_upstream_sum<i64><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const i64 *)a.dev, (i64)nodata_i, has_nodata, (i64 *)o.dev); break;
    case PFD_F32: k_upstream_sum<float><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const float *)a.dev, (float)nodata_f, has_nodata, (float *)o.dev); break;
    default: k_upstream_sum<double><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const double *)a.dev, nodata_f, has_nodata, (double *)o.dev); break;
  }
  KCHK();
  pfd_seg_end(h, 1);
  return o.finish(h->stream);
}

extern "C" int pfd_stream_order_classic(pfd_raster *h, int idx_dtype, const void *idxs_us_main, const uint8_t *mask,
                                        uint8_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  const size_t es = idx_bytes(idx_dtype);
  if (!idxs_us_main || !out || !es) {
    pfd_set_error("pfd_stream_order_classic: bad arguments (index dtype %d)", idx_dtype);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  if (h->gen) return pfd_gen_classic(h, idx_dtype, idxs_us_main, mask, out, memspace);
  PFDCHK(ensure_sweep_structure(h));
  InArg mu, m;
  PFDCHK(mu.bind(idxs_us_main, (size_t)h->n * es, memspace, h->stream));
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  DevBuf flag;
  PFDCHK(flag.alloc((size_t)h->n));
  pfd_seg_begin(h, "init");
  if (h->xplan_state != 1) HIPCHK(hipMemsetAsync(o.dev, 0, (size_t)h->n, h->stream));
  const u32 grid = cdiv_u32((u64)h->n, 256);
  if (idx_dtype == PFD_I32)
    k_trib_flag<i32><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const i32 *)mu.dev, (const u8 *)m.dev, flag.as<u8>());
  else if (idx_dtype == PFD_U32)
    k_trib_flag<u32><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const u32 *)mu.dev, (const u8 *)m.dev, flag.as<u8>());
  else
    k_trib_flag<i64><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const i64 *)mu.dev, (const u8 *)m.dev, flag.as<u8>());
  KCHK();
  pfd_seg_end(h, 2);
  Classic op{h->ncode, h->geo, flag.as<u8>(), (const u8 *)m.dev, (u8 *)o.dev};
  PFDCHK(sweep_down(h, op, "sweep_classic_order", "exact_classic_order"));
  return o.finish(h->stream);  // (synchronises: `flag` may be released afterwards)
}

// ---------------------------------------------------------------------------------------------
// classic stream order over ROW BLOCKS (reference pyflwdir/streams.py:191-225 + core.main_upstream core.py:191-219):
// what a cell needs to know of its DOWNSTREAM cell p is one byte — which of p's neighbours is p's main upstream cell
// (slot 0-7, 15: none) and whether p has more than one upstream cell inside the mask (bit 4) — no index array at all:
// a cell that drains into p through direction k is p's neighbour (k + 4) & 7.  pfd_trib_info_block computes that byte
// for every cell of the block's device raster from the upstream masks that include the halo cells (the block's plan /
// level structure); the byte of a HALO cell is incomplete (its own upstream cells lie beyond the block) and is
// replaced by the neighbouring block's boundary row before pfd_stream_order_classic_block reads it
// (pyflwdir_amd/dist.py classic_blocks: one exchange of one byte per boundary cell, then the seeded down-sweeps).
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_trib_info(const u8 *__restrict__ ncode, Geo g, const u8 *__restrict__ kids,
                                                   const T *__restrict__ uparea, T upa_min, const u8 *__restrict__ mask,
                                                   u8 *__restrict__ tinfo) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.n) return;
  u32 slot = 15u, nup = 0;
  if (ncode[x] != D8_MV) {
    const u32 m = kids[x];
    T best = upa_min;
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // ascending index: the first maximum wins (core.py:216, strict >)
      const int k = PFD_SLOT_ASC[q];
      if (m & (1u << k)) {
        const u32 nb = nb_of(g, x, k);
        const T a = uparea[nb];
        if (a > best) best = a, slot = (u32)k;
        nup += (mask == nullptr || mask[nb]) ? 1u : 0u;
      }
    }
  }
  tinfo[x] = (u8)(slot | (nup > 1u ? 0x10u : 0u));
}
__global__ void __launch_bounds__(256) k_trib_flag_info(const u8 *__restrict__ ncode, Geo g, const u8 *__restrict__ tinfo,
                                                        u8 *__restrict__ flag) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.n) return;
  u8 f = 0;
  const u32 code = ncode[x];
  if (d8_is_dir(code)) {
    const u32 t = tinfo[d8_down(g, x, code)];
    f = ((t & 0x10u) && (t & 0xFu) != (((u32)d8_slot(code) + 4u) & 7u)) ? 1 : 0;
  }
  flag[x] = f;
}
extern "C" int pfd_trib_info_block(pfd_raster *h, int dtype, const void *uparea, double upa_min, const uint8_t *mask,
                                   uint8_t *tinfo, int memspace) {
  PFDCHK(up_block_prepare(h, "pfd_trib_info_block"));
  const size_t ps = payload_bytes(dtype);
  if (!uparea || !tinfo || !ps) {
    pfd_set_error("pfd_trib_info_block: bad arguments (dtype %d)", dtype);
    return PFD_EINVAL;
  }
  PFDCHK(ensure_sweep_structure(h, true));
  const u8 *kids = nullptr;  // per cell: the neighbours draining into it, halo cells included
  if (h->xplan_state == 1) {
    kids = ((ExactPlan *)h->xplan)->kids;
  } else {
    PFDCHK(pfd_ensure_seq_aux(h));
    kids = h->cell_kids;
  }
  InArg a, m;
  PFDCHK(a.bind(uparea, (size_t)h->n * ps, memspace, h->stream));
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(tinfo, (size_t)h->n, memspace));
  const u32 grid = cdiv_u32((u64)h->n, 256);
  pfd_seg_begin(h, "trib_info");
  switch (dtype) {
    case PFD_I32: k_trib_info<i32><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, kids, (const i32 *)a.dev, (i32)upa_min, (const u8 *)m.dev, (u8 *)o.dev); break;
    case PFD_I64: k_trib_info<i64><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, kids, (const i64 *)a.dev, (i64)upa_min, (const u8 *)m.dev, (u8 *)o.dev); break;
    case PFD_F32: k_trib_info<float><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, kids, (const float *)a.dev, (float)upa_min, (const u8 *)m.dev, (u8 *)o.dev); break;
    default: k_trib_info<double><<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, kids, (const double *)a.dev, upa_min, (const u8 *)m.dev, (u8 *)o.dev); break;
  }
  KCHK();
  pfd_seg_end(h, 1);
  return o.finish(h->stream);
}
extern "C" int pfd_stream_order_classic_block(pfd_raster *h, const uint8_t *tinfo, const uint8_t *mask,
                                              const uint8_t *halo_seed_host, int verify, uint8_t *out, int memspace,
                                              uint8_t *boundary_rows_host, int64_t *n_bad) {
  PFDCHK(up_block_prepare(h, "pfd_stream_order_classic_block"));
  if (!tinfo || !out || !halo_seed_host) {
    pfd_set_error("pfd_stream_order_classic_block: bad arguments");
    return PFD_EINVAL;
  }
  PFDCHK(ensure_sweep_structure(h, true));
  InArg ti, m, sd;
  PFDCHK(ti.bind(tinfo, (size_t)h->n, memspace, h->stream));
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  PFDCHK(sd.bind(halo_seed_host, 2 * (size_t)h->ncol, h->block_seed_space, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  if (verify && memspace == PFD_HOST) HIPCHK(hipMemcpyAsync(o.dev, out, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  DevBuf flag;
  PFDCHK(flag.alloc((size_t)h->n));
  pfd_seg_begin(h, "init");
  if (!verify && h->xplan_state != 1) HIPCHK(hipMemsetAsync(o.dev, 0, (size_t)h->n, h->stream));
  k_trib_flag_info<<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo, (const u8 *)ti.dev, flag.as<u8>());
  KCHK();
  pfd_seg_end(h, 2);
  Classic op{h->ncode, h->geo, flag.as<u8>(), (const u8 *)m.dev, (u8 *)o.dev};
  PFDCHK(down_block_run(h, op, (u8 *)o.dev, (const u8 *)sd.dev, verify, boundary_rows_host, n_bad, "sweep_classic_block"));
  if (verify) {
    HIPCHK(hipStreamSynchronize(h->stream));
    return PFD_OK;
  }
  return o.finish(h->stream);  // (synchronises: `flag` may be released afterwards)
}

extern "C" int pfd_stream_distance(pfd_raster *h, const uint8_t *mask, int real_length, const float *step_lengths,
                                   void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!out || (real_length && !step_lengths)) {
    pfd_set_error("pfd_stream_distance: bad arguments");
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  if (h->gen) return pfd_gen_stream_distance(h, mask, real_length, step_lengths, out, memspace);
  PFDCHK(ensure_sweep_structure(h));
  InArg m, tab;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  if (real_length)  // the table always comes from the host
    PFDCHK(tab.bind(step_lengths, 3 * (size_t)(2 * h->nrow - 1) * sizeof(float), PFD_HOST, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * 4, memspace));
  if (h->xplan_state != 1) {
    pfd_seg_begin(h, "init");
    if (real_length)
      k_fill<float><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((float *)o.dev, h->geo.n, -9999.0f);
    else
      k_fill<i32><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((i32 *)o.dev, h->geo.n, -9999);
    KCHK();
    pfd_seg_end(h, 1);
  }
  if (real_length) {
    Dist<float> op{h->ncode, h->geo, (const u8 *)m.dev, (const float *)tab.dev, (float *)o.dev};
    PFDCHK(sweep_down(h, op, "sweep_stream_distance", "exact_stream_distance"));
  } else {
    Dist<i32> op{h->ncode, h->geo, (const u8 *)m.dev, nullptr, (i32 *)o.dev};
    PFDCHK(sweep_down(h, op, "sweep_stream_distance", "exact_stream_distance"));
  }
  return o.finish(h->stream);  // (synchronises: the staged table may be released afterwards)
}

// stream_distance of a ROW BLOCK (see pfd_accuflux_block, direction "down"): the halo cells the block drains into hold
// the neighbour's distances (`halo_seed_host`: 2 * ncol int32 / float32); `step_lengths` covers the rows of the block's
// device raster (3 * (2 * nrow - 1) floats, the slice of the whole raster's table that starts at its first device row).
extern "C" int pfd_stream_distance_block(pfd_raster *h, const uint8_t *mask, int real_length, const float *step_lengths,
                                         const void *halo_seed_host, int verify, void *out, int memspace,
                                         void *boundary_rows_host, int64_t *n_bad) {
  PFDCHK(up_block_prepare(h, "pfd_stream_distance_block"));
  if (!out || !halo_seed_host || (real_length && !step_lengths)) {
    pfd_set_error("pfd_stream_distance_block: bad arguments");
    return PFD_EINVAL;
  }
  InArg m, tab, sd;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  if (real_length) PFDCHK(tab.bind(step_lengths, 3 * (size_t)(2 * h->nrow - 1) * sizeof(float), PFD_HOST, h->stream));
  PFDCHK(sd.bind(halo_seed_host, 2 * (size_t)h->ncol * 4, h->block_seed_space, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * 4, memspace));
  if (verify && memspace == PFD_HOST) HIPCHK(hipMemcpyAsync(o.dev, out, (size_t)h->n * 4, hipMemcpyHostToDevice, h->stream));
  if (!verify && h->xplan_state != 1) {
    pfd_seg_begin(h, "init");
    if (real_length)
      k_fill<float><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((float *)o.dev, h->geo.n, -9999.0f);
    else
      k_fill<i32><<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>((i32 *)o.dev, h->geo.n, -9999);
    KCHK();
    pfd_seg_end(h, 1);
  }
  if (real_length) {
    Dist<float> op{h->ncode, h->geo, (const u8 *)m.dev, (const float *)tab.dev, (float *)o.dev};
    PFDCHK(down_block_run(h, op, (float *)o.dev, (const float *)sd.dev, verify, (float *)boundary_rows_host, n_bad,
                          "sweep_stream_distance_block"));
  } else {
    Dist<i32> op{h->ncode, h->geo, (const u8 *)m.dev, nullptr, (i32 *)o.dev};
    PFDCHK(down_block_run(h, op, (i32 *)o.dev, (const i32 *)sd.dev, verify, (i32 *)boundary_rows_host, n_bad,
                          "sweep_stream_distance_block"));
  }
  return verify ? PFD_OK : o.finish(h->stream);
}

// ---------------------------------------------------------------------------------------------
// dem.floodplains (reference pyflwdir/dem.py:333-379; FlwdirRaster.floodplains pyflwdir.py:1513-1545)
// ---------------------------------------------------------------------------------------------
__global__ void k_flood_init(const u8 *__restrict__ ncode, u32 n, FloodV *__restrict__ st) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st[i] = FloodV{-9999.f, -9999.f, -1, 0};  // cells off the sequence keep -1
}
__global__ void k_flood_out(const FloodV *__restrict__ st, u32 n, int8_t *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int8_t)st[i].flag;
}
template <class E>
static int floodplains_t(pfd_raster *h, const u8 *stream, const float *hin, const void *elev, FloodV *st) {
  Flood<E> op{h->ncode, h->geo, stream, hin, (const E *)elev, st};
  return sweep_down(h, op, "sweep_floodplains", "exact_floodplains");
}
extern "C" int pfd_floodplains(pfd_raster *h, int elev_dtype, const void *elevtn, const uint8_t *is_stream,
                               const float *stream_h, int8_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "floodplains"));
  if (!elevtn || !is_stream || !stream_h || !out || (elev_dtype != PFD_F32 && elev_dtype != PFD_F64)) {
    pfd_set_error("pfd_floodplains: bad arguments (elevation dtype code %d)", elev_dtype);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  PFDCHK(ensure_sweep_structure(h));
  InArg el, sm, hh;
  PFDCHK(el.bind(elevtn, (size_t)h->n * (elev_dtype == PFD_F32 ? 4 : 8), memspace, h->stream));
  PFDCHK(sm.bind(is_stream, (size_t)h->n, memspace, h->stream));
  PFDCHK(hh.bind(stream_h, (size_t)h->n * sizeof(float), memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  DevBuf st;
  PFDCHK(st.alloc((size_t)h->n * sizeof(FloodV) + 64));
  const u32 grid = cdiv_u32((u64)h->n, 256);
  k_flood_init<<<grid, 256, 0, h->stream>>>(h->ncode, h->geo.n, st.as<FloodV>());
  KCHK();
  if (elev_dtype == PFD_F32)
    PFDCHK(floodplains_t<float>(h, (const u8 *)sm.dev, (const float *)hh.dev, el.dev, st.as<FloodV>()));
  else
    PFDCHK(floodplains_t<double>(h, (const u8 *)sm.dev, (const float *)hh.dev, el.dev, st.as<FloodV>()));
  k_flood_out<<<grid, 256, 0, h->stream>>>(st.as<FloodV>(), h->geo.n, (int8_t *)o.dev);
  KCHK();
  return o.finish(h->stream);  // (synchronises: `st` may be released afterwards)
}

// the int8 flags of a row block's OWN rows from its device-resident state (pfd_floodplains_block): what a caller that
// assembles the raster on the host downloads — 1 byte per cell instead of the 16-byte records
extern "C" int pfd_floodplains_block_flags(pfd_raster *h, const void *state_dev, int8_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!state_dev || !out) {
    pfd_set_error("pfd_floodplains_block_flags: NULL state or out");
    return PFD_EINVAL;
  }
  const u32 n_own = (u32)((size_t)h->own_rows * h->ncol);
  OutArg o;
  PFDCHK(o.bind(out, (size_t)n_own, memspace));
  if (n_own) k_flood_out<<<cdiv_u32(n_own, 256), 256, 0, h->stream>>>((const FloodV *)state_dev + (size_t)h->halo_top * h->ncol, n_own, (int8_t *)o.dev);
  KCHK();
  return o.finish(h->stream);
}

// dem.floodplains of a ROW BLOCK (reference pyflwdir/dem.py:333-379), the down- to upstream twin of pfd_hand_block /
// pfd_stream_distance_block: a halo cell the block drains into holds the neighbouring block's STATE — (z, h, flag) of the
// floodplain that cell is in, 16 bytes — as `halo_seed_host` gives it; the caller exchanges the boundary rows of the
// state and sweeps again until no row changes (pyflwdir_amd/dist.py floodplains_blocks).  `state` covers the block's
// device raster (own + halo rows, FloodV = {float z, float h, int32 flag, int32 pad} per cell) and IS the result: flag
// = 1 floodplain, 0 not, -1 nodata.  Lifts the 2^32-cell limit of pfd_floodplains for rasters cut into row blocks.
extern "C" int pfd_floodplains_block(pfd_raster *h, int elev_dtype, const void *elevtn, const uint8_t *is_stream,
                                     const float *stream_h, const void *halo_seed_host, int verify, void *state,
                                     int memspace, void *boundary_rows_host, int64_t *n_bad) {
  PFDCHK(up_block_prepare(h, "pfd_floodplains_block"));
  if (!elevtn || !is_stream || !stream_h || !state || !halo_seed_host || (elev_dtype != PFD_F32 && elev_dtype != PFD_F64)) {
    pfd_set_error("pfd_floodplains_block: bad arguments (elevation dtype code %d)", elev_dtype);
    return PFD_EINVAL;
  }
  PFDCHK(ensure_sweep_structure(h, true));
  InArg el, sm, hh, sd;
  PFDCHK(el.bind(elevtn, (size_t)h->n * (elev_dtype == PFD_F32 ? 4 : 8), memspace, h->stream));
  PFDCHK(sm.bind(is_stream, (size_t)h->n, memspace, h->stream));
  PFDCHK(hh.bind(stream_h, (size_t)h->n * sizeof(float), memspace, h->stream));
  PFDCHK(sd.bind(halo_seed_host, 2 * (size_t)h->ncol * sizeof(FloodV), h->block_seed_space, h->stream));
  OutArg o;
  PFDCHK(o.bind(state, (size_t)h->n * sizeof(FloodV), memspace));
  if (verify && memspace == PFD_HOST)
    HIPCHK(hipMemcpyAsync(o.dev, state, (size_t)h->n * sizeof(FloodV), hipMemcpyHostToDevice, h->stream));
  if (!verify) {
    pfd_seg_begin(h, "init");
    k_flood_init<<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo.n, (FloodV *)o.dev);
    KCHK();
    pfd_seg_end(h, 1);
  }
  if (elev_dtype == PFD_F32) {
    Flood<float> op{h->ncode, h->geo, (const u8 *)sm.dev, (const float *)hh.dev, (const float *)el.dev, (FloodV *)o.dev};
    PFDCHK(down_block_run(h, op, (FloodV *)o.dev, (const FloodV *)sd.dev, verify, (FloodV *)boundary_rows_host, n_bad,
                          "sweep_floodplains_block"));
  } else {
    Flood<double> op{h->ncode, h->geo, (const u8 *)sm.dev, (const float *)hh.dev, (const double *)el.dev, (FloodV *)o.dev};
    PFDCHK(down_block_run(h, op, (FloodV *)o.dev, (const FloodV *)sd.dev, verify, (FloodV *)boundary_rows_host, n_bad,
                          "sweep_floodplains_block"));
  }
  return verify ? PFD_OK : o.finish(h->stream);
}
