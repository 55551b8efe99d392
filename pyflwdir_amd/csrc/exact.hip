// exact.hip — builds the plan of the exact-order engine (exact.h): leaf steps per tile, heavy chains
// of the trunk, slot layout.  The sweeps themselves are templates over the operation (exact_sweep.h,
// instantiated in sweeps.hip).
//
// Everything here is integer work that is exact in any execution order; the plan only decides WHERE
// and WHEN a cell is computed, never in which order its operands are combined.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "exact.h"
#include "tiled.h"

int pfd_path_rank(pfd_raster *h, const u8 *codes, u32 *out_dev, int *complete);                         // paths.hip
int pfd_path_labels(pfd_raster *h, const u8 *codes, const u32 *seed_dev, u32 *out_dev, int *complete);  // paths.hip
int pfd_path_rank_tails(pfd_raster *h, const u8 *codes, u32 *hops_dev, u32 *tails_dev, int *complete);   // paths.hip

// ---------------------------------------------------------------------------------------------
// P2: leaf steps of one tile.  A cell is a leaf of step s if all its upstream cells lie in the tile and
// are leaves of steps < s (s = 1 + max), s <= XCAP; headwaters are step 0.  Found by in-degree counting
// in LDS: a finished cell decrements its downstream cell; a cell whose counter reaches zero joins the
// next step.  A cell with an upstream cell outside the tile never starts counting down ("blocked").
// ---------------------------------------------------------------------------------------------
// Row blocks (halo_raw != nullptr): a cell of a halo row is neither leaf nor trunk — its value is given (XL_HALO) —
// but where its code AS GIVEN points at a valid own cell it is one of that cell's upstream cells (the normalised codes
// hold sinks in the halo rows): the staged image is patched with those directions, and a cell with an upstream halo
// cell never starts counting down (blocked -> trunk), like one with an upstream cell outside the tile.
__global__ void __launch_bounds__(256) k_plan_tile(const u8 *__restrict__ ncode, u32 nrow, u32 ncol, u32 ntc,
                                                   u8 *__restrict__ lh, u8 *__restrict__ kids_out,
                                                   uint16_t *__restrict__ tord, uint16_t *__restrict__ toff,
                                                   const u8 *__restrict__ halo_raw, u32 row_first, u32 row_last) {
  __shared__ __attribute__((aligned(16))) u8 code[HW * CP];
  __shared__ u32 cnt[XTC];         // unresolved upstream cells | 0x100 blocked | 0x200 nodata
  __shared__ uint16_t ord[XTC];
  __shared__ u32 s_n;              // entries in ord after the headwaters
  __shared__ u32 s_add[XOFF];      // entries appended by step s (a counter per step: one barrier per step)
  __shared__ uint16_t off[XOFF];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_, tr = by_;
  const i64 r0 = (i64)tr * XT, c0 = (i64)tc * XT;
  {
    u32 v[5];
    stage_load_auto(ncode, nrow, ncol, r0, c0, tid, v);
    stage_store(code, tid, v);
  }
  if (tid == 0) s_n = 0;
  if (tid < XOFF) off[tid] = 0, s_add[tid] = 0;
  __syncthreads();
  // image rows that are halo rows of the block (-2: none)
  int hrow[2] = {-2, -2};
  if (halo_raw) {
    if (row_first > 0 && (i64)row_first - 1 >= r0 - 1 && (i64)row_first - 1 <= r0 + XT) hrow[0] = (int)((i64)row_first - 1 - r0);
    if (row_last + 1 < nrow && (i64)row_last + 1 >= r0 - 1 && (i64)row_last + 1 <= r0 + XT) hrow[1] = (int)((i64)row_last + 1 - r0);
    for (u32 t = tid; t < 2u * (XT + 2); t += 256u) {
      const int side = (int)(t / (XT + 2)), lc = (int)(t % (XT + 2)) - 1, lr = hrow[side];
      const i64 gc = c0 + lc;
      if (lr == -2 || gc < 0 || gc >= (i64)ncol) continue;
      const u32 raw = halo_raw[(size_t)side * ncol + (size_t)gc];
      if (!d8_is_dir(raw)) continue;
      const int k = d8_slot(raw);
      if (d8_dr(k) != (side ? -1 : 1)) continue;  // (only a step into the own rows links the halo cell to the block)
      const int tr_ = lr + d8_dr(k), tc_ = lc + d8_dc(k);
      const i64 tgc = gc + d8_dc(k);
      if (tr_ < -1 || tr_ > XT || tgc < 0 || tgc >= (i64)ncol) continue;
      if (CODE(lr, lc) != D8_MV && CODE(tr_, tc_) != D8_MV) CODE(lr, lc) = (u8)raw;
    }
    __syncthreads();
  }
  auto is_halo_row = [&](int lr) { return lr == hrow[0] || lr == hrow[1]; };
  // append to ord, one LDS atomic per wave (every lane of the wave calls it: same-address atomics with a return value
  // are served one lane after the other)
  const u32 lane = tid & 63u;
  auto append = [&](bool f, u32 val, u32 *counter, u32 first) {
    const u64 bm = __ballot(f);
    if (!bm) return;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(counter, (u32)__popcll(bm));
    base = first + (u32)__shfl((int)base, 0);
    if (f) ord[base + (u32)__popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)val;
  };
  u32 mykids[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const u32 l = tid + 256u * j;
    const int lr = l >> 6, lc = l & 63;
    const u32 c = CODE(lr, lc);
    u32 m = 0, blocked = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
      if (CODE(nr, nc) == (1u << ((k + 4) & 7))) {
        m |= 1u << k;
        if ((unsigned)nr >= XT || (unsigned)nc >= XT || is_halo_row(nr)) blocked = 0x100u;
      }
    }
    const u32 nkids = (u32)__popc(m);  // (a halo cell keeps its count: the leaves draining into it count it down)
    if (is_halo_row(lr)) {
      m = 0;
      blocked = 0x400u;
    }
    if (c == D8_MV) {
      m = 0;
      blocked = 0x200u;
    }
    mykids[j] = m;
    const u32 v = (c == D8_MV ? 0u : nkids) | blocked;
    cnt[l] = v == 0 ? 0x1000u : v;  // (0x1000 | step: the cell is a leaf of that step — headwaters: step 0)
    append(v == 0, l, &s_n, 0u);
  }
  __syncthreads();
  u32 begin = 0, end = s_n;
  int s = 0;
  // (off[0] = 0 already)
  while (end > begin && s < XCAP) {
    if (tid == 0) off[s + 1] = (uint16_t)end;
    for (u32 jb = begin; jb < end; jb += 256u) {  // (uniform trip count: append() is called by whole waves)
      const u32 j = jb + tid;
      bool last = false;
      u32 p = 0;
      if (j < end) {
        const u32 x = ord[j];
        const int lr = x >> 6, lc = x & 63;
        const u32 c = CODE(lr, lc);
        if (d8_is_dir(c)) {
          const int k = d8_slot(c);
          const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
          if ((unsigned)nr < XT && (unsigned)nc < XT) {
            p = (u32)(nr * XT + nc);
            last = atomicSub(&cnt[p], 1u) == 1u;  // last upstream cell done: nobody else touches cnt[p] any more
            if (last) cnt[p] = 0x1000u | (u32)(s + 1);
          }
        }
      }
      append(last, p, &s_add[s], end);
    }
    __syncthreads();
    begin = end;
    end += s_add[s];  // (the next step counts in its own word: no second barrier)
    ++s;
  }
  // cells appended by the last executed step (step index s) stay leaves only if s <= XCAP: the loop
  // stops at s == XCAP with [begin, end) = the cells of step XCAP, which are kept; their parents are not
  // appended any more (the loop did not run for them) -> trunk.
  const u32 total = end;
  // steps 0 .. s hold cells ([off[t], off[t+1])), off[s+1] = total and so are the entries behind it
  if (tid >= (u32)s + 1u && tid < XOFF) off[tid] = (uint16_t)total;
  __syncthreads();
  const size_t tile = (size_t)tr * ntc + tc;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const u32 l = tid + 256u * j;
    const i64 gr = r0 + (l >> 6), gc = c0 + (l & 63);
    if (gr < (i64)nrow && gc < (i64)ncol) {
      const size_t g = (size_t)gr * ncol + (size_t)gc;
      const u32 v = cnt[l];
      lh[g] = (u8)((v & 0x1000u) ? (v & 0xFFu) : ((v & 0x200u) ? XL_NODATA : ((v & 0x400u) ? XL_HALO : XL_TRUNK)));
      kids_out[g] = (u8)mykids[j];
    }
    // entry = the leaf's cell (12 bits) | slot of its downstream cell << 12 | "is a pit" << 15: the down-sweep of a
    // tile then needs no code lookup per leaf (k_xtile_down)
    uint16_t ent = 0;
    if (l < total) {
      const u32 x = ord[l];
      const u32 c = CODE((int)(x >> 6), (int)(x & 63u));
      ent = (uint16_t)(x | (d8_is_dir(c) ? (u32)d8_slot(c) << 12 : 0x8000u));
    }
    tord[tile * XTC + l] = ent;
  }
  if (tid < XOFF) toff[tile * XOFF + tid] = off[tid];
}

// ---------------------------------------------------------------------------------------------
// P3: heavy links of the trunk.  heavy child of x = its upstream TRUNK cell with the largest upstream
// area (first maximum in ascending index).  hcode = the forest of heavy links only (light trunk cells
// and pits become path ends, everything else nodata): a chain end is a pit of that forest.
// ---------------------------------------------------------------------------------------------
// One workgroup per 64 x 64 tile: upstream areas and marks of the tile and TWO rings in LDS (the heavy flag of a cell
// asks for the heavy slot of its downstream cell, which may lie in the first ring, whose own upstream cells reach into
// the second), coalesced loads instead of up to 34 gathers per trunk cell in raster order (10.9 ms at 30000^2; as two
// raster-order passes with gathers only of the cells that drain in: 8.9 ms; this form: see DESIGN.md 4.5).
#define HPW (XT + 4)  // staged edge with two rings
#define HRW (XT + 2)  // region whose heavy slots are computed: the tile and one ring
__global__ void __launch_bounds__(256) k_plan_heavy_tile(const u8 *__restrict__ ncode, u32 nrow, u32 ncol,
                                                         const u8 *__restrict__ lh, const u8 *__restrict__ kids,
                                                         const u32 *__restrict__ upa, uint16_t *__restrict__ hinfo,
                                                         u8 *__restrict__ hcode) {
  __shared__ u32 sU[HPW * HPW];
  __shared__ u8 sL[HPW * HPW];
  __shared__ u8 sK[HRW * HRW], sH[HRW * HRW];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const i64 r0 = (i64)by_ * XT, c0 = (i64)bx_ * XT;
  // (all loads of the staging first, from clamped addresses: one round trip, not one per loop iteration)
  constexpr int NU = (HPW * HPW + 255) / 256, NK = (HRW * HRW + 255) / 256;
  u32 ua[NU], la[NU], ka[NK];
  u32 inu = 0, ink = 0;
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const u32 idx = tid + 256u * (u32)i;
    const i64 gr = r0 - 2 + (i64)(idx / HPW), gc = c0 - 2 + (i64)(idx % HPW);
    const bool inb = idx < HPW * HPW && gr >= 0 && gr < (i64)nrow && gc >= 0 && gc < (i64)ncol;
    const size_t g = inb ? (size_t)gr * ncol + (size_t)gc : 0;
    ua[i] = upa[g];
    la[i] = lh[g];
    inu |= inb ? 1u << i : 0u;
  }
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const u32 idx = tid + 256u * (u32)i;
    const i64 gr = r0 - 1 + (i64)(idx / HRW), gc = c0 - 1 + (i64)(idx % HRW);
    const bool inb = idx < HRW * HRW && gr >= 0 && gr < (i64)nrow && gc >= 0 && gc < (i64)ncol;
    ka[i] = kids[inb ? (size_t)gr * ncol + (size_t)gc : 0];
    ink |= inb ? 1u << i : 0u;
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const u32 idx = tid + 256u * (u32)i;
    if (idx < HPW * HPW) {
      const bool inb = (inu >> i) & 1u;
      sU[idx] = inb ? ua[i] : 0u;
      sL[idx] = (u8)(inb ? la[i] : XL_NODATA);
    }
  }
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const u32 idx = tid + 256u * (u32)i;
    if (idx < HRW * HRW) sK[idx] = (u8)(((ink >> i) & 1u) ? ka[i] : 0u);
  }
  __syncthreads();
  // heavy slot of every cell of the region: its upstream TRUNK cell with the largest upstream area, first maximum in
  // ascending linear index (8: none, or no trunk cell)
  for (u32 idx = tid; idx < HRW * HRW; idx += 256u) {
    const u32 lr = idx / HRW + 1u, lc = idx % HRW + 1u;  // staged coordinates
    u32 hs = 8;
    if (xl_trunk(sL[lr * HPW + lc])) {
      const u32 m = sK[idx];
      // (the eight neighbours' marks and areas first, unconditionally — they lie inside the staged area —, then the
      //  selection in registers: under `if (m & bit)` every one of the 16 LDS loads was waited for on the spot)
      u32 nl[8], nu[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const u32 j = (u32)((int)lr + d8_dr(k)) * HPW + (u32)((int)lc + d8_dc(k));
        nl[k] = sL[j];
        nu[k] = sU[j];
      }
      u32 best = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = (q == 0) ? 5 : (q == 1) ? 6 : (q == 2) ? 7 : (q == 3) ? 4 : (q == 4) ? 0 : (q == 5) ? 3 : (q == 6) ? 2 : 1;  // PFD_SLOT_ASC
        const bool take = (m & (1u << k)) && xl_trunk(nl[k]) && nu[k] > best;
        best = take ? nu[k] : best;
        hs = take ? (u32)k : hs;
      }
    }
    sH[idx] = (u8)hs;
  }
  __syncthreads();
  u32 ca[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {  // (the codes of the thread's cells, loaded together)
    const u32 l = tid + 256u * (u32)jj;
    const i64 gr = r0 + (l >> 6), gc = c0 + (l & 63u);
    ca[jj] = ncode[(gr < (i64)nrow && gc < (i64)ncol) ? (size_t)gr * ncol + (size_t)gc : 0];
  }
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const u32 l = tid + 256u * (u32)jj;
    const u32 lr = l >> 6, lc = l & 63u;
    const i64 gr = r0 + lr, gc = c0 + lc;
    if (gr >= (i64)nrow || gc >= (i64)ncol) continue;
    const size_t g = (size_t)gr * ncol + (size_t)gc;
    u32 info = 0, hc = D8_MV;
    if (xl_trunk(sL[(lr + 2u) * HPW + lc + 2u])) {
      const u32 ri = (lr + 1u) * HRW + lc + 1u;
      const u32 m = sK[ri], hs = sH[ri];
      // post cells: upstream cells the serial loop adds after the heavy one = those of lower linear index
      u32 npost = 0;
      if (hs < 8) {
        const i64 hoff = (i64)d8_dr((int)hs) * (i64)ncol + d8_dc((int)hs);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const i64 off = (i64)d8_dr(k) * (i64)ncol + d8_dc(k);
          if ((m & (1u << k)) && off < hoff) ++npost;
        }
      }
      info = m | (hs << 8) | (npost << 12);
      // heavy link: the heavy slot of the downstream cell holds this cell (a halo cell of a row block is in no chain —
      // it is no trunk cell: the cell draining into it ends its own chain)
      const u32 c = ca[jj];
      bool heavy = false;
      if (d8_is_dir(c)) {
        const int k = d8_slot(c);
        const u32 pr = (u32)((int)lr + 1 + d8_dr(k)), pc = (u32)((int)lc + 1 + d8_dc(k));  // region coordinates
        const u32 ps = sH[pr * HRW + pc];
        heavy = ps < 8 && ((k + 4) & 7) == (int)ps && xl_trunk(sL[(pr + 1u) * HPW + pc + 1u]);
      }
      hc = heavy ? c : 0u;
    }
    hinfo[g] = (uint16_t)info;
    hcode[g] = (u8)hc;
  }
}
// chain ends (pits of the heavy forest: hcode 0) per 4096 cells in raster order — what the list of chain ends is offset by
// — and the list itself: position = ends in the groups before (exclusive scan of the counts) + ends before the cell in
// its own group.  16 cells (one 16-byte load) per thread: with one byte per thread the two passes cost 2.1 + 1.8 ms at
// 30000 x 30000, most of it dispatching 3.5 M workgroups.  (hcode carries 64 bytes of slack behind its n cells.)
#define TLG 4096u
__device__ __forceinline__ u32 zero_bytes(u32 x) {  // bit 7 of every byte that is 0 (exact: no carry crosses a byte)
  return ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x)) & 0x80808080u;
}
__device__ __forceinline__ void tl_load(const u8 *__restrict__ hcode, u32 n, u32 x0, u32 (&z)[4]) {
  uint4 v = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
  if (x0 < n) v = *reinterpret_cast<const uint4 *>(hcode + x0);  // (x0 is a multiple of 16: aligned)
  const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    z[i] = zero_bytes(w[i]);
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (x0 + 4u * i + b >= n) z[i] &= ~(0x80u << (8 * b));  // (cells past the raster)
  }
}
// TILED = one workgroup per 64 x 64 tile (16 cells of a tile row per thread, the list in tile-major order: the chains of a
// tile — and with them their slots in every chain-order array — lie together, see k_plan_tail order in DESIGN 4.5);
// !TILED = 4096 cells of the raster in linear order per workgroup (the list in raster order, PFD_TAILS_RASTER).
template <bool TILED>
__device__ __forceinline__ u32 tl_load_any(const u8 *__restrict__ hcode, u32 n, u32 nrow, u32 ncol, u32 ntc, u32 (&z)[4]) {
  if (!TILED) {
    const u32 x0 = blockIdx.x * TLG + 16u * threadIdx.x;
    tl_load(hcode, n, x0, z);
    return x0;
  }
  const u32 ty = blockIdx.x / ntc, tx = blockIdx.x - ty * ntc;
  const u32 gr = ty * XT + (threadIdx.x >> 2), gc = tx * XT + 16u * (threadIdx.x & 3u);
  const u32 x0 = gr * ncol + gc;  // (< n whenever the thread has a cell)
  uint4 v = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
  const bool in = gr < nrow && gc < ncol;
  if (in) __builtin_memcpy(&v, hcode + x0, 16);  // (unaligned; hcode carries 64 bytes of slack behind its n cells)
  const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    z[i] = zero_bytes(w[i]);
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (!in || gc + 4u * i + b >= ncol) z[i] &= ~(0x80u << (8 * b));  // (cells of the next row / past the raster)
  }
  return x0;
}
template <bool TILED>
__global__ void __launch_bounds__(256) k_plan_count_ends(const u8 *__restrict__ hcode, u32 n, u32 nrow, u32 ncol, u32 ntc,
                                                         u32 *__restrict__ bcount) {
  __shared__ u32 wcnt[4];
  u32 z[4];
  tl_load_any<TILED>(hcode, n, nrow, ncol, ntc, z);
  u32 c = (u32)__popc(z[0]) + (u32)__popc(z[1]) + (u32)__popc(z[2]) + (u32)__popc(z[3]);
  for (int o = 32; o > 0; o >>= 1) c += (u32)__shfl_down((int)c, o);
  if ((threadIdx.x & 63u) == 0u) wcnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) bcount[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}
template <bool TILED>
__global__ void __launch_bounds__(256) k_plan_tail_list(const u8 *__restrict__ hcode, u32 n, u32 nrow, u32 ncol, u32 ntc,
                                                        const u32 *__restrict__ boff, u32 *__restrict__ tails) {
  __shared__ u32 wcnt[4];
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  u32 z[4];
  const u32 x0 = tl_load_any<TILED>(hcode, n, nrow, ncol, ntc, z);
  const u32 c = (u32)__popc(z[0]) + (u32)__popc(z[1]) + (u32)__popc(z[2]) + (u32)__popc(z[3]);
  u32 incl = c;  // inclusive prefix over the wave's lanes
  for (int o = 1; o < 64; o <<= 1) {
    const u32 y = (u32)__shfl_up((int)incl, o);
    if (lane >= (u32)o) incl += y;
  }
  if (lane == 63u) wcnt[wave] = incl;
  __syncthreads();
  if (!c) return;
  u32 pos = boff[blockIdx.x] + incl - c;
  for (u32 w = 0; w < wave; ++w) pos += wcnt[w];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32 m = z[i];
    while (m) {
      const u32 b = (u32)__ffs((int)m) - 1u;  // bit 7 of byte b / 8
      tails[pos++] = x0 + 4u * (u32)i + (b >> 3);
      m &= m - 1u;
    }
  }
}

// the head of a chain (a trunk cell without heavy child) knows the length of its chain
// (per-chain arrays are indexed by the chain end's number j in the raster-ordered list `tails`;
//  tidx_at[cell] = j for a chain end)
__global__ void __launch_bounds__(256) k_plan_tidx(const u32 *__restrict__ tails, u32 nt, u32 *__restrict__ tidx_at) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nt) tidx_at[tails[j]] = j;
}
// (8 cells = one 16-byte load of hinfo per thread: with one cell per thread the pass spent most of its 2.2 ms at
//  30000 x 30000 dispatching 3.5 M workgroups for 1.8 GB)
__device__ __forceinline__ void hinfo8(const uint16_t *__restrict__ hinfo, u32 x0, u32 n, u32 (&inf)[8]) {
  if ((u64)x0 + 8ull <= (u64)n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(hinfo + x0);  // (x0 is a multiple of 8: aligned)
    const u32 d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) inf[2 * i] = d[i] & 0xFFFFu, inf[2 * i + 1] = d[i] >> 16;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) inf[i] = (u64)x0 + (u64)i < (u64)n ? (u32)hinfo[x0 + i] : 0u;
  }
}
__global__ void __launch_bounds__(256) k_plan_len(const uint16_t *__restrict__ hinfo, const u32 *__restrict__ hops,
                                                  const u32 *__restrict__ tailnum, const u32 *__restrict__ tidx_at, u32 n,
                                                  u32 *__restrict__ len_of) {
  const u64 x64 = 8ull * ((u64)blockIdx.x * blockDim.x + threadIdx.x);  // (n may lie within 2048 cells of 2^32)
  if (x64 >= (u64)n) return;
  const u32 x0 = (u32)x64;
  u32 inf[8];
  hinfo8(hinfo, x0, n, inf);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (((inf[i] >> 8) & 0xFu) != 8u) continue;  // (hinfo is 0 on cells that are no trunk cells; 8: no heavy child = head)
    const u32 tn = tailnum[x0 + i];
    if (tn) len_of[tidx_at[tn - 1]] = hops[x0 + i] + 1;
  }
}

// Round of a chain = 31 - (number of light cells between its last cell and the pit): the chain a
// tributary joins runs one round later, and all the main stems (chains ending in a pit) share the LAST
// round, so independent long chains never queue up behind one another.  With heavy = largest upstream
// area a path crosses at most log2(n) <= 32 light cells.  The count is a pointer doubling over the chain
// ends only (6 rounds cover 64 links): D(t) += D(P(t)), P(t) = P(P(t)), gather-only, double-buffered.
__global__ void __launch_bounds__(256) k_plan_tails(const u8 *__restrict__ ncode, Geo g, const u32 *__restrict__ tails,
                                                    u32 nt, const u32 *__restrict__ tailnum,
                                                    const u32 *__restrict__ tidx_at, u32 *__restrict__ D,
                                                    u32 *__restrict__ P, u32 *__restrict__ par) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  const u32 x = tails[j];
  u32 d = 0, p = NONE32;
  const u32 c = ncode[x];
  if (d8_is_dir(c) && ncode[d8_down(g, x, c)] != D8_HALO) {  // (into a halo cell of a row block: like a pit, nothing to join)
    d = 1;
    p = tidx_at[tailnum[d8_down(g, x, c)] - 1u];  // the end of the chain this one joins (a trunk cell: never 0)
  }
  D[j] = d;
  P[j] = p;
  par[j] = p;
}
__global__ void __launch_bounds__(256) k_plan_depth_round(u32 nt, const u32 *__restrict__ Di, const u32 *__restrict__ Pi,
                                                          u32 *__restrict__ Do, u32 *__restrict__ Po) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  u32 d = Di[j], p = Pi[j];
  if (p != NONE32) {
    d += Di[p];
    p = Pi[p];
  }
  Do[j] = d;
  Po[j] = p;
}

// ---- rounds of the chains (round 6) ----------------------------------------------------------------------------------
// Any labelling with label(chain) < label(chain it joins) is a valid order of the rounds.  "As late as possible" — 31 minus
// the light links below the chain — puts all main stems into the last round, which is what the few LONG chains need (each
// is folded serially by one wave: they must run side by side, not one round after the other).  But it spreads the ~1.3e7
// short headwater chains of a 30000 x 30000 raster over all 13 rounds by how deep in the tree they happen to sit, and every round is a
// launch of its own: a 64 x 64 tile's chains are then gathered in 8 - 13 different launches and every launch fetches
// nearly all lines of the tile's payload for a few hundred slots.  So: a chain of >= XPIN cells keeps the late label
// (PINNED, bit 7 set so that an atomicMax below 128 never moves it); every other chain takes the EARLIEST round its
// tributaries allow, 1 + max(label of the chains joining it), starting at the first round in use.  A pinned tributary's
// label is below its parent's late label, so by induction no label exceeds the late one: the order stays valid and the
// number of rounds the same.  Pass k pushes the labels of the chains whose label is k (final by then: everything below
// pushed in an earlier pass; a push from a label that rises later is only a lower bound pushed too early).
#define XPIN 512u
#define XLAB_PIN 0x80u
__global__ void __launch_bounds__(256) k_plan_maxdepth(u32 nt, const u32 *__restrict__ depth, u32 *__restrict__ maxd) {
  __shared__ u32 s_m;
  if (threadIdx.x == 0) s_m = 0;
  __syncthreads();
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  u32 d = j < nt ? min(depth[j], 31u) : 0u;
  for (int o = 32; o > 0; o >>= 1) d = max(d, (u32)__shfl_down((int)d, o));
  if ((threadIdx.x & 63u) == 0u && d) atomicMax(&s_m, d);
  __syncthreads();
  if (threadIdx.x == 0 && s_m) atomicMax(maxd, s_m);
}
__global__ void __launch_bounds__(256) k_plan_label_init(u32 nt, const u32 *__restrict__ depth, const u32 *__restrict__ len_of,
                                                         const u32 *__restrict__ maxd, u32 *__restrict__ lab) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  lab[j] = len_of[j] >= XPIN ? (XLAB_PIN | (31u - min(depth[j], 31u))) : 31u - *maxd;
}
__global__ void __launch_bounds__(256) k_plan_label_pins(u32 nt, const u32 *__restrict__ par, u32 *__restrict__ lab) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  const u32 l = lab[j];
  if (!(l & XLAB_PIN)) return;
  const u32 p = par[j];
  if (p != NONE32) atomicMax(&lab[p], (l & 31u) + 1u);  // (a pinned parent holds >= 128: unchanged)
}
__global__ void __launch_bounds__(256) k_plan_label_pass(u32 nt, u32 k, const u32 *__restrict__ par, u32 *__restrict__ lab) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  if (lab[j] != k) return;
  const u32 p = par[j];
  if (p != NONE32) atomicMax(&lab[p], k + 1u);
}

// chains in layout order = the chain ends sorted by round (stable: raster order inside a round);
// the sort carries the list numbers j, rank_of[j] = chain id
__global__ void __launch_bounds__(256) k_plan_keys(u32 nt, const u32 *__restrict__ depth, const u32 *__restrict__ lab, u32 *__restrict__ keys,
                                                   u32 *__restrict__ iota, u32 *__restrict__ hist) {
  __shared__ u32 s_h[32];
  if (threadIdx.x < 32) s_h[threadIdx.x] = 0;
  __syncthreads();
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nt) {
    const u32 k = lab ? (lab[j] & 31u) : 31u - min(depth[j], 31u);
    keys[j] = k;
    iota[j] = j;
    atomicAdd(&s_h[k], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 32 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_plan_chain_lens(const u32 *__restrict__ cj, u32 nchain,
                                                         const u32 *__restrict__ tails, const u32 *__restrict__ len_of,
                                                         u32 *__restrict__ rank_of, u32 *__restrict__ ctail,
                                                         u32 *__restrict__ clen_pos, u32 *__restrict__ cpos_in,
                                                         u32 *__restrict__ tidx_at) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nchain) return;
  u32 l = 0;
  if (c < nchain) {
    const u32 j = cj[c];
    l = len_of[j];
    rank_of[j] = c;
    ctail[c] = tails[j];
    clen_pos[c] = l;
    tidx_at[tails[j]] = c;  // (from here on the chain end's cell names its chain, not its list number)
  }
  cpos_in[c] = l;  // (exclusive scan in place -> first position of the chain)
}

// the LAST position of a chain, where its end cell finds it: k_plan_scatter's cells look their chain up through their end
// cell (tailnum) — with this array the position is one gather beside the chain id instead of two behind it
__global__ void __launch_bounds__(256) k_plan_pend(const u32 *__restrict__ ctail, const u32 *__restrict__ cpos,
                                                   const u32 *__restrict__ clen_pos, u32 nchain, u32 *__restrict__ pend_at) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nchain) pend_at[ctail[c]] = cpos[c] + clen_pos[c] - 1u;
}
// position of every trunk cell: chain base + distance from the chain head; w = slots the cell needs.
// One workgroup per 64 x 64 TILE (round 5; a strip of 256 cells of a raster row before): the records land in chain
// order, and a chain crosses a tile in a run of ~64 consecutive positions — 1 KB of 16-byte records that the L2 merges
// before they leave — where a row strip meets every chain once and pays a sector per record (the same observation as
// k_xtrunk_unscatter's: 7.5 -> see DESIGN 4.5).  The position also goes into ptmp[x] (the plan's cslot array, coalesced):
// k_plan_cslot turns it into the slot once the slots are known, so that nobody stores into raster order from chain order.
__global__ void __launch_bounds__(256) k_plan_scatter(const uint16_t *__restrict__ hinfo, const u32 *__restrict__ hops,
                                                      const u32 *__restrict__ tailnum, const u32 *__restrict__ tidx_at,
                                                      const u32 *__restrict__ pend_at, u32 nrow,
                                                      u32 ncol, uint4 *__restrict__ urec, u32 *__restrict__ ptmp,
                                                      u32 *__restrict__ tl_cnt) {
  __shared__ u32 s_tc[4];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 r0 = by_ * XT, c0 = bx_ * XT;
  u32 mine = 0;  // trunk cells with a position among the thread's 16 cells (-> tl_cnt[tile])
  u32 inf[4][4], tn[4][4], hp[4][4], x0s[4];
  bool any[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // all loads of the four quads first
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 gr = r0 + (l0 >> 6), gc = c0 + (l0 & 63);
    x0s[j] = gr * ncol + gc;
    any[j] = false;
#pragma unroll
    for (int b = 0; b < 4; ++b) inf[j][b] = tn[j][b] = hp[j][b] = 0;
    if (gr >= nrow || gc >= ncol) continue;
    if (gc + 3 < ncol) {
      uint2 i4;
      __builtin_memcpy(&i4, hinfo + x0s[j], 8);
      inf[j][0] = i4.x & 0xFFFFu, inf[j][1] = i4.x >> 16, inf[j][2] = i4.y & 0xFFFFu, inf[j][3] = i4.y >> 16;
    } else {
      for (u32 b = 0; b < 4u && gc + b < ncol; ++b) inf[j][b] = hinfo[x0s[j] + b];
    }
    any[j] = (inf[j][0] | inf[j][1] | inf[j][2] | inf[j][3]) != 0u;
  }
  // (NOTES item 1 once more: a load under a per-cell branch is waited for on the spot — 2 x 16 serialised round trips per
  //  thread here, then 2 x 16 dependent ones.  A quad that holds a trunk cell loads its four tail numbers and hops as two
  //  16-byte vectors; the chain id / chain end of a cell come from a clamped index, all of a quad's in flight, the mark selects)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!any[j]) continue;
    if (c0 + (4u * tid & 63u) + 3u < ncol) {
      uint4 t4, h4;
      __builtin_memcpy(&t4, tailnum + x0s[j], 16);
      __builtin_memcpy(&h4, hops + x0s[j], 16);
      tn[j][0] = t4.x, tn[j][1] = t4.y, tn[j][2] = t4.z, tn[j][3] = t4.w;
      hp[j][0] = h4.x, hp[j][1] = h4.y, hp[j][2] = h4.z, hp[j][3] = h4.w;
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (inf[j][b]) tn[j][b] = tailnum[x0s[j] + b], hp[j][b] = hops[x0s[j] + b];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) tn[j][b] = inf[j][b] ? tn[j][b] : 0u;
  }
  u32 cid[4][4], pe[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int b = 0; b < 4; ++b) cid[j][b] = pe[j][b] = 0;
    if (!any[j]) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u32 t = tn[j][b] ? tn[j][b] - 1u : 0u;  // (clamped: the loads are unconditional inside the quad)
      cid[j][b] = tidx_at[t];
      pe[j][b] = pend_at[t];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!any[j]) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (!inf[j][b]) continue;
      const u32 x = x0s[j] + (u32)b;
      if (!tn[j][b]) {  // (a trunk cell whose heavy path has no end: cannot happen on an acyclic forest)
        ptmp[x] = NONE32;
        continue;
      }
      const u32 c = cid[j][b];  // (chain ids since k_plan_chain_lens)
      const u32 p = pe[j][b] - hp[j][b];
      // one 16-byte record per position — cell, chain, hinfo — so that k_plan_expand, which runs in position order, finds
      // everything in one coalesced load instead of five dependent gathers per cell
      urec[p] = make_uint4(x, c, inf[j][b], 0u);
      ptmp[x] = p;
      ++mine;
    }
  }
  for (int o = 32; o > 0; o >>= 1) mine += (u32)__shfl_down((int)mine, o);
  if ((tid & 63u) == 0u) s_tc[tid >> 6] = mine;
  __syncthreads();
  if (tid == 0) tl_cnt[by_ * gridDim.x + bx_] = s_tc[0] + s_tc[1] + s_tc[2] + s_tc[3];
}
// cslot[x]: position -> slot (spos[p], written by k_plan_expand in position order), and the number of post slots into
// the trunk mark; tile-shaped like the scatter (the gather from chain order hits the runs the tile holds)
__global__ void __launch_bounds__(256) k_plan_cslot(const uint16_t *__restrict__ hinfo, const u32 *__restrict__ spos, u32 nrow, u32 ncol, u32 *__restrict__ cslot,
                                                    u8 *__restrict__ lh, const u32 *__restrict__ tl_off, uint2 *__restrict__ tlist) {
  __shared__ u32 s_tw[4];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 r0 = by_ * XT, c0 = bx_ * XT;
  u32 esl[16], eps[16], emask = 0;  // slot / post slots of the thread's cells that join the list (static indices only)
#pragma unroll
  for (int k = 0; k < 16; ++k) esl[k] = eps[k] = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 gr = r0 + (l0 >> 6), gc = c0 + (l0 & 63);
    if (gr >= nrow || gc >= ncol) continue;
    const u32 x0 = gr * ncol + gc;
    u32 inf[4] = {0, 0, 0, 0};
    if (gc + 3 < ncol) {
      uint2 i4;
      __builtin_memcpy(&i4, hinfo + x0, 8);
      inf[0] = i4.x & 0xFFFFu, inf[1] = i4.x >> 16, inf[2] = i4.y & 0xFFFFu, inf[3] = i4.y >> 16;
    } else {
      for (u32 b = 0; b < 4u && gc + b < ncol; ++b) inf[b] = hinfo[x0 + b];
    }
    if (!(inf[0] | inf[1] | inf[2] | inf[3])) continue;
    u32 pp[4];
    if (gc + 3 < ncol) {
      __builtin_memcpy(pp, cslot + x0, 16);  // (positions of the trunk cells of the quad; whatever the others hold)
    } else {
      for (u32 b = 0; b < 4u; ++b) pp[b] = gc + b < ncol ? cslot[x0 + b] : NONE32;
    }
    u32 sl[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {  // (unconditional gathers from clamped positions, the select afterwards)
      const bool ok = inf[b] && pp[b] != NONE32;
      const u32 v = spos[ok ? pp[b] : 0u];
      sl[b] = ok ? v : 0u;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (!inf[b] || pp[b] == NONE32) continue;
      cslot[x0 + b] = sl[b];  // (the sweeps find a trunk cell's value in chain order through it: no scatter pass per round)
      if ((inf[b] >> 12) & 7u) lh[x0 + b] = (u8)(XL_TRUNK + ((inf[b] >> 12) & 7u));  // (up-sweeps: the value sits behind the post slots)
      esl[4 * j + b] = sl[b], eps[4 * j + b] = (inf[b] >> 12) & 7u;
      emask |= 1u << (4 * j + b);
    }
  }
  const u32 ne = (u32)__popc(emask);
  // the tile's dense list: exclusive prefix of the threads' counts (the order inside a tile does not matter)
  const u32 lane = tid & 63u, wave = tid >> 6;
  u32 incl = ne;
  for (int o = 1; o < 64; o <<= 1) {
    const u32 y = (u32)__shfl_up((int)incl, o);
    if (lane >= (u32)o) incl += y;
  }
  if (lane == 63u) s_tw[wave] = incl;
  __syncthreads();
  u32 pos = tl_off[by_ * gridDim.x + bx_] + incl - ne;
  for (u32 w = 0; w < wave; ++w) pos += s_tw[w];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if ((emask >> k) & 1u) {
      const u32 local = 4u * tid + 1024u * (u32)(k >> 2) + (u32)(k & 3);
      tlist[pos++] = make_uint2(esl[k], local | (eps[k] << 12));
    }
  }
}
struct XRecSlots {  // slots a position needs: 1 + its post slots (0 for a position nobody wrote)
  __device__ u32 operator()(const uint4 &r) const { return r.z ? 1u + ((r.z >> 12) & 7u) : 0u; }
};

// chain c occupies the unpadded slots [S[cpos], S[cpos + len)); every chain starts on a multiple of 4 slots
// (the lane that folds it loads and stores 4 slots per instruction): cpad = its padded length
__global__ void __launch_bounds__(256) k_plan_chain_len(const u32 *__restrict__ cpos, const u32 *__restrict__ clen_pos,
                                                        const u32 *__restrict__ S, u32 nchain, u32 *__restrict__ cpad) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nchain) return;
  u32 v = 0;
  if (c < nchain) {
    const u32 p = cpos[c];
    v = (S[p + clen_pos[c]] - S[p] + 3u) & ~3u;
  }
  cpad[c] = v;
}
// cstart = padded start; clen = slots | trailing post slots of the last cell << 29; adj_at[tail] = what
// turns an unpadded slot number of this chain into the padded one
__global__ void __launch_bounds__(256) k_plan_chains(const u32 *__restrict__ cpos, const u32 *__restrict__ clen_pos,
                                                     const u32 *__restrict__ ctail, const u32 *__restrict__ S,
                                                     const u32 *__restrict__ cstart_pad,
                                                     const uint16_t *__restrict__ hinfo, u32 nchain,
                                                     u32 *__restrict__ cstart, u32 *__restrict__ clen,
                                                     u32 *__restrict__ adj) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchain) return;
  const u32 p = cpos[c], l = clen_pos[c];
  const u32 u0 = S[p];
  const u32 t = ctail[c];
  cstart[c] = cstart_pad[c];
  clen[c] = (S[p + l] - u0) | ((((u32)hinfo[t] >> 12) & 7u) << 29);
  adj[c] = cstart_pad[c] - u0;
}
// (S is read at the thread's own position only: the padded slot of the position replaces it in place — k_plan_cslot
//  carries it to the cell in raster order)
__global__ void __launch_bounds__(256) k_plan_expand(const uint4 *__restrict__ urec, u32 *__restrict__ S,
                                                     const u32 *__restrict__ adj, Geo g, u32 npos, u32 *__restrict__ scell,
                                                     uint16_t *__restrict__ sinfo, u32 *__restrict__ spost) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npos) return;
  const uint4 rec = urec[p];
  const u32 x = rec.x;
  const u32 info = rec.z;
  if (!info) return;  // (a position nobody wrote: cannot happen on a consistent forest)
  u32 s = S[p] + adj[rec.y];
  S[p] = s;
  scell[s] = x;
  sinfo[s] = (uint16_t)info;
  const u32 hs = (info >> 8) & 0xFu;
  if (hs < 8 && (info >> 12)) {
    const i64 hoff = (i64)d8_dr((int)hs) * (i64)g.ncol + d8_dc((int)hs);
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // descending linear index = the serial loop's order
      const int k = PFD_SLOT_DESC[q];
      const i64 off = (i64)d8_dr(k) * (i64)g.ncol + d8_dc(k);
      if ((info & (1u << k)) && off < hoff) {
        ++s;
        scell[s] = (u32)((i64)x + off);
        sinfo[s] = (uint16_t)XS_POST;
        atomicOr(&spost[s >> 5], 1u << (s & 31u));
      }
    }
  }
}
__global__ void __launch_bounds__(256) k_plan_long_lens(const u32 *__restrict__ longc, u32 nl, const u32 *__restrict__ clen,
                                                        u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nl) out[i] = clen[longc[i]] & XC_LEN;
}
__global__ void k_plan_pick(const u32 *__restrict__ S, const u32 *__restrict__ idx, u32 k, u32 *__restrict__ out) {
  const u32 t = threadIdx.x;
  if (t < k) out[t] = S[idx[t]];
}

// debugging aid (builds with DEVTOOLS=1 only, env PFD_XPLAN_DIGEST): 64-bit sum of a device array after a build step, printed to stderr —
// two builds of the same raster must print identical lines (tools/plan_determinism.py)
__global__ void __launch_bounds__(256) k_digest(const u32 *__restrict__ v, u64 nwords, unsigned long long *__restrict__ res) {
  unsigned long long s = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (u64)gridDim.x * 256) s += (unsigned long long)v[i] * (i % 1000003ull + 1ull);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(res, s);
}
static void xdigest(pfd_raster *h, const char *name, const void *p, size_t bytes) {
#ifndef PFD_DEVTOOLS
  (void)h, (void)name, (void)p, (void)bytes;
  return;
#else
  if (!getenv("PFD_XPLAN_DIGEST") || !p) return;
  unsigned long long *acc = nullptr, host = 0;
  (void)hipStreamSynchronize(h->stream);
  if (hipMalloc((void **)&acc, 8) != hipSuccess) return;
  (void)hipMemset(acc, 0, 8);
  k_digest<<<2048, 256>>>((const u32 *)p, (u64)(bytes / 4), acc);
  (void)hipMemcpy(&host, acc, 8, hipMemcpyDeviceToHost);
  (void)hipFree(acc);
  fprintf(stderr, "[xdigest] %-10s %016llx\n", name, host);
#endif
}

// ---------------------------------------------------------------------------------------------
// incremental re-sweeps of a row block (exact.h): chain of a slot, chain below a chain, chain below a halo cell
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_xinc_schain(const u32 *__restrict__ cstart, u32 nchain, u32 nslot, u32 *__restrict__ schain) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslot) return;
  u32 lo = 0, hi = nchain;  // the last chain that starts at or before s (cstart ascends; padding belongs to the chain before it)
  while (hi - lo > 1u) {
    const u32 mid = (lo + hi) >> 1;
    if (cstart[mid] <= s) lo = mid;
    else hi = mid;
  }
  schain[s] = lo;
}
__global__ void __launch_bounds__(256) k_xinc_dchain(const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                     const u32 *__restrict__ scell, const u8 *__restrict__ ncode,
                                                     const u8 *__restrict__ lh, const u32 *__restrict__ cslot,
                                                     const u32 *__restrict__ schain, Geo g, u32 nchain, u32 *__restrict__ dchain) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchain) return;
  const u32 cl = clen[c], np = cl >> 29;
  const u32 x = scell[cstart[c] + (cl & XC_LEN) - 1u - np];  // the chain's last cell
  const u32 code = ncode[x];
  u32 d = 0xFFFFFFFFu;
  if (d8_is_dir(code)) {
    const u32 y = d8_down(g, x, code);
    if (xl_trunk(lh[y])) d = schain[cslot[y]];  // (a halo cell or a pit below: nothing to fold again)
  }
  dchain[c] = d;
}
// ctrl[0] counts halo cells that drain into an own cell which is no trunk cell (k_plan_tile blocks those: must stay 0)
__global__ void __launch_bounds__(256) k_xinc_hfeed(const u8 *__restrict__ halo_raw, const u8 *__restrict__ ncode,
                                                    const u8 *__restrict__ lh, const u32 *__restrict__ cslot,
                                                    const u32 *__restrict__ schain, u32 ncol, u32 halo_top, u32 halo_bot,
                                                    u32 own_rows, u32 *__restrict__ hfeed, unsigned long long *__restrict__ ctrl) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2u * ncol) return;
  const u32 side = i >= ncol ? 1u : 0u, gc = i - side * ncol;
  u32 f = 0xFFFFFFFFu;
  if (side ? halo_bot : halo_top) {
    const u32 hr = side ? halo_top + own_rows : halo_top - 1u;
    const u32 raw = halo_raw[i];
    const u32 x = hr * ncol + gc;
    if (ncode[x] != D8_MV && d8_is_dir(raw)) {
      const int k = d8_slot(raw);
      const int tgc = (int)gc + d8_dc(k);
      if (d8_dr(k) == (side ? -1 : 1) && tgc >= 0 && tgc < (int)ncol) {  // (only a step into the own rows links the halo cell)
        const u32 y = (u32)((int)hr + d8_dr(k)) * ncol + (u32)tgc;
        if (ncode[y] != D8_MV) {
          if (xl_trunk(lh[y])) f = schain[cslot[y]];
          else atomicAdd(ctrl, 1ull);
        }
      }
    }
  }
  hfeed[i] = f;
}
template <class W>
__global__ void __launch_bounds__(256) k_xinc_mark(const W *__restrict__ seed, W *__restrict__ prev, const u32 *__restrict__ hfeed,
                                                   u32 n2, u8 *__restrict__ dirty) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  const W v = seed[i];
  if (v == prev[i]) return;  // (bitwise: W is an unsigned integer of the element's size)
  prev[i] = v;
  const u32 f = hfeed[i];
  if (f != 0xFFFFFFFFu) dirty[f] = 1;
}

void pfd_xinc_drop(pfd_raster *h) {
  ExactPlan *p = (ExactPlan *)h->xplan;
  if (!p) return;
  pfd_dfree(p->incE);
  pfd_dfree(p->incR);
  pfd_dfree(p->incSeed);
  p->incE = p->incR = p->incSeed = nullptr;
  h->bytes_held -= std::min(h->bytes_held, p->inc_bytes);
  p->inc_bytes = 0;
  p->inc_valid = false;
  p->inc_out = nullptr;
}
// releases the static maps of the incremental sweeps (a failed or half-run prepare must not look like a finished one)
static void xinc_free_maps(pfd_raster *h, ExactPlan *p) {
  pfd_dfree(p->schain), pfd_dfree(p->dchain), pfd_dfree(p->hfeed), pfd_dfree(p->dirty);
  p->schain = p->dchain = p->hfeed = nullptr, p->dirty = nullptr;
  p->bytes -= std::min(p->bytes, p->xinc_map_bytes), h->bytes_held -= std::min(h->bytes_held, p->xinc_map_bytes);
  p->xinc_map_bytes = 0;
  p->xinc_ready = false;
}
static int xinc_build_maps(pfd_raster *h, ExactPlan *p, size_t nsl, size_t nch, u64 *odd) {
  pfd_seg_begin(h, "xinc_prepare");
  HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(u64), h->stream));
  HIPCHK(hipMemsetAsync(p->schain, 0, nsl * sizeof(u32), h->stream));
  if (p->nchain) {
    k_xinc_schain<<<cdiv_u32((u64)p->nslot, 256), 256, 0, h->stream>>>(p->cstart, (u32)p->nchain, (u32)p->nslot, p->schain);
    k_xinc_dchain<<<cdiv_u32((u64)p->nchain, 256), 256, 0, h->stream>>>(p->cstart, p->clen, p->scell, h->ncode, p->lh, p->cslot,
                                                                        p->schain, h->geo, (u32)p->nchain, p->dchain);
  }
  k_xinc_hfeed<<<cdiv_u32(2 * (u64)h->ncol, 256), 256, 0, h->stream>>>(h->halo_raw, h->ncode, p->lh, p->cslot, p->schain,
                                                                       (u32)h->ncol, (u32)h->halo_top, (u32)h->halo_bot,
                                                                       (u32)h->own_rows, p->hfeed, (unsigned long long *)h->ctrl);
  KCHK();
  pfd_seg_end(h, 3);
  HIPCHK(hipMemcpyAsync(odd, h->ctrl, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}
int pfd_xinc_prepare(pfd_raster *h) {
  ExactPlan *p = (ExactPlan *)h->xplan;
  if (!p || h->xplan_state != 1 || !h->halo_raw) {
    pfd_set_error("incremental block sweeps need the exact-order plan of a row block");
    return PFD_EINVAL;
  }
  if (p->xinc_ready) return PFD_OK;  // (set only after the maps were built AND checked)
  if (p->xinc_refused) {             // (a property of the plan: asking again builds the same maps)
    pfd_set_error("incremental block sweeps: a halo cell drains into a cell that is not a trunk cell of the block's plan");
    return PFD_EUNSUPPORTED;
  }
  xinc_free_maps(h, p);  // (whatever an earlier, failed attempt left)
  const size_t nsl = std::max<size_t>((size_t)p->nslot, 4), nch = std::max<size_t>((size_t)p->nchain, 1);
  int rc;
  if ((rc = pfd_dmalloc((void **)&p->schain, nsl * sizeof(u32))) != PFD_OK ||
      (rc = pfd_dmalloc((void **)&p->dchain, nch * sizeof(u32))) != PFD_OK ||
      (rc = pfd_dmalloc((void **)&p->hfeed, 2 * (size_t)h->ncol * sizeof(u32))) != PFD_OK ||
      (rc = pfd_dmalloc((void **)&p->dirty, nch)) != PFD_OK) {
    xinc_free_maps(h, p);
    return rc;
  }
  p->xinc_map_bytes = nsl * 4 + nch * 5 + 2 * (size_t)h->ncol * 4;
  p->bytes += p->xinc_map_bytes, h->bytes_held += p->xinc_map_bytes;
  u64 odd = 0;
  rc = xinc_build_maps(h, p, nsl, nch, &odd);
  if (rc != PFD_OK) {
    (void)hipStreamSynchronize(h->stream);  // (nothing may still write the maps when they are freed)
    xinc_free_maps(h, p);
    return rc;
  }
  if (odd) {
    xinc_free_maps(h, p);
    p->xinc_refused = true;
    pfd_set_error("internal: %llu halo cells drain into a cell that is not a trunk cell of the block's plan", (unsigned long long)odd);
    return PFD_EUNSUPPORTED;
  }
  p->xinc_ready = true;
  return PFD_OK;
}
int pfd_xinc_mark(pfd_raster *h, const void *seed_dev, size_t elem) {
  ExactPlan *p = (ExactPlan *)h->xplan;
  const u32 n2 = 2u * (u32)h->ncol;
  HIPCHK(hipMemsetAsync(p->dirty, 0, std::max<size_t>((size_t)p->nchain, 1), h->stream));
  const u32 grid = cdiv_u32(n2, 256);
  if (elem == 1) k_xinc_mark<u8><<<grid, 256, 0, h->stream>>>((const u8 *)seed_dev, (u8 *)p->incSeed, p->hfeed, n2, p->dirty);
  else if (elem == 4) k_xinc_mark<u32><<<grid, 256, 0, h->stream>>>((const u32 *)seed_dev, (u32 *)p->incSeed, p->hfeed, n2, p->dirty);
  else if (elem == 8) k_xinc_mark<u64><<<grid, 256, 0, h->stream>>>((const u64 *)seed_dev, (u64 *)p->incSeed, p->hfeed, n2, p->dirty);
  else {
    pfd_set_error("incremental block sweep: element size %zu", elem);
    return PFD_EINVAL;
  }
  KCHK();
  return PFD_OK;
}

void pfd_free_xplan(pfd_raster *h) {
  ExactPlan *p = (ExactPlan *)h->xplan;
  if (p) {
    pfd_xinc_drop(h);
    pfd_dfree(p->schain);
    pfd_dfree(p->dchain);
    pfd_dfree(p->hfeed);
    pfd_dfree(p->dirty);
    pfd_dfree(p->lh);
    pfd_dfree(p->kids);
    pfd_dfree(p->tord);
    pfd_dfree(p->toff);
    pfd_dfree(p->cslot);
    pfd_dfree(p->tlist);
    pfd_dfree(p->tl_off);
    pfd_dfree(p->scell);
    pfd_dfree(p->sinfo);
    pfd_dfree(p->spost);
    pfd_dfree(p->cstart);
    pfd_dfree(p->clen);
    pfd_dfree(p->longc);
    h->bytes_held -= std::min(h->bytes_held, p->bytes);
    delete p;
  }
  h->xplan = nullptr;
  h->xplan_state = 0;
}

int pfd_ensure_xplan(pfd_raster *h, bool allow_block) {
  if ((h->halo_top || h->halo_bot) && !allow_block) return pfd_require_whole(h, "the exact-order plan");
  if (h->xplan_state != 0) return PFD_OK;
  h->xplan_state = -1;
  if (h->gen) return PFD_OK;
  if (h->n > 4294967294ll || pfd_knob("PFD_EXACT_LEVELS")) return PFD_OK;
  const bool block = h->halo_top || h->halo_bot;
  // (row blocks: the cells of the halo rows hold given values, see k_plan_tile; the codes of the halo rows as given
  //  must still be around)
  if (block && (!h->halo_raw || pfd_knob("PFD_BLOCK_LEVELS"))) return PFD_OK;
  if (h->acyclic < 0) return PFD_OK;
  const u32 n = h->geo.n;
  const u32 ntr = cdiv_u32((u64)h->nrow, XT), ntc = cdiv_u32((u64)h->ncol, XT);
  const size_t ntiles = (size_t)ntr * ntc;
  if (ntr > 65535u) return PFD_OK;
  pfd_seg_begin(h, "exact_plan");
  DevBuf upa;
  PFDCHK(upa.alloc((size_t)n * sizeof(u32)));
  int complete = 0;
  {
    const bool prof = h->profiling;  // (the tiled pass records its own segments: keep ours intact)
    h->profiling = false;
    // (a row block: the count inside the block, nothing entering from the neighbours — the heavy links only need SOME
    //  consistent weight; the result covers the own rows, the halo rows stay 0 and are never heavy)
    if (block) HIPCHK(hipMemsetAsync(upa.p, 0, (size_t)n * sizeof(u32), h->stream));
    const int rc = pfd_upstream_area_cell_tiled(h, (i32 *)upa.p + (size_t)h->halo_top * (size_t)h->ncol, &complete);
    h->profiling = prof;
    PFDCHK(rc);
  }
  if (!complete) {  // cycles: the level engine keeps the reference's semantics for them
    h->acyclic = -1;
    pfd_seg_end(h, 0);
    return PFD_OK;
  }
  h->acyclic = 1;
  ExactPlan *p = new ExactPlan();
  h->xplan = p;  // (freed by pfd_free_xplan also when a later step fails)
  p->ntr = ntr;
  p->ntc = ntc;
  int rc = PFD_OK;
  auto fail = [&](int code) {
    pfd_free_xplan(h);
    h->xplan_state = -1;
    return code;
  };
  if ((rc = pfd_dmalloc((void **)&p->lh, (size_t)n + 64)) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->kids, (size_t)n + 64)) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->tord, ntiles * XTC * sizeof(uint16_t))) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->toff, ntiles * XOFF * sizeof(uint16_t))) != PFD_OK) return fail(rc);
  k_plan_tile<<<dim3(ntc, ntr), 256, 0, h->stream>>>(h->ncode, (u32)h->nrow, (u32)h->ncol, ntc, p->lh, p->kids, p->tord,
                                                     p->toff, block ? h->halo_raw : nullptr, (u32)h->halo_top,
                                                     (u32)(h->halo_top + h->own_rows - 1));
  XDBG(h, "k_plan_tile");
  xdigest(h, "upa", upa.p, (size_t)n * 4);
  xdigest(h, "ncode", h->ncode, (size_t)n);
  xdigest(h, "lh", p->lh, (size_t)n);
  xdigest(h, "kids", p->kids, (size_t)n);
  xdigest(h, "toff", p->toff, ntiles * XOFF * 2);
  if (hipGetLastError() != hipSuccess) return fail(PFD_EHIP);
  DevBuf hcode, tidxbuf, hinfo, hops, tailnum;
  if ((rc = hcode.alloc((size_t)n + 64)) != PFD_OK) return fail(rc);
  if ((rc = tidxbuf.alloc((size_t)n * sizeof(u32) + 64)) != PFD_OK) return fail(rc);  // (written at the chain ends only)
  if ((rc = hinfo.alloc((size_t)n * sizeof(uint16_t) + 64)) != PFD_OK) return fail(rc);
  DevBuf bcount;
  // (the chain-end count / list kernels: a 64 x 64 tile per workgroup, or 4096 cells in linear order)
  const bool tails_tiled = !pfd_knob("PFD_TAILS_RASTER");
  const u32 gridT = tails_tiled ? (u32)ntiles : cdiv_u32(n, TLG);
  if ((rc = bcount.alloc(((size_t)gridT + 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  if (hipMemsetAsync(bcount.as<u32>() + gridT, 0, sizeof(u32), h->stream) != hipSuccess) return fail(PFD_EHIP);
  k_plan_heavy_tile<<<dim3(ntc, ntr), 256, 0, h->stream>>>(h->ncode, (u32)h->nrow, (u32)h->ncol, p->lh, p->kids, upa.as<u32>(),
                                                           hinfo.as<uint16_t>(), hcode.as<u8>());
  if (tails_tiled)
    k_plan_count_ends<true><<<gridT, 256, 0, h->stream>>>(hcode.as<u8>(), n, (u32)h->nrow, (u32)h->ncol, ntc, bcount.as<u32>());
  else
    k_plan_count_ends<false><<<gridT, 256, 0, h->stream>>>(hcode.as<u8>(), n, (u32)h->nrow, (u32)h->ncol, ntc, bcount.as<u32>());
  XDBG(h, "k_plan_heavy");
  xdigest(h, "hcode", hcode.p, (size_t)n);
  xdigest(h, "hinfo", hinfo.p, (size_t)n * 2);
  if (hipGetLastError() != hipSuccess) return fail(PFD_EHIP);
  if ((rc = hops.alloc((size_t)n * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = tailnum.alloc((size_t)n * sizeof(u32))) != PFD_OK) return fail(rc);
  // hops to the end of the chain and the end itself (its index + 1), one path query for both
  if ((rc = pfd_path_rank_tails(h, hcode.as<u8>(), hops.as<u32>(), tailnum.as<u32>(), &complete)) != PFD_OK) return fail(rc);
  if (!complete) return fail(PFD_OK);
  xdigest(h, "hops", hops.p, (size_t)n * 4);
  xdigest(h, "tailnum", tailnum.p, (size_t)n * 4);
  // the chain ends (pits of the heavy forest), in raster order (selection by scan: no same-address atomics).  The list
  // lives in the upstream-area buffer, which is no longer needed.
  DevBuf dA, dB, pA, pB, tmp, cnt, len_of, rank_of, parb, labb;
  u32 *tails = upa.as<u32>();
  if ((rc = cnt.alloc(64 * sizeof(u32))) != PFD_OK) return fail(rc);
  size_t tmp_bytes = 0;
  {
    if (rocprim::exclusive_scan(nullptr, tmp_bytes, bcount.as<u32>(), bcount.as<u32>(), 0u, (size_t)gridT + 1, rocprim::plus<u32>(),
                                h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
    if (rocprim::exclusive_scan(tmp.p, tmp_bytes, bcount.as<u32>(), bcount.as<u32>(), 0u, (size_t)gridT + 1, rocprim::plus<u32>(),
                                h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    if (tails_tiled)
      k_plan_tail_list<true><<<gridT, 256, 0, h->stream>>>(hcode.as<u8>(), n, (u32)h->nrow, (u32)h->ncol, ntc, bcount.as<u32>(), tails);
    else
      k_plan_tail_list<false><<<gridT, 256, 0, h->stream>>>(hcode.as<u8>(), n, (u32)h->nrow, (u32)h->ncol, ntc, bcount.as<u32>(), tails);
    if (hipGetLastError() != hipSuccess) return fail(PFD_EHIP);
  }
  hcode.alloc(0);
  u32 nt32 = 0;
  if (hipMemcpyAsync(&nt32, bcount.as<u32>() + gridT, sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  const unsigned long long nchain = nt32;
  const size_t nc1 = std::max<size_t>(nchain, 1);
  u32 *tidx_at = tidxbuf.as<u32>();
  if (nchain) k_plan_tidx<<<cdiv_u32(nchain, 256), 256, 0, h->stream>>>(tails, nt32, tidx_at);
  XDBG(h, "k_plan_tidx");
  // rounds of the chains (see k_plan_tails)
  if ((rc = dA.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = dB.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = pA.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = pB.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = parb.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  u32 *Dc = dA.as<u32>(), *Dn = dB.as<u32>(), *Pc = pA.as<u32>(), *Pn = pB.as<u32>();
  if (nchain) {
    k_plan_tails<<<cdiv_u32(nchain, 256), 256, 0, h->stream>>>(h->ncode, h->geo, tails, nt32, tailnum.as<u32>(), tidx_at, Dc, Pc,
                                                               parb.as<u32>());
  XDBG(h, "k_plan_tails");
    for (int r = 0; r < 6; ++r) {
      k_plan_depth_round<<<cdiv_u32(nchain, 256), 256, 0, h->stream>>>(nt32, Dc, Pc, Dn, Pn);
  XDBG(h, "k_plan_depth_round");
      std::swap(Dc, Dn);
      std::swap(Pc, Pn);
    }
  }
  const u32 *depth = Dc;
  if ((rc = len_of.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = rank_of.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  k_plan_len<<<cdiv_u32(n, 2048), 256, 0, h->stream>>>(hinfo.as<uint16_t>(), hops.as<u32>(), tailnum.as<u32>(), tidx_at, n,
                                                       len_of.as<u32>());
  XDBG(h, "k_plan_len");
  xdigest(h, "tails", tails, (size_t)nchain * 4);
  xdigest(h, "depth", depth, (size_t)nchain * 4);
  xdigest(h, "len_of", len_of.p, (size_t)nchain * 4);
  // labels = rounds of the chains (see k_plan_label_pass): built, measured (profiles/r06_ab_rounds_early.txt) and OFF —
  // PFD_ROUNDS_EARLY turns them on; by default every chain keeps "as late as possible"
  const u32 *lab = nullptr;
  if (nchain && pfd_knob("PFD_ROUNDS_EARLY")) {
    if ((rc = labb.alloc((nc1 + 1) * sizeof(u32))) != PFD_OK) return fail(rc);
    u32 *L = labb.as<u32>(), *maxd = L + nc1;
    if (hipMemsetAsync(maxd, 0, sizeof(u32), h->stream) != hipSuccess) return fail(PFD_EHIP);
    const u32 gc = cdiv_u32(nchain, 256);
    k_plan_maxdepth<<<gc, 256, 0, h->stream>>>(nt32, depth, maxd);
    k_plan_label_init<<<gc, 256, 0, h->stream>>>(nt32, depth, len_of.as<u32>(), maxd, L);
    k_plan_label_pins<<<gc, 256, 0, h->stream>>>(nt32, parb.as<u32>(), L);
    u32 md = 0;
    if (hipMemcpyAsync(&md, maxd, sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    for (u32 k = 31u - md; k < 31u; ++k) k_plan_label_pass<<<gc, 256, 0, h->stream>>>(nt32, k, parb.as<u32>(), L);
    XDBG(h, "k_plan_label_pass");
    lab = L;
  }
  // chains in layout order: stable sort of the chain ends by round; chain id = rank in that order
  DevBuf ucell, w, cpos, clenp, ctail, cpad, pick, keys, keys2, iota, cj, adj;
  if ((rc = keys.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = keys2.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = iota.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = cj.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = adj.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = ctail.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = cpos.alloc((nchain + 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = clenp.alloc(nc1 * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = cpad.alloc((nchain + 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  if (hipMemsetAsync(cnt.p, 0, 64 * sizeof(u32), h->stream) != hipSuccess) return fail(PFD_EHIP);
  if (nchain) {
    k_plan_keys<<<cdiv_u32(nchain, 256), 256, 0, h->stream>>>(nt32, depth, lab, keys.as<u32>(), iota.as<u32>(), cnt.as<u32>());
  XDBG(h, "k_plan_keys");
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<u32>(), keys2.as<u32>(), iota.as<u32>(), cj.as<u32>(),
                                  (size_t)nchain, 0u, 5u, h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
    if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.as<u32>(), keys2.as<u32>(), iota.as<u32>(), cj.as<u32>(),
                                  (size_t)nchain, 0u, 5u, h->stream) != hipSuccess)
      return fail(PFD_EHIP);
  }
  k_plan_chain_lens<<<cdiv_u32(nchain + 1, 256), 256, 0, h->stream>>>(cj.as<u32>(), nt32, tails, len_of.as<u32>(),
                                                                     rank_of.as<u32>(), ctail.as<u32>(), clenp.as<u32>(),
                                                                     cpos.as<u32>(), tidx_at);
  XDBG(h, "k_plan_chain_lens");
  xdigest(h, "cj", cj.p, (size_t)nchain * 4);
  xdigest(h, "ctail", ctail.p, (size_t)nchain * 4);
  if (rocprim::exclusive_scan(nullptr, tmp_bytes, cpos.as<u32>(), cpos.as<u32>(), 0u, (size_t)nchain + 1,
                              rocprim::plus<u32>(), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
  if (rocprim::exclusive_scan(tmp.p, tmp_bytes, cpos.as<u32>(), cpos.as<u32>(), 0u, (size_t)nchain + 1,
                              rocprim::plus<u32>(), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  u32 hist[32], npos32 = 0;
  if (hipMemcpyAsync(hist, cnt.p, sizeof(hist), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipMemcpyAsync(&npos32, cpos.as<u32>() + nchain, sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  const unsigned long long npos = npos32;
  {
    i64 c = 0;
    for (int b = 0; b < 32; ++b) {
      p->b_chain[b] = c;
      c += hist[b];
    }
    p->b_chain[32] = c;
  }
  p->ntrunk = (i64)npos;
  p->nchain = (i64)nchain;
  upa.alloc(0);  // (the list of chain ends is no longer needed: ctail holds them in layout order)
  if ((rc = ucell.alloc((npos + 1) * sizeof(uint4))) != PFD_OK) return fail(rc);
  if ((rc = w.alloc((npos + 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  if (hipMemsetAsync(ucell.p, 0, (npos + 1) * sizeof(uint4), h->stream) != hipSuccess) return fail(PFD_EHIP);
  if ((rc = pfd_dmalloc((void **)&p->cslot, ((size_t)n + 64) * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->tl_off, (ntiles + 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  if (hipMemsetAsync(p->tl_off + ntiles, 0, sizeof(u32), h->stream) != hipSuccess) return fail(PFD_EHIP);
  DevBuf pend;  // (written at the chain ends only, like tidx_at)
  if ((rc = pend.alloc((size_t)n * sizeof(u32) + 64)) != PFD_OK) return fail(rc);
  if (nchain)
    k_plan_pend<<<cdiv_u32(nchain, 256), 256, 0, h->stream>>>(ctail.as<u32>(), cpos.as<u32>(), clenp.as<u32>(), nt32, pend.as<u32>());
  k_plan_scatter<<<dim3(ntc, ntr), 256, 0, h->stream>>>(hinfo.as<uint16_t>(), hops.as<u32>(), tailnum.as<u32>(), tidx_at,
                                                        pend.as<u32>(), (u32)h->nrow, (u32)h->ncol,
                                                        ucell.as<uint4>(), p->cslot, p->tl_off);
  // first list entry of every tile (exclusive scan of the tiles' trunk counts, in place)
  if (rocprim::exclusive_scan(nullptr, tmp_bytes, p->tl_off, p->tl_off, 0u, ntiles + 1, rocprim::plus<u32>(), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
  if (rocprim::exclusive_scan(tmp.p, tmp_bytes, p->tl_off, p->tl_off, 0u, ntiles + 1, rocprim::plus<u32>(), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  if ((rc = pfd_dmalloc((void **)&p->tlist, (npos + 1) * sizeof(uint2))) != PFD_OK) return fail(rc);
  XDBG(h, "k_plan_scatter");
  if (hipGetLastError() != hipSuccess) return fail(PFD_EHIP);
  // slot of a position = exclusive scan of the slots the positions before it need (in place)
  auto wneed = rocprim::make_transform_iterator((const uint4 *)ucell.p, XRecSlots());
  if (rocprim::exclusive_scan(nullptr, tmp_bytes, wneed, w.as<u32>(), 0u, (size_t)npos + 1, rocprim::plus<u32>(),
                              h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
  if (rocprim::exclusive_scan(tmp.p, tmp_bytes, wneed, w.as<u32>(), 0u, (size_t)npos + 1, rocprim::plus<u32>(),
                              h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  // padded chain starts: exclusive scan of the padded chain lengths; round offsets = starts of their first chains
  k_plan_chain_len<<<cdiv_u32(nchain + 1, 256), 256, 0, h->stream>>>(cpos.as<u32>(), clenp.as<u32>(), w.as<u32>(),
                                                                    (u32)nchain, cpad.as<u32>());
  XDBG(h, "k_plan_chain_len");
  if (rocprim::exclusive_scan(nullptr, tmp_bytes, cpad.as<u32>(), cpad.as<u32>(), 0u, (size_t)nchain + 1,
                              rocprim::plus<u32>(), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
  if (rocprim::exclusive_scan(tmp.p, tmp_bytes, cpad.as<u32>(), cpad.as<u32>(), 0u, (size_t)nchain + 1,
                              rocprim::plus<u32>(), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  u32 pidx[33], pval[33];
  for (int b = 0; b <= 32; ++b) pidx[b] = (u32)p->b_chain[b];
  if ((rc = pick.alloc(2 * 33 * sizeof(u32))) != PFD_OK) return fail(rc);
  if (hipMemcpyAsync(pick.p, pidx, sizeof(pidx), hipMemcpyHostToDevice, h->stream) != hipSuccess) return fail(PFD_EHIP);
  k_plan_pick<<<1, 64, 0, h->stream>>>(cpad.as<u32>(), pick.as<u32>(), 33, pick.as<u32>() + 33);
  XDBG(h, "k_plan_pick");
  if (hipMemcpyAsync(pval, pick.as<u32>() + 33, sizeof(pval), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  for (int b = 0; b <= 32; ++b) p->b_slot[b] = (i64)pval[b];
  p->nslot = (i64)pval[32];
  const size_t nsl = std::max<size_t>((size_t)p->nslot, 4);
  const size_t nwords = nsl / 32 + 4;
  if ((rc = pfd_dmalloc((void **)&p->scell, nsl * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->sinfo, nsl * sizeof(uint16_t) + 16)) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->spost, nwords * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->cstart, std::max<size_t>(nchain, 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  if ((rc = pfd_dmalloc((void **)&p->clen, std::max<size_t>(nchain, 1) * sizeof(u32))) != PFD_OK) return fail(rc);
  // padding slots: post slots that carry cell 0 (never folded: they lie beyond their chain's length)
  if (hipMemsetAsync(p->scell, 0, nsl * sizeof(u32), h->stream) != hipSuccess ||
      hipMemsetAsync(p->sinfo, 0x80, nsl * sizeof(uint16_t), h->stream) != hipSuccess ||
      hipMemsetAsync(p->spost, 0, nwords * sizeof(u32), h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  if (npos) {
    k_plan_chains<<<cdiv_u32(nchain, 256), 256, 0, h->stream>>>(cpos.as<u32>(), clenp.as<u32>(), ctail.as<u32>(),
                                                                w.as<u32>(), cpad.as<u32>(), hinfo.as<uint16_t>(),
                                                                (u32)nchain, p->cstart, p->clen, adj.as<u32>());
  XDBG(h, "k_plan_chains");
    k_plan_expand<<<cdiv_u32(npos, 256), 256, 0, h->stream>>>(ucell.as<uint4>(), w.as<u32>(), adj.as<u32>(), h->geo, (u32)npos,
                                                              p->scell, p->sinfo, p->spost);
  XDBG(h, "k_plan_expand");
    k_plan_cslot<<<dim3(ntc, ntr), 256, 0, h->stream>>>(hinfo.as<uint16_t>(), w.as<u32>(), (u32)h->nrow,
                                                        (u32)h->ncol, p->cslot, p->lh, p->tl_off, p->tlist);
  XDBG(h, "k_plan_cslot");
  xdigest(h, "ucell", ucell.p, (size_t)npos * 16);
  xdigest(h, "scell", p->scell, (size_t)p->nslot * 4);
  xdigest(h, "sinfo", p->sinfo, (size_t)p->nslot * 2);
  xdigest(h, "spost", p->spost, (size_t)(p->nslot / 32) * 4);
  xdigest(h, "cstart", p->cstart, (size_t)nchain * 4);
  xdigest(h, "clen", p->clen, (size_t)nchain * 4);
  }
  // the long chains (one wave each in the sweeps): their ids in layout order, offsets per round
  if (nchain) {
    const size_t cap = (size_t)p->nslot / XLONG + 1;
    if ((rc = pfd_dmalloc((void **)&p->longc, cap * sizeof(u32))) != PFD_OK) return fail(rc);
    rocprim::counting_iterator<u32> ids(0u);
    auto flags = rocprim::make_transform_iterator(p->clen, [] __device__(u32 cl) { return (cl & XC_LEN) >= XLONG; });
    if (rocprim::select(nullptr, tmp_bytes, ids, flags, p->longc, cnt.as<u32>(), (size_t)nchain, h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    if ((rc = tmp.alloc(std::max<size_t>(tmp_bytes, 16))) != PFD_OK) return fail(rc);
    if (rocprim::select(tmp.p, tmp_bytes, ids, flags, p->longc, cnt.as<u32>(), (size_t)nchain, h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    u32 nl = 0;
    if (hipMemcpyAsync(&nl, cnt.p, sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
      return fail(PFD_EHIP);
    p->nlong = nl;
    std::vector<u32> lc(nl);
    if (nl && hipMemcpy(lc.data(), p->longc, (size_t)nl * sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) return fail(PFD_EHIP);
    for (int b = 0; b <= 32; ++b)
      p->b_long[b] = (i64)(std::lower_bound(lc.begin(), lc.end(), (u32)std::min<i64>(p->b_chain[b], 0xFFFFFFFFll)) - lc.begin());
    // the longest chain of every round (what a round costs when it is latency: see run_exact_up / run_exact_down)
    if (nl) {
      DevBuf ll;
      if ((rc = ll.alloc((size_t)nl * sizeof(u32))) != PFD_OK) return fail(rc);
      k_plan_long_lens<<<cdiv_u32(nl, 256), 256, 0, h->stream>>>(p->longc, nl, p->clen, ll.as<u32>());
      std::vector<u32> lens(nl);
      if (hipMemcpyAsync(lens.data(), ll.p, (size_t)nl * sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
          hipStreamSynchronize(h->stream) != hipSuccess)
        return fail(PFD_EHIP);
      for (int b = 0; b < 32; ++b)
        for (i64 i = p->b_long[b]; i < p->b_long[b + 1]; ++i) p->b_maxlen[b] = std::max(p->b_maxlen[b], lens[(size_t)i]);
    }
  }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(PFD_EHIP);
  p->bytes = 6 * ((size_t)n + 64) + ntiles * (XTC + XOFF) * sizeof(uint16_t) + (size_t)p->nslot * 6 + (size_t)p->nslot / 8 + (size_t)nchain * 8 + (size_t)npos * 8 + ntiles * 4;
  h->bytes_held += p->bytes;
  h->xplan_state = 1;
  pfd_seg_end(h, 14);
#ifdef PFD_DEVTOOLS
  if (getenv("PFD_DEBUG"))
    fprintf(stderr, "[xplan] %lld cells: %lld trunk (%.1f%%) in %lld chains, %lld slots\n", (long long)h->n_valid,
            (long long)p->ntrunk, 100.0 * (double)p->ntrunk / (double)std::max<i64>(h->n_valid, 1), (long long)p->nchain,
            (long long)p->nslot);
#endif
  return PFD_OK;
}

// debug export (tests / tools): plan summary; not part of the C-ABI contract
extern "C" int pfd_debug_xplan(pfd_raster *h, int64_t info[8], uint8_t *lh_host) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_ensure_xplan(h));
  for (int k = 0; k < 8; ++k) info[k] = 0;
  info[0] = h->xplan_state;
  if (h->xplan_state != 1) return PFD_OK;
  ExactPlan *p = (ExactPlan *)h->xplan;
  info[1] = p->ntrunk;
  info[2] = p->nchain;
  info[3] = p->nslot;
  int nb = 0;
  i64 longest = 0;
  for (int b = 0; b < 32; ++b) nb += p->b_chain[b + 1] > p->b_chain[b];
  info[4] = nb;
  (void)longest;
  if (lh_host) HIPCHK(hipMemcpy(lh_host, p->lh, (size_t)h->n, hipMemcpyDeviceToHost));
  return PFD_OK;
}
