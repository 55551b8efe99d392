// tile_fast.h — the tile passes for INTERIOR tiles (included by tiled.hip).
//
// A tile is interior when it, its halo ring and the 4 staging columns either side lie inside the device raster and
// none of those rows is a halo row or a boundary row of a row block: no bounds handling, no halo sinks, no
// boundary-row bookkeeping.  That is all but a one-tile frame of a raster (99.7 % of the tiles at 90000 x 90000),
// so these kernels ARE the tile passes; k_tile (tiled.hip) keeps the general form for the frame.
//
// Both passes sit on the VALU (SQ counters, profiles/archive/r02y_sq_counters_tile.csv: 123 and 89 VALU instructions per
// cell in round 2, LDS and HBM far from their limits), so everything here is about instructions per cell:
//  * decode by byte permute: the step of a direction code c = 1 << k inside the tile (dr * 64 + dc) and inside the
//    staged codes (dr * 72 + dc) are 8-entry byte tables looked up with v_perm_b32 (selector k); "the step stays in
//    the tile" is one AND with a per-cell constant mask of the directions that leave it; the raw pass classifies a
//    byte by its population count (0 / 8: pit, 1: direction, else nodata or a bad code), again through a byte table;
//  * pointers are LDS byte offsets, so that a gather needs no address arithmetic at all (local pass) or one shift
//    (final pass), and "saturated" is a range test on the pointer itself instead of a flag bit that every use has to
//    mask away;
//  * local pass: a root is not "the cell where the path ends" but the perimeter SLOT of that cell — the 256 slots
//    (+ one "nobody asks": pits, nodata) have pointer words of their own behind the 4096 cells, written by one
//    thread per slot — so that after the pointer jumping a cell holds the address of its exit's counter and the
//    count per exit costs three instructions per cell instead of a slot computation per cell;
//  * final pass: a saturated cell points at a per-lane sink word behind the tile's counts, with a pointer word behind
//    the tile's pointers that points at itself: no compare / select per cell and round;
//  * one (local) / two (final) barriers per round: the "is any pointer still moving" vote goes through two
//    alternating LDS flag rows written by the wave leaders instead of three barriers of __syncthreads_or.
#pragma once
#include <type_traits>

#define FX_SLOT0 (2u * TCELLS)              // local pass: P byte offset of the pointer word of perimeter slot 0
#define FX_NOBODY (FX_SLOT0 + 2u * PSL)     // ... of the root nobody asks about (pit, nodata, cell of a cycle)
#define FX_PN (TCELLS + PSL + 8)            // P entries of the local pass
#ifndef FY_COMBINE
#define FY_COMBINE true
#endif
#ifndef FX_COMBINE
#define FX_COMBINE true
#endif
#ifndef FY_LEAVES
#define FY_LEAVES 0  // (1: the experiment of k_tile_final_fast with the in-tile leaves retired before the rounds)
#endif
#ifndef FXP
#define FXP 4  // count words per perimeter slot (replicas picked by lane: a wave's atomics on one word are serialised)
#endif
#define FY_SINK0 (4u * TCELLS)              // final pass: A byte offset of sink word 0 (64 of them, one per lane)

// step tables (selector k = position of the code's bit): E, SE, S, SW | W, NW, N, NE
#define FX_TP_LO 0x3F404101u  // tile index steps  +1 +65 +64 +63
#define FX_TP_HI 0xC1C0BFFFu  //                   -1 -65 -64 -63
#define FX_TC_LO 0x47484901u  // staged-code steps +1 +73 +72 +71   (row pitch CP = 72)
#define FX_TC_HI 0xB9B8B7FFu  //                   -1 -73 -72 -71
// normalised code of a cell that is no direction, by population count of its raw byte: 0 (code 0) and 8 (code 255:
// selector 8 replicates the sign bit of table byte 1 = 0) are pits -> 0; 1 is a direction (-> 0 when its target is
// nodata); everything else is nodata (247 itself, 7 bits) or a bad code -> 247
#define FX_T0_LO 0xF7F70000u
#define FX_T0_HI 0xF7F7F7F7u

__device__ __forceinline__ u32 fx_ffbl(u32 x) {  // position of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32)
  u32 r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ int fx_sext8(u32 x) { return (int)(int8_t)(x & 0xFFu); }

// directions that leave the tile from a cell in row lr / column lc (only perimeter cells have any)
__device__ __forceinline__ u32 fx_rowmask(u32 lr) { return (lr == 0u ? 0xE0u : 0u) | (lr == TS - 1u ? 0x0Eu : 0u); }
__device__ __forceinline__ u32 fx_colmask(u32 lc) { return (lc == 0u ? 0x38u : 0u) | (lc == TS - 1u ? 0x83u : 0u); }

// wave leaders publish "one of my lanes still moves a pointer"; everybody reads the four flags after the barrier.
// Two alternating rows: a wave can only reach the write of round r + 2 after every wave has read round r's row.
__device__ __forceinline__ bool fx_vote(u32 (*s_flag)[4], int round, u32 tid, bool live) {
  const bool any = __ballot(live) != 0ull;
  if ((tid & 63u) == 0u) s_flag[round & 1][tid >> 6] = any ? 1u : 0u;
  __syncthreads();
  const uint4 f = *(const uint4 *)s_flag[round & 1];
  return (f.x | f.y | f.z | f.w) != 0u;
}

// Where the neighbours of perimeter cell p that lie OUTSIDE the tile live: per slot p up to 5 candidates (3 along an
// edge, 5 at a corner), each 12 bits pslot | tile delta << 8 (xr_t12) plus the direction k << 12; unused entries name
// the cell's own slot with k = 8 (a bit no source mask has).  A per-lane constant of the final pass (16 bytes): an
// entry pulls the totals of the exits that drain into it, and because the candidates are known up front all five
// loads go out with the first instructions of the kernel — speculatively, the record's source mask (loaded beside them)
// picks the ones that count.
struct NbrTab {
  uint16_t v[PSL][8];
};
constexpr NbrTab make_nbr_tab() {
  NbrTab t{};
  for (int p = 0; p < PSL; ++p) {
    int lr = 0, lc = 0;  // pslot_inv
    if (p < TS) lr = 0, lc = p;
    else if (p < 2 * TS) lr = TS - 1, lc = p - TS;
    else if (p < 2 * TS + (TS - 2)) lr = p - 2 * TS + 1, lc = 0;
    else lr = p - (2 * TS + (TS - 2)) + 1, lc = TS - 1;
    int cnt = 0;
    for (int k = 0; k < 8 && p < NPERIM; ++k) {
      const int dr = (k >= 1 && k <= 3) ? 1 : (k >= 5 ? -1 : 0);
      const int dc = (k == 0 || k == 1 || k == 7) ? 1 : ((k >= 3 && k <= 5) ? -1 : 0);
      const int nr = lr + dr, nc = lc + dc;
      if (nr >= 0 && nr < TS && nc >= 0 && nc < TS) continue;
      const int qr = nr & (TS - 1), qc = nc & (TS - 1);  // pslot
      const int ps = qr == 0 ? qc : (qr == TS - 1 ? TS + qc : (qc == 0 ? 2 * TS + (qr - 1) : 2 * TS + (TS - 2) + (qr - 1)));
      const int d = 3 * ((nr < 0 ? -1 : (nr >= TS ? 1 : 0)) + 1) + (nc < 0 ? -1 : (nc >= TS ? 1 : 0)) + 1;
      t.v[p][cnt++] = (uint16_t)(ps | (d << 8) | (k << 12));
    }
    for (; cnt < 8; ++cnt) t.v[p][cnt] = (uint16_t)((p < NPERIM ? p : 0) | (4 << 8) | (8 << 12));
  }
  return t;
}
static __device__ __constant__ const NbrTab NBR_TAB = make_nbr_tab();

// Weight of the cell in column c of a row with quantised area base + f / 2^32: base + floor((c + 1) f) - floor(c f)
// (Bresenham along the row).  Rounding every cell of a row the same way would give a run of k cells the error
// k * (rounding of the row); sharing the fraction out keeps the error of ANY run of consecutive cells of a row
// below one quantum — and the upstream set of a cell is made of such runs.
__device__ __forceinline__ u64 w_cell(u64 base, u32 f, u32 c) { return base + (u64)(__umulhi(c + 1u, f) - __umulhi(c, f)); }


template <int NW>
__device__ __forceinline__ bool fx_vote_n(u32 (*s_flag)[8], int round, u32 tid, bool live) {
  const bool any = __ballot(live) != 0ull;
  if ((tid & 63u) == 0u) s_flag[round & 1][tid >> 6] = any ? 1u : 0u;
  __syncthreads();
  const uint4 f = *(const uint4 *)s_flag[round & 1];
  u32 acc = f.x | f.y | f.z | f.w;
  if (NW == 8) {
    const uint4 g = *(const uint4 *)&s_flag[round & 1][4];
    acc |= g.x | g.y | g.z | g.w;
  }
  return acc != 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// local pass of an interior tile
// ---------------------------------------------------------------------------------------------------------------
// WIDE (wide.h, the fixed-point upstream area): beside the count, the 64-bit weights of the cells per exit (a.xT64)
template <bool RAW, bool WEIGHTS, bool WIDE = false>
__global__ void __launch_bounds__(256) k_tile_local_fast(TileArgs a) {
  __shared__ __attribute__((aligned(16))) u32 A[PSL * FXP];      // FXP count words per perimeter slot
  __shared__ __attribute__((aligned(16))) u64 A64[WIDE ? PSL * 4 : 2];  // WIDE: 4 sum words per perimeter slot
  __shared__ __attribute__((aligned(16))) uint16_t P[FX_PN];     // byte offset into P of an ancestor / of a root word
  __shared__ __attribute__((aligned(16))) u8 code[HW * CP];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][4];
  __shared__ u64 s_cnt[4];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_ + a.tc_lo, tr = by_ + a.tr_lo;
  const u32 sbase = sslot_base(tr, tc, a.nstc);
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  bool mvq = false;  // RAW: one of the staged bytes is nodata (zero-byte test of v ^ 247 x 4)
  {
    u32 v[5];
    stage_load_interior(RAW ? a.raw : a.ncode, a.ncol, r0, c0, tid, v);
    if (RAW) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const u32 x = v[k] ^ 0xF7F7F7F7u;
        mvq |= ((x - 0x01010101u) & ~x & 0x80808080u) != 0u;
      }
    }
    stage_store(code, tid, v);
  }
  *(uint4 *)&A[FXP * tid] = make_uint4(0u, 0u, 0u, 0u);
  if (FXP == 8) *(uint4 *)&A[FXP * tid + 4] = make_uint4(0u, 0u, 0u, 0u);
  if (WIDE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) A64[tid + 256u * k] = 0;
  }
  P[TCELLS + tid] = (uint16_t)(FX_SLOT0 + 2u * tid);  // a root word points at itself
  if (tid == 0) P[TCELLS + PSL] = (uint16_t)FX_NOBODY;
  // a tile without a nodata byte in its staging area (most tiles of a land raster) has no "flow into nodata ends
  // here" to look for: its decode skips the per-cell read of the target's code
  const bool tile_mv = RAW ? (__syncthreads_or(mvq ? 1 : 0) != 0) : (__syncthreads(), false);

  // ---- decode (+ normalise) the thread's 4 quads, initial pointers ------------------------------------------
  const u32 qs = (tid >> 3) & 3u;      // register slot s of a quad holds logical cell s ^ qs (swizzle, see k_tile)
  const u32 lcq = 4u * (tid & 15u);    // first column of the thread's quads
  u32 sh[4], cm[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const u32 b = (u32)s ^ qs;
    sh[s] = 8u * b;
    cm[s] = fx_colmask(lcq + b);
  }
  u32 ndir = 0, npit = 0, nbad = 0;
  auto decode = [&](auto chk) {
  constexpr bool CHKMV = decltype(chk)::value;
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const u32 lr = (tid >> 4) + 16u * j;
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 ca0 = (lr + 1u) * CP + lcq + 4u;  // byte offset of CODE(lr, lcq)
    const u32 c4 = *(const u32 *)&code[ca0];
    const u32 rm = (j == 0 || j == QPT - 1) ? fx_rowmask(lr) : 0u;
    u32 p[4], n4 = 0, badq = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32 c = (c4 >> sh[s]) & 0xFFu;
      const u32 k = fx_ffbl(c);
      const u32 pcnt = __popc(c);
      bool isdir = pcnt == 1u;
      if (RAW) {
        if (CHKMV) {
          const u32 t = code[(u32)((int)(ca0 + (sh[s] >> 3)) + fx_sext8(__builtin_amdgcn_perm(FX_TC_HI, FX_TC_LO, k)))];
          isdir = isdir && t != D8_MV;  // flow into nodata ends here (interior tile: never off the raster)
        }
        const u32 t0 = __builtin_amdgcn_perm(FX_T0_HI, FX_T0_LO, pcnt);
        const u32 n = isdir ? c : t0;
        badq |= t0 & ~c;  // != 0 exactly for a byte that is neither a code nor 247 (247 & ~c == 0 <=> c in {247, 255})
        n4 |= n << sh[s];
        ndir += isdir ? 1u : 0u;
        npit += n == 0u ? 1u : 0u;
      }
      const bool go = isdir && (c & (rm | cm[s])) == 0u;
      const u32 lt = (u32)((int)(l0 + (sh[s] >> 3)) + fx_sext8(__builtin_amdgcn_perm(FX_TP_HI, FX_TP_LO, k)));
      p[s] = go ? (PHYS(lt) << 1) : FX_NOBODY;  // (an exit's own word is set by its slot's thread below)
    }
    *(uint2 *)&P[l0] = make_uint2(p[0] | (p[1] << 16), p[2] | (p[3] << 16));
    if (RAW) {
      if (badq) {  // rare: count the bad bytes of the quad exactly (the error message quotes the number)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32 c = (c4 >> sh[s]) & 0xFFu;
          nbad += (__builtin_amdgcn_perm(FX_T0_HI, FX_T0_LO, __popc(c)) & ~c) ? 1u : 0u;
        }
      }
      *(u32 *)&code[ca0] = n4;  // (readers of a raw byte only ask "== nodata": unchanged unless the raster is rejected)
      __builtin_memcpy(a.ncode_w + (size_t)(r0 + lr) * a.ncol + (size_t)(c0 + lcq), &n4, 4);  // possibly unaligned dword
    }
  }
  };
  if (RAW && tile_mv) decode(std::true_type{});
  else decode(std::false_type{});
  if (RAW) {  // counts of the tile -> tcnt (summed by k_tile_counts: no same-address atomics)
    u64 pk = (u64)(ndir + npit) | ((u64)npit << 16) | ((u64)nbad << 32);
    for (int o = 32; o > 0; o >>= 1) pk += __shfl_down(pk, o);
    if ((tid & 63u) == 0) s_cnt[tid >> 6] = pk;
  }
  __syncthreads();
  if (RAW && tid == 0) a.tcnt[(size_t)tr * a.ntc + tc] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];

  // ---- one thread per perimeter slot: an exit's pointer word names its slot -----------------------------------
  u32 xt12 = XR_NONE;  // target of the exit on this slot (xr_t12)
  int plr = 0, plc = 0;
  if (tid < NPERIM) {
    pslot_inv((int)tid, &plr, &plc);
    const u32 c = CODE(plr, plc);
    if (d8_is_dir(c)) {
      const int k = d8_slot(c);
      const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
      if ((unsigned)nr >= TS || (unsigned)nc >= TS) {  // (the target is inside the raster and valid: normalised codes)
        xt12 = xr_t12(nr, nc);
        P[PHYS((u32)(plr * TS + plc))] = (uint16_t)(FX_SLOT0 + 2u * tid);
      }
    }
  }
  __syncthreads();

  // ---- pointer jumping, gather-only: J <- J o J until every pointer sits on a root word ------------------------
  // (the "quad j still moves" flags are separate bools on purpose: the compiler keeps them as lane masks in
  //  scalar registers, so that testing and clearing them costs no VALU instruction)
  u32 pc[QPT * 4];
  bool lv[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const uint2 pp = *(const uint2 *)&P[4u * tid + 1024u * j];
    pc[4 * j + 0] = pp.x & 0xFFFFu;
    pc[4 * j + 1] = pp.x >> 16;
    pc[4 * j + 2] = pp.y & 0xFFFFu;
    pc[4 * j + 3] = pp.y >> 16;
    lv[j] = !(pp.x & pp.y & (pp.x >> 16) & (pp.y >> 16) & FX_SLOT0);  // (offsets < 0x4000: bit 13 <=> root word)
  }
  int round = 0;
#pragma nounroll
  for (; round < MAXROUNDS_TILE; ++round) {
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      if (lv[j]) {
        // two jumps per round (a root word points at itself, so jumping from a root stays there): half the barriers
        // and pointer write-backs per hop
        u32 q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = *(const uint16_t *)((const u8 *)P + pc[4 * j + b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = *(const uint16_t *)((const u8 *)P + q[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) pc[4 * j + b] = q[b];
        lv[j] = !(q[0] & q[1] & q[2] & q[3] & FX_SLOT0);
        *(uint2 *)&P[4u * tid + 1024u * j] = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
      }
    }
    if (!fx_vote(s_flag, round, tid, lv[0] | lv[1] | lv[2] | lv[3])) break;
  }
  const u32 live = (lv[0] ? 1u : 0u) + (lv[1] ? 1u : 0u) + (lv[2] ? 1u : 0u) + (lv[3] ? 1u : 0u);
  if ((a.ablate & 32) && tid == 0) {  // pfd_set_profiling(h, 2): rounds this tile needed (max and sum over the tiles)
    const unsigned long long r = (unsigned long long)min(round + 1, MAXROUNDS_TILE);
    const u32 w = (tr * a.ntc + tc) & 255u;
    atomicMax((unsigned long long *)&a.rcnt[w], r);
    atomicAdd((unsigned long long *)&a.rcnt[256 + w], r);
  }
  // a cell on or upstream of a cycle never reaches a root word: count their quads (normally zero)
  if (live) atomicAdd((unsigned long long *)&a.ctrl[T_UNSAT], (unsigned long long)live);

  // ---- every cell adds its weight to the counter of its exit (PREP replicas per slot, picked by lane) ---------
  // The four cells of a quad are neighbours in a row and mostly share their exit: they are combined in registers
  // first (a wave's atomics on one counter word are served one lane after the other, and a tile has few exits that
  // collect most of its cells).
  const u32 rep = 4u * (tid & (FXP - 1u));
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    u32 x[4], w[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      x[s] = pc[4 * j + s] - FX_SLOT0;  // 2 x slot; >= 2 * PSL: nobody asks (a nodata cell is its own root)
      w[s] = x[s] < 2u * PSL ? 1u : 0u;
      if (WEIGHTS) {
        const u32 lr = (tid >> 4) + 16u * j;
        w[s] = x[s] < 2u * PSL ? (u32)a.weights[(size_t)(r0 + lr) * a.ncol + (size_t)(c0 + lcq + (sh[s] >> 3))] : 0u;
      }
    }
    if (FX_COMBINE) {
      const bool e10 = x[1] == x[0], e20 = x[2] == x[0], e21 = x[2] == x[1], e30 = x[3] == x[0], e31 = x[3] == x[1], e32 = x[3] == x[2];
      w[0] += (e10 ? w[1] : 0u) + (e20 ? w[2] : 0u) + (e30 ? w[3] : 0u);
      w[1] = e10 ? 0u : w[1] + ((!e20 && e21) ? w[2] : 0u) + ((!e30 && e31) ? w[3] : 0u);
      w[2] = (e20 || e21) ? 0u : w[2] + ((!e30 && !e31 && e32) ? w[3] : 0u);
      w[3] = (e30 || e31 || e32) ? 0u : w[3];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (w[s]) atomicAdd((u32 *)((u8 *)A + (x[s] << (FXP == 8 ? 4 : 3)) + rep), w[s]);  // word slot * FXP + replica
    if (WIDE) {  // the same walk with the cells' 64-bit weights (a quad shares its row: one pair of row words)
      const u32 lr = (tid >> 4) + 16u * j;
      const u64 wr = a.wrow[(u32)(r0 + lr)];
      const u32 wf = a.wfrac[(u32)(r0 + lr)];
      u64 v[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = x[s] < 2u * PSL ? w_cell(wr, wf, (u32)c0 + lcq + (sh[s] >> 3)) : 0ull;
      const bool e10 = x[1] == x[0], e20 = x[2] == x[0], e21 = x[2] == x[1], e30 = x[3] == x[0], e31 = x[3] == x[1], e32 = x[3] == x[2];
      v[0] += (e10 ? v[1] : 0ull) + (e20 ? v[2] : 0ull) + (e30 ? v[3] : 0ull);
      v[1] = e10 ? 0ull : v[1] + ((!e20 && e21) ? v[2] : 0ull) + ((!e30 && e31) ? v[3] : 0ull);
      v[2] = (e20 || e21) ? 0ull : v[2] + ((!e30 && !e31 && e32) ? v[3] : 0ull);
      v[3] = (e30 || e31 || e32) ? 0ull : v[3];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (v[s]) atomicAdd((unsigned long long *)&A64[(x[s] >> 1) * 4u + (tid & 3u)], (unsigned long long)v[s]);
    }
  }
  __syncthreads();

  // ---- perimeter records for the exit graph -------------------------------------------------------------------
  u32 xt = 0, link = XR_NONE, inmask = 0;
  if (tid < NPERIM) {
    if (xt12 != XR_NONE) {
      const uint4 lo = *(const uint4 *)&A[tid * FXP];
      xt = lo.x + lo.y + lo.z + lo.w;
      if (FXP == 8) {
        const uint4 hi = *(const uint4 *)&A[tid * FXP + 4];
        xt += hi.x + hi.y + hi.z + hi.w;
      }
    }
    const u32 c = CODE(plr, plc);
    if (c != D8_MV) {  // entry?  (neighbours outside the tile that drain into this cell: the sources of its inflow)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
        if (((unsigned)nr >= TS || (unsigned)nc >= TS) && CODE(nr, nc) == (1u << ((k + 4) & 7))) inmask |= 1u << k;
      }
    }
    if (inmask) {  // the exit its in-tile path reaches
      const u32 x = (u32)P[PHYS((u32)(plr * TS + plc))] - FX_SLOT0;
      if (x < 2u * PSL) link = x >> 1;
    }
  }
  a.xT[sbase + tid] = xt;
  a.xrec[sbase + tid] = xr_pack(xt12 & 0xFFu, xt12 >> 8, link, inmask);
  if (WIDE) a.xT64[sbase + tid] = (tid < NPERIM && xt12 != XR_NONE) ? A64[4u * tid] + A64[4u * tid + 1] + A64[4u * tid + 2] + A64[4u * tid + 3] : 0ull;
  const u64 xm = __ballot(xt12 != XR_NONE);  // (a wave = 64 consecutive slots)
  if ((tid & 63u) == 0u) a.xmask[(sbase + tid) >> 6] = xm;
}

// ---------------------------------------------------------------------------------------------------------------
// final pass of an interior tile: the doubling with values; entries start with 1 + inflow
// ---------------------------------------------------------------------------------------------------------------
template <bool WEIGHTS, int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 6 : 8) k_tile_final_fast(TileArgs a) {
  constexpr int QF = TCELLS / 4 / NT;   // quads per thread (4 with 256 threads, 2 with 512)
  constexpr u32 QSTR = 4u * NT;         // cells between a thread's quads
  constexpr u32 RSTR = NT / 16;         // rows between them
  constexpr int NW = NT / 64;
  __shared__ __attribute__((aligned(16))) u32 A[TCELLS + 64];        // running count of the cell; 64 sink words
  __shared__ __attribute__((aligned(16))) uint16_t P[TCELLS + 64];   // A byte offset of an ancestor / of a sink word
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][8];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_ + a.tc_lo, tr = by_ + a.tr_lo;
  const u32 sbase = sslot_base(tr, tc, a.nstc);
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  const u32 lcq = 4u * (tid & 15u);
  u32 cq[QF];
#pragma unroll
  for (int j = 0; j < QF; ++j)
    __builtin_memcpy(&cq[j], a.ncode + (size_t)(r0 + (tid >> 4) + RSTR * j) * a.ncol + (size_t)(c0 + lcq), 4);
  // flow entering at this perimeter cell: pulled from the exits that drain into it.  All candidates are loaded right
  // away (NBR_TAB: no load waits for another), the record's source mask selects after the decode below.
  const u32 ptid = tid & 255u;  // (NT = 512: threads 256.. repeat the loads of 0..255 and drop them)
  const u32 rec = tid < 256u ? a.xrec[sbase + ptid] : 0u;  // (256 slots per tile; slots 252..255 carry no source mask)
  u32 xc[5], xk[5];
  {
    const uint4 nb = *reinterpret_cast<const uint4 *>(NBR_TAB.v[ptid]);
    // slot base of the neighbouring tile with delta code (lane & 15), fetched per candidate with a lane permute
    const u32 dl = min(tid & 15u, 8u), ql = (dl * 11u) >> 5;
    const u32 nbase = sslot_base(tr + ql - 1u, tc + (dl - 3u * ql) - 1u, a.nstc);
    const u32 e5[5] = {nb.x & 0xFFFFu, nb.x >> 16, nb.y & 0xFFFFu, nb.y >> 16, nb.z & 0xFFFFu};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const u32 bs = (u32)__shfl((int)nbase, (int)(((tid & 48u) | ((e5[i] >> 8) & 15u))));
      xc[i] = a.xtot[bs + (e5[i] & 0xFFu)];
      xk[i] = e5[i] >> 12;
    }
  }
  const u32 sink = FY_SINK0 + 4u * (tid & 63u);
  if (tid < 64u) P[TCELLS + tid] = (uint16_t)sink;  // a sink's pointer word points at the sink

  const u32 qs = (tid >> 3) & 3u;
  u32 sh[4], cm[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const u32 b = (u32)s ^ qs;
    sh[s] = 8u * b;
    cm[s] = fx_colmask(lcq + b);
  }
  u32 pc[QF * 4], qn[QF * 4];
  bool lv[QF];
#pragma unroll
  for (int j = 0; j < QF; ++j) {
    const u32 lr = (tid >> 4) + RSTR * j;
    const u32 l0 = 4u * tid + QSTR * j;
    const u32 c4 = cq[j];
    const u32 rm = (j == 0 || j == QF - 1) ? fx_rowmask(lr) : 0u;
    u32 w4[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32 c = (c4 >> sh[s]) & 0xFFu;
      // a direction (one bit) whose step stays in the tile; 247 / 254 keep >= 4 bits whatever the mask removes
      const bool go = __popc(c & ~(rm | cm[s])) == 1u;
      const u32 lt = (u32)((int)(l0 + (sh[s] >> 3)) + fx_sext8(__builtin_amdgcn_perm(FX_TP_HI, FX_TP_LO, fx_ffbl(c))));
      pc[4 * j + s] = go ? (PHYS(lt) << 2) : sink;
      u32 wv = 1u;
      if (WEIGHTS) wv = (u32)a.weights[(size_t)(r0 + lr) * a.ncol + (size_t)(c0 + lcq + (sh[s] >> 3))];
      w4[s] = c != D8_MV ? wv : 0u;
    }
    *(uint4 *)&A[l0] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    *(uint2 *)&P[l0] = make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
    lv[j] = !(pc[4 * j + 0] & pc[4 * j + 1] & pc[4 * j + 2] & pc[4 * j + 3] & FY_SINK0);
  }
  u32 inf = 0;
  {
    const u32 m = (rec >> 16) & 0xFFu;
#pragma unroll
    for (int i = 0; i < 5; ++i) inf += ((m >> xk[i]) & 1u) ? xc[i] : 0u;
  }
  __syncthreads();
  if (FY_LEAVES && !WEIGHTS) {
    // EXPERIMENT (off by default, same-box A/B + SQ counters in profiles/r05_ab_tile_leaves.txt): the
    // doubling without the in-tile leaves — cells no cell of the tile drains into, a third of them.  A leaf's count is
    // final from the start (1) and what its ancestors get from it reaches the parent ONCE; the forest of the other cells,
    // with the leaves absorbed, has the same counts.  Leaves are found by a mark: bit 31 of a cell's count word (never
    // set by a count) is raised by every cell that drains into it, and by the entry that receives flow from outside.
    // A retired leaf points at its lane's sink like a saturated cell: no instruction less, fewer LDS conflicts.
#pragma unroll
    for (int i = 0; i < QF * 4; ++i)
      if (pc[i] < FY_SINK0) ((u8 *)A)[pc[i] + 3u] = 0x80u;
    int plr = 0, plc = 0;
    pslot_inv((int)ptid, &plr, &plc);
    const u32 eoff = 4u * PHYS((u32)(plr * TS + plc));
    if (inf) ((u8 *)A)[eoff + 3u] = 0x80u;
    __syncthreads();
    u32 leafm = 0;
#pragma unroll
    for (int j = 0; j < QF; ++j) {
      const u32 l0 = 4u * tid + QSTR * j;
      uint4 a4 = *(const uint4 *)&A[l0];
      const u32 m4[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (!(m4[b] >> 31) && pc[4 * j + b] < FY_SINK0) leafm |= 1u << (4 * j + b);
      a4.x &= 0x7FFFFFFFu, a4.y &= 0x7FFFFFFFu, a4.z &= 0x7FFFFFFFu, a4.w &= 0x7FFFFFFFu;
      *(uint4 *)&A[l0] = a4;
    }
    __syncthreads();
    if (inf) atomicAdd(&A[eoff >> 2], inf);  // (an entry with inflow is marked: never a leaf; leaves that drain into it push beside this)
#pragma unroll
    for (int i = 0; i < QF * 4; ++i)
      if ((leafm >> i) & 1u) atomicAdd((u32 *)((u8 *)A + pc[i]), 1u);  // (a leaf's count is its own cell: 1)
#pragma unroll
    for (int j = 0; j < QF; ++j) {
      if (!((leafm >> (4 * j)) & 15u)) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if ((leafm >> (4 * j + b)) & 1u) pc[4 * j + b] = sink;
      *(uint2 *)&P[4u * tid + QSTR * j] = make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
      lv[j] = !(pc[4 * j + 0] & pc[4 * j + 1] & pc[4 * j + 2] & pc[4 * j + 3] & FY_SINK0);
    }
    __syncthreads();
  } else {
  if (inf) {  // (one slot per perimeter cell: no two threads share a word)
    int plr, plc;
    pslot_inv((int)ptid, &plr, &plc);
    A[PHYS((u32)(plr * TS + plc))] += inf;
  }
  __syncthreads();
  }

  // ---- doubling: A[J(z)] += A(z); J(z) <- J(J(z)).  A saturated cell adds to its lane's sink word --------------
  // One round reads (own counts, the pointers of the ancestors), waits for everybody's reads, then writes.  Two
  // rounds per trip with the pointer registers swapping roles, so that no register copies are needed.
#define FY_ROUND(PC, QN, COMBINE)                                                                                       \
  {                                                                                                               \
    u32 av[QF * 4];                                                                                              \
    _Pragma("unroll") for (int j = 0; j < QF; ++j) {                                                              \
      if (lv[j]) {                                                                                                \
        const uint4 a4 = *(const uint4 *)&A[4u * tid + QSTR * j];                                                  \
        av[4 * j + 0] = a4.x, av[4 * j + 1] = a4.y, av[4 * j + 2] = a4.z, av[4 * j + 3] = a4.w;                   \
        _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                             \
            QN[4 * j + b] = *(const uint16_t *)((const u8 *)P + (PC[4 * j + b] >> 1));                           \
      }                                                                                                           \
    }                                                                                                             \
    __syncthreads(); /* every read of this round precedes every write of this round */                           \
    _Pragma("unroll") for (int j = 0; j < QF; ++j) {                                                              \
      if (lv[j]) {                                                                                                \
        /* the four cells of a quad are neighbours in a row and, after a few rounds, mostly share their target: */ \
        /* combine them in registers then (same-address LDS atomics are served one lane after the other)        */ \
        if (COMBINE) {                                                                                            \
          const u32 t0 = PC[4 * j], t1 = PC[4 * j + 1], t2 = PC[4 * j + 2], t3 = PC[4 * j + 3];                   \
          u32 w0 = av[4 * j], w1 = av[4 * j + 1], w2 = av[4 * j + 2], w3 = av[4 * j + 3];                         \
          const bool e10 = t1 == t0, e20 = t2 == t0, e21 = t2 == t1, e30 = t3 == t0, e31 = t3 == t1, e32 = t3 == t2; \
          w0 += (e10 ? w1 : 0u) + (e20 ? w2 : 0u) + (e30 ? w3 : 0u);                                              \
          w1 = e10 ? 0u : w1 + ((!e20 && e21) ? w2 : 0u) + ((!e30 && e31) ? w3 : 0u);                             \
          w2 = (e20 || e21) ? 0u : w2 + ((!e30 && !e31 && e32) ? w3 : 0u);                                        \
          w3 = (e30 || e31 || e32) ? 0u : w3;                                                                     \
          atomicAdd((u32 *)((u8 *)A + t0), w0);                                                                   \
          if (w1) atomicAdd((u32 *)((u8 *)A + t1), w1);                                                           \
          if (w2) atomicAdd((u32 *)((u8 *)A + t2), w2);                                                           \
          if (w3) atomicAdd((u32 *)((u8 *)A + t3), w3);                                                           \
        } else {                                                                                                  \
          _Pragma("unroll") for (int b = 0; b < 4; ++b) atomicAdd((u32 *)((u8 *)A + PC[4 * j + b]), av[4 * j + b]); \
        }                                                                                                         \
        lv[j] = !(QN[4 * j + 0] & QN[4 * j + 1] & QN[4 * j + 2] & QN[4 * j + 3] & FY_SINK0);                      \
        *(uint2 *)&P[4u * tid + QSTR * j] =                                                                       \
            make_uint2(QN[4 * j + 0] | (QN[4 * j + 1] << 16), QN[4 * j + 2] | (QN[4 * j + 3] << 16));             \
      }                                                                                                           \
    }                                                                                                             \
  }
  auto anylive = [&]() {
    bool v = false;
    _Pragma("unroll") for (int j = 0; j < QF; ++j) v |= lv[j];
    return v;
  };
  int round = 0;
#ifndef FY_COPY
#pragma nounroll
  for (; round < MAXROUNDS_TILE; round += 2) {
    FY_ROUND(pc, qn, FY_COMBINE)
    if (!fx_vote_n<NW>(s_flag, 0, tid, anylive())) break;
    FY_ROUND(qn, pc, FY_COMBINE)
    if (!fx_vote_n<NW>(s_flag, 1, tid, anylive())) {
      ++round;
      break;
    }
  }
#else
#pragma nounroll
  for (; round < MAXROUNDS_TILE; ++round) {
    FY_ROUND(pc, qn, FY_COMBINE)
    _Pragma("unroll") for (int i = 0; i < QF * 4; ++i) pc[i] = qn[i];
    if (!fx_vote_n<NW>(s_flag, round, tid, anylive())) break;
  }
#endif
#undef FY_ROUND
  u32 live = 0;
  _Pragma("unroll") for (int j = 0; j < QF; ++j) live += lv[j] ? 1u : 0u;
  if ((a.ablate & 32) && tid == 0) {
    const unsigned long long r = (unsigned long long)min(round + 1, MAXROUNDS_TILE);
    const u32 w = (tr * a.ntc + tc) & 255u;
    atomicMax((unsigned long long *)&a.rcnt[512 + w], r);
    atomicAdd((unsigned long long *)&a.rcnt[768 + w], r);
  }
  if (live) atomicAdd((unsigned long long *)&a.ctrl[T_UNSAT], (unsigned long long)live);

  // ---- write the finished tile (16 B per lane) -----------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < QF; ++j) {
    const u32 l0 = 4u * tid + QSTR * j;
    const u32 c4 = cq[j];
    const uint4 a4 = *(const uint4 *)&A[l0];
    const u32 x0 = (qs & 1u) ? a4.y : a4.x, x1 = (qs & 1u) ? a4.x : a4.y;  // undo the swizzle: logical cell k sits in slot k ^ qs
    const u32 x2 = (qs & 1u) ? a4.w : a4.z, x3 = (qs & 1u) ? a4.z : a4.w;
    i32 o4[4] = {(i32)((qs & 2u) ? x2 : x0), (i32)((qs & 2u) ? x3 : x1), (i32)((qs & 2u) ? x0 : x2),
                 (i32)((qs & 2u) ? x1 : x3)};
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (((c4 >> (8 * b)) & 0xFFu) == D8_MV) o4[b] = -9999;
    i32 *dst = a.out + (size_t)(r0 + (tid >> 4) + RSTR * j - a.row_first) * a.ncol + (size_t)(c0 + lcq);
    if ((((size_t)dst) & 15) == 0) {
      *(int4 *)dst = make_int4(o4[0], o4[1], o4[2], o4[3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) dst[b] = o4[b];
    }
  }
}
