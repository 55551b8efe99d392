// tiled.h — shared declarations of the LDS-tiled accumulation engine (tiled.hip) and its
// multi-GPU driver (dist.hip).
#pragma once
#include "common.h"

#define TS 64               // tile edge (cells)
#define TCELLS (TS * TS)    // 4096
#define HW (TS + 2)         // halo'd row pitch in LDS
#define PSL 256             // perimeter slots per tile (252 used)
#define SG 8                // supertile edge in tiles (supertile = 512 x 512 cells)
#define SSL (SG * SG * PSL) // slots per supertile (16384)
#define SSHIFT 14           // log2(SSL)
#define SDONE 0x8000u       // supertile pointer saturated (local slot ids use 14 bits)
#define MAXROUNDS_SUPER 15
#define HG 4                // hypertile edge in supertiles (hypertile = 2048 x 2048 cells)
#define HCAP 24576          // super-exit ids per hypertile kept in LDS (6 B per node = 144 KB)
#define HDONE 0x8000u       // hypertile pointer saturated (local ids use 15 bits)
#define NPERIM (2 * TS + 2 * (TS - 2))
#define NONE32 0xFFFFFFFFu
#define PDONE 0x8000u       // in-tile pointer saturated at its root
#define XDONE 0x80000000u   // coarse pointer saturated
#define MAXROUNDS_TILE 13   // 2^13 > 4096 cells: more rounds mean a cycle
#define CPT (TCELLS / 256)  // cells per thread
// "the path ends on a halo sink" encoding (row blocks): bit 31 | side << 30 | column
#define ENC_SINK 0x80000000u
#define ENC_SIDE1 0x40000000u
#define ENC_COL 0x3FFFFFFFu

// ctrl slots (u64): 8..10 live for a whole pass, 12..14 are reset at the start of every exit-graph solve
enum { T_UNSAT = 8, T_SLIVE = 9, T_OVERFLOW = 10, T_XACTIVE = 12, T_NSUPER = 13, T_NHYPER = 14, T_MISS = 15 };  // ctrl slots (u64)
// T_MISS: sticky "an earlier stage of this pass fell short" bits of a pass that runs without host round trips between its
// stages (row blocks, dist.hip): the stage's own word (T_OVERFLOW, T_XACTIVE) is reused by the next stage
enum { MISS_OVERFLOW = 1, MISS_ROUNDS4 = 2, MISS_IFACE = 4 };

// slot numbering: [supertile][tile within supertile][perimeter slot] so that the exits of one
// 8x8-tile supertile are 16384 consecutive ids (the level-2 solve keeps them in LDS)
__host__ __device__ inline u32 sslot_base(u32 tr, u32 tc, u32 nstc) {
  return ((((tr >> 3) * nstc + (tc >> 3)) << SSHIFT) | ((((tr & 7) << 3) | (tc & 7)) << 8));
}
__host__ __device__ inline void sslot_inv(u32 s, u32 nstc, u32 *tr, u32 *tc, u32 *p) {
  const u32 st = s >> SSHIFT, tl = (s >> 8) & 63;
  *tr = (st / nstc) * 8 + (tl >> 3);
  *tc = (st % nstc) * 8 + (tl & 7);
  *p = s & 255;
}

// Per-slot record of the local tile pass (one u32, xrec):
//   bits 0..7   EXIT sitting on this perimeter slot: perimeter slot (0..251) of the cell it drains into, inside the
//               neighbouring tile named by bits 24..27; XR_NONE if the slot holds no exit
//   bits 8..15  ENTRY: local slot (0..251) of the exit its in-tile path reaches, XR_NONE if none (no entry, or the
//               path ends in a pit / halo sink)
//   bits 16..23 ENTRY: mask of the neighbour positions k OUTSIDE the tile whose cell drains into this one — the
//               exits whose totals the final tile pass pulls (slot_inflow()); no delivery pass, no atomics
//   bits 24..27 EXIT: the tile it drains into, 3 * (dtr + 1) + (dtc + 1)
#define XR_NONE 0xFFu
#define SBN 2048u            // cells on the boundary of a supertile (4 * 512 - 4), padded
#define SB_VALID 0x80000000u
#define SC (SG * TS)         // supertile edge in cells (512)
__host__ __device__ inline u32 xr_pack(u32 tslot, u32 tdelta, u32 link, u32 inmask) {
  return tslot | (link << 8) | (inmask << 16) | (tdelta << 24);
}
// target slot of an exit of tile (tr, tc) from the 12 bits that describe it: pslot | delta << 8
__host__ __device__ inline u32 xr_target12(u32 tr, u32 tc, u32 t12, u32 nstc);

__host__ __device__ inline u32 xr_target12(u32 tr, u32 tc, u32 t12, u32 nstc) {
  const u32 d = t12 >> 8;  // 3 * (dtr + 1) + (dtc + 1)
  const u32 q = (d * 11u) >> 5;  // d / 3 for d < 9
  return sslot_base(tr + q - 1u, tc + (d - 3u * q) - 1u, nstc) + (t12 & 0xFFu);
}

// the cells on the boundary of a supertile, numbered like the perimeter slots of a tile: row 0, row SC-1, then the
// inner cells of column 0 and of column SC-1
__host__ __device__ inline u32 sb_index(u32 R, u32 C) {
  return R == 0u ? C : (R == SC - 1u ? SC + C : (C == 0u ? 2u * SC + (R - 1u) : 2u * SC + (SC - 2u) + (R - 1u)));
}
__host__ __device__ inline void sb_cell(u32 t, u32 *R, u32 *C) {  // (t < 4 * SC - 4; the padding maps into row SC-1.. harmlessly)
  if (t < SC) *R = 0u, *C = t;
  else if (t < 2u * SC) *R = SC - 1u, *C = t - SC;
  else if (t < 2u * SC + (SC - 2u)) *R = t - 2u * SC + 1u, *C = 0u;
  else *R = (t - (2u * SC + (SC - 2u)) + 1u) & (SC - 1u), *C = SC - 1u;
}

// level-2 (supertile) solve arguments
struct SuperArgs {
  u32 nst;          // number of supertiles
  const u32 *xT;    // [nslots] start values of the solve (tile-local count of the exit on the slot)
  const u32 *xrec;  // [nslots] records of the local tile pass (above)
  // exit lists, built once per pass by k_exit_lists (one entry per exit, in slot order, at offset st << SSHIFT).
  // Everything the solves keep PER EXIT lives in list order (dense, coalesced), not per slot:
  const u64 *xmask;           // [nslots / 64] exit bitmasks written by the local tile pass
  uint16_t *xcb;              // [nslots / 64] exits of the supertile before this 64-slot word: list index of a slot =
                              // xcb[slot >> 6] + popcount(xmask[slot >> 6] below the slot)  (xl_index())
  uint16_t *xl_slot;          // [nslots] slot of the e-th exit of the supertile (local: 14 bits) | XL_SX
  uint16_t *xl_next;          // [nslots] list index of the exit its flow reaches next inside the supertile | SDONE
  u32 *scount;                // [nst] exits of the supertile
  u32 *nflag, *flagged;       // supertiles with more exits than scap (contrived rasters): count, list
  u32 *sb;          // [nst * SBN] per cell of the supertile's boundary (index: sb_index) that receives flow from OUTSIDE
                    // the supertile: SB_VALID | source mask << 16 | list index of the exit its in-tile path reaches; else 0
  uint16_t *R2L;    // [nslots, list order] list index of the last exit of the exit's path inside its supertile
  u32 *sxidL;       // [nslots, list order] dense id of a super-exit (drains into another supertile), else NONE32
  u32 *sx_slot;     // [nsuper] slot of the super-exit
  u32 *sx_n1;       // [nsuper] list position (global: supertile base + list index) of the exit the flow through the
                    // super-exit reaches first in the supertile it enters, NONE32 if none; written by k_link3
  u32 *T3;          // [nsuper] level-3 start value (= supertile-local total of the super-exit)
  u32 *xtot;        // [nslots] (final pass) total of the exit on the slot: pulled by the tile entries it drains into
  u64 *ctrl;
  u32 nstc, nhtc;   // supertiles / hypertiles per row
  u32 *hcnt;        // [nht] super-exits per hypertile (hmode 1: ids = ht*HCAP + rank)
  int hmode;        // 1: per-hypertile ids (level 3 solved in LDS), 0: one flat id range
  u32 edge_nstr;    // final pass: != 0 -> only the first and last of the edge_nstr supertile rows are solved
  u32 ntr, ntc;     // tiles per column / row (slots of tiles beyond them do not exist)
  u32 hcap;         // super-exits per hypertile that fit in LDS (HCAP; lowered by tests via PFD_TEST_HCAP)
  u8 *sover;        // [nst] set by k_exit_lists: the supertile holds more exits than the dense form keeps in LDS
  u32 scap;         // that capacity (SCAP; lowered by tests via PFD_TEST_SCAP)
  int ablate;       // DEVTOOLS experiments (PFD_SUPER_ABLATE): 1 skip the rounds, 2 skip the outputs, 4 no start-value gather
  u32 st0 = 0;      // first supertile of the launch (k_exit_lists / k_boundary_records / k_super over a band of supertile rows)
};

// level-3 (hypertile = 4x4 supertiles) solve arguments; node ids k = ht*HCAP + i, i < hcnt[ht]
struct HyperArgs {
  const u32 *sx_slot;  // [nht*HCAP] slot of the super-exit
  u32 *xtot;           // (final) the total of a super-exit also goes to its slot: the supertile it drains into pulls it
  u32 nht;
  const u32 *hcnt;
  const u32 *T3;    // [nht*HCAP] start value (supertile-local total of the super-exit)
  const u32 *J3;    // [nht*HCAP] next super-exit on the path | XDONE
  const u32 *xin3;  // [nht*HCAP] (final) flow entering the hypertile at this node
  u32 *T3out;       // [nht*HCAP] hypertile-local total (pass 1) / total (final)
  u32 *R3;          // [nht*HCAP] last super-exit of the path inside the hypertile
  u32 *hx_id;       // [nht*HCAP] dense id of a hyper-exit (drains into another hypertile), else NONE32
  u32 *hx_node;     // [nhyper] node of the hyper-exit
  u32 *T4;          // [nhyper] level-4 start value
  u64 *ctrl;
  u32 edge_nstr, nhtc;  // final pass: != 0 -> only the hypertile rows that feed the edge supertile rows
};

struct TileArgs {
  const u8 *ncode;
  const u8 *raw;   // deferred handle: raw codes; the first tile pass normalises them into ncode_w
  u8 *ncode_w;
  u64 *tcnt;       // [ntr*ntc] per tile: valid | pits << 16 | bad << 32 (raw pass only)
  u64 ntot;        // nrow * ncol
  u32 *hcnt;       // [nht] super-exit counters of the hypertiles, cleared by the local pass
  u32 nht;
  const i32 *weights;  // [n] integer payload to accumulate (accuflux), nullptr = unit weights
  u32 nrow, ncol, ntr, ntc;
  u32 row_first, row_last;  // owned rows (inclusive) of the device raster; the rest are halo rows
  u32 nstc;        // supertiles per row; slot ids are supertile-major (sslot_base)
  u32 *xT;         // [nslots] tile-local count of the exit sitting on this perimeter slot
  u32 *xrec;       // [nslots] exit direction | entry link | entry source mask (xr_pack, above)
  const u32 *xtot; // [nslots] (final pass) totals of the exits, pulled by the entries they drain into
  u64 *xmask;      // [nslots / 64] bit = the slot holds an exit (one wave ballot per 64 slots of a tile): what the
                   // exit lists of the supertile solve are built from (k_exit_lists)
  u32 *esink;      // [2*ntc*PSL] first/last tile row: halo sink an entry's in-tile path ends on
  u32 *brow_first; // [2*ncol] boundary rows: where the in-tile path of the cell ends (exit id / sink)
  u32 *haloA;      // [2*ncol] flow that reached a halo sink inside its tile
  u32 *brow_inflow;// [2*ncol] flow entering the boundary rows from the neighbouring row blocks
  u64 *ctrl;
  i32 *out;
  u32 tr_lo, tr_hi, tc_lo, tc_hi;  // the rectangle of INTERIOR tiles [tr_lo, tr_hi) x [tc_lo, tc_hi) (k_tile_*_fast,
                   // tile_fast.h); k_tile runs on the frame around it (1-D grid, frame_tile())
  u64 *rcnt;       // pfd_set_profiling(h, 2): [4][256] doubling rounds of the tile passes (local max, local sum, final
                   // max, final sum), spread over 256 words by tile id — one same-address atomic per tile costs ~12 ns
  u64 *stamps;     // DEVTOOLS: [1024][8] cycle stamps, spread over 1024 rows against same-address atomics
  int ablate;      // profiling knob (env PFD_TILE_ABLATE): bit0 skip doubling, bit4 cycle stamps; bit5 (set by
                   // pfd_set_profiling(h, 2)) counts the doubling rounds per tile into ctrl[48..51]
  // the fixed-point upstream area (wide.h): when xT64 is set, the local pass of the INTERIOR tiles also sums the 64-bit
  // weights of its cells per exit (k_tile_local_fast<.., WIDE>) — the roots are in its registers anyway
  const u64 *wrow = nullptr;   // [nrow] integer part of a cell's quantised area
  const u32 *wfrac = nullptr;  // [nrow] the fraction the columns of the row share out (w_cell)
  u64 *xT64 = nullptr;         // [nslots] 64-bit tile-local sum of the exit on the slot
};

// ---- device helpers shared by the tile kernels (tiled.hip, paths.hip) ---------------------------
// the id-th tile of the frame around the interior rectangle: top strip, bottom strip, then the left and right
// pieces of the rows in between
__host__ __device__ inline u32 frame_tiles(u32 ntr, u32 ntc, u32 tr_lo, u32 tr_hi, u32 tc_lo, u32 tc_hi) {
  return ntr * ntc - (tr_hi - tr_lo) * (tc_hi - tc_lo);
}
__host__ __device__ inline void frame_tile(u32 id, u32 ntr, u32 ntc, u32 tr_lo, u32 tr_hi, u32 tc_lo, u32 tc_hi, u32 *tr,
                                           u32 *tc) {
  const u32 top = tr_lo * ntc, bot = (ntr - tr_hi) * ntc;
  if (id < top) {
    *tr = id / ntc, *tc = id % ntc;
  } else if (id < top + bot) {
    id -= top;
    *tr = tr_hi + id / ntc, *tc = id % ntc;
  } else {
    id -= top + bot;
    const u32 side = tc_lo + (ntc - tc_hi);
    const u32 x = id % side;
    *tr = tr_lo + id / side;
    *tc = x < tc_lo ? x : tc_hi + (x - tc_lo);
  }
}

#ifdef __HIPCC__
__device__ __forceinline__ int pslot(int lr, int lc) {
  if (lr == 0) return lc;
  if (lr == TS - 1) return TS + lc;
  if (lc == 0) return 2 * TS + (lr - 1);
  if (lc == TS - 1) return 2 * TS + (TS - 2) + (lr - 1);
  return -1;
}
__device__ __forceinline__ void pslot_inv(int p, int *lr, int *lc) {
  if (p < TS) {
    *lr = 0;
    *lc = p;
  } else if (p < 2 * TS) {
    *lr = TS - 1;
    *lc = p - TS;
  } else if (p < 2 * TS + (TS - 2)) {
    *lr = p - 2 * TS + 1;
    *lc = 0;
  } else {
    *lr = p - (2 * TS + (TS - 2)) + 1;
    *lc = TS - 1;
  }
}

// slot of the cell that neighbours perimeter cell (lr, lc) of tile (tr, tc) in direction k and lies OUTSIDE the tile
// (the caller knows it does): the cell sits on the perimeter of one of the 8 neighbouring tiles
__device__ __forceinline__ u32 nbr_slot(u32 tr, u32 tc, int lr, int lc, int k, u32 nstc) {
  const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);  // in [-1, TS]
  const u32 ttr = tr + (u32)(nr >> 6), ttc = tc + (u32)(nc >> 6);  // (arithmetic shift: -1 -> -1, 0..63 -> 0, 64 -> 1)
  return sslot_base(ttr, ttc, nstc) + (u32)pslot(nr & (TS - 1), nc & (TS - 1));
}
// the 12 bits that name the cell an exit drains into: perimeter slot inside its tile | tile delta << 8
__device__ __forceinline__ u32 xr_t12(int nr, int nc) {  // (nr, nc) in [-1, TS], outside [0, TS)
  return (u32)pslot(nr & (TS - 1), nc & (TS - 1)) | ((u32)(3 * ((nr >> 6) + 1) + (nc >> 6) + 1) << 8);
}
// target slot of the exit on slot `s` with record `rec`
__device__ __forceinline__ u32 xr_target(u32 s, u32 rec, u32 nstc) {
  u32 tr, tc, p;
  sslot_inv(s, nstc, &tr, &tc, &p);
  return xr_target12(tr, tc, (rec & 0xFFu) | ((rec >> 16) & 0xF00u), nstc);
}
// flow entering the tile at perimeter cell (lr, lc): the totals of the exits named by the record's source mask
__device__ __forceinline__ u32 slot_inflow(const u32 *__restrict__ xtot, u32 rec, u32 tr, u32 tc, int lr, int lc, u32 nstc) {
  u32 m = (rec >> 16) & 0xFFu, v = 0;
  while (m) {
    const int k = __ffs((int)m) - 1;
    m &= m - 1u;
    v += xtot[nbr_slot(tr, tc, lr, lc, k, nstc)];
  }
  return v;
}
// list index (inside its supertile) of the exit on slot s
__device__ __forceinline__ u32 xl_index(const u64 *__restrict__ xmask, const uint16_t *__restrict__ xcb, u32 s) {
  return (u32)xcb[s >> 6] + (u32)__popcll(xmask[s >> 6] & ((1ull << (s & 63u)) - 1ull));
}

// LDS layout of the staged codes: 66 rows (1-cell halo) x 72 bytes; column lc in [-1, 64] lives
// at byte lc + 4 of its row, so that the 64 own columns start on a dword boundary and the halo'd
// row is exactly 18 dwords [c0-4, c0+68) of the raster row.
#define CP 72
// swizzled LDS index of tile cell z (an involution; see k_tile)
#define PHYS(z) ((z) ^ (((z) >> 5) & 3u))
#define CODE(lr, lc) code[((lr) + 1) * CP + (lc) + 4]
#define QPT (TCELLS / 4 / 256)  // quads (4 consecutive cells) per thread


// issue the (<= 5 per thread) unconditional, possibly unaligned dword loads of a tile's halo'd
// codes; v[k] is dword idx = tid + 256*k of the 66 x 18 staging area, out-of-raster bytes = nodata
// GUARD: the buffer has no slack behind its last byte (a caller's raw raster)
template <bool GUARD = false>
__device__ __forceinline__ void stage_load(const u8 *__restrict__ ncode, u32 nrow, u32 ncol, i64 r0, i64 c0, u32 tid,
                                           u32 (&v)[5], u64 ntot = 0) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const u32 idx = tid + 256u * k;  // dword idx of the 66 x 18 staging area (1188 used)
    const u32 hr = idx / 18u, d = idx - hr * 18u;
    const i64 gr = r0 + (i64)hr - 1;
    const i64 cs = c0 - 4 + 4 * (i64)d;  // first raster column of this dword
    // a load inside a branch would be waited for on the spot: load from a clamped address and
    // mask afterwards.  Reading up to 3 bytes past a row end is fine: the bytes are masked and the
    // allocation carries slack.
    const i64 crr = gr < 0 ? 0 : (gr >= (i64)nrow ? (i64)nrow - 1 : gr);
    const i64 ccs = cs < 0 ? 0 : (cs >= (i64)ncol ? (i64)ncol - 1 : cs);
    u32 w;
    size_t off = (size_t)crr * ncol + (size_t)ccs;
    u32 sh = 0;
    if (GUARD && off + 4 > ntot) {  // last bytes of the raster: load the final dword, shift down
      sh = 8u * (u32)(off + 4 - ntot);
      off = ntot - 4;
    }
    __builtin_memcpy(&w, ncode + off, 4);
    if (GUARD) w >>= sh;
    const bool rowok = idx < HW * 18u && gr >= 0 && gr < (i64)nrow;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const i64 col = cs + b;
      if (!rowok || col < 0 || col >= (i64)ncol) w = (w & ~(0xFFu << (8 * b))) | (D8_MV << (8 * b));
    }
    v[k] = w;
  }
}
// the same for an INTERIOR tile (the tile and its halo ring lie inside the raster, at least 4 columns from its
// left and right edges): no clamping, no masking — a fifth of the VALU work of the general form, and the tile
// kernels are VALU-bound
__device__ __forceinline__ void stage_load_interior(const u8 *__restrict__ ncode, u32 ncol, i64 r0, i64 c0, u32 tid,
                                                    u32 (&v)[5]) {
  const u8 *base = ncode + (size_t)(r0 - 1) * ncol + (size_t)(c0 - 4);
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const u32 idx = tid + 256u * k;
    u32 hr = idx / 18u;
    const u32 d = idx - hr * 18u;
    hr = hr < HW - 1u ? hr : HW - 1u;  // (dwords past the staging area re-read its last row and are dropped)
    __builtin_memcpy(&v[k], base + (size_t)hr * ncol + 4u * d, 4);
  }
}
// the cheap form wherever the staging window lies inside the raster (a workgroup-uniform branch), else the general one
__device__ __forceinline__ void stage_load_auto(const u8 *__restrict__ ncode, u32 nrow, u32 ncol, i64 r0, i64 c0, u32 tid,
                                                u32 (&v)[5]) {
  if (r0 >= 1 && c0 >= 4 && r0 + TS + 1 <= (i64)nrow && c0 + TS + 4 <= (i64)ncol) stage_load_interior(ncode, ncol, r0, c0, tid, v);
  else stage_load(ncode, nrow, ncol, r0, c0, tid, v);
}
__device__ __forceinline__ void stage_store(u8 *code, u32 tid, const u32 (&v)[5]) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const u32 idx = tid + 256u * k;
    if (idx < HW * 18u) ((u32 *)code)[idx] = v[k];
  }
}
#endif  // __HIPCC__

struct TiledRun {
  pfd_raster *h = nullptr;
  u32 ntr = 0, ntc = 0, nexits = 0;
  size_t nslots = 0;
  bool supported = false, is_block = false, coarse_done = false, force_flat = false;
  DevBuf slots, sx, esink, bnd;  // per-slot arrays (7 x nslots), per-super-exit arrays (5 x cap)
  u32 nst = 0, nstc = 0, nsuper = 0, nht = 0, nhtc = 0, nhyper = 0;
  DevBuf sbbuf, l3, l4, hcntbuf, tcntbuf, iface_buf, stampbuf, soverbuf, rcntbuf, xmaskbuf, xlbuf, scountbuf, flaggedbuf, xcbbuf;
  bool fused_norm = false;  // this run's first tile pass normalises a deferred handle
#ifndef PFD_PATCH_DEFAULT
#define PFD_PATCH_DEFAULT 0
#endif
  bool use_patch = PFD_PATCH_DEFAULT != 0;  // local pass of the interior tiles: k_tile_local_patch (tile_patch.h) instead of k_tile_local_fast
  int rounds4 = 0, extra_rounds = 0;  // level-4 rounds issued without a host check / added after a miss
  bool short_of_rounds = false;
  u32 *xT = nullptr, *xrec = nullptr, *xtot = nullptr, *sxidL = nullptr, *sx_slot = nullptr,
      *sx_n1 = nullptr;
  uint16_t *R2L = nullptr;
  SuperArgs sa{};
  u32 *brow_first = nullptr, *haloA = nullptr, *haloL = nullptr, *brow_sink = nullptr, *brow_inflow = nullptr;
  u32 *Tc = nullptr, *Tn = nullptr, *Jc = nullptr, *Jn = nullptr, *xin3 = nullptr, *R3 = nullptr, *hx_id = nullptr;
  size_t n3cap = 0, n4cap = 0;
  u32 *J4fin = nullptr;
  int solve_exits(const u32 *start, i64 *launches, bool cleared = false, bool edge_down = false);
  // phase A with the supertile-local part of the exit graph (exit lists, boundary records, first supertile solve) of a BAND
  // of supertile rows running on the handle's second stream beside the local tile pass of the next band (PFD_BANDS)
  int phase_a_bands(int bands);
  bool setup_only = false, up_done = false;  // solve_exits: set up the solve and return / the first supertile solve has run
  hipEvent_t band_ev[16] = {};
  ~TiledRun() {
    for (hipEvent_t e : band_ev)
      if (e) (void)hipEventDestroy(e);
  }
  bool edge_down_now = false;
  int level3_flat(i64 *launches);
  int level3_flat_nosync(i64 *launches);
  bool flat_nosync = false;  // this solve ran level 3 as one flat forest without host round trips (few hypertiles)
  int level3_hyper(i64 *launches);
  int level4_down(i64 *launches);
  int resolve_with_inflow(i64 *launches);
  TileArgs a{};
  int init(pfd_raster *hh, i32 *out_dev);
  bool overflowed = false;
  int phase_a();
  int phase_a_checked();
  int phase_a_check();
  int phase_b(int *complete);
  // phase_b in two halves: everything up to the final tile pass issued on the stream; the verdict from the pass's
  // control words (48 u64, copied by the caller behind whatever else it wants to see at the same synchronisation)
  int phase_b_issue();
  int phase_b_collect(const u64 *c0, int *complete);
  // stream-ordered stage checks of a pass without host round trips: phase A's overflow / level-4 budget folded into
  // ctrl[T_MISS]; the whole pass folded into one word (2 = fine, 1 = redo the pass, 0 = failed) for an agreement
  int stage_verdict_a();
  int block_verdict(int *flag_dev, bool host_ok);
  bool iface_doubling = false;  // the interface forest by doubling rounds (after a chase ran out of hops)
  u64 last_miss = 0;            // ctrl[T_MISS] of the pass phase_b_collect last looked at
  // phase A's control words 8..15 (copied by the caller): does phase A have to run again (flat level 3 after an
  // overflow, more level-4 rounds after a short budget)?  Adjusts the run's settings when it does.
  bool phase_a_needs_redo(const u64 *c8, int tries);
};

int pfd_doubling_rounds(pfd_raster *h, u32 *T[3], u32 *J[2], u32 n, int first_batch, bool check, bool *done,
                        int *rounds_issued, i64 *launches, const u64 *ncnt = nullptr, bool prepared = false);
