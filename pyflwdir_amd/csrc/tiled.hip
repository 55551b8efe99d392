// tiled.hip — LDS-tiled fast path for FlwdirRaster.upstream_area(unit="cell")
// (reference pyflwdir/pyflwdir.py:770-801 = core.idxs_seq + streams.accuflux on unit weights).
//
// Integer accumulation is associative (int32 wrap-around included), so the serial ordering of
// the reference is not needed.  The raster is cut into 64x64-cell tiles; one 256-thread
// workgroup owns one tile and keeps its whole state in LDS (160 KB/CU on MI355X, 5-6 tiles
// resident per CU).  Inside a tile — and again on the graph of tile exits — subtree sums are
// computed by POINTER DOUBLING instead of a dependent walk:
//
//     A_0(y) = w(y),  J_0(z) = downstream cell of z
//     round k:  A[J_k(z)] += A_k(z)  for every z whose 2^k-th ancestor J_k(z) exists,
//               J_{k+1}(z) = J_k(J_k(z))
//     => A_k(y) = sum of w over the upstream cells of y closer than 2^k   (exact, any order)
//
// log2(longest in-tile path) rounds, every lane busy, no dependent chains: the work per round
// is a handful of LDS reads, one non-returning ds_add_u32 and one ds_write_b16 per cell (final
// pass; the local pass needs no values, see phase 1).
// A pointer that runs off the end of its path saturates at the path's last cell ("root": the
// cell where the flow leaves the tile, or a pit) and is flagged done; the roots of the
// perimeter cells are exactly the links the coarse graph needs.
//
//   phase 1  k_tile<false>   per tile: per perimeter slot the local count of every cell that
//                            drains out of the tile ("exit") and the slot it drains into; for
//                            every perimeter cell that receives flow from outside ("entry") the
//                            exit its in-tile path ends at ("link").  Only roots and counts per
//                            root are needed here, so this pass jumps pointers WITHOUT values
//                            (gather-only) and adds every cell's weight to its root afterwards.
//                            On a deferred handle (RAW) the pass also decodes / validates /
//                            counts the raw codes and writes the normalised ones.
//   phase 2  exit graph      the exits form a forest ~50x smaller than the raster:
//                            exit e -> link(target(e)).  Solved hierarchically with the same
//                            doubling, bottom-up then top-down:
//                              level 2  k_super   8x8-tile supertile in LDS (16384 slots)
//                              level 3  k_hyper   4x4-supertile hypertile in LDS (dense ids of
//                                                 the exits that leave their supertile)
//                              level 4  k_coarse_round  global memory, one launch per round,
//                                                 three rotating value buffers, fixed round budget
//                            -> TOTAL count at every exit, inflow at every tile entry.
//   phase 3  k_tile<true>    per tile: the doubling again with entries weighted 1 + inflow;
//                            the finished tile is written to HBM once, coalesced.
//
// HBM traffic: 2 x 1 B/cell (codes) + 4 B/cell (result) + ~1 B/cell of perimeter records.
// Cells on or upstream of a cycle never saturate; the run counts saturated cells and exits, and
// pfd_upstream_area_cell falls back to the level engine when a count is short.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "common.h"
#include "tiled.h"

#define TSTAMP(slot)                                                                                        \
  if (a.ablate & 16) {                                                                                      \
    __syncthreads();                                                                                        \
    if (tid == 0) {                                                                                         \
      const u64 t_ = __builtin_readcyclecounter();                                                          \
      atomicAdd((unsigned long long *)&a.stamps[(((tr * a.ntc + tc) & 1023u) << 3) + (FINAL ? 4 : 0) + slot], \
                (unsigned long long)(t_ - tprev));                                                        \
      tprev = t_;                                                                                           \
    }                                                                                                       \
  }

// RAW (first pass of a deferred handle): the staged bytes are the caller's raw codes; the tile
// normalises its own cells on the fly (decode + pit rule + validation + counts = k_normalise,
// order.hip), keeps the result in LDS and writes it to the handle's ncode for every later pass.
// A neighbour's raw byte serves wherever a normalised one would: normalisation only ever turns a
// cell into a pit (0) when its TARGET is nodata, and the halo ring is only asked "are you nodata"
// and "do you point at this valid cell".
// INT: the tile is an interior one — it lies, halo ring and the 4 staging columns either side included, inside
// the raster, and the raster is no row block.  All bounds handling compiles away; the branch is per workgroup.
// PERIM (local pass of a WHOLE raster): only the exits of the tile need a count, so A holds PREP words per perimeter
// slot instead of one per cell (8 KB instead of 16.6 KB of LDS: a sixth workgroup per CU) — cells whose root is a pit
// issue no atomic at all, and the PREP replicas of a slot (picked by lane) spread the same-address ones.
#define PREP 8
template <bool FINAL, bool RAW, bool INT, bool PERIM>
__device__ __forceinline__ void tile_body(const TileArgs &a, u32 *A, uint16_t *P, u8 *code, u64 *s_cnt, const u32 tr,
                                          const u32 tc) {
  u64 tprev = __builtin_readcyclecounter();
  const u32 tid = threadIdx.x;
  const u32 sbase = sslot_base(tr, tc, a.nstc);  // first of this tile's 256 slot ids
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  if (!FINAL && tc == 0 && tr == 0)  // (the exit-graph solve starts after this kernel: no memset launch)
    for (u32 q = tid; q < a.nht; q += 256u) a.hcnt[q] = 0;

  // ---- stage the tile's codes (+halo) as dwords: all loads in flight before the first store ----
  u32 cq[QPT];      // FINAL: the codes of the thread's quads
  {
    u32 v[5];
    if (FINAL) {
#pragma unroll
      for (int j = 0; j < QPT; ++j) {  // own quads straight from HBM (clamped address, masked afterwards)
        const u32 l0 = 4u * tid + 1024u * j;
        const i64 gr = r0 + (l0 >> 6), gc0 = c0 + (l0 & 63);
        u32 w;
        if (INT) {
          __builtin_memcpy(&w, a.ncode + (size_t)gr * a.ncol + (size_t)gc0, 4);
        } else {
          const i64 crr = gr >= (i64)a.nrow ? (i64)a.nrow - 1 : gr;
          const i64 ccs = gc0 >= (i64)a.ncol ? (i64)a.ncol - 1 : gc0;
          __builtin_memcpy(&w, a.ncode + (size_t)crr * a.ncol + (size_t)ccs, 4);  // (ncode carries slack)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (gr >= (i64)a.nrow || gc0 + b >= (i64)a.ncol) w = (w & ~(0xFFu << (8 * b))) | (D8_MV << (8 * b));
        }
        cq[j] = w;
      }
    } else if (INT) {
      stage_load_interior(RAW ? a.raw : a.ncode, a.ncol, r0, c0, tid, v);
    } else if (RAW) {
      stage_load<true>(a.raw, a.nrow, a.ncol, r0, c0, tid, v, a.ntot);
    } else {
      stage_load(a.ncode, a.nrow, a.ncol, r0, c0, tid, v);
    }
    u32 nbad = 0, cnt = 0;  // RAW: cnt = valid | pits << 10 | bad << 20 of this thread's 16 cells
    if (RAW && !INT && (a.row_first > 0 || a.row_last + 1 < a.nrow)) {
      // halo rows of a row block: weightless sinks (D8_HALO) wherever they are seen, ring included
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const u32 idx = tid + 256u * k;
        const u32 hr = idx / 18u, d = idx - hr * 18u;
        const i64 gr = r0 + (i64)hr - 1;
        if (idx < HW * 18u && gr >= 0 && gr < (i64)a.nrow && (gr < (i64)a.row_first || gr > (i64)a.row_last)) {
          const bool own = hr >= 1 && hr <= TS && d >= 1 && d <= 16;
          u32 w = v[k];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const u32 x = (w >> (8 * b)) & 0xFFu;
            if (x == D8_MV) continue;
            const bool ok = (x & (x - 1)) == 0u || x == 255u;
            if (!ok && own) ++nbad;
            w = (w & ~(0xFFu << (8 * b))) | ((ok ? D8_HALO : D8_MV) << (8 * b));
          }
          v[k] = w;
        }
      }
    }
    u32 inf = 0;
    if (FINAL && tid < NPERIM) {  // flow entering at this perimeter cell: pulled from the exits that drain into it
      int lr, lc;
      pslot_inv((int)tid, &lr, &lc);
      inf = slot_inflow(a.xtot, a.xrec[sbase + tid], tr, tc, lr, lc, a.nstc);
    }
    if (!FINAL) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const u32 idx = tid + 256u * k;
        if (idx < HW * 18u) ((u32 *)code)[idx] = v[k];
      }
      __syncthreads();
    }
    TSTAMP(0)

    // ---- initial weights and downstream pointers -------------------------------------------
    // a thread owns QPT quads of 4 consecutive cells: l0 = 4*tid + 1024*j (16 lanes = one row).
    // A and P are stored SWIZZLED (PHYS): with the linear layout the lanes of a half-wave touch
    // words 4 apart for a given register slot -> 8 banks, 4-way conflicts on every gather and
    // atomic of the doubling.  PHYS xors the two low index bits with bits 5..6, a per-lane
    // constant qs for own cells: register slot b holds logical cell l0 + (b ^ qs), the quad is
    // still one aligned 16-byte vector, and pointers are kept in physical form throughout.
    const u32 qs = (tid >> 3) & 3u;
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const u32 l0 = 4u * tid + 1024u * j;
      const int lr = l0 >> 6, lc0 = l0 & 63;
      const u32 c4 = FINAL ? cq[j] : *(const u32 *)&CODE(lr, lc0);
      u32 w4[4], p4[4];
      // integer accuflux: the payload instead of unit weights (clamped address, masked by the code)
      const i64 wrow = (INT ? (i64)(r0 + lr) : (i64)min((i64)(r0 + lr), (i64)a.nrow - 1)) * (i64)a.ncol;
      u32 n4 = 0;  // RAW: the normalised codes of the quad
      const bool halorow = !INT && ((i64)r0 + lr < (i64)a.row_first || (i64)r0 + lr > (i64)a.row_last);
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // branch-free: (dr, dc) from two packed 2-bit tables
        const u32 b = (u32)s ^ qs;   // logical position in the quad of register slot s
        u32 c = (c4 >> (8 * b)) & 0xFFu;
        const u32 l = l0 + (u32)s;   // physical index of that cell
        const int k = (int)__builtin_ctz(c | 0x100u);                 // slot of a direction code (8 for 0)
        const int dr = (int)((0x101A9u >> (2 * k)) & 3u) - 1;         // dr+1 per slot, slot 8 (no direction) -> 0
        const int dc = (int)((0x1901Au >> (2 * k)) & 3u) - 1;         // dc+1 per slot, slot 8 -> 0
        const int nr = lr + dr, nc = lc0 + (int)b + dc;
        bool isdir = d8_is_dir(c);
        if (RAW) {  // normalise (k_normalise, order.hip): pit rule, validation, counts
          const u32 t = CODE(nr, nc);  // code of the target, the cell itself when c is no direction
          const bool special = c == D8_MV || (c == D8_HALO && halorow);  // (254 elsewhere is a bad code)
          const bool ispit = c == 0u || c == 255u;
          const bool isbad = !special && !isdir && !ispit;
          isdir = isdir && t != D8_MV;  // flow off the raster / into nodata ends here
          const bool valid = !special && !isbad;
          cnt += (valid ? 1u : 0u) + ((valid && !isdir) ? 1u << 10 : 0u) + (isbad ? 1u << 20 : 0u);
          c = special ? c : (isbad ? (u32)D8_MV : (isdir ? c : 0u));
          n4 |= c << (8 * b);
        }
        const bool go = isdir && (unsigned)nr < TS && (unsigned)nc < TS;
        // nodata, pit, halo sink, or flow leaves the tile: the cell is its own root
        // pointers are kept as BYTE offsets into P (2 x the physical index): the gather needs one AND, the
        // atomic's address one shift more — the tile kernels are VALU-bound
        p4[s] = go ? PHYS((u32)(nr * TS + nc)) << 1 : ((l << 1) | PDONE);
        u32 wv = 1u;
        if (a.weights != nullptr)
          wv = (u32)a.weights[wrow + (INT ? (i64)(c0 + lc0) + (i64)b : min((i64)(c0 + lc0) + (i64)b, (i64)a.ncol - 1))];
        w4[s] = (c != D8_MV && c != D8_HALO) ? wv : 0u;
      }
      if (FINAL) {
        *(uint4 *)&A[l0] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
      } else if (!PERIM) {
        *(uint4 *)&A[l0] = make_uint4(0u, 0u, 0u, 0u);  // (the weights are added to the roots after the pointer jumping)
      } else if (j < 2) {
        *(uint4 *)&A[4u * tid + 1024u * j] = make_uint4(0u, 0u, 0u, 0u);  // 256 x PREP words
      }
      *(uint2 *)&P[l0] = make_uint2(p4[0] | (p4[1] << 16), p4[2] | (p4[3] << 16));
      if (RAW) {
        *(u32 *)&CODE(lr, lc0) = n4;  // (readers of the raw byte only ask "== nodata": unchanged)
        const i64 gr = r0 + lr, gc0 = c0 + lc0;
        if (INT || (gr < (i64)a.nrow && gc0 < (i64)a.ncol)) {
          u8 *dst = a.ncode_w + (size_t)gr * a.ncol + (size_t)gc0;
          if (INT || gc0 + 3 < (i64)a.ncol) {
            __builtin_memcpy(dst, &n4, 4);  // (possibly unaligned) dword store
          } else {
            for (int k = 0; k < 4 && gc0 + k < (i64)a.ncol; ++k) dst[k] = (u8)(n4 >> (8 * k));
          }
        }
      }
    }
    if (RAW) {  // counts of the tile -> tcnt (summed by k_tile_counts: no same-address atomics)
      u64 pk = (u64)(cnt & 1023u) | ((u64)((cnt >> 10) & 1023u) << 16) | ((u64)((cnt >> 20) + nbad) << 32);
      for (int o = 32; o > 0; o >>= 1) pk += __shfl_down(pk, o);
      if ((tid & 63u) == 0) s_cnt[tid >> 6] = pk;
      __syncthreads();
      if (tid == 0) a.tcnt[(size_t)tr * a.ntc + tc] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    if (FINAL && !INT && (a.row_first > 0 || a.row_last + 1 < a.nrow)) {
      // row blocks: flow entering the owned boundary rows from the neighbouring GPUs
      __syncthreads();
      if (tid < 2 * TS) {
        const u32 side = tid >> 6, lc = tid & 63;
        const i64 gr = side ? (i64)a.row_last : (i64)a.row_first;
        const i64 lr = gr - r0, gc = c0 + lc;
        if (lr >= 0 && lr < TS && gc < (i64)a.ncol && !(side && a.row_last == a.row_first)) {
          const u32 c = a.ncode[(size_t)gr * a.ncol + (size_t)gc];
          if (c != D8_MV && c != D8_HALO) {
            u32 v = a.brow_inflow[side * a.ncol + gc];
            if (!side && a.row_last == a.row_first) v += a.brow_inflow[a.ncol + gc];  // one-row block: both sides
            if (v) A[PHYS((u32)(lr * TS + lc))] += v;
          }
        }
      }
    }
    if (FINAL) {
      __syncthreads();
      if (tid < NPERIM && inf) {  // flow entering the tile from its neighbours
        int lr, lc;
        pslot_inv((int)tid, &lr, &lc);
        A[PHYS((u32)(lr * TS + lc))] += inf;
      }
    }
  }
  __syncthreads();
  TSTAMP(1)

  // ---- pointer doubling ------------------------------------------------------------------------
  u32 pc[QPT * 4];   // current pointer word (2 x ancestor | PDONE) of own cell 4*j+b
  u32 live = 0;      // bit j: own quad j still has an unsaturated pointer
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const uint2 pp = *(const uint2 *)&P[l0];
    pc[4 * j + 0] = pp.x & 0xFFFFu;
    pc[4 * j + 1] = pp.x >> 16;
    pc[4 * j + 2] = pp.y & 0xFFFFu;
    pc[4 * j + 3] = pp.y >> 16;
    if (!(pp.x & pp.y & (pp.x >> 16) & (pp.y >> 16) & PDONE)) live |= 1u << j;
  }
  int round = 0;
  if (!(a.ablate & 1)) {
    // Branches are per QUAD only: inside a live quad all four cells run the same instruction
    // stream; a saturated cell re-reads its root's pointer (its own value) and adds to a sink word,
    // which is cheaper than four exec-mask regions per quad and round.
    const u32 sink = (TCELLS + (tid & 63u)) * 4u;  // byte offset of this lane's sink word of A
    const i64 r0_ = r0;                            // (the tile's first row; r0..r3 below are roots)
    if (!FINAL) {
      // The local pass only needs ROOTS (where does an entry's path end, which exit does a cell drain to) and
      // the count per root (the local count of an exit) — not the count of every cell.  So it jumps pointers
      // without carrying values, J_{k+1}(z) = J_k(J_k(z)): gather-only, half the LDS work of the value-carrying
      // doubling, and a read that sees a pointer already advanced by its owner just jumps further (every value
      // a pointer ever holds is an ancestor), so one barrier per round is enough.  Afterwards every cell adds
      // its weight to its root.
      for (; round < MAXROUNDS_TILE; ++round) {
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
          if (live & (1u << j)) {
            u32 q[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) q[b] = *(const uint16_t *)((const u8 *)P + (pc[4 * j + b] & 0x1FFEu));
#pragma unroll
            for (int b = 0; b < 4; ++b) pc[4 * j + b] = q[b];
            if (q[0] & q[1] & q[2] & q[3] & PDONE) live &= ~(1u << j);
            *(uint2 *)&P[4u * tid + 1024u * j] = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
          }
        }
        if (!__syncthreads_or((int)live)) break;
      }
      // The four cells of a quad are neighbours in a row and mostly share their root: combine them in registers
      // first (same-address LDS atomics are served one lane per cycle).
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        u32 r0 = pc[4 * j + 0], r1 = pc[4 * j + 1], r2 = pc[4 * j + 2], r3 = pc[4 * j + 3];
        // the weights again (from the staged codes; keeping 16 of them in registers through the rounds costs a
        // wave of occupancy): register slot s holds logical cell s ^ qs of the quad
        u32 wq[4];
        {
          const u32 l0 = 4u * tid + 1024u * j;
          const int lr = l0 >> 6, lc0 = l0 & 63;
          const u32 qs = (tid >> 3) & 3u;
          const u32 c4 = *(const u32 *)&CODE(lr, lc0);
          const i64 wrow = (INT ? (i64)(r0_ + lr) : (i64)min((i64)(r0_ + lr), (i64)a.nrow - 1)) * (i64)a.ncol;
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) {
            const u32 b = (u32)sl ^ qs;
            const u32 c = (c4 >> (8 * b)) & 0xFFu;
            u32 wv = 1u;
            if (a.weights != nullptr)
              wv = (u32)a.weights[wrow + (INT ? (i64)(c0 + lc0) + (i64)b : min((i64)(c0 + lc0) + (i64)b, (i64)a.ncol - 1))];
            wq[sl] = (c != D8_MV && c != D8_HALO) ? wv : 0u;
          }
        }
        u32 w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
        {
          const bool e10 = r1 == r0;
          w0 += e10 ? w1 : 0u;
          w1 = e10 ? 0u : w1;
          const bool e20 = r2 == r0, e21 = r2 == r1;
          w0 += e20 ? w2 : 0u;
          w1 += (!e20 && e21) ? w2 : 0u;
          w2 = (e20 || e21) ? 0u : w2;
          const bool e30 = r3 == r0, e31 = r3 == r1, e32 = r3 == r2;
          w0 += e30 ? w3 : 0u;
          w1 += (!e30 && e31) ? w3 : 0u;
          w2 += (!e30 && !e31 && e32) ? w3 : 0u;
          w3 = (e30 || e31 || e32) ? 0u : w3;
        }
        // (a cell that never saturated sits on or upstream of a cycle: the pass is redone by the level engine)
        auto push = [&](u32 r, u32 w) {
          if (!w || r < PDONE) return;
          if (!PERIM) {
            atomicAdd((u32 *)((u8 *)A + ((r & 0x1FFEu) << 1)), w);
          } else {
            // perimeter slot of the root, branch-free (pslot(), tiled.h): row 0 -> lc, row 63 -> 64 + lc,
            // column 0 -> 127 + lr, column 63 -> 189 + lr; (x + 1) & 62 == 0 exactly for x in {0, 63}
            const u32 L = PHYS((r & 0x1FFEu) >> 1);  // logical index of the root
            const u32 lr = L >> 6, lc = L & 63u;
            const bool tb = ((lr + 1u) & 62u) == 0u, lrc = ((lc + 1u) & 62u) == 0u;
            const u32 s_tb = lc + ((lr + 1u) & 64u);
            const u32 s_lr = 127u + lr + (((lc + 1u) & 64u) - (((lc + 1u) >> 5) & 2u));
            const u32 ps = tb ? s_tb : s_lr;
            if (tb || lrc) atomicAdd(&A[ps * PREP + (tid & (PREP - 1u))], w);  // (a pit inside the tile: nobody asks)
          }
        };
        push(r0, w0);
        push(r1, w1);
        push(r2, w2);
        push(r3, w3);
      }
    } else
    for (; round < MAXROUNDS_TILE; ++round) {
      u32 av[QPT * 4], q[QPT * 4];
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (live & (1u << j)) {
          const uint4 a4 = *(const uint4 *)&A[4u * tid + 1024u * j];
          av[4 * j + 0] = a4.x;
          av[4 * j + 1] = a4.y;
          av[4 * j + 2] = a4.z;
          av[4 * j + 3] = a4.w;
#pragma unroll
          for (int b = 0; b < 4; ++b) q[4 * j + b] = *(const uint16_t *)((const u8 *)P + (pc[4 * j + b] & 0x1FFEu));
        }
      }
      __syncthreads();  // every read of this round precedes every write of this round
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (live & (1u << j)) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const u32 p = pc[4 * j + b];
            // a saturated cell adds to a per-lane sink word nobody reads: adding to its root would pile
            // same-address LDS atomics onto the few roots of the tile (n-way bank conflicts).  Its pointer
            // needs no select: a root points at itself, so a saturated pointer re-reads its own value.
            atomicAdd((u32 *)((u8 *)A + (p >= PDONE ? sink : (p & 0x1FFEu) << 1)), av[4 * j + b]);
            pc[4 * j + b] = q[4 * j + b];
          }
          if (pc[4 * j + 0] & pc[4 * j + 1] & pc[4 * j + 2] & pc[4 * j + 3] & PDONE) live &= ~(1u << j);
          *(uint2 *)&P[4u * tid + 1024u * j] =
              make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
        }
      }
      if (!__syncthreads_or((int)live)) break;
    }
    if ((a.ablate & 32) && tid == 0) {  // pfd_set_profiling(h, 2): rounds this tile needed (max and sum over the tiles)
      const unsigned long long r = (unsigned long long)min(round + 1, MAXROUNDS_TILE);
      const u32 w = (tr * a.ntc + tc) & 255u;
      atomicMax((unsigned long long *)&a.rcnt[(FINAL ? 512 : 0) + w], r);
      atomicAdd((unsigned long long *)&a.rcnt[(FINAL ? 768 : 256) + w], r);
    }
  }
  TSTAMP(2)
  // a cell on or upstream of a cycle never saturates: count their quads (normally zero, so that no
  // same-address global atomic is issued at all — 25k of them would cost ~0.3 ms)
  if (live) atomicAdd((unsigned long long *)&a.ctrl[T_UNSAT], (unsigned long long)__popc(live));
  __syncthreads();

  if (FINAL) {
    // ---- write the owned rows of the finished tile (16 B per lane) ---------------------------
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const u32 l0 = 4u * tid + 1024u * j;
      const int lr = l0 >> 6, lc0 = l0 & 63;
      const i64 gr = r0 + lr, gc0 = c0 + lc0;
      if (!INT && (gr < (i64)a.row_first || gr > (i64)a.row_last || gc0 >= (i64)a.ncol)) continue;
      const u32 c4 = cq[j];
      const uint4 a4 = *(const uint4 *)&A[l0];
      const u32 qs = (tid >> 3) & 3u;  // undo the swizzle: logical cell k sits in slot k ^ qs
      const u32 x0 = (qs & 1u) ? a4.y : a4.x, x1 = (qs & 1u) ? a4.x : a4.y;
      const u32 x2 = (qs & 1u) ? a4.w : a4.z, x3 = (qs & 1u) ? a4.z : a4.w;
      i32 o4[4] = {(i32)((qs & 2u) ? x2 : x0), (i32)((qs & 2u) ? x3 : x1), (i32)((qs & 2u) ? x0 : x2),
                   (i32)((qs & 2u) ? x1 : x3)};
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (((c4 >> (8 * b)) & 0xFFu) == D8_MV) o4[b] = -9999;
      i32 *dst = a.out + (size_t)(gr - a.row_first) * a.ncol + (size_t)gc0;
      if ((INT || gc0 + 3 < (i64)a.ncol) && (((size_t)dst) & 15) == 0) {
        *(int4 *)dst = make_int4(o4[0], o4[1], o4[2], o4[3]);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (gc0 + b < (i64)a.ncol) dst[b] = o4[b];
      }
    }
    TSTAMP(3)
    return;
  }

  if (a.ablate & 8) return;
  // ---- perimeter records for the coarse graph ------------------------------------------------
  u32 xt = 0, xt12 = XR_NONE, inmask = 0;  // xt12: target of the exit (xr_t12)
  int plr = 0, plc = 0;
  if (tid < NPERIM) {
    pslot_inv((int)tid, &plr, &plc);
    const u32 c = CODE(plr, plc);
    if (c != D8_MV && c != D8_HALO) {
      if (d8_is_dir(c)) {  // exit?
        const int k = d8_slot(c);
        const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
        if ((unsigned)nr >= TS || (unsigned)nc >= TS) {  // (the target is inside the raster and valid: normalised codes)
          xt12 = xr_t12(nr, nc);
          if (!PERIM) {
            xt = A[PHYS((u32)(plr * TS + plc))];
          } else {
            const uint4 lo = *(const uint4 *)&A[tid * PREP], hi = *(const uint4 *)&A[tid * PREP + 4];
            xt = lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
          }
        }
      }
    }
    if (c != D8_MV) {  // entry?  (a neighbour outside the tile drains into this cell; may be a halo sink)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
        if (((unsigned)nr >= TS || (unsigned)nc >= TS) && CODE(nr, nc) == (1u << ((k + 4) & 7)))
          inmask |= 1u << k;
      }
    }
  }
  const bool entry = inmask != 0u;
  __syncthreads();
  // where does the in-tile path of a cell end?  -> exit slot, halo sink (row block), or nothing
  auto path_end = [&](u32 l) -> u32 {
    const u32 root = PHYS((P[PHYS(l)] & 0x1FFEu) >> 1);
    const int rr = root >> 6, rc = root & 63;
    const u32 cr = CODE(rr, rc);
    if (cr == D8_HALO) return ENC_SINK | (((u32)r0 + (u32)rr > a.row_last) ? ENC_SIDE1 : 0u) | ((u32)c0 + (u32)rc);
    if (d8_is_dir(cr)) {
      const int k = d8_slot(cr);
      const int nr = rr + d8_dr(k), nc = rc + d8_dc(k);
      if ((unsigned)nr >= TS || (unsigned)nc >= TS) return sbase + (u32)pslot(rr, rc);
    }
    return NONE32;
  };
  if (tid < PSL) {
    u32 link = NONE32, esink = NONE32;
    if (entry) {
      const u32 e = path_end((u32)(plr * TS + plc));
      if (e != NONE32 && (e & ENC_SINK))
        esink = e;
      else if (e != NONE32)
        link = e;  // slot of the exit the entry's path reaches
    }
    a.xT[sbase + tid] = xt;
    a.xrec[sbase + tid] = xr_pack(xt12 & 0xFFu, xt12 >> 8, link != NONE32 ? link - sbase : XR_NONE, inmask);
    const u64 xm = __ballot(xt12 != XR_NONE);  // (tid < PSL = all 256 threads: a wave = 64 consecutive slots)
    if ((tid & 63u) == 0u) a.xmask[(sbase + tid) >> 6] = xm;
    if (tr == 0) a.esink[(size_t)tc * PSL + tid] = esink;
    if (tr == a.ntr - 1 && a.ntr > 1) a.esink[((size_t)a.ntc + tc) * PSL + tid] = esink;
  }
  // row-block bookkeeping: flow collected by the halo sinks, first hop of the boundary rows — only the
  // tiles that hold a halo row or a boundary row of the block have any
  if (!INT && !PERIM && ((a.row_first > 0 && (u32)r0 <= a.row_first) || (a.row_last + 1 < a.nrow && (u32)r0 + TS > a.row_last))) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const u32 l = tid + 256u * j;
      const u32 gr = (u32)r0 + (l >> 6), gc = (u32)c0 + (l & 63);
      if (gr >= a.nrow || gc >= a.ncol) continue;
      const u32 c = CODE((int)(l >> 6), (int)(l & 63));
      if (c == D8_HALO) a.haloA[(gr > a.row_last ? a.ncol : 0u) + gc] = A[PHYS(l)];
      if (c != D8_MV && c != D8_HALO) {
        if (gr == a.row_first) a.brow_first[gc] = path_end(l);
        if (gr == a.row_last) a.brow_first[a.ncol + gc] = path_end(l);
      }
    }
  }
  TSTAMP(3)
}

// the general form runs on the FRAME of tiles around the interior rectangle (1-D grid, frame_tile()): tiles that touch
// the raster edge, a halo row or a boundary row of a row block.  Interior tiles: tile_fast.h.
template <bool FINAL, bool RAW = false, bool PERIM = false>
__global__ void __launch_bounds__(256) k_tile(TileArgs a) {
  // running subtree count of the cell (+64 sink words) — or, PERIM, PREP count words per perimeter slot
  __shared__ __attribute__((aligned(16))) u32 A[PERIM ? 256 * PREP : TCELLS + 64];
  __shared__ __attribute__((aligned(16))) uint16_t P[TCELLS];  // 2 x (2^k-th ancestor) | PDONE once saturated
  // codes with a 1-cell halo.  The final pass needs neither the halo ring nor lookups of other
  // cells' codes: it keeps its own quads' codes in registers (cq) and stages nothing.
  __shared__ __attribute__((aligned(16))) u8 code[FINAL ? 16 : HW * CP];
  __shared__ u64 s_cnt[4];
  u32 tr, tc;
  frame_tile(blockIdx.x, a.ntr, a.ntc, a.tr_lo, a.tr_hi, a.tc_lo, a.tc_hi, &tr, &tc);
  tile_body<FINAL, RAW, false, PERIM>(a, A, P, code, s_cnt, tr, tc);
}

#ifndef FY_NT
#define FY_NT 256  // threads of the final pass of an interior tile (tile_fast.h; 512 = 8 waves per tile was measured: 20.5 vs 17.6 ms — tiles in flight per CU count, not waves)
#endif
#include "tile_fast.h"
#include "tile_patch.h"

// per-tile counts of a raw pass -> the counters k_normalise would have left in ctrl
__global__ void __launch_bounds__(1024) k_tile_counts(const u64 *__restrict__ tcnt, u32 ntiles, u64 *ctrl) {
  __shared__ u64 s[3][16];
  u64 v = 0, p = 0, b = 0;
  for (u32 i = blockIdx.x * 1024u + threadIdx.x; i < ntiles; i += gridDim.x * 1024u) {
    const u64 x = tcnt[i];
    v += x & 0xFFFFu;
    p += (x >> 16) & 0xFFFFu;
    b += x >> 32;
  }
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_down(v, o);
    p += __shfl_down(p, o);
    b += __shfl_down(b, o);
  }
  if ((threadIdx.x & 63u) == 0) {
    s[0][threadIdx.x >> 6] = v;
    s[1][threadIdx.x >> 6] = p;
    s[2][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    v = p = b = 0;
    for (int k = 0; k < 16; ++k) v += s[0][k], p += s[1][k], b += s[2][k];
    // (ctrl was cleared when the handle was set up; the host sums the 16 copies of each counter)
    const u32 copy = blockIdx.x & 15u;
    if (v) atomicAdd((unsigned long long *)&ctrl[16 + copy], (unsigned long long)v);
    if (p) atomicAdd((unsigned long long *)&ctrl[32 + copy], (unsigned long long)p);
    if (b) atomicAdd((unsigned long long *)&ctrl[2], (unsigned long long)b);  // C_BAD
  }
}

// ---------------------------------------------------------------------------------------------
// coarse graph: exit e -> exit reached from the cell it drains into.  Pointer doubling in global
// memory (a launch is the round barrier; k_coarse_round further down).
// ---------------------------------------------------------------------------------------------
// raise the "a pointer is still unsaturated" flag: one store per wave at most, and none once
// the flag is visible (millions of same-address stores would serialise in L2)
__device__ __forceinline__ void flag_active(u64 *ctrl) {
  const u64 m = __ballot(1);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
    if (__hip_atomic_load(&ctrl[T_XACTIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
      __hip_atomic_store(&ctrl[T_XACTIVE], (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// level 2: one 1024-thread workgroup per supertile (8x8 tiles) keeps the supertile's 16384 slot
// records in LDS and runs the same pointer doubling over the exits, restricted to the hops that
// stay inside the supertile.  Exits that drain into another supertile ("super-exits") are the
// only nodes left for the global (level-3) solve: ~8x fewer nodes, ~8x shorter paths, and the
// global atomics of the previous single-level solve become LDS atomics.
//   FINAL == false: R2 = last exit of the path inside the supertile,
//                   dense ids + start values for the super-exits
//   FINAL == true : exits start with their tile-local count + the flow entering the supertile
//                   at them (xin, from level 3); every exit delivers its total to its tile entry
// ---------------------------------------------------------------------------------------------
// "does any lane of the workgroup still move a pointer": wave leaders publish their ballot in one of two alternating
// flag rows, everybody reads the row after ONE barrier (__syncthreads_or costs three)
template <int NWAVES>
__device__ __forceinline__ bool wg_vote(u32 (*s_flag)[NWAVES], int round, u32 tid, bool live) {
  const bool any = __ballot(live) != 0ull;
  if ((tid & 63u) == 0u) s_flag[round & 1][tid >> 6] = any ? 1u : 0u;
  __syncthreads();
  u32 acc = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; w += 4) {
    const uint4 f = *(const uint4 *)&s_flag[round & 1][w];
    acc |= f.x | f.y | f.z | f.w;
  }
  return acc != 0u;
}

// The exit lists.  The supertile solve wants its exits — a third of the slots on real rasters — as a dense list: only
// exits get an LDS word (72 KB instead of 96 KB: two workgroups per CU), the rounds run without idle lanes, and the
// solve's own memory accesses become one coalesced, unpredicated, INDEPENDENT batch (round 3 measured the
// positional kernel: of 79 us per supertile 17 went into streaming 64 KB of xtgt to find the exits, 25 into the
// dependent loads xtgt -> elink[xtgt] of a third of the lanes, 19 into its scattered outputs).  The lists are built
// once per pass by one workgroup per TILE (fully parallel, high occupancy): the list index of an exit is the number
// of exits before it in the supertile — the local tile pass left one ballot word per 64 slots (xmask), a prefix over
// the supertile's 256 words gives the index; the exit's next hop (elink[xtgt]) is resolved here as well, as the list
// index of that exit, so that both passes of every solve of the pass (row blocks solve twice) just read it.
#define XL_SX 0x8000u  // xl_slot: the exit drains into another supertile (slots use 14 bits)
__global__ void __launch_bounds__(256) k_exit_lists(SuperArgs s) {
  __shared__ u64 maskw[SSL / 64];
  __shared__ u32 cbase[SSL / 64];
  __shared__ u32 wsum[4];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 st = (blockIdx.x >> 4) + s.st0, part = blockIdx.x & 15u;  // 16 workgroups per supertile, 1024 slots (4 tiles) each
  const u32 base = st << SSHIFT;
  const u32 i0 = (part << 10) + 4u * tid;  // own slots i0 .. i0 + 3
  // (unconditional: slots of tiles beyond the raster edge are allocated, never written and have no mask bit)
  const uint4 rc = *reinterpret_cast<const uint4 *>(s.xrec + base + i0);
  {
    const u64 m = s.xmask[(base >> 6) + tid];  // the 256 ballot words of the supertile
    maskw[tid] = m;
    const u32 c = (u32)__popcll(m);
    u32 incl = c;
    for (int o = 1; o < 64; o <<= 1) {
      const u32 y = __shfl_up(incl, o);
      if (lane >= (u32)o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    cbase[tid] = incl - c;
  }
  __syncthreads();
  {
    u32 woff = 0;
    for (u32 w = 0; w < wave; ++w) woff += wsum[w];
    cbase[tid] += woff;
    if (part == 0) s.xcb[(base >> 6) + tid] = (uint16_t)cbase[tid];  // slot -> list index for everybody else (xl_index)
  }
  if (part == 0 && tid == 0) {
    const u32 n = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    s.scount[st] = n;
    s.sover[st] = n > s.scap ? 1 : 0;  // more exits than the LDS form of the solve keeps: the full-size form takes it
    if (n > s.scap) s.flagged[atomicAdd(s.nflag, 1u)] = st;
  }
  __syncthreads();
  auto dense = [&](u32 i) -> u32 { return cbase[i >> 6] + (u32)__popcll(maskw[i >> 6] & ((1ull << (i & 63u)) - 1ull)); };
  const u32 own = (u32)(maskw[i0 >> 6] >> (i0 & 63u)) & 15u;  // (4 | i0: the four bits sit in one word)
  // the four slots share their tile: position of the tile
  const u32 tl = i0 >> 8;
  const u32 tr = (st / s.nstc) * SG + (tl >> 3), tc = (st % s.nstc) * SG + (tl & 7u);
  const u32 rec[4] = {rc.x, rc.y, rc.z, rc.w};
  if (!own) return;
  u32 tgt[4], l[4], t12[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // the exit reached from the tile entry it drains into
    tgt[j] = NONE32, l[j] = XR_NONE;
    t12[j] = (rec[j] & 0xFFu) | ((rec[j] >> 16) & 0xF00u);
    if ((own >> j) & 1u) {
      tgt[j] = xr_target12(tr, tc, t12[j], s.nstc);
      if ((tgt[j] >> SSHIFT) == st) l[j] = (s.xrec[tgt[j]] >> 8) & 0xFFu;
    }
  }
  u32 d = dense(i0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!((own >> j) & 1u)) continue;
    const bool sx = (tgt[j] >> SSHIFT) != st;
    // (a super-exit is a root of its supertile: its entry carries the 12 bits that name the exit's target instead)
    const u32 nx = sx ? (SDONE | t12[j]) : (l[j] != XR_NONE ? dense((tgt[j] & (SSL - 1) & ~255u) | l[j]) : (d | SDONE));
    s.xl_slot[base + d] = (uint16_t)((i0 + j) | (sx ? XL_SX : 0u));
    s.xl_next[base + d] = (uint16_t)nx;
    ++d;
  }
}

// Cells on a supertile's boundary that receive flow from outside the supertile: the final solve pulls the totals of
// those super-exits itself, so every boundary cell gets one record (sb) naming its sources and the list index of the
// exit its in-tile path reaches.  One thread per boundary cell, after k_exit_lists (xcb).
__global__ void __launch_bounds__(256) k_boundary_records(SuperArgs s) {
  const u32 st = (blockIdx.x >> 3) + s.st0, t = ((blockIdx.x & 7u) << 8) + threadIdx.x;
  u32 v = 0;
  if (t < 4u * SC - 4u) {
    u32 R, C;
    sb_cell(t, &R, &C);
    const u32 trl = R >> 6, tcl = C >> 6;
    const u32 tr = (st / s.nstc) * SG + trl, tc = (st % s.nstc) * SG + tcl;
    if (tr < s.ntr && tc < s.ntc) {  // (tiles beyond the raster edge hold nothing)
      const u32 slot0 = (st << SSHIFT) | (((trl << 3) | tcl) << 8);
      const u32 rec = s.xrec[slot0 + (u32)pslot((int)(R & 63u), (int)(C & 63u))];
      const u32 out = (R == 0u ? 0xE0u : 0u) | (R == SC - 1u ? 0x0Eu : 0u) | (C == 0u ? 0x38u : 0u) | (C == SC - 1u ? 0x83u : 0u);
      const u32 m = (rec >> 16) & out, xe = (rec >> 8) & 0xFFu;
      if (m && xe != XR_NONE) v = SB_VALID | (m << 16) | xl_index(s.xmask, s.xcb, slot0 | xe);
    }
  }
  s.sb[(size_t)st * SBN + t] = v;
}

// The supertile solve over its exit list.  CAP = exits kept in LDS: SCAP (72 KB, two workgroups per CU) for the
// supertiles k_exit_lists did not flag, SSL (every slot an exit: 96 KB, one per CU) for the others — contrived
// rasters only; same code.
// SCAP / SNT: exits kept in LDS and threads of the regular form.  8192 exits (48 KB) and 512 threads run THREE supertiles per
// CU (24 waves) where 12288 exits and 1024 threads ran two (32 waves): what these barrier-synchronised chains of LDS
// round trips respond to is workgroups in flight (§5).  Natural rasters hold 5500-7200 exits per supertile (33-44 % of
// the slots; synthetic regimes: max 7144); fuller supertiles take the flagged form.
#ifndef SCAP
#define SCAP 8192u
#endif
#ifndef SNT
#define SNT 512u
#endif
template <bool FINAL, u32 CAP, u32 NT>
__device__ __forceinline__ void super_solve(const SuperArgs &s, const u32 st) {
  constexpr int NW = NT / 64;
  __shared__ u32 T[CAP];
  __shared__ uint16_t P[CAP];
  __shared__ u32 wtot[NW];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][NW];
  __shared__ u32 s_base;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 base = st << SSHIFT;
  constexpr int DPT = CAP / NT;  // exits per thread
  if (FINAL && s.edge_nstr) {
    // row blocks, first solve: only the totals pulled by the first and last TILE row are needed yet; they
    // belong to exits in tile rows 0..1 and ntr-2..ntr-1 -> supertile row 0 and the rows of those two
    const u32 row = st / s.nstc;
    if (row != 0 && row != (s.ntr - 1) / SG && row != (s.ntr >= 2 ? (s.ntr - 2) / SG : 0u)) return;
  }
  const u32 n = s.scount[st];
  constexpr int NSB = SBN / NT;
  u32 sbr[NSB];  // FINAL: the boundary records of the supertile (asked for first: their totals are a dependent load)
#pragma unroll
  for (int hh = 0; hh < NSB; ++hh) sbr[hh] = FINAL ? s.sb[(size_t)st * SBN + NT * hh + tid] : 0u;
  // ---- the exits: list entry -> slot, start value, next hop; dense loads but for the start value ----
  u32 sxbits = 0;  // bit k: own exit k (tid + 1024 k) drains into another supertile
#pragma unroll 4
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    if (e >= n) continue;
    const u32 w = s.xl_slot[base + e];
    u32 nx = s.xl_next[base + e];
    const u32 t = (s.ablate & 4) ? 1u : s.xT[base + (w & (SSL - 1))];
    if (w & XL_SX) {  // a super-exit is a root; its list entry names its target instead of a next hop
      if (!FINAL) sxbits |= 1u << k;
      nx = e | SDONE;
    }
    T[e] = t;
    P[e] = (uint16_t)nx;
  }
  __syncthreads();
  if (FINAL) {
    // flow entering the supertile: the totals of the super-exits that drain into its boundary cells (left on their
    // slots by the level-3 solve), added to the exit the cell's in-tile path reaches
    const u32 str = st / s.nstc, stc = st % s.nstc;
#pragma unroll
    for (int h = 0; h < NSB; ++h) {
      const u32 r = sbr[h];
      if (!(r & SB_VALID)) continue;
      u32 R, C;
      sb_cell(tid + NT * h, &R, &C);
      u32 m = (r >> 16) & 0xFFu, v = 0;
      while (m) {
        const int k = __ffs((int)m) - 1;
        m &= m - 1u;
        v += s.xtot[nbr_slot(str * SG + (R >> 6), stc * SG + (C >> 6), (int)(R & 63u), (int)(C & 63u), k, s.nstc)];
      }
      atomicAdd(&T[r & (SSL - 1)], v);
    }
    __syncthreads();
  }
  // ---- doubling over the exits ----
  {
    u32 y[DPT];
    u32 live = 0;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const u32 e = tid + NT * k;
      u32 p = P[e];  // (unconditional, then a select: the branchy form made the compiler spill whole copies of y[])
      p = e < n ? p : SDONE;
      y[k] = p & (SDONE - 1u);
      live |= (p & SDONE) ? 0u : 1u << k;
    }
    if (!FINAL) {
      // The first solve only needs ROOTS (the last exit of every path inside the supertile) and the total per root
      // (the start value of a super-exit at level 3) — like the local tile pass it jumps pointers without values:
      // gather-only, two jumps per round, one barrier per round (a read that sees a pointer its owner has already
      // advanced just jumps further), and every exit adds its start value to its root afterwards.
      u32 nonroot = live;
#pragma nounroll
      for (int round = 0; round < ((s.ablate & 1) ? 0 : MAXROUNDS_SUPER); ++round) {
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
          if (live & (1u << k)) {
            u32 q = P[y[k]];
            q = P[q & (SDONE - 1u)];
            P[tid + NT * k] = (uint16_t)q;
            y[k] = q & (SDONE - 1u);
            if (q & SDONE) live &= ~(1u << k);
          }
        }
        if (!wg_vote<NW>(s_flag, round, tid, live != 0u)) break;
      }
#pragma unroll
      for (int k = 0; k < DPT; ++k)  // (nobody adds to the word of an exit that is no root: reading it here is safe)
        if (nonroot & ~live & (1u << k)) atomicAdd(&T[y[k]], T[tid + NT * k]);
      __syncthreads();
    } else {
#pragma nounroll
      for (int round = 0; round < ((s.ablate & 1) ? 0 : MAXROUNDS_SUPER); ++round) {
        u32 av[DPT], q[DPT];
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
          if (live & (1u << k)) {
            av[k] = T[tid + NT * k];
            q[k] = P[y[k]];
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
          if (live & (1u << k)) {
            atomicAdd(&T[y[k]], av[k]);
            P[tid + NT * k] = (uint16_t)q[k];
            y[k] = q[k] & (SDONE - 1u);
            if (q[k] & SDONE) live &= ~(1u << k);
          }
        }
        if (!wg_vote<NW>(s_flag, round, tid, live != 0u)) break;
      }
    }
    if (live && !(s.ablate & 1)) atomicAdd((unsigned long long *)&s.ctrl[T_SLIVE], 1ull);  // a cycle inside the supertile
  }
  if (s.ablate & 2) return;
  if (FINAL) {  // the total of every exit, where the tile entries it drains into will pull it
#pragma unroll 4
    for (int k = 0; k < DPT; ++k) {
      const u32 e = tid + NT * k;
      if (e < n) s.xtot[base + (s.xl_slot[base + e] & (SSL - 1))] = T[e];
    }
    return;
  }
  // ---- records of the pass: super-exits (exits that drain into another supertile) get dense ids (order is
  //      irrelevant) and start values for level 3; per exit (list order) the last exit of its path inside the supertile
  u32 wcnt = 0;  // (wave-uniform) super-exits of this wave
#pragma unroll
  for (int k = 0; k < DPT; ++k) wcnt += (u32)__popcll(__ballot((sxbits >> k) & 1u));
  if (lane == 0) wtot[wave] = wcnt;
  __syncthreads();
  const u32 str = st / s.nstc, stc = st % s.nstc;
  const u32 ht = (str / HG) * s.nhtc + stc / HG;
  if (tid == 0) {
    u32 tot = 0;
    for (int w = 0; w < NW; ++w) {
      const u32 c = wtot[w];
      wtot[w] = tot;
      tot += c;
    }
    if (!tot) {
      s_base = 0;
    } else if (s.hmode) {  // ids of one hypertile are consecutive: its level-3 solve runs in LDS
      const u32 b = atomicAdd(&s.hcnt[ht], tot);
      if (b + tot > s.hcap) s.ctrl[T_OVERFLOW] = 1;  // host falls back to the flat id range
      s_base = ht * HCAP + (b + tot > s.hcap ? 0u : b);
    } else {
      s_base = (u32)atomicAdd((unsigned long long *)&s.ctrl[T_NSUPER], (unsigned long long)tot);
    }
  }
  __syncthreads();
  u32 run = s_base + wtot[wave];
#pragma unroll
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    const bool sx = (sxbits >> k) & 1u;
    const u64 m = __ballot(sx);
    u32 id = NONE32;
    if (sx) {
      id = run + (u32)__popcll(m & ((1ull << lane) - 1ull));
      const u32 sl = (u32)s.xl_slot[base + e] & (SSL - 1);
      s.sx_slot[id] = base + sl;
      // (sx_n1 doubles as the target slot of the super-exit until k_link3 turns it into a list position)
      s.sx_n1[id] = xr_target12(str * SG + (sl >> 11), stc * SG + ((sl >> 8) & 7u), (u32)s.xl_next[base + e] & 0xFFFu, s.nstc);
      s.T3[id] = T[e];
    }
    run += (u32)__popcll(m);
    if (e < n) {
      // (on a cycle the pointer is not saturated and names an ancestor instead of a root; the pass is discarded then)
      s.R2L[base + e] = (uint16_t)(P[e] & (SDONE - 1u));
      s.sxidL[base + e] = id;
    }
  }
}

template <bool FINAL>
__global__ void __launch_bounds__(SNT, SNT == 512u ? 6 : 8) k_super(SuperArgs s) {
  if (s.sover[blockIdx.x + s.st0]) return;  // (more exits than SCAP: k_super_flagged takes it)
  super_solve<FINAL, SCAP, SNT>(s, blockIdx.x + s.st0);
}
// the supertiles k_exit_lists flagged (contrived rasters only: normally none, and a grid of this 96 KB kernel over all
// supertiles costs 60-90 us just to find that out): a small fixed grid walks their list
#define SFLAG_GRID 64u
template <bool FINAL>
__global__ void __launch_bounds__(1024, 4) k_super_flagged(SuperArgs s) {
  const u32 nf = *s.nflag;
  for (u32 f = blockIdx.x; f < nf; f += gridDim.x) {
    super_solve<FINAL, SSL, 1024u>(s, s.flagged[f]);
    __syncthreads();  // (the LDS image is reused)
  }
}

// level 3 links: super-exit -> next super-exit on its path (through the supertile it enters)
__device__ __forceinline__ bool sx_active(const SuperArgs &s, u32 k) {
  return !s.hmode || (k % HCAP) < s.hcnt[k / HCAP];
}
__global__ void __launch_bounds__(256) k_link3(SuperArgs s, u32 nsuper, u32 *__restrict__ J3,
                                               u32 *__restrict__ clear = nullptr) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nsuper) return;
  if (clear) clear[k] = 0;  // (hyper mode: the level-3 inflow array, filled by k_push4 later on)
  if (s.hmode && s.ctrl[T_OVERFLOW]) return;  // ids are invalid: the host redoes the pass flat
  if (!sx_active(s, k)) {
    J3[k] = k | XDONE;
    return;
  }
  const u32 tgt = s.sx_n1[k];                    // target slot (left there by the supertile solve)
  const u32 xe = (s.xrec[tgt] >> 8) & 0xFFu;     // exit the target entry's in-tile path reaches
  u32 j = k | XDONE, pos = NONE32;
  if (xe != XR_NONE) {
    const u32 n1 = (tgt & ~255u) | xe, b1 = n1 & ~(u32)(SSL - 1);
    pos = b1 + xl_index(s.xmask, s.xcb, n1);
    const u32 id = s.sxidL[b1 + s.R2L[pos]];    // the supertile's last exit on that path: a super-exit?
    if (id != NONE32) j = id;
  }
  s.sx_n1[k] = pos;
  J3[k] = j;
}
// flat level 3: the total of every super-exit goes to its slot, where the supertile it drains into pulls it
// (hyper mode: k_hyper<true> does this itself)
__global__ void __launch_bounds__(256) k_sx_totals(SuperArgs s, u32 nsuper, const u32 *__restrict__ T3final) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nsuper) s.xtot[s.sx_slot[k]] = T3final[k];
}

// The same two for a whole raster of few hypertiles (TiledRun::level3_flat_nosync): the number of super-exits stays on the
// device (ctrl[T_NSUPER]), a bounded grid strides over it, and the first round's buffer is cleared on the way — no host look
__global__ void __launch_bounds__(256) k_link3_flat(SuperArgs s, u32 cap, u32 *__restrict__ J3, u32 *__restrict__ Tnext) {
  const u32 n = min(cap, (u32)s.ctrl[T_NSUPER]);
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    Tnext[k] = 0;
    const u32 tgt = s.sx_n1[k];
    const u32 xe = (s.xrec[tgt] >> 8) & 0xFFu;
    u32 j = k | XDONE, pos = NONE32;
    if (xe != XR_NONE) {
      const u32 n1 = (tgt & ~255u) | xe, b1 = n1 & ~(u32)(SSL - 1);
      pos = b1 + xl_index(s.xmask, s.xcb, n1);
      const u32 id = s.sxidL[b1 + s.R2L[pos]];
      if (id != NONE32) j = id;
    }
    s.sx_n1[k] = pos;
    J3[k] = j;
  }
}
__global__ void __launch_bounds__(256) k_sx_totals_flat(SuperArgs s, u32 cap, const u32 *__restrict__ T3final) {
  const u32 n = min(cap, (u32)s.ctrl[T_NSUPER]);
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) s.xtot[s.sx_slot[k]] = T3final[k];
}

// ---------------------------------------------------------------------------------------------
// level 3: one 1024-thread workgroup per hypertile (4x4 supertiles = 2048 x 2048 cells) keeps the
// <= HCAP super-exits of the hypertile in LDS; same doubling restricted to the hops that stay
// inside the hypertile.  Only the super-exits that leave their hypertile remain for the global
// (level-4) rounds.  FINAL: start values + flow entering the hypertile -> totals of all nodes.
// ---------------------------------------------------------------------------------------------
template <bool FINAL>
__global__ void __launch_bounds__(1024) k_hyper(HyperArgs s) {
  __shared__ u32 T[HCAP];
  __shared__ uint16_t P[HCAP];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][16];
  __shared__ u32 s_cnt, s_base;
  const u32 tid = threadIdx.x;
  const u32 ht = blockIdx.x;
  const u32 base = ht * HCAP;
  if (s.ctrl[T_OVERFLOW]) return;  // (uniform) ids are invalid: the host redoes the pass flat
  if (FINAL && s.edge_nstr) {
    // hypertile rows holding the supertile rows that feed those edge rows: 0..1 and the last three
    const u32 hr = ht / s.nhtc, n_ = s.edge_nstr;
    if (hr != 0 && hr < (n_ >= 3 ? (n_ - 3) / HG : 0u)) return;
  }
  const u32 n = s.hcnt[ht];
  constexpr int SPT = HCAP / 1024;
  if (tid == 0) s_cnt = 0;
  u32 y[SPT], ext = 0, live = 0;  // ext bit j: node drains into another hypertile
  // (this workgroup has its CU to itself — 144 KB of LDS — so nobody fills the gaps of a load that is waited for under
  //  its `if (i < n)`: 24 serialised round trips were most of the kernel.  Loads are unconditional — the id range of a
  //  hypertile is allocated in full — in batches of 8, slices of 1024 ids past n are skipped as a whole)
#pragma unroll
  for (int j0 = 0; j0 < SPT; j0 += 8) {
    u32 j3s[8], ts[8];
    if (1024u * j0 < n) {  // (uniform)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const u32 i = tid + 1024u * (j0 + q);
        j3s[q] = s.J3[base + i];
        ts[q] = s.T3[base + i];
        if (FINAL) ts[q] += s.xin3[base + i];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = j0 + q;
      const u32 i = tid + 1024u * j;
      u32 p = i | HDONE, t = 0;
      if (i < n) {
        const u32 j3 = j3s[q];
        t = ts[q];
        if (!(j3 & XDONE)) {
          if (j3 / HCAP == ht)
            p = j3 % HCAP;
          else
            ext |= 1u << j;
        }
      }
      T[i] = t;
      P[i] = (uint16_t)p;
      y[j] = p & 0x7FFFu;
      if (!(p & HDONE)) live |= 1u << j;
    }
  }
  __syncthreads();
  for (int round = 0; round < 16; ++round) {
    u32 av[SPT], q[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      if (live & (1u << j)) {
        av[j] = T[tid + 1024u * j];
        q[j] = P[y[j]];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      if (live & (1u << j)) {
        atomicAdd(&T[y[j]], av[j]);
        P[tid + 1024u * j] = (uint16_t)q[j];
        y[j] = q[j] & 0x7FFFu;
        if (q[j] & HDONE) live &= ~(1u << j);
      }
    }
    if (!wg_vote<16>(s_flag, round, tid, live != 0u)) break;
  }
  if (live) atomicAdd((unsigned long long *)&s.ctrl[T_SLIVE], 1ull);  // a cycle inside the hypertile
  if (FINAL) {
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const u32 i = tid + 1024u * j;
      if (i < n) {
        s.T3out[base + i] = T[i];
        s.xtot[s.sx_slot[base + i]] = T[i];  // (the supertile the super-exit drains into pulls it from there)
      }
    }
    return;
  }
  u32 rank[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) rank[j] = (ext & (1u << j)) ? atomicAdd(&s_cnt, 1u) : NONE32;
  __syncthreads();
  if (tid == 0) s_base = s_cnt ? (u32)atomicAdd((unsigned long long *)&s.ctrl[T_NHYPER], (unsigned long long)s_cnt) : 0u;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const u32 i = tid + 1024u * j;
    if (i >= n) continue;
    s.T3out[base + i] = T[i];
    s.R3[base + i] = base + (P[i] & 0x7FFFu);
    u32 id = NONE32;
    if (rank[j] != NONE32) {
      id = s_base + rank[j];
      s.hx_node[id] = base + i;
      s.T4[id] = T[i];
    }
    s.hx_id[base + i] = id;
  }
}
// level-4 links: hyper-exit -> next hyper-exit on its path (through the hypertile it enters)
__global__ void __launch_bounds__(256) k_link4(HyperArgs s, u32 cap, u32 *__restrict__ J4, u32 *__restrict__ Tnext) {
  if (s.ctrl[T_OVERFLOW]) return;
  const u32 n = min(cap, (u32)s.ctrl[T_NHYPER]);  // (device-side count, bounded grid: see k_coarse_round)
  for (u32 m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
    Tnext[m] = 0;  // the first round accumulates into a cleared buffer (pfd_doubling_rounds, prepared)
    const u32 n1 = s.J3[s.hx_node[m]];  // first node inside the entered hypertile
    const u32 id = s.hx_id[s.R3[n1]];
    const u32 j = (id != NONE32) ? id : (m | XDONE);
    J4[m] = j;
  }
}
__global__ void __launch_bounds__(256) k_push4(HyperArgs s, u32 cap, const u32 *__restrict__ T4final, u32 *xin3) {
  if (s.ctrl[T_OVERFLOW]) return;
  const u32 n = min(cap, (u32)s.ctrl[T_NHYPER]);
  for (u32 m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x)
    atomicAdd(&xin3[s.J3[s.hx_node[m]]], T4final[m]);
}

// round prologue: Tnew = Told (the adds of the round go on top) and reset the activity flag
// (`ncnt`, if given, is the device-side node count: the host launches for the capacity and does
// not have to wait for the count)
// One doubling round, one launch: T_{k+1}[v] = T_k[v] + sum of T_k[e] over the unsaturated e with
// J_k(e) = v.  Three T buffers rotate: every thread carries its own value into the (zeroed) next
// buffer with the same atomics that deliver its contribution, and zeroes its word of the buffer
// after next — no separate copy/clear launch per round.  ctrl[T_XACTIVE] keeps the number of the
// last round that still moved a pointer.
__global__ void __launch_bounds__(256) k_coarse_round(const u32 *__restrict__ Told, u32 *__restrict__ Tnew,
                                                      u32 *__restrict__ Tzero, const u32 *__restrict__ Jold,
                                                      u32 *__restrict__ Jnew, u32 nexits, u64 *ctrl,
                                                      const u64 *__restrict__ ncnt, u32 round) {
  // As the pointers converge, thousands of nodes of a large basin push to the same few ancestors in one round, and
  // same-address global atomics are served one after the other in L2 (~12 ns each: 0.4 ms per round at 90000^2).
  // The workgroup therefore combines its pushes per target in a small LDS hash table first (integer adds: any
  // order) and issues one global atomic per distinct target.
  __shared__ u32 hk[512], hv[512];
  const u32 tid = threadIdx.x;
  if (ncnt) {
    // device-side node count (level 4).  After a hypertile overflow the level-4 graph is only
    // partly built (the pass is about to be redone flat): touch nothing.
    nexits = ctrl[T_OVERFLOW] ? 0u : min(nexits, (u32)*ncnt);
  }
  // (the node count is only known on the device and the capacity is ~10x the count at 90000^2: a bounded grid strides
  //  over the nodes — a grid sized for the capacity spent 40 of a round's 49 us dispatching workgroups without a node)
  bool moving = false;
  for (u32 base = blockIdx.x * blockDim.x; base < nexits; base += gridDim.x * blockDim.x) {
  hk[tid] = NONE32, hk[tid + 256u] = NONE32;
  hv[tid] = 0, hv[tid + 256u] = 0;
  const u32 e = base + tid;
  __syncthreads();
  if (e < nexits) {
    const u32 j = Jold[e];
    const u32 t = Told[e];
    Tzero[e] = 0;
    if (t) atomicAdd(&Tnew[e], t);  // (its own word: no contention)
    if (j & XDONE) {
      Jnew[e] = j;
    } else {
      const u32 q = Jold[j];
      Jnew[e] = q;
      moving |= !(q & XDONE);
      if (t) {
        u32 slot = (j * 2654435761u) >> 23;  // 9 bits
        for (;;) {
          const u32 prev = atomicCAS(&hk[slot], NONE32, j);
          if (prev == NONE32 || prev == j) {
            atomicAdd(&hv[slot], t);
            break;
          }
          slot = (slot + 1u) & 511u;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const u32 key = hk[tid + 256u * k], val = hv[tid + 256u * k];
    if (key != NONE32 && val) atomicAdd(&Tnew[key], val);
  }
  __syncthreads();  // (the table is reused by the next stride)
  }
  if (round && moving) {  // at most one store per wave, none once this round's mark is visible
    const u64 m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
      if (__hip_atomic_load(&ctrl[T_XACTIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)round)
        __hip_atomic_store(&ctrl[T_XACTIVE], (u64)round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// "did the fixed number of rounds saturate every pointer?" — asked once, after the rounds (rounds that
// mark themselves make ~1000 waves load and store ONE address: measured 27 us in a 34 us kernel)
__global__ void __launch_bounds__(256) k_check_saturated(const u32 *__restrict__ J, u32 nexits, u64 *ctrl,
                                                         const u64 *__restrict__ ncnt, u32 mark) {
  if (ncnt) nexits = ctrl[T_OVERFLOW] ? 0u : min(nexits, (u32)*ncnt);
  for (u32 e = blockIdx.x * blockDim.x + threadIdx.x; e < nexits; e += gridDim.x * blockDim.x)
    if (!(J[e] & XDONE)) ctrl[T_XACTIVE] = (u64)mark;  // (rare: only when the budget was short)
}

// generic pointer doubling driver on T[3] / J[2] rotating buffers (T[0], J[0] hold the input; on
// return T[0] / J[0] hold the result).  Rounds are idempotent once every pointer is saturated, so
// they are issued in batches.  check == true: the "last active round" mark is read between batches
// (one host round trip per batch) until a batch ends saturated.  check == false: exactly
// first_batch rounds are issued and *done is left to the caller, who compares ctrl[T_XACTIVE]
// with the returned round count at its next synchronisation.
int pfd_doubling_rounds(pfd_raster *h, u32 *T[3], u32 *J[2], u32 n, int first_batch, bool check, bool *done,
                        int *rounds_issued, i64 *launches, const u64 *ncnt, bool prepared) {
  const u32 grid = std::min(cdiv_u32(n, 256), 4096u);  // (the kernels stride)
  *done = false;
  if (!prepared) {  // (prepared: the caller's kernels have cleared T[1] and the round mark already)
    HIPCHK(hipMemsetAsync(T[1], 0, (size_t)n * sizeof(u32), h->stream));
    HIPCHK(hipMemsetAsync(h->ctrl + T_XACTIVE, 0, sizeof(u64), h->stream));
  }
  int batch = first_batch, rounds = 0;
  while (rounds < 48 && !*done) {
    for (int b = 0; b < batch; ++b) {
      ++rounds;
      k_coarse_round<<<grid, 256, 0, h->stream>>>(T[0], T[1], T[2], J[0], J[1], n, h->ctrl, ncnt,
                                                  check ? (u32)rounds : 0u);
      ++*launches;
      u32 *t0 = T[0];
      T[0] = T[1], T[1] = T[2], T[2] = t0;
      std::swap(J[0], J[1]);
    }
    KCHK();
    if (!check) {  // one question after the rounds: mark = rounds iff some pointer is still unsaturated
      k_check_saturated<<<grid, 256, 0, h->stream>>>(J[0], n, h->ctrl, ncnt, (u32)rounds);
      ++*launches;
      break;
    }
    u64 last = 0;
    HIPCHK(hipMemcpyAsync(&last, h->ctrl + T_XACTIVE, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *done = last < (u64)rounds;  // the last round issued moved no pointer
    batch = 2;
  }
  *rounds_issued = rounds;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// row-block (multi-GPU) helpers
// ---------------------------------------------------------------------------------------------
// flow collected by a halo sink = what reached it inside its tile (haloA) + the inflow of the
// tile entries whose in-tile path ends on it
__global__ void __launch_bounds__(256) k_halo_collect(const u32 *__restrict__ esink, const u32 *__restrict__ xrec,
                                                      const u32 *__restrict__ xtot, u32 ntc, u32 ntr, u32 nstc, u32 ncol,
                                                      u32 *__restrict__ haloL, u32 n) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const u32 e = esink[t];
  if (e == NONE32) return;
  // region 0 = tile row 0, region 1 = last tile row
  const u32 reg = t / (ntc * PSL), tc = (t % (ntc * PSL)) / PSL, p = t % PSL;
  const u32 tr = reg ? ntr - 1 : 0;
  int lr, lc;
  pslot_inv((int)p, &lr, &lc);
  const u32 v = slot_inflow(xtot, xrec[sslot_base(tr, tc, nstc) + p], tr, tc, lr, lc, nstc);
  if (v) atomicAdd(&haloL[((e & ENC_SIDE1) ? ncol : 0u) + (e & ENC_COL)], v);
}
// last exit on the path of the exit on slot f, by O(1) lookups in the hierarchy built by the solve:
// supertile root -> (level 3) last super-exit [-> (level 4) last hyper-exit -> hypertile root] ->
// the supertile the path finally enters -> its root
struct LastExitArgs {
  const u64 *xmask;
  const uint16_t *xcb, *xl_slot, *R2L;
  const u32 *sxidL, *sx_slot, *sx_n1, *xrec;
  const u32 *J3;       // hyper mode: level-3 links; flat mode: saturated level-3 pointers
  const u32 *R3, *hx_id, *hx_node, *J4fin;  // hyper mode only
  int hmode;
};
__device__ __forceinline__ u32 last_exit(const LastExitArgs &q, u32 f) {
  const u32 b = f & ~(u32)(SSL - 1);
  const u32 r2 = q.R2L[b + xl_index(q.xmask, q.xcb, f)];  // list index of the supertile's last exit on the path
  const u32 k = q.sxidL[b + r2];
  if (k == NONE32) return b + (q.xl_slot[b + r2] & (SSL - 1));  // the path ends inside this supertile
  u32 k3;
  if (q.hmode) {
    k3 = q.R3[k];  // last super-exit inside the hypertile
    const u32 m = q.hx_id[k3];
    if (m != NONE32) {  // the path leaves the hypertile: last hyper-exit, then the root of the hypertile it enters
      const u32 mlast = q.J4fin[m] & ~XDONE;
      k3 = q.R3[q.J3[q.hx_node[mlast]] & ~XDONE];
    }
  } else {
    k3 = q.J3[k] & ~XDONE;
  }
  const u32 pos = q.sx_n1[k3];  // where the flow through that super-exit enters the next supertile
  if (pos == NONE32) return q.sx_slot[k3];
  const u32 b2 = pos & ~(u32)(SSL - 1);
  return b2 + (q.xl_slot[b2 + q.R2L[pos]] & (SSL - 1));
}
// where does the flow entering at a boundary-row cell leave the block?  (halo sink or nowhere)
__global__ void __launch_bounds__(256) k_brow_sink(const u32 *__restrict__ brow_first, LastExitArgs q,
                                                   const u32 *__restrict__ esink, u32 ntc, u32 ntr, u32 nstc, u32 ncol,
                                                   u32 *__restrict__ brow_sink, const u64 *__restrict__ ctrl) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  if (q.hmode && ctrl[T_OVERFLOW]) return;  // the hierarchy is only partly built: the pass is redone flat
  u32 f = brow_first[t];
  if (f != NONE32 && !(f & ENC_SINK)) {
    const u32 le = last_exit(q, f);
    u32 tr, tc, p;
    sslot_inv(xr_target(le, q.xrec[le], nstc), nstc, &tr, &tc, &p);
    f = NONE32;
    if (tr == 0)
      f = esink[(size_t)tc * PSL + p];
    else if (tr == ntr - 1)
      f = esink[((size_t)ntc + tc) * PSL + p];
  }
  brow_sink[t] = f;
}
// flow entering at the boundary rows becomes the start value of the exit it reaches first
__global__ void __launch_bounds__(256) k_brow_scatter(const u32 *__restrict__ brow_first, const u32 *__restrict__ brow_inflow,
                                                      u32 ncol, u32 *__restrict__ start) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  const u32 v = brow_inflow[t], e = brow_first[t];
  if (v && e != NONE32 && !(e & ENC_SINK)) atomicAdd(&start[e], v);
}

// same, and the super-exit / hyper-exit start values of the second solve (k_brow_delta) in one launch
__global__ void __launch_bounds__(256) k_brow_scatter_delta(const u32 *__restrict__ brow_first, const u32 *__restrict__ brow_inflow,
                                                            u32 ncol, u32 *__restrict__ start, const u64 *__restrict__ xmask,
                                                            const uint16_t *__restrict__ xcb, const uint16_t *__restrict__ R2L,
                                                            const u32 *__restrict__ sxidL, u32 *__restrict__ T3,
                                                            const u32 *__restrict__ R3, const u32 *__restrict__ hx_id,
                                                            u32 *__restrict__ T4start, const u64 *__restrict__ ctrl) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  // (a pass without host round trips runs on after a hypertile overflowed its id range — it is about to be redone flat —
  //  and R3 / hx_id of that hypertile were never written: a lookup through them is an address from uninitialised memory.
  //  Found by the suite under allocation poisoning, test_a_stage_that_falls_short_is_redone_by_every_rank[PFD_TEST_HCAP])
  if (ctrl[T_OVERFLOW]) return;
  const u32 v = brow_inflow[t], e = brow_first[t];
  if (!v || e == NONE32 || (e & ENC_SINK)) return;
  atomicAdd(&start[e], v);
  const u32 b = e & ~(u32)(SSL - 1);
  const u32 id = sxidL[b + R2L[b + xl_index(xmask, xcb, e)]];  // the super-exit the path leaves the supertile through
  if (id == NONE32) return;
  atomicAdd(&T3[id], v);
  const u32 m = hx_id[R3[id]];  // ... and the hyper-exit it leaves the hypertile through
  if (m != NONE32) atomicAdd(&T4start[m], v);
}
// level 4 starts again: T4 <- saved start values (+ what k_brow_scatter_delta added), xin3 <- 0, round mark <- 0
__global__ void __launch_bounds__(256) k_l4_restart(const u32 *__restrict__ T4saved, u32 *__restrict__ T4, u32 n4, u32 *__restrict__ xin3,
                                                    u32 n3, u64 *__restrict__ ctrl) {
  const u32 i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (u32 i = i0; i < n4; i += stride) T4[i] = T4saved[i];
  for (u32 i = i0; i < n3; i += stride) xin3[i] = 0;
  if (i0 == 0) ctrl[T_XACTIVE] = 0;
}
// every clear a pass starts with, in one launch (eight memsets cost eight host calls and eight kernel boundaries):
// control words (all 64, or 8..63 when the handle's counters 0..7 are live), the five boundary arrays of a row block
// (brow_first / brow_sink = NONE, haloA / haloL / brow_inflow = 0) and the exit bitmasks
__global__ void __launch_bounds__(256) k_pass_clear(u64 *__restrict__ ctrl, u32 ctrl_from, u32 *__restrict__ bnd, u32 nb,
                                                    uint4 *__restrict__ xmask16, u32 n16) {
  const u32 i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (i0 < 64u && i0 >= ctrl_from) ctrl[i0] = 0;
  for (u32 i = i0; i < 5u * nb; i += stride) {
    const u32 reg = i / nb;  // brow_first | haloA | haloL | brow_sink | brow_inflow
    bnd[i] = (reg == 0u || reg == 3u) ? NONE32 : 0u;
  }
  for (u32 i = i0; i < n16; i += stride) xmask16[i] = make_uint4(0u, 0u, 0u, 0u);
}
// stage checks of a pass without host round trips (see T_MISS)
__global__ void k_stage_verdict_a(u64 *ctrl, u32 hmode, u32 rounds4) {
  u64 m = 0;
  if (ctrl[T_OVERFLOW]) m |= MISS_OVERFLOW;
  if (hmode && rounds4 && ctrl[T_XACTIVE] >= (u64)rounds4) m |= MISS_ROUNDS4;
  if (m) ctrl[T_MISS] |= m;
}
__global__ void k_block_verdict(const u64 *ctrl, u32 hmode, u32 rounds4, u32 host_ok, int *flag) {
  const bool redo = ctrl[T_OVERFLOW] != 0 || ctrl[T_MISS] != 0 || (hmode && rounds4 && ctrl[T_XACTIVE] >= (u64)rounds4);
  const bool fine = host_ok && ctrl[T_SLIVE] == 0 && ctrl[T_UNSAT] == 0 && ctrl[2] == 0;  // (ctrl[2]: bad D8 codes)
  *flag = redo ? 1 : (fine ? 2 : 0);
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
int TiledRun::init(pfd_raster *hh, i32 *out_dev) {
  h = hh;
  ntr = cdiv_u32((u64)h->nrow, TS);
  ntc = cdiv_u32((u64)h->ncol, TS);
  nstc = cdiv_u32(ntc, SG);
  const u32 nstr = cdiv_u32(ntr, SG);
  nst = nstr * nstc;
  nhtc = cdiv_u32(nstc, HG);
  nht = cdiv_u32(nstr, HG) * nhtc;
  nslots = (size_t)nst * SSL;
  supported = !(nslots >= 0x3FFFFFFFull || ntr > 65535u || (u64)h->ncol >= ENC_SIDE1 ||
                (size_t)nht * HCAP >= 0x7FFFFFFFull);
  if (!supported) return PFD_OK;  // ids are 30 bit: such rasters go through the level engine
  const size_t nb = 2 * (size_t)h->ncol;
  const size_t sxcap = (size_t)nst * 4 * SG * TS;  // super-exits sit on the supertile perimeter
  n3cap = std::max(sxcap, (size_t)nht * HCAP);
  n4cap = (size_t)nht * 4 * HG * SG * TS;           // hyper-exits sit on the hypertile perimeter
  // per slot: xT | xrec | xtot (u32, slot order), sxidL (u32, list order), R2L (u16, list order)
  PFDCHK(slots.alloc(4 * nslots * sizeof(u32) + nslots * sizeof(uint16_t)));
  PFDCHK(sbbuf.alloc((size_t)nst * SBN * sizeof(u32)));
  PFDCHK(l3.alloc(9 * n3cap * sizeof(u32)));
  PFDCHK(l4.alloc(7 * n4cap * sizeof(u32)));  // hx_node | T4 x2 | J4 x2 | T4 | saved level-4 start values (row blocks)
  PFDCHK(hcntbuf.alloc((size_t)nht * sizeof(u32)));
  PFDCHK(esink.alloc(2 * (size_t)ntc * PSL * sizeof(u32)));
  PFDCHK(bnd.alloc(5 * nb * sizeof(u32)));  // brow_first | haloA | haloL | brow_sink | brow_inflow
  u32 *q = slots.as<u32>();
  // (nothing needs clearing: every tile writes its 256 slots, the list-order arrays are written before they are read,
  //  xtot is only read where an entry's source mask names an exit)
  xT = q, xrec = q + nslots, xtot = q + 2 * nslots, sxidL = q + 3 * nslots;
  R2L = (uint16_t *)(q + 4 * nslots);
  u32 *x = l3.as<u32>();
  sx_slot = x;
  Tc = x + n3cap, Tn = x + 2 * n3cap, Jc = x + 3 * n3cap, Jn = x + 4 * n3cap;
  xin3 = x + 5 * n3cap, R3 = x + 6 * n3cap, hx_id = x + 7 * n3cap;
  sx_n1 = x + 8 * n3cap;
  u32 *b = bnd.as<u32>();
  brow_first = b;
  haloA = b + nb;
  haloL = b + 2 * nb;
  brow_sink = b + 3 * nb;
  brow_inflow = b + 4 * nb;
  a = TileArgs{h->ncode, nullptr, h->ncode, nullptr, (u64)h->n, hcntbuf.as<u32>(), nht, nullptr, (u32)h->nrow, (u32)h->ncol, ntr, ntc, (u32)h->halo_top,
               (u32)(h->halo_top + h->own_rows - 1), nstc, xT, xrec, xtot, nullptr, esink.as<u32>(),
               brow_first, haloA, brow_inflow, h->ctrl, out_dev, 0};
  sa = SuperArgs{};
  sa.nst = nst, sa.xT = xT, sa.xrec = xrec, sa.sb = sbbuf.as<u32>(), sa.R2L = R2L, sa.sxidL = sxidL, sa.sx_slot = sx_slot,
  sa.sx_n1 = sx_n1, sa.T3 = Tc, sa.xtot = xtot, sa.ctrl = h->ctrl, sa.nstc = nstc, sa.nhtc = nhtc,
  sa.hcnt = hcntbuf.as<u32>(), sa.ntr = ntr, sa.ntc = ntc, sa.hcap = HCAP, sa.scap = SCAP;
  PFDCHK(soverbuf.alloc((size_t)nst));
  sa.sover = soverbuf.as<u8>();
  PFDCHK(xmaskbuf.alloc(nslots / 64 * sizeof(u64) + 8 + 16));  // (+ the count of flagged supertiles: one clear for both, in 16-byte stores)
  PFDCHK(xlbuf.alloc(2 * nslots * sizeof(uint16_t)));
  PFDCHK(scountbuf.alloc((size_t)nst * sizeof(u32)));
  PFDCHK(xcbbuf.alloc(nslots / 64 * sizeof(uint16_t)));
  sa.xcb = xcbbuf.as<uint16_t>();
  a.xmask = xmaskbuf.as<u64>();
  sa.xmask = xmaskbuf.as<u64>();
  sa.xl_slot = xlbuf.as<uint16_t>();
  sa.xl_next = xlbuf.as<uint16_t>() + nslots;
  sa.scount = scountbuf.as<u32>();
  sa.nflag = (u32 *)(xmaskbuf.as<u64>() + nslots / 64);
  PFDCHK(flaggedbuf.alloc((size_t)nst * sizeof(u32)));
  sa.flagged = flaggedbuf.as<u32>();
  if (const char *e = pfd_knob("PFD_TEST_HCAP")) sa.hcap = (u32)std::min(atoi(e), HCAP);
  if (const char *e = pfd_knob("PFD_TEST_SCAP")) sa.scap = (u32)std::min<u32>((u32)atoi(e), SCAP);
  if (const char *e = pfd_knob("PFD_SUPER_ABLATE")) sa.ablate = atoi(e);
  if (const char *e = pfd_knob("PFD_TILE_PATCH")) use_patch = atoi(e) != 0;
  a.stamps = nullptr;
#ifdef PFD_DEVTOOLS
  if (const char *e = getenv("PFD_TILE_ABLATE")) a.ablate = atoi(e);
  if (a.ablate & 16) {
    PFDCHK(stampbuf.alloc(8192 * sizeof(u64)));
    a.stamps = stampbuf.as<u64>();
    HIPCHK(hipMemsetAsync(a.stamps, 0, 8192 * sizeof(u64), h->stream));
  }
#endif
  if (h->count_rounds) {
    a.ablate |= 32;
    PFDCHK(rcntbuf.alloc(1024 * sizeof(u64)));
    HIPCHK(hipMemsetAsync(rcntbuf.p, 0, 1024 * sizeof(u64), h->stream));
    a.rcnt = rcntbuf.as<u64>();
  }
  is_block = h->halo_top || h->halo_bot;
  // interior tiles (tile_fast.h): the tile, its ring and 4 staging columns either side inside the device raster, and
  // none of those rows a halo row or a boundary row of a row block:  r0 >= row_first + 1,  r0 + TS <= row_last,
  // c0 >= 4,  c0 + TS + 4 <= ncol
  a.tr_lo = (a.row_first + 1u + TS - 1u) / TS;
  a.tr_hi = a.row_last >= TS ? a.row_last / TS : 0u;
  a.tc_lo = 1u;
  a.tc_hi = a.ncol >= TS + 4u ? (a.ncol - 4u) / TS : 0u;
  if (a.tr_hi <= a.tr_lo || a.tc_hi <= a.tc_lo || pfd_knob("PFD_TILE_GENERAL")) a.tr_lo = a.tr_hi = a.tc_lo = a.tc_hi = 0u;
  return PFD_OK;
}

// level 3 over one flat id range: global doubling over all super-exits (small rasters, and the
// fallback when a hypertile holds more than HCAP super-exits)
int TiledRun::level3_flat(i64 *launches) {
  const u32 g3 = cdiv_u32(nsuper, 256);
  k_link3<<<g3, 256, 0, h->stream>>>(sa, nsuper, Jc);
  ++*launches;
  int batch = 1;
  for (u32 span = 1; span < (ntr + ntc) / SG + 2; span <<= 1) ++batch;  // ~log2 of a path in supertiles
  bool done3 = false;
  u32 *T[3] = {Tc, Tn, xin3}, *J[2] = {Jc, Jn};  // (xin3 is only used in hyper mode)
  int rounds = 0;
  PFDCHK(pfd_doubling_rounds(h, T, J, nsuper, batch + 2, true, &done3, &rounds, launches, nullptr));
  Tc = T[0], Tn = T[1], Jc = J[0], Jn = J[1];
  coarse_done = coarse_done && done3;
  k_sx_totals<<<g3, 256, 0, h->stream>>>(sa, nsuper, Tc);
  ++*launches;
  KCHK();
  return PFD_OK;
}

// The flat level 3 WITHOUT host round trips, for a whole raster of at most FLAT_MAX_HT hypertiles (10000^2: 25): the two
// hypertile solves of such a raster are 25 single-workgroup kernels on 256 CUs — 2 x 50 us of a 0.80 ms pass, plus the
// level-4 launches behind them — where one flat forest of its ~3e5 super-exits takes ~10 rounds of ~5 us.  Same machinery as
// level 4: device-side count, bounded grids, a fixed round budget whose shortfall shows at the pass's only synchronisation
// (rounds4 / short_of_rounds: the pass is redone with more rounds).
#define FLAT_MAX_HT 64u
int TiledRun::level3_flat_nosync(i64 *launches) {
  const u32 cap3 = (u32)std::min<size_t>(n3cap, 0x7FFFFFFF);
  const u32 g3 = std::min(cdiv_u32(cap3, 256), 4096u);
  k_link3_flat<<<g3, 256, 0, h->stream>>>(sa, cap3, Jc, Tn);
  // (a river that runs along a supertile edge crosses it again and again: its chain of super-exits is several times the
  //  raster's width in supertiles — 9 rounds fell short at 10000^2, and a miss costs a whole pass)
  int batch = 7 + extra_rounds;
  for (u32 span = 1; span < (ntr + ntc) / SG + 2; span <<= 1) ++batch;  // ~log2 of a path in supertiles
  if (const char *e = pfd_knob("PFD_TEST_ROUNDS4")) batch = atoi(e) + extra_rounds;  // (tests: force a miss)
  bool done3 = false;
  u32 *T[3] = {Tc, Tn, xin3}, *J[2] = {Jc, Jn};
  PFDCHK(pfd_doubling_rounds(h, T, J, cap3, batch, false, &done3, &rounds4, launches, h->ctrl + T_NSUPER, true));
  Tc = T[0], Tn = T[1], xin3 = T[2], Jc = J[0], Jn = J[1];
  k_sx_totals_flat<<<g3, 256, 0, h->stream>>>(sa, cap3, Tc);
  *launches += 2;
  KCHK();
  return PFD_OK;
}

// level 3 per hypertile in LDS + level 4 (global doubling over the hyper-exits only)
int TiledRun::level3_hyper(i64 *launches) {
  const u32 n3 = nht * HCAP;
  const u32 g3 = cdiv_u32(n3, 256);
  u32 *T3 = Tc, *J3 = Jc, *T3out = Tn;  // Jn is free: level 4 has its own buffers
  u32 *y = l4.as<u32>();
  u32 *hx_node = y, *T4c = y + n4cap;
  k_link3<<<g3, 256, 0, h->stream>>>(sa, n3, J3, xin3);
  HyperArgs ha{sx_slot, xtot, nht, hcntbuf.as<u32>(), T3, J3, xin3, T3out, R3, hx_id, hx_node, T4c, h->ctrl,
               edge_down_now ? cdiv_u32(ntr, SG) : 0u, nhtc};
  k_hyper<false><<<nht, 1024, 0, h->stream>>>(ha);
  KCHK();
  *launches += 2;
  if (is_block)  // (the second solve of a row block starts level 4 from these values plus what enters the block)
    HIPCHK(hipMemcpyAsync(y + 6 * n4cap, T4c, n4cap * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
  return level4_down(launches);
}

// level 4 (global doubling over the hyper-exits, start values in place) and the way down through level 3
int TiledRun::level4_down(i64 *launches) {
  u32 *T3 = Tc, *J3 = Jc, *T3out = Tn;
  u32 *y = l4.as<u32>();
  u32 *hx_node = y, *T4c = y + n4cap, *J4c = y + 3 * n4cap;
  u32 *T4[3] = {y + n4cap, y + 2 * n4cap, y + 5 * n4cap}, *J4[2] = {y + 3 * n4cap, y + 4 * n4cap};
  HyperArgs ha{sx_slot, xtot, nht, hcntbuf.as<u32>(), T3, J3, xin3, T3out, R3, hx_id, hx_node, T4c, h->ctrl,
               edge_down_now ? cdiv_u32(ntr, SG) : 0u, nhtc};
  {  // level 4: the number of hyper-exits stays on the device (grids are sized for the capacity)
    const u32 cap4 = (u32)std::min<size_t>(n4cap, 0x7FFFFFFF);
    const u32 g4 = std::min(cdiv_u32(cap4, 256), 4096u);  // (the kernels stride over the device-side count)
    k_link4<<<g4, 256, 0, h->stream>>>(ha, cap4, J4c, T4[1]);
    // No host round trip here: a fixed number of rounds is issued (rounds past saturation are
    // idempotent) and the "last round that moved a pointer" mark is compared with it at the pass's
    // final synchronisation; a miss redoes the pass with more rounds.
    int batch = 5 + extra_rounds;
    for (u32 span = 1; span < (ntr + ntc) / (SG * HG) + 2; span <<= 1) ++batch;  // ~log2 of a path in hypertiles
    if (const char *e = pfd_knob("PFD_TEST_ROUNDS4")) batch = atoi(e) + extra_rounds;  // (tests: force a miss)
    bool done4 = false;
    PFDCHK(pfd_doubling_rounds(h, T4, J4, cap4, batch, false, &done4, &rounds4, launches, h->ctrl + T_NHYPER, true));
    J4fin = J4[0];
    k_push4<<<g4, 256, 0, h->stream>>>(ha, cap4, T4[0], xin3);
    *launches += 2;
  }
  k_hyper<true><<<nht, 1024, 0, h->stream>>>(ha);
  *launches += 1;
  KCHK();
  return PFD_OK;
}

// Second solve of a row block (the flow entering from the other blocks as extra start values), when the first solve
// built the hierarchy per hypertile: roots, ids and links do not depend on the start values, so levels 2 and 3 are
// not solved upwards again — the inflow of every boundary-row cell is added to the start value of the super-exit
// its path leaves its supertile through (T3) and of the hyper-exit that path leaves its hypertile through (T4), by
// O(1) lookups; then level 4 runs again and the totals come down as usual.
int TiledRun::resolve_with_inflow(i64 *launches) {
  edge_down_now = false;
  sa.edge_nstr = 0;
  sa.xT = xT;
  const size_t nb = 2 * (size_t)h->ncol;
  u32 *y = l4.as<u32>();
  // (k_brow_scatter + k_brow_delta in one launch; the copy of the level-4 start values and the two clears in another)
  k_brow_scatter_delta<<<cdiv_u32(nb, 256), 256, 0, h->stream>>>(brow_first, brow_inflow, (u32)h->ncol, xT, sa.xmask, sa.xcb, R2L,
                                                                sxidL, Tc, R3, hx_id, y + 6 * n4cap, h->ctrl);
  const u32 n3 = (u32)((size_t)nht * HCAP), n4 = (u32)n4cap;
  k_l4_restart<<<std::min(cdiv_u32(std::max(n3, n4), 256), 2048u), 256, 0, h->stream>>>(y + 6 * n4cap, y + n4cap, n4, xin3, n3, h->ctrl);
  *launches += 2;
  PFDCHK(level4_down(launches));
  k_super<true><<<nst, SNT, 0, h->stream>>>(sa);
  k_super_flagged<true><<<std::min<u32>(nst, SFLAG_GRID), 1024, 0, h->stream>>>(sa);
  *launches += 2;
  KCHK();
  return PFD_OK;
}

// hierarchical solve of the exit graph for the start values `start` (one u32 per slot): totals of
// all exits, delivered (added) to the tile entries they drain into
// (cleared: the hypertile counters and the ctrl words of the solve are zero already — first solve of a
//  pass: the ctrl memset of phase_a and the local tile pass have done it)
// (edge_down: row blocks, first solve — the deliveries of the down-pass are only needed in the first
//  and last supertile row yet, where the halo sinks collect them; the second solve delivers everything)
int TiledRun::solve_exits(const u32 *start, i64 *launches, bool cleared, bool edge_down) {
  edge_down_now = edge_down;
  sa.edge_nstr = edge_down ? cdiv_u32(ntr, SG) : 0u;
  if (!cleared) {
    HIPCHK(hipMemsetAsync(hcntbuf.p, 0, (size_t)nht * sizeof(u32), h->stream));
    HIPCHK(hipMemsetAsync(h->ctrl + T_XACTIVE, 0, 3 * sizeof(u64), h->stream));  // T_XACTIVE, T_NSUPER, T_NHYPER
  }
  sa.xT = start;
  Tc = l3.as<u32>() + n3cap, Tn = Tc + n3cap, Jc = Tn + n3cap, Jn = Jc + n3cap;  // undo earlier buffer rotations
  sa.T3 = Tc;
  // level 3 runs per hypertile in LDS when the raster spans several hypertiles, else flat
  sa.hmode = (nht > 1 && !force_flat && !pfd_knob("PFD_FLAT_L3")) ? 1 : 0;
  // a whole raster of few hypertiles: one flat forest, no host look (level3_flat_nosync); PFD_FLAT_L3=0 keeps the hypertiles
  const char *hs = pfd_knob("PFD_HYPER_SMALL");
  flat_nosync = !is_block && sa.hmode && nht <= FLAT_MAX_HT && !(hs && atoi(hs) != 0);
  if (flat_nosync) sa.hmode = 0;
  xin3 = l3.as<u32>() + 5 * n3cap;  // (undo a rotation of level3_flat_nosync)
  if (setup_only) return PFD_OK;  // (phase_a in bands: the caller launches the first solve band by band)
  if (!up_done) k_super<false><<<nst, SNT, 0, h->stream>>>(sa);
  up_done = false;
  k_super_flagged<false><<<std::min<u32>(nst, SFLAG_GRID), 1024, 0, h->stream>>>(sa);  // (normally none)
  KCHK();
  *launches += 2;
  if (flat_nosync) {
    PFDCHK(level3_flat_nosync(launches));
  } else if (!sa.hmode) {  // the flat level-3 rounds are sized by the number of super-exits
    u64 c[8];
    HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    nsuper = (u32)c[T_NSUPER - 8];
  }
  // (hyper mode runs on without a host round trip; T_OVERFLOW — a hypertile with more super-exits
  //  than fit in LDS — is looked at when the pass is checked, and the pass is redone flat)
  if (flat_nosync) {
  } else if (sa.hmode)
    PFDCHK(level3_hyper(launches));
  else if (nsuper)
    PFDCHK(level3_flat(launches));
  k_super<true><<<nst, SNT, 0, h->stream>>>(sa);
  k_super_flagged<true><<<std::min<u32>(nst, SFLAG_GRID), 1024, 0, h->stream>>>(sa);
  *launches += 2;
  KCHK();
  return PFD_OK;
}

// phase A: local tile pass + hierarchical exit-graph solve with zero flow from other row blocks
int TiledRun::phase_a() {
  coarse_done = true;
  const size_t nb = 2 * (size_t)h->ncol;
  // One launch clears what the pass starts from: the control words (a deferred handle has never run a kernel: its
  // counters 0..7, C_BAD among them, still hold whatever the block held — cleared too), the boundary arrays of a row
  // block (brow_first / brow_sink = NONE, haloA / haloL = 0, brow_inflow = 0: read by the final tile pass), the exit
  // bitmasks (tiles beyond the raster edge of a partial supertile write none; 8 bytes per 64 slots + the flagged count).
  // No slot array needs clearing: every tile writes its 256 slots, slots of tiles beyond the raster edge are never read.
  {
    const size_t xbytes = nslots / 64 * sizeof(u64) + 8;  // (xmaskbuf is rounded up to 16 bytes in init)
    const u32 n16 = (u32)((xbytes + 15) / 16);
    const u32 work = std::max<u32>(n16, is_block ? (u32)(5 * nb) : 0u);
    k_pass_clear<<<std::max(1u, std::min(cdiv_u32(work, 256), 4096u)), 256, 0, h->stream>>>(
        h->ctrl, h->normalised ? 8u : 0u, bnd.as<u32>(), is_block ? (u32)nb : 0u, (uint4 *)a.xmask, n16);
  }
  // interior tiles: k_tile_local_fast; the frame around them (raster edge, halo and boundary rows): k_tile
  const dim3 gridi(a.tc_hi - a.tc_lo, a.tr_hi - a.tr_lo);
  const bool have_i = gridi.x && gridi.y;
  const u32 gridf = frame_tiles(ntr, ntc, a.tr_lo, a.tr_hi, a.tc_lo, a.tc_hi);
  if (const char *e = pfd_knob("PFD_BANDS")) {
    const int bands = atoi(e);
    if (bands > 1 && !is_block && have_i && !a.weights && !a.xT64 && !use_patch) return phase_a_bands(std::min(bands, 16));
  }
  pfd_seg_begin(h, "tile_local");
  if (!h->normalised) {  // deferred handle: decode + validate + count inside the tile pass
    if (!tcntbuf.p) PFDCHK(tcntbuf.alloc((size_t)ntr * ntc * sizeof(u64)));
    a.raw = h->raw;
    a.tcnt = tcntbuf.as<u64>();
    if (have_i) {
      if (a.weights) k_tile_local_fast<true, true><<<gridi, 256, 0, h->stream>>>(a);
      else if (a.xT64) k_tile_local_fast<true, false, true><<<gridi, 256, 0, h->stream>>>(a);
      else if (use_patch) k_tile_local_patch<true><<<gridi, 256, 0, h->stream>>>(a);
      else k_tile_local_fast<true, false><<<gridi, 256, 0, h->stream>>>(a);
      pfd_seg_end(h, 1);
      pfd_seg_begin(h, "tile_local_frame");
    }
    if (is_block)
      k_tile<false, true><<<gridf, 256, 0, h->stream>>>(a);
    else
      k_tile<false, true, true><<<gridf, 256, 0, h->stream>>>(a);
    fused_norm = true;
    KCHK();
    pfd_seg_end(h, 1);
    k_tile_counts<<<std::min<u32>(cdiv_u32((u64)ntr * ntc, 4096), 256u), 1024, 0, h->stream>>>(a.tcnt, ntr * ntc, h->ctrl);
  } else {
    if (have_i) {
      if (a.weights) k_tile_local_fast<false, true><<<gridi, 256, 0, h->stream>>>(a);
      else if (a.xT64) k_tile_local_fast<false, false, true><<<gridi, 256, 0, h->stream>>>(a);
      else if (use_patch) k_tile_local_patch<false><<<gridi, 256, 0, h->stream>>>(a);
      else k_tile_local_fast<false, false><<<gridi, 256, 0, h->stream>>>(a);
      pfd_seg_end(h, 1);
      pfd_seg_begin(h, "tile_local_frame");
    }
    if (is_block)
      k_tile<false><<<gridf, 256, 0, h->stream>>>(a);
    else
      k_tile<false, false, true><<<gridf, 256, 0, h->stream>>>(a);
    KCHK();
    pfd_seg_end(h, 1);
  }

  pfd_seg_begin(h, "exit_graph");
  i64 launches = 1;
  k_exit_lists<<<nst * 16, 256, 0, h->stream>>>(sa);  // (valid for every solve of the pass)
  k_boundary_records<<<nst * (SBN / 256), 256, 0, h->stream>>>(sa);
  ++launches;
  PFDCHK(solve_exits(xT, &launches, true, is_block));
  if (is_block) {  // what leaves through the halo rows, and where boundary-row inflow would leave
    HIPCHK(hipMemcpyAsync(haloL, haloA, nb * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
    const u32 ne = (ntr > 1 ? 2u : 1u) * ntc * PSL;
    k_halo_collect<<<cdiv_u32(ne, 256), 256, 0, h->stream>>>(esink.as<u32>(), xrec, xtot, ntc, ntr, nstc, (u32)h->ncol,
                                                            haloL, ne);
    LastExitArgs le{sa.xmask, sa.xcb, sa.xl_slot, R2L, sxidL, sx_slot, sx_n1, xrec, Jc, R3, hx_id, l4.as<u32>(), J4fin, sa.hmode};
    k_brow_sink<<<cdiv_u32(nb, 256), 256, 0, h->stream>>>(brow_first, le, esink.as<u32>(), ntc, ntr, nstc, (u32)h->ncol,
                                                         brow_sink, h->ctrl);
    launches += 3;
    KCHK();
  }
  pfd_seg_end(h, launches);
  return PFD_OK;
}

int TiledRun::phase_a_bands(int bands) {
  PFDCHK(pfd_aux_stream(h));
  const bool raw = !h->normalised;
  const u32 gridf = frame_tiles(ntr, ntc, a.tr_lo, a.tr_hi, a.tc_lo, a.tc_hi);
  pfd_seg_begin(h, "tile_local_frame");
  if (raw) {
    if (!tcntbuf.p) PFDCHK(tcntbuf.alloc((size_t)ntr * ntc * sizeof(u64)));
    a.raw = h->raw;
    a.tcnt = tcntbuf.as<u64>();
    k_tile<false, true, true><<<gridf, 256, 0, h->stream>>>(a);
    fused_norm = true;
  } else {
    k_tile<false, false, true><<<gridf, 256, 0, h->stream>>>(a);
  }
  KCHK();
  pfd_seg_end(h, 1);
  i64 launches = 0;
  setup_only = true;
  PFDCHK(solve_exits(xT, &launches, true, false));  // (pointers and modes of the solve: the first supertile solve reads them)
  setup_only = false;
  pfd_seg_begin(h, "tile_local");
  const u32 nstr = cdiv_u32(ntr, SG);
  const u32 nb_ = std::min<u32>((u32)bands, nstr);
  i64 nl = 0;
  for (u32 b = 0; b < nb_; ++b) {
    const u32 sr0 = (u32)((u64)nstr * b / nb_), sr1 = (u32)((u64)nstr * (b + 1) / nb_);
    const u32 lo = std::max(a.tr_lo, sr0 * SG), hi = std::min(a.tr_hi, sr1 * SG);
    if (hi > lo) {
      TileArgs ab = a;
      ab.tr_lo = lo, ab.tr_hi = hi;
      const dim3 g(a.tc_hi - a.tc_lo, hi - lo);
      if (raw) k_tile_local_fast<true, false><<<g, 256, 0, h->stream>>>(ab);
      else k_tile_local_fast<false, false><<<g, 256, 0, h->stream>>>(ab);
      ++nl;
    }
    if (!band_ev[b]) HIPCHK(hipEventCreateWithFlags(&band_ev[b], hipEventDisableTiming));
    HIPCHK(hipEventRecord(band_ev[b], h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, band_ev[b], 0));
    SuperArgs sb_ = sa;
    sb_.st0 = sr0 * nstc;
    const u32 nsb = (sr1 - sr0) * nstc;
    k_exit_lists<<<nsb * 16, 256, 0, h->stream2>>>(sb_);
    k_boundary_records<<<nsb * (SBN / 256), 256, 0, h->stream2>>>(sb_);
    k_super<false><<<nsb, SNT, 0, h->stream2>>>(sb_);
    launches += 3;
  }
  KCHK();
  HIPCHK(hipEventRecord(h->ev_join, h->stream2));
  HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  pfd_seg_end(h, nl);
  if (raw) k_tile_counts<<<std::min<u32>(cdiv_u32((u64)ntr * ntc, 4096), 256u), 1024, 0, h->stream>>>(a.tcnt, ntr * ntc, h->ctrl);
  pfd_seg_begin(h, "exit_graph");
  up_done = true;
  PFDCHK(solve_exits(xT, &launches, true, false));
  pfd_seg_end(h, launches);
  return PFD_OK;
}

// phase B: add the flow arriving from the other row blocks (brow_inflow, already on the device),
// final tile pass, completeness check
int TiledRun::phase_b(int *complete) {
  PFDCHK(phase_b_issue());
  u64 c0[48];
  HIPCHK(hipMemcpyAsync(c0, h->ctrl, sizeof(c0), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return phase_b_collect(c0, complete);
}
int TiledRun::stage_verdict_a() {
  k_stage_verdict_a<<<1, 1, 0, h->stream>>>(h->ctrl, (u32)sa.hmode, (u32)rounds4);
  KCHK();
  return PFD_OK;
}
int TiledRun::block_verdict(int *flag_dev, bool host_ok) {
  k_block_verdict<<<1, 1, 0, h->stream>>>(h->ctrl, (u32)sa.hmode, (u32)rounds4, host_ok && coarse_done ? 1u : 0u, flag_dev);
  KCHK();
  return PFD_OK;
}
int TiledRun::phase_b_issue() {
  const size_t nb = 2 * (size_t)h->ncol;
  if (is_block) {
    // The flow entering from the other row blocks is one more set of start values on the same exit
    // graph (added to the local counts of the exits it reaches first): the graph is solved once more,
    // now with the full down-pass.  The first solve left totals in the edge supertile rows only (that
    // is all the halo sinks needed); every exit's total is written again.
    pfd_seg_begin(h, "block_inflow");
    i64 launches = 0;
    if (sa.hmode && !pfd_knob("PFD_BLOCK_FULL_RESOLVE")) {
      PFDCHK(resolve_with_inflow(&launches));  // (structure of the first solve reused: values only)
    } else {
      k_brow_scatter<<<cdiv_u32(nb, 256), 256, 0, h->stream>>>(brow_first, brow_inflow, (u32)h->ncol, xT);
      KCHK();
      ++launches;
      PFDCHK(solve_exits(xT, &launches));
    }
    pfd_seg_end(h, launches);
  }
  const dim3 gridi(a.tc_hi - a.tc_lo, a.tr_hi - a.tr_lo);
  const bool have_i = gridi.x && gridi.y;
  pfd_seg_begin(h, "tile_final");
  if (have_i) {  // (segments: the interior kernel alone, then the frame around it)
    if (a.weights) k_tile_final_fast<true, FY_NT><<<gridi, FY_NT, 0, h->stream>>>(a);
    else k_tile_final_fast<false, FY_NT><<<gridi, FY_NT, 0, h->stream>>>(a);
    pfd_seg_end(h, 1);
    pfd_seg_begin(h, "tile_final_frame");
  }
  // (tried: the frame kernels on a side stream under the interior ones — the fork / join costs what the overlap
  //  saves: 5.132 vs 5.135 ms at 30000^2, 3.05 vs 2.95 ms for phase A of an 11250 x 90000 row block)
  k_tile<true><<<frame_tiles(ntr, ntc, a.tr_lo, a.tr_hi, a.tc_lo, a.tc_hi), 256, 0, h->stream>>>(a);
  KCHK();
  pfd_seg_end(h, 1);
  return PFD_OK;
}
int TiledRun::phase_b_collect(const u64 *c0, int *complete) {
  const u64 *c = c0 + 8;
  if (a.ablate & 32) {
    std::vector<u64> rc(1024);
    HIPCHK(hipMemcpy(rc.data(), a.rcnt, 1024 * sizeof(u64), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) {
      u64 v = 0;
      for (int w = 0; w < 256; ++w) v = (k & 1) ? v + rc[256 * k + w] : std::max(v, rc[256 * k + w]);
      h->tile_rounds[k] = (i64)v;
    }
  }
  if (fused_norm && !h->normalised) PFDCHK(pfd_adopt_counts(h, c0));  // bad codes / no pits surface here
  if (a.ablate & 16) {
    std::vector<u64> st(8192);
    HIPCHK(hipMemcpy(st.data(), a.stamps, 8192 * sizeof(u64), hipMemcpyDeviceToHost));
    u64 t[8] = {0};
    for (int r = 0; r < 1024; ++r)
      for (int k = 0; k < 8; ++k) t[k] += st[8 * r + k];
    const double nt = (double)ntr * ntc;
    for (int ph = 0; ph < 2; ++ph) {
      const u64 *q = t + 4 * ph;
      fprintf(stderr, "[k_tile<%d>] cycles/tile: load %.0f init %.0f doubling %.0f out %.0f\n", ph, q[0] / nt,
              q[1] / nt, q[2] / nt, q[3] / nt);
    }
  }
  // T_UNSAT: cells left unsaturated by a tile pass; T_SLIVE: supertile/hypertile solves that did not
  // saturate; T_OVERFLOW: a hypertile held more super-exits than fit in LDS (result invalid: redo flat)
  last_miss = c[T_MISS - 8];
  *complete = coarse_done && c[T_SLIVE - 8] == 0 && c[T_UNSAT - 8] == 0;
  overflowed = c[T_OVERFLOW - 8] != 0;
  // level 4 ran a fixed number of rounds: saturated iff the last one moved no pointer
  short_of_rounds = (sa.hmode || flat_nosync) && rounds4 > 0 && c[T_XACTIVE - 8] >= (u64)rounds4;
#ifdef PFD_DEVTOOLS
  if (getenv("PFD_DEBUG_ROUNDS")) fprintf(stderr, "[level4] rounds issued %d, last active %llu\n", rounds4, (unsigned long long)c[T_XACTIVE - 8]);
#endif
  if (overflowed || short_of_rounds) *complete = 0;
  return PFD_OK;
}

// phase A for row blocks: the boundary records leave the GPU right after it, so an overflow of
// the per-hypertile id range has to be known (and repaired by a flat re-run) before that
int TiledRun::phase_a_checked() {
  PFDCHK(phase_a());
  return phase_a_check();
}
bool TiledRun::phase_a_needs_redo(const u64 *c8, int tries) {
  if (!sa.hmode || tries >= 3) return false;
  if (c8[T_OVERFLOW - 8]) {
    force_flat = true;
    return true;
  }
  if (rounds4 > 0 && c8[T_XACTIVE - 8] >= (u64)rounds4 && tries < 2) {
    extra_rounds += 8;  // (a cyclic exit graph never saturates: give up after two extensions)
    return true;
  }
  return false;
}
// the check alone: callers that run phase A of several blocks concurrently issue all of them first
int TiledRun::phase_a_check() {
  for (int tries = 0; sa.hmode && tries < 3; ++tries) {
    u64 c[8];
    HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!phase_a_needs_redo(c, tries)) break;
    PFDCHK(phase_a());
  }
  return PFD_OK;
}

// returns PFD_OK and *complete = 1 when every valid cell was finalised (no cycles)
// `weights` (optional): integer payload accumulated instead of unit weights (int32 wrap like the
// reference's int32 accuflux; the caller guarantees that the nodata rule cannot interfere)
int pfd_upstream_area_cell_tiled(pfd_raster *h, i32 *out_dev, int *complete, const i32 *weights) {
  *complete = 0;
  TiledRun run;
  PFDCHK(run.init(h, out_dev));
  if (!run.supported) return PFD_OK;
  run.a.weights = weights;
  PFDCHK(run.phase_a());
  PFDCHK(run.phase_b(complete));
  for (int tries = 0; tries < 3 && (run.overflowed || (run.short_of_rounds && tries < 2)); ++tries) {
    // rare: a hypertile overflowed its LDS id range (redo with one flat id range for level 3), or
    // level 4 needed more rounds than were issued (a cyclic exit graph never saturates: two
    // extensions, then the level engine takes over)
    if (run.overflowed)
      run.force_flat = true;
    else
      run.extra_rounds += 8;
    PFDCHK(run.phase_a());
    PFDCHK(run.phase_b(complete));
  }
  if (*complete && !run.is_block) h->acyclic = 1;  // every valid cell was finalised: no cycles
  return PFD_OK;
}

#include "wide.h"
