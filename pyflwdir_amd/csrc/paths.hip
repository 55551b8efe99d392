// paths.hip — LDS-tiled PATH queries by pointer doubling: "how far is the pit" (rank) and
// "which is the first outlet downstream" (basin labels).  Both are properties of the path from
// a cell to its pit, both are pure integer/copy operations (bit-exact in any order), and both
// use the two-pass tile scheme of tiled.hip with GATHER-only doubling (no atomics at all):
//
//     rank :  V_0(z) = 1 (0 at the path end),  V(z) += V(J(z)),  J(z) = J(J(z))
//     label:  outlets are absorbing path ends;  after saturation J(z) = first outlet or the end
//
//   pass 1  k_path<MODE, false>  per 64x64 tile in LDS; per perimeter slot: the slot an exit drains
//                                 into, and for an entry the exit its in-tile path reaches plus the
//                                 hops to it (rank) / the outlet met on the way (label)
//   exits   k_xinit + k_xround    the exits form chains; the same gather doubling over them gives
//                                 every exit its rank / its first outlet
//   pass 2  k_path<MODE, true>    per tile again: every cell = own in-tile part + its exit's value
//
// Replaces, on rasters without cycles:
//   core.rank                 pyflwdir/core.py:17-47   (and, with a radix sort of the cells by rank,
//   core.idxs_seq             pyflwdir/core.py:87-117   the level structure of the level engine)
//   basins.basins             pyflwdir/basins.py:12-18 + core.fillnodata_upstream core.py:120-146
#include <string.h>

#include <cstring>
#include <unordered_set>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"
#include "tiled.h"

enum { MODE_RANK = 0, MODE_LABEL = 1 };
enum { P_UNSAT = 8, P_XACTIVE = 10, P_MAXRANK = 13 };  // ctrl slots (u64)
#define KEY_INVALID 0xFFFFFFFFu

struct PathArgs {
  const u8 *ncode;
  u32 nrow, ncol, ntr, ntc, nstc;
  u32 *xtgt;        // [nslots] slot the exit on this slot drains into, NONE32 if no exit
  u32 *elink;       // [nslots] entry: slot of the exit its in-tile path reaches, NONE32 if it ends in the tile
  u32 *eval;        // [nslots] entry: hops to the end of its in-tile path (rank) / outlet met there (label)
  const u64 *xres;  // [nslots] pass 2: low word = rank of the exit cell / outlet the exit finally reaches
  const u32 *seed;  // [n] label mode: outlet number (1-based) seeded on the cell, 0 = none
  u32 *out;         // [n] pass 2: rank (KEY_INVALID on nodata) / outlet number per cell
  const u8 *tflag;  // label mode, nullable: [ntr * ntc] tile holds a seed; the seeds of the other tiles are never read
  const u32 *ids32; // label mode, nullable: the outlets' 32-bit labels — `out` then receives ids32[number - 1] (0: none)
  u64 *ctrl;
  // rank mode with TAIL (the chains of exact.hip): the END of every cell's path as well — linear index + 1 of the pit
  // it reaches — from the same doubling: an entry whose in-tile path ends in the tile records that pit (eend), the exit
  // rounds leave in every exit's pointer the LAST exit of its chain, whose target entry holds the pit
  u32 *eend;        // [nslots]
  u32 *out2;        // [n]
  // a pass over some tile rows only, into a window of rows (row blocks of basins: the boundary rows first, the rest when
  // the labels of the halo cells are known): tile row = blockIdx.y + tr_off; `out` holds the rows [out_row0, out_row0 + out_nrows)
  u32 tr_off = 0, out_row0 = 0, out_nrows = 0xFFFFFFFFu;
  const u32 *halo32 = nullptr;  // label mode with ids32: labels of the halo cells of a row block ([2 * ncol], tag -> label)
};

template <int MODE, bool FINAL, bool TAIL = false>
__global__ void __launch_bounds__(256) k_path(PathArgs a) {
  __shared__ __attribute__((aligned(16))) u32 V[TCELLS];       // rank: hops so far; label: seed number
  __shared__ __attribute__((aligned(16))) uint16_t P[TCELLS];  // 2^k-th ancestor | PDONE once saturated
  __shared__ __attribute__((aligned(16))) u8 code[HW * CP];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_, tr = by_ + a.tr_off;
  const u32 sbase = sslot_base(tr, tc, a.nstc);
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  {
    u32 v[5];
    stage_load_auto(a.ncode, a.nrow, a.ncol, r0, c0, tid, v);
    stage_store(code, tid, v);
  }
  __syncthreads();

  // ---- initial pointers / values -----------------------------------------------------------------
  // (a few outlets on a large raster: most tiles hold none, and 4 bytes of seed per cell and pass are a third of the
  //  query's traffic — the caller flags the tiles that hold one and zeroes only those)
  const bool seeded = MODE == MODE_LABEL && (a.tflag == nullptr || a.tflag[(size_t)tr * a.ntc + tc] != 0);
  u32 pc[QPT * 4];
  u32 live = 0;
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const int lr = l0 >> 6, lc0 = l0 & 63;
    const u32 c4 = *(const u32 *)&CODE(lr, lc0);
    u32 s4[4] = {0, 0, 0, 0};
    if (MODE == MODE_LABEL && seeded) {  // outlets seeded on the cells of this quad (nodata cells may be seeded too)
      const i64 gr = r0 + lr, gc0 = c0 + lc0;
      const i64 crr = gr >= (i64)a.nrow ? (i64)a.nrow - 1 : gr;
      const i64 ccs = gc0 >= (i64)a.ncol ? (i64)a.ncol - 1 : gc0;
      // unconditional 16-byte load from a clamped address (the seed array carries slack past its end)
      __builtin_memcpy(s4, a.seed + (size_t)crr * a.ncol + (size_t)ccs, 16);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gr >= (i64)a.nrow || gc0 + b >= (i64)a.ncol) s4[b] = 0;
    }
    u32 v4[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u32 c = (c4 >> (8 * b)) & 0xFFu;
      const u32 l = l0 + b;
      // branch-free decode: (dr, dc) from two packed 2-bit tables; an outlet is the end of its path
      const int k = (int)__builtin_ctz(c | 0x100u);
      const int nr = lr + (int)((0x101A9u >> (2 * k)) & 3u) - 1;
      const int nc = lc0 + b + (int)((0x1901Au >> (2 * k)) & 3u) - 1;
      const bool go = d8_is_dir(c) && !(MODE == MODE_LABEL && s4[b]) && (unsigned)nr < TS && (unsigned)nc < TS;
      const u32 p = go ? (u32)(nr * TS + nc) : (l | PDONE);
      pc[4 * j + b] = p;
      if (!(p & PDONE)) live |= 1u << (4 * j + b);
      v4[b] = (MODE == MODE_RANK) ? ((p & PDONE) ? 0u : 1u) : s4[b];
    }
    *(uint4 *)&V[l0] = make_uint4(v4[0], v4[1], v4[2], v4[3]);
    *(uint2 *)&P[l0] = make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
  }
  __syncthreads();

  // ---- gather doubling ---------------------------------------------------------------------------
  // "does any pointer still move": wave leaders publish a flag, everybody reads the four flags after ONE barrier (two
  // alternating rows: a wave reaches the write of round r + 2 only after every wave has read round r's row) — a
  // __syncthreads_or is three barriers and a reduction
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][4];
  auto vote = [&](int round) -> bool {
    const bool any = __ballot(live != 0u) != 0ull;
    if ((tid & 63u) == 0u) s_flag[round & 1][tid >> 6] = any ? 1u : 0u;
    __syncthreads();
    const uint4 f = *(const uint4 *)s_flag[round & 1];
    return (f.x | f.y | f.z | f.w) != 0u;
  };
  if (MODE == MODE_LABEL) {
    // labels travel with the ROOT (read at the end), so the loop moves pointers only: every value a pointer ever holds
    // is an ancestor of its cell, whoever advanced it when — no barrier between reads and writes, and two jumps per
    // round (4 rounds instead of 6 on a river raster)
    // (the four gathers of a quad go out together, unconditionally: a load under a per-cell branch is waited for on the
    //  spot — 8 serialised LDS round trips per quad and round; a cell that has arrived re-reads its root's word, which
    //  names the root)
    for (int round = 0; round < MAXROUNDS_TILE; ++round) {
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (!(live & (0xFu << (4 * j)))) continue;
        u32 q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = P[pc[4 * j + b] & 0xFFFu];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = P[q[b] & 0xFFFu];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          pc[4 * j + b] = q[b];
          if (q[b] & PDONE) live &= ~(1u << (4 * j + b));
        }
        *(uint2 *)&P[4u * tid + 1024u * j] = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
      }
      if (!vote(round)) break;
    }
  } else {
    for (int round = 0; round < MAXROUNDS_TILE; ++round) {
      u32 q[QPT * 4], dv[QPT * 4];
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (!(live & (0xFu << (4 * j)))) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {  // (unconditional inside a live quad; a cell that has arrived is not updated below)
          const u32 t = pc[4 * j + b] & 0xFFFu;
          q[4 * j + b] = P[t];
          dv[4 * j + b] = V[t];
        }
      }
      __syncthreads();  // every read of this round precedes every write of this round
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (live & (0xFu << (4 * j))) {
          const u32 l0 = 4u * tid + 1024u * j;
          uint4 v4 = *(const uint4 *)&V[l0];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const bool lv = (live >> (4 * j + b)) & 1u;
            ((u32 *)&v4)[b] += lv ? dv[4 * j + b] : 0u;
            pc[4 * j + b] = lv ? q[4 * j + b] : pc[4 * j + b];
            if (lv && (q[4 * j + b] & PDONE)) live &= ~(1u << (4 * j + b));
          }
          *(uint4 *)&V[l0] = v4;
          *(uint2 *)&P[l0] = make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
        }
      }
      if (!vote(round)) break;
    }
  }
  if (live) atomicAdd((unsigned long long *)&a.ctrl[P_UNSAT], (unsigned long long)__popc(live));  // cycles
  __syncthreads();

  // is the end of an in-tile path an exit?  -> its slot, else NONE32
  auto exit_slot_of = [&](u32 root) -> u32 {
    const int rr = root >> 6, rc = root & 63;
    const u32 cr = CODE(rr, rc);
    if (d8_is_dir(cr) && !(MODE == MODE_LABEL && V[root])) {
      const int k = d8_slot(cr);
      const int nr = rr + d8_dr(k), nc = rc + d8_dc(k);
      if ((unsigned)nr >= TS || (unsigned)nc >= TS) return sbase + (u32)pslot(rr, rc);
    }
    return NONE32;
  };

  if (FINAL) {
    // The result of an exit (hops from there to the pit / outlet met further down) goes into the exit cell's own
    // word of V — a root, whose word holds 0 so far — by the <= 252 perimeter threads; every cell then reads its
    // root's word from LDS instead of decoding the root and gathering from global memory itself.
    if (tid < NPERIM) {
      int plr, plc;
      pslot_inv((int)tid, &plr, &plc);
      const u32 l = (u32)(plr * TS + plc);
      if (exit_slot_of(l) != NONE32) V[l] = (u32)a.xres[sbase + tid];
    }
    __syncthreads();
    u32 mx = 0;
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const u32 l0 = 4u * tid + 1024u * j;
      const int lr = l0 >> 6, lc0 = l0 & 63;
      const i64 gr = r0 + lr, gc0 = c0 + lc0;
      if (gr >= (i64)a.nrow || gc0 >= (i64)a.ncol) continue;
      const u32 c4 = *(const u32 *)&CODE(lr, lc0);
      u32 o4[4], vr[4];
      const uint4 own4 = *(const uint4 *)&V[l0];  // (rank: the cells' own hops to their roots)
      const u32 own[4] = {own4.x, own4.y, own4.z, own4.w};
#pragma unroll
      for (int b = 0; b < 4; ++b) vr[b] = V[pc[4 * j + b] & 0xFFFu];  // (the pointer registers ARE P[l]: no re-read; all four gathers in flight)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const u32 c = (c4 >> (8 * b)) & 0xFFu;
        const u32 l = l0 + b;
        const u32 root = pc[4 * j + b] & 0xFFFu;
        u32 val;
        if (MODE == MODE_RANK) {
          val = KEY_INVALID;
          if (c != D8_MV) {
            val = own[b] + (root != l ? vr[b] : 0u);  // hops to the root + the root's hops from there on
            mx = max(mx, val);
          }
        } else {
          val = vr[b];  // outlet at the end of the in-tile path (the cell's own seed, or what the exit reaches)
          // (32-bit labels straight from the table: saves the pass that maps numbers to labels — 8 bytes per cell)
          if (a.ids32) {  // (unconditional load from a clamped index, then the select: 16 loads in flight, not 16 round trips)
            const bool tag = a.halo32 != nullptr && (val & 0x80000000u) != 0u;  // (BTAG: the path leaves through a halo cell)
            const u32 id = a.ids32[(val && !tag) ? val - 1u : 0u];
            u32 hl = 0;
            if (a.halo32) hl = a.halo32[tag ? ((val >> 30) & 1u) * a.ncol + (val & 0x3FFFFFFFu) : 0u];
            val = tag ? hl : (val ? id : 0u);
          }
        }
        o4[b] = val;
      }
      if (a.out == nullptr) continue;  // (statistics only: pfd_graph_stats)
      const i64 orow = gr - (i64)a.out_row0;
      if (orow < 0 || orow >= (i64)a.out_nrows) continue;  // (rows outside the window: another pass writes them)
      u32 *dst = a.out + (size_t)orow * a.ncol + (size_t)gc0;
      if (gc0 + 3 < (i64)a.ncol && (((size_t)dst) & 15) == 0) {
        *(uint4 *)dst = make_uint4(o4[0], o4[1], o4[2], o4[3]);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (gc0 + b < (i64)a.ncol) dst[b] = o4[b];
      }
    }
    if (TAIL) {
      __syncthreads();  // every rank is out: V is free
      // a root names the end of its path: itself when it is a pit ...
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const u32 l0 = 4u * tid + 1024u * j;
        const int lr = l0 >> 6, lc0 = l0 & 63;
        const u32 c4 = *(const u32 *)&CODE(lr, lc0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const u32 l = l0 + b;
          if ((P[l] & 0xFFFu) == l)
            V[l] = ((c4 >> (8 * b)) & 0xFFu) == 0u ? (u32)((r0 + lr) * (i64)a.ncol + c0 + lc0 + b) + 1u : 0u;
        }
      }
      __syncthreads();
      // ... or, for an exit, what the entry behind the last exit of its chain recorded
      if (tid < NPERIM) {
        int plr, plc;
        pslot_inv((int)tid, &plr, &plc);
        const u32 l = (u32)(plr * TS + plc);
        if (exit_slot_of(l) != NONE32) V[l] = a.eend[a.xtgt[(u32)(a.xres[sbase + tid] >> 32) & ~XDONE]];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const u32 l0 = 4u * tid + 1024u * j;
        const int lr = l0 >> 6, lc0 = l0 & 63;
        const i64 gr = r0 + lr, gc0 = c0 + lc0;
        if (gr >= (i64)a.nrow || gc0 >= (i64)a.ncol) continue;
        const u32 c4 = *(const u32 *)&CODE(lr, lc0);
        u32 o4[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) o4[b] = ((c4 >> (8 * b)) & 0xFFu) != D8_MV ? V[P[l0 + b] & 0xFFFu] : 0u;
        u32 *dst = a.out2 + (size_t)gr * a.ncol + (size_t)gc0;
        if (gc0 + 3 < (i64)a.ncol && (((size_t)dst) & 15) == 0) {
          *(uint4 *)dst = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (gc0 + b < (i64)a.ncol) dst[b] = o4[b];
        }
      }
    }
    if (MODE == MODE_RANK) {
      // the longest path: one candidate per tile, and the atomic only if it can still raise the maximum (same-address
      // global atomics are served one after the other: four per tile cost 10 ms at 30000^2)
      __shared__ u32 s_mx[4];
      for (int o = 32; o > 0; o >>= 1) mx = max(mx, (u32)__shfl_down(mx, o));
      if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
      __syncthreads();
      if (tid == 0) {
        const u32 m = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        if ((u64)m > __hip_atomic_load(&a.ctrl[P_MAXRANK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          atomicMax((unsigned long long *)&a.ctrl[P_MAXRANK], (unsigned long long)m);
      }
    }
    return;
  }

  // ---- pass 1: perimeter records ------------------------------------------------------------------
  if (tid < PSL) {
    u32 tgt = NONE32, link = NONE32, ev = 0, end = 0;
    if (tid < NPERIM) {
      int plr, plc;
      pslot_inv((int)tid, &plr, &plc);
      const u32 l = (u32)(plr * TS + plc);
      const u32 c = CODE(plr, plc);
      if (c != D8_MV) {
        if (d8_is_dir(c) && !(MODE == MODE_LABEL && V[l])) {  // exit?
          const int k = d8_slot(c);
          const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
          if ((unsigned)nr >= TS || (unsigned)nc >= TS) {
            const i64 gr = r0 + nr, gc = c0 + nc;
            tgt = sslot_base((u32)(gr >> 6), (u32)(gc >> 6), a.nstc) + (u32)pslot((int)(gr & 63), (int)(gc & 63));
          }
        }
        bool entry = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
          if (((unsigned)nr >= TS || (unsigned)nc >= TS) && CODE(nr, nc) == (1u << ((k + 4) & 7))) entry = true;
        }
        if (entry) {
          const u32 root = P[l] & 0xFFFu;
          link = exit_slot_of(root);
          ev = (MODE == MODE_RANK) ? V[l] : V[root];
          if (TAIL && link == NONE32 && CODE((int)(root >> 6), (int)(root & 63u)) == 0u)
            end = (u32)((r0 + (i64)(root >> 6)) * (i64)a.ncol + c0 + (i64)(root & 63u)) + 1u;
        }
      }
    }
    a.xtgt[sbase + tid] = tgt;
    a.elink[sbase + tid] = link;
    a.eval[sbase + tid] = ev;
    if (TAIL) a.eend[sbase + tid] = end;
  }
}

// ---------------------------------------------------------------------------------------------
// exit chains: W(e) = value accumulated from exit e to the end of its path, J(e) = next exit
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void flag_active_p(u64 *ctrl) {
  const u64 m = __ballot(1);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
    if (__hip_atomic_load(&ctrl[P_XACTIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
      __hip_atomic_store(&ctrl[P_XACTIVE], (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The exit graph of a path query: per perimeter slot ONE 64-bit word (value | pointer << 32), doubled IN PLACE.  A
// word always states "the value holds up to the slot the pointer names" (rank: hops; label: no outlet met yet), so
// composing it with ANY consistent word of that slot — old or already advanced by this very round — gives another
// true statement: no ping-pong buffers, no copy of the finished two thirds of the slots per round (round 2's form
// moved 32 bytes per slot and round; this one reads 8 and writes 8 where something moved), and pointers that were
// advanced earlier in the round shorten the path further.  8-byte relaxed atomic accesses keep the pairs untorn.
__device__ __forceinline__ u64 wj_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wj_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int MODE>
__global__ void __launch_bounds__(256) k_xinit(const u32 *__restrict__ xtgt, const u32 *__restrict__ elink,
                                               const u32 *__restrict__ eval, u64 *__restrict__ WJ, u32 nslots, u64 *ctrl) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const u32 q = xtgt[s];
  u32 w = 0, j = s | XDONE;
  if (q != NONE32) {
    const u32 l = elink[q];
    if (MODE == MODE_RANK) {
      w = 1 + eval[q];  // one hop into the next tile + the hops to the end of the path in there
      if (l != NONE32) j = l;
    } else {
      w = eval[q];  // outlet met inside the next tile (0: none)
      if (!w && l != NONE32) j = l;
    }
  }
  WJ[s] = (u64)w | ((u64)j << 32);
}

// FLAG: the round reports whether a pointer is still moving — only the LAST round of a batch does (the flag word is
// one address for every wave that has something to report: not free, and an earlier round's report says nothing about
// the state after the batch)
template <int MODE, bool FLAG>
__global__ void __launch_bounds__(256) k_xround(u64 *__restrict__ WJ, u32 nslots, u64 *ctrl, u8 *__restrict__ wdone) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  // a wave whose 64 slots are all saturated (or hold no exit) has nothing left to do in any later round: one byte per
  // wave instead of 512 bytes of slot words (two thirds of the slots hold no exit, and most chains are short)
  if (wdone[s >> 6]) return;  // (wave-uniform: nslots is a multiple of 64)
  bool moving = false;
  if (s < nslots) {
    const u64 own = wj_load(WJ + s);
    const u32 j = (u32)(own >> 32);
    if (!(j & XDONE)) {
      u32 w = (u32)own;
      const u64 oth = wj_load(WJ + j);
      const u32 wj = (u32)oth;
      u32 q = (u32)(oth >> 32);
      if (MODE == MODE_RANK) {
        w += wj;
      } else if (wj) {  // the first outlet on the path wins; the chain ends there
        w = wj;
        q = s | XDONE;
      }
      // a second jump in the same round (the composed word is a true statement again): half the launches and half the
      // passes over the slots' own words for the same gathers
      if (!(q & XDONE)) {
        const u64 oth2 = wj_load(WJ + q);
        const u32 wj2 = (u32)oth2;
        q = (u32)(oth2 >> 32);
        if (MODE == MODE_RANK) {
          w += wj2;
        } else if (wj2) {
          w = wj2;
          q = s | XDONE;
        }
      }
      wj_store(WJ + s, (u64)w | ((u64)q << 32));
      moving = !(q & XDONE);
    }
  }
  const bool any = __ballot(moving) != 0ull;
  if (!any && (threadIdx.x & 63u) == 0u) wdone[s >> 6] = 1;
  if (FLAG && any) flag_active_p(ctrl);
}

// one complete path query; on return *complete = 0 means cycles were found (caller falls back)
template <int MODE>
static int run_paths(pfd_raster *h, const u32 *seed_dev, u32 *out_dev, int *complete, u32 *maxrank,
                     const u8 *codes = nullptr, const u32 *ids32 = nullptr, const u8 *tflag = nullptr, u32 *tails_dev = nullptr) {
  const bool tail = MODE == MODE_RANK && tails_dev != nullptr;  // (see PathArgs::eend)
  *complete = 0;
  if (!codes) codes = h->ncode;  // (exact.hip queries a derived forest: heavy links only)
  const u32 ntr = cdiv_u32((u64)h->nrow, TS), ntc = cdiv_u32((u64)h->ncol, TS);
  const u32 nstc = cdiv_u32(ntc, SG);
  const size_t nslots = (size_t)cdiv_u32(ntr, SG) * nstc * SSL;
  if (nslots >= 0x3FFFFFFFull || ntr > 65535u) return PFD_OK;
  DevBuf buf;
  PFDCHK(buf.alloc((tail ? 6 : 5) * nslots * sizeof(u32)));
  u32 *b = buf.as<u32>();
  u32 *xtgt = b + 2 * nslots, *elink = b + 3 * nslots, *eval = b + 4 * nslots;
  u64 *WJ = buf.as<u64>();  // (first: 8-byte aligned)
  HIPCHK(hipMemsetAsync(h->ctrl + 8, 0, 8 * sizeof(u64), h->stream));
  HIPCHK(hipMemsetAsync(xtgt, 0xFF, nslots * sizeof(u32), h->stream));  // slots of tiles that do not exist
  PathArgs a{codes, (u32)h->nrow, (u32)h->ncol, ntr, ntc, nstc, xtgt, elink, eval, nullptr, seed_dev, out_dev, tflag, ids32, h->ctrl,
             tail ? b + 5 * nslots : nullptr, tails_dev};
  const dim3 grid(ntc, ntr);
  i64 launches = 2;
  if (tail) k_path<MODE_RANK, false, true><<<grid, 256, 0, h->stream>>>(a);
  else k_path<MODE, false><<<grid, 256, 0, h->stream>>>(a);
  const u32 sgrid = cdiv_u32(nslots, 256);
  k_xinit<MODE><<<sgrid, 256, 0, h->stream>>>(xtgt, elink, eval, WJ, (u32)nslots, h->ctrl);
  KCHK();
  DevBuf wdone;  // one byte per 64 slots: the wave's slots are saturated (k_xround)
  PFDCHK(wdone.alloc(nslots / 64 + 64));
  HIPCHK(hipMemsetAsync(wdone.p, 0, nslots / 64 + 64, h->stream));
  bool done = false;
  int batch = 2;
  for (u32 span = 1; span < ntr + ntc; span <<= 2) ++batch;  // (two jumps per round: a round covers a factor of 4)
  for (int rounds = 0; rounds < 40 && !done;) {
    for (int r = 0; r + 1 < batch; ++r, ++rounds) {
      k_xround<MODE, false><<<sgrid, 256, 0, h->stream>>>(WJ, (u32)nslots, h->ctrl, wdone.as<u8>());
      ++launches;
    }
    // the last round of the batch reports: saturated iff it left no pointer moving
    HIPCHK(hipMemsetAsync(h->ctrl + P_XACTIVE, 0, sizeof(u64), h->stream));
    k_xround<MODE, true><<<sgrid, 256, 0, h->stream>>>(WJ, (u32)nslots, h->ctrl, wdone.as<u8>());
    ++launches, ++rounds;
    KCHK();
    u64 active = 0;
    HIPCHK(hipMemcpyAsync(&active, h->ctrl + P_XACTIVE, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    done = active == 0;
    batch = 2;
  }
  a.xres = WJ;
  if (tail) k_path<MODE_RANK, true, true><<<grid, 256, 0, h->stream>>>(a);
  else k_path<MODE, true><<<grid, 256, 0, h->stream>>>(a);
  KCHK();
  u64 c[6];
  HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *complete = done && c[P_UNSAT - 8] == 0;
  if (maxrank) *maxrank = (u32)c[P_MAXRANK - 8];
  (void)launches;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// basins on a raster row-tiled over several GPUs (BASELINE config 5): the mirror image of the
// accumulation protocol of dist.hip.  A label is a property of the downstream path, so per block:
//   begin   local label query with the halo cells as extra path ends carrying a TAG (side, column).  Every
//           cell of the block then holds the number of a local outlet, 0 (its path ends at a pit of the
//           block), or the tag of the halo cell through which its path leaves the block.  The record that
//           leaves the GPU holds, per cell of the own first / last row: the final label, or that tag.
//   exchange  all-gather of the records (6 * ncol words per block; any transport)
//   finish  the halo cell (b, side, c) IS a boundary-row cell of block b -/+ 1: following the tags through
//           the gathered records (host, O(interface cells), path compression) gives every halo cell of this
//           block its final label; one streaming pass writes the block's labels.
// Reference: basins.basins + core.fillnodata_upstream (pyflwdir/basins.py:12-18, core.py:120-146).
// ---------------------------------------------------------------------------------------------
#define BTAG 0x80000000u
__global__ void __launch_bounds__(256) k_zero_seed_tiles(const u8 *__restrict__ tflag, u32 nrow, u32 ncol, u32 ntc, u32 *__restrict__ seed);
// The label query in two halves (round 5): start() = local tile pass + exit rounds; final_rows() = the final tile pass
// over some tile rows into a window of rows.  A row block runs the final pass on the two tile rows that hold its
// boundary rows first (their numbers / tags make the record that leaves the GPU) and on everything only when the labels
// of its halo cells have come back — so the block's labels are written ONCE, by the tile pass itself (before: the
// numbers of every cell, then a pass of their own that mapped numbers and tags to labels: 8 bytes per cell more, and
// a 4-byte-per-cell memset of the seeds that the tile flags make unnecessary).
struct LabelRun {
  DevBuf buf, wdone, seed, tflag;
  PathArgs a{};
  u32 ntr = 0, ntc = 0;
  bool done = false;
  int start(pfd_raster *h) {  // seed / tflag are filled by the caller
    ntr = cdiv_u32((u64)h->nrow, TS), ntc = cdiv_u32((u64)h->ncol, TS);
    const u32 nstc = cdiv_u32(ntc, SG);
    const size_t nslots = (size_t)cdiv_u32(ntr, SG) * nstc * SSL;
    if (nslots >= 0x3FFFFFFFull || ntr > 65535u) {
      pfd_set_error("the block is too large for the tiled label query");
      return PFD_EUNSUPPORTED;
    }
    PFDCHK(buf.alloc(5 * nslots * sizeof(u32)));
    u32 *b = buf.as<u32>();
    u32 *xtgt = b + 2 * nslots, *elink = b + 3 * nslots, *eval = b + 4 * nslots;
    u64 *WJ = buf.as<u64>();
    HIPCHK(hipMemsetAsync(h->ctrl + 8, 0, 8 * sizeof(u64), h->stream));
    HIPCHK(hipMemsetAsync(xtgt, 0xFF, nslots * sizeof(u32), h->stream));  // slots of tiles that do not exist
    a = PathArgs{h->ncode, (u32)h->nrow, (u32)h->ncol, ntr, ntc, nstc, xtgt, elink, eval, nullptr, seed.as<u32>(), nullptr,
                 tflag.as<u8>(), nullptr, h->ctrl, nullptr, nullptr};
    k_path<MODE_LABEL, false><<<dim3(ntc, ntr), 256, 0, h->stream>>>(a);
    const u32 sgrid = cdiv_u32(nslots, 256);
    k_xinit<MODE_LABEL><<<sgrid, 256, 0, h->stream>>>(xtgt, elink, eval, WJ, (u32)nslots, h->ctrl);
    KCHK();
    PFDCHK(wdone.alloc(nslots / 64 + 64));
    HIPCHK(hipMemsetAsync(wdone.p, 0, nslots / 64 + 64, h->stream));
    done = false;
    int batch = 2;
    for (u32 span = 1; span < ntr + ntc; span <<= 1) ++batch;
    for (int rounds = 0; rounds < 40 && !done;) {
      for (int r = 0; r + 1 < batch; ++r, ++rounds) k_xround<MODE_LABEL, false><<<sgrid, 256, 0, h->stream>>>(WJ, (u32)nslots, h->ctrl, wdone.as<u8>());
      HIPCHK(hipMemsetAsync(h->ctrl + P_XACTIVE, 0, sizeof(u64), h->stream));
      k_xround<MODE_LABEL, true><<<sgrid, 256, 0, h->stream>>>(WJ, (u32)nslots, h->ctrl, wdone.as<u8>());
      ++rounds;
      KCHK();
      u64 active = 0;
      HIPCHK(hipMemcpyAsync(&active, h->ctrl + P_XACTIVE, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      done = active == 0;
      batch = 2;
    }
    a.xres = WJ;
    return PFD_OK;
  }
  // final tile pass over the tile rows [tr0, tr0 + ntrs) into `out` = the rows [row0, row0 + nrows) of the block
  int final_rows(pfd_raster *h, u32 tr0, u32 ntrs, u32 *out, u32 row0, u32 nrows, const u32 *ids32, const u32 *halo32) {
    PathArgs f = a;
    f.tr_off = tr0, f.out = out, f.out_row0 = row0, f.out_nrows = nrows, f.ids32 = ids32, f.halo32 = halo32;
    k_path<MODE_LABEL, true><<<dim3(ntc, ntrs), 256, 0, h->stream>>>(f);
    KCHK();
    return PFD_OK;
  }
  int unsaturated(pfd_raster *h, bool *bad) {  // (synchronises)
    u64 c[6];
    HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *bad = !done || c[P_UNSAT - 8] != 0;
    return PFD_OK;
  }
};
struct BasinsPending {
  DevBuf num, ids, brows;  // num: only for labels that are not 32 bits wide; brows: the numbers of the two boundary tile rows
  LabelRun run;
  OutArg out;
  u32 k = 0;
  int id_size = 0;
};
void pfd_free_pending_basins(pfd_raster *h) {
  delete (BasinsPending *)h->pending_basins;
  h->pending_basins = nullptr;
}
__global__ void k_seed_block(const i64 *__restrict__ idx, u32 k, u32 row_off, u32 ncol, u32 *__restrict__ seed) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) seed[(size_t)idx[t] + (size_t)row_off * ncol] = t + 1;
}
__global__ void k_seed_halo(const u8 *__restrict__ ncode, u32 ncol, u32 halo_top, u32 halo_bot, u32 nrow,
                            u32 *__restrict__ seed) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  const u32 side = t / ncol, col = t % ncol;
  if ((side == 0 && !halo_top) || (side == 1 && !halo_bot)) return;
  const size_t cell = (size_t)(side ? nrow - 1 : 0) * ncol + col;
  if (ncode[cell] == D8_HALO) seed[cell] = BTAG | (side << 30) | col;
}
// sparse seeding of a row block (see k_flag_seed_tiles): the tiles that hold an outlet (indices relative to the own rows)
__global__ void k_flag_seed_tiles_block(const i64 *__restrict__ idx, u32 k, u32 row_off, u32 ncol, u32 ntc, u8 *__restrict__ tflag) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const u32 x = (u32)idx[t], r = x / ncol + row_off, c = x % ncol;
  tflag[(size_t)(r >> 6) * ntc + (c >> 6)] = 1;
}
template <class L>
__device__ __forceinline__ u64 label_bits(const L *ids, u32 num) { return num ? (u64)ids[num - 1] : 0ull; }
// (num_first / num_last: the numbers of the first / last OWN row, wherever the caller keeps them)
template <class L>
__global__ void k_basin_record(const u8 *__restrict__ ncode, const u32 *__restrict__ num_first, const u32 *__restrict__ num_last,
                               const L *__restrict__ ids, u32 ncol, u32 halo_top, u32 own_rows, u32 *__restrict__ rec) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncol) return;
  const u32 side = t / ncol, col = t % ncol;
  const size_t cell = (size_t)(halo_top + (side ? own_rows - 1 : 0)) * ncol + col;
  u32 kind = 0;
  u64 val = 0;
  if (ncode[cell] != D8_MV) {
    const u32 v = side ? num_last[col] : num_first[col];
    if (v & BTAG) {
      kind = 1;
      val = v & ~BTAG;
    } else {
      val = label_bits(ids, v);
    }
  }
  rec[t] = kind;
  rec[2 * ncol + t] = (u32)val;
  rec[4 * ncol + t] = (u32)(val >> 32);
}
template <class L>
__global__ void k_labels_out_block(const u32 *__restrict__ num, const L *__restrict__ ids,
                                   const u64 *__restrict__ halo_label, u32 ncol, u32 n_own, L *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  const u32 v = num[i];
  L r = 0;
  if (v & BTAG)
    r = (L)halo_label[((v >> 30) & 1u) * ncol + (v & 0x3FFFFFFFu)];
  else if (v)
    r = ids[v - 1];
  out[i] = r;
}

extern "C" int pfd_basins_begin(pfd_raster *h, const int64_t *outlets, const void *ids, int64_t k, int id_size,
                                void *out, int memspace, uint32_t *record_host) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "the multi-block basins query"));
  if (!out || !record_host || k < 0 || (k > 0 && (!outlets || !ids)) ||
      (id_size != 1 && id_size != 2 && id_size != 4 && id_size != 8) || k >= 0x7FFFFFFFll) {
    pfd_set_error("pfd_basins_begin: bad arguments (k=%lld, id_size=%d)", (long long)k, id_size);
    return PFD_EINVAL;
  }
  if ((unsigned __int128)h->nrow * (unsigned __int128)h->ncol > 4294967294ull || (u64)h->ncol >= 0x40000000ull) {
    pfd_set_error("pfd_basins_begin: the block is too large (32-bit cell indices)");
    return PFD_EUNSUPPORTED;
  }
  pfd_free_pending_basins(h);
  pfd_seg_clear(h);
  const u32 ncol = (u32)h->ncol, n = h->geo.n;
  const i64 n_own = h->own_rows * h->ncol;
  // numpy's `basins[idxs] = ids` keeps the LAST id of a repeated index
  std::vector<i64> uidx;
  std::vector<unsigned char> uids;
  {
    std::unordered_set<i64> seen;
    for (i64 j = k - 1; j >= 0; --j) {
      const i64 i = outlets[j];
      if (i < 0 || i >= n_own) {
        pfd_set_error("pfd_basins_begin: outlet index %lld outside the block's own rows", (long long)i);
        return PFD_EINVAL;
      }
      if (!seen.insert(i).second) continue;
      uidx.push_back(i);
      const unsigned char *src = (const unsigned char *)ids + (size_t)j * id_size;
      uids.insert(uids.end(), src, src + id_size);
    }
  }
  const u32 ku = (u32)uidx.size();
  BasinsPending *p = new BasinsPending();
  h->pending_basins = p;
  p->k = ku;
  p->id_size = id_size;
  LabelRun &run = p->run;
  const u32 ntr = cdiv_u32((u64)h->nrow, TS), ntc = cdiv_u32((u64)h->ncol, TS);
  InArg di;
  int rc = di.bind(ku ? uidx.data() : nullptr, (size_t)ku * sizeof(i64), PFD_HOST, h->stream);
  if (rc == PFD_OK) rc = p->ids.alloc(std::max<size_t>((size_t)ku * id_size, 8));
  if (rc == PFD_OK && ku &&
      hipMemcpyAsync(p->ids.p, uids.data(), (size_t)ku * id_size, hipMemcpyHostToDevice, h->stream) != hipSuccess)
    rc = PFD_EHIP;
  if (rc == PFD_OK) rc = p->out.bind(out, (size_t)n_own * id_size, memspace);
  if (rc == PFD_OK) rc = run.seed.alloc((size_t)n * sizeof(u32) + 64);  // (+ slack: quads are loaded 16 bytes at a time)
  if (rc == PFD_OK) rc = run.tflag.alloc((size_t)ntr * ntc);
  if (rc == PFD_OK && id_size != 4) rc = p->num.alloc((size_t)n * sizeof(u32));  // (32-bit labels are written by the tile pass itself)
  if (rc == PFD_OK) rc = p->brows.alloc(2 * (size_t)TS * ncol * sizeof(u32));
  if (rc != PFD_OK) {
    pfd_free_pending_basins(h);
    return rc;
  }
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(h->stream);
    pfd_free_pending_basins(h);
    return code;
  };
  if (h->acyclic == 0) {  // (see pfd_basins_tiled: the label query must not hide a cycle)
    DevBuf rk;
    if ((rc = rk.alloc((size_t)n * sizeof(u32))) != PFD_OK) return fail(rc);
    int ok_rank = 0;
    if ((rc = run_paths<MODE_RANK>(h, nullptr, rk.as<u32>(), &ok_rank, nullptr)) != PFD_OK) return fail(rc);
    h->acyclic = ok_rank ? 1 : -1;
  }
  if (h->acyclic < 0) {
    pfd_set_error("the block holds cells that never reach a pit (cycles); the multi-block basins path requires a "
                  "valid flow direction raster (FlwdirRaster.isvalid)");
    return fail(PFD_EUNSUPPORTED);
  }
  // seeds: the outlets' numbers and the halo cells' tags, in the tiles that hold one — the other tiles never read theirs
  u8 *tflag = run.tflag.as<u8>();
  if (hipMemsetAsync(tflag, 0, (size_t)ntr * ntc, h->stream) != hipSuccess) return fail(PFD_EHIP);
  if (ku) k_flag_seed_tiles_block<<<cdiv_u32(ku, 256), 256, 0, h->stream>>>((const i64 *)di.dev, ku, (u32)h->halo_top, ncol, ntc, tflag);
  if (h->halo_top && hipMemsetAsync(tflag, 1, ntc, h->stream) != hipSuccess) return fail(PFD_EHIP);
  if (h->halo_bot && hipMemsetAsync(tflag + (size_t)(ntr - 1) * ntc, 1, ntc, h->stream) != hipSuccess) return fail(PFD_EHIP);
  k_zero_seed_tiles<<<dim3(ntc, ntr), 256, 0, h->stream>>>(tflag, (u32)h->nrow, ncol, ntc, run.seed.as<u32>());
  if (ku) k_seed_block<<<cdiv_u32(ku, 256), 256, 0, h->stream>>>((const i64 *)di.dev, ku, (u32)h->halo_top, ncol, run.seed.as<u32>());
  k_seed_halo<<<cdiv_u32(2 * ncol, 256), 256, 0, h->stream>>>(h->ncode, ncol, (u32)h->halo_top, (u32)h->halo_bot,
                                                              (u32)h->nrow, run.seed.as<u32>());
  if (hipGetLastError() != hipSuccess) return fail(PFD_EHIP);
  pfd_seg_begin(h, "tile_labels");
  if ((rc = run.start(h)) != PFD_OK) return fail(rc);
  // the numbers of the two boundary rows: the final pass on the tile rows that hold them, into a window of TS rows each
  const u32 rf = (u32)h->halo_top, rl = (u32)(h->halo_top + h->own_rows - 1);
  const u32 trf = rf / TS, trl = rl / TS;
  u32 *bw = p->brows.as<u32>();
  if ((rc = run.final_rows(h, trf, 1, bw, trf * TS, TS, nullptr, nullptr)) != PFD_OK) return fail(rc);
  if ((rc = run.final_rows(h, trl, 1, bw + (size_t)TS * ncol, trl * TS, TS, nullptr, nullptr)) != PFD_OK) return fail(rc);
  pfd_seg_end(h, 4);
  if (!run.done) {
    pfd_set_error("pfd_basins_begin: the label query did not converge (cycles)");
    return fail(PFD_EUNSUPPORTED);
  }
  DevBuf rec;
  if ((rc = rec.alloc(6 * (size_t)ncol * sizeof(u32))) != PFD_OK) return fail(rc);
  const u32 g = cdiv_u32(2 * ncol, 256);
  const u32 *nf = bw + (size_t)(rf - trf * TS) * ncol, *nl = bw + (size_t)TS * ncol + (size_t)(rl - trl * TS) * ncol;
  switch (id_size) {
    case 1: k_basin_record<u8><<<g, 256, 0, h->stream>>>(h->ncode, nf, nl, p->ids.as<u8>(), ncol, (u32)h->halo_top, (u32)h->own_rows, rec.as<u32>()); break;
    case 2: k_basin_record<uint16_t><<<g, 256, 0, h->stream>>>(h->ncode, nf, nl, p->ids.as<uint16_t>(), ncol, (u32)h->halo_top, (u32)h->own_rows, rec.as<u32>()); break;
    case 4: k_basin_record<u32><<<g, 256, 0, h->stream>>>(h->ncode, nf, nl, p->ids.as<u32>(), ncol, (u32)h->halo_top, (u32)h->own_rows, rec.as<u32>()); break;
    default: k_basin_record<u64><<<g, 256, 0, h->stream>>>(h->ncode, nf, nl, p->ids.as<u64>(), ncol, (u32)h->halo_top, (u32)h->own_rows, rec.as<u32>()); break;
  }
  if (hipMemcpyAsync(record_host, rec.p, 6 * (size_t)ncol * sizeof(u32), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return fail(PFD_EHIP);
  return PFD_OK;
}

extern "C" int pfd_basins_finish(pfd_raster *h, const uint32_t *all_records_host, int nblocks, int block, int *complete) {
  PFDCHK(pfd_check_handle_lazy(h));
  BasinsPending *p = (BasinsPending *)h->pending_basins;
  if (!p || !all_records_host || nblocks < 1 || block < 0 || block >= nblocks || !complete) {
    pfd_set_error("pfd_basins_finish: no query in flight on this handle, or bad arguments");
    return PFD_EINVAL;
  }
  *complete = 1;
  const u32 ncol = (u32)h->ncol;
  const size_t recw = 6 * (size_t)ncol, nn = (size_t)nblocks * 2 * ncol;
  // interface nodes: (blk, side, col) = the cell `col` of the first (side 0) / last (side 1) own row of blk
  auto kind = [&](size_t nd) { return all_records_host[(nd / (2 * (size_t)ncol)) * recw + nd % (2 * (size_t)ncol)]; };
  auto value = [&](size_t nd) {
    const size_t b = nd / (2 * (size_t)ncol), t = nd % (2 * (size_t)ncol);
    return (u64)all_records_host[b * recw + 2 * ncol + t] | ((u64)all_records_host[b * recw + 4 * ncol + t] << 32);
  };
  auto next = [&](size_t nd) -> i64 {  // the cell the path continues in: the halo cell named by the tag
    const i64 b = (i64)(nd / (2 * (size_t)ncol));
    const u64 v = value(nd);
    const u32 hs = (u32)(v >> 30) & 1u, hc = (u32)v & 0x3FFFFFFFu;
    const i64 nb = b + (hs ? 1 : -1);
    if (nb < 0 || nb >= nblocks || hc >= ncol) return -1;  // (hc comes from a peer's record: never index with it unchecked)
    return (nb * 2 + (1 - (i64)hs)) * (i64)ncol + hc;
  };
  std::vector<u64> lab(nn, 0);
  std::vector<unsigned char> state(nn, 0);  // 0 unknown, 1 on the current walk, 2 final
  std::vector<size_t> walk;
  for (size_t s0 = 0; s0 < nn; ++s0) {
    if (state[s0] == 2) continue;
    walk.clear();
    size_t cur = s0;
    u64 res = 0;
    for (;;) {
      if (state[cur] == 2) {
        res = lab[cur];
        break;
      }
      if (state[cur] == 1) {  // a cycle through several blocks
        *complete = 0;
        res = 0;
        break;
      }
      if (kind(cur) == 0) {
        res = value(cur);
        state[cur] = 2;
        lab[cur] = res;
        break;
      }
      state[cur] = 1;
      walk.push_back(cur);
      const i64 nx = next(cur);
      if (nx < 0) {  // (cannot happen: a halo row only exists towards an existing neighbour)
        res = 0;
        break;
      }
      cur = (size_t)nx;
    }
    for (size_t w : walk) {
      lab[w] = res;
      state[w] = 2;
    }
  }
  // labels of this block's halo cells: halo (side, c) = boundary cell (block -/+ 1, other side, c)
  std::vector<u64> halo((size_t)2 * ncol, 0);
  for (u32 side = 0; side < 2; ++side) {
    const i64 nb = (i64)block + (side ? 1 : -1);
    if (nb < 0 || nb >= nblocks) continue;
    for (u32 c = 0; c < ncol; ++c) halo[(size_t)side * ncol + c] = lab[(size_t)(nb * 2 + (1 - (i64)side)) * ncol + c];
  }
  // the final tile pass, now that every path end has a label: 32-bit labels are written by the pass itself (numbers and
  // tags through the two tables), other widths through the numbers of every cell and one mapping pass
  int rc = PFD_OK;
  const u32 ntr = p->run.ntr;
  bool bad = false;
  if (p->id_size == 4) {
    std::vector<u32> halo32(halo.size());
    for (size_t i = 0; i < halo.size(); ++i) halo32[i] = (u32)halo[i];
    InArg hl;
    rc = hl.bind(halo32.data(), halo32.size() * sizeof(u32), PFD_HOST, h->stream);
    pfd_seg_begin(h, "tile_labels_final");
    if (rc == PFD_OK)
      rc = p->run.final_rows(h, 0, ntr, (u32 *)p->out.dev, (u32)h->halo_top, (u32)h->own_rows, p->ids.as<u32>(), (const u32 *)hl.dev);
    pfd_seg_end(h, 1);
    if (rc == PFD_OK) rc = p->run.unsaturated(h, &bad);  // (synchronises: hl is released on return)
  } else {
    InArg hl;
    rc = hl.bind(halo.data(), halo.size() * sizeof(u64), PFD_HOST, h->stream);
    pfd_seg_begin(h, "tile_labels_final");
    if (rc == PFD_OK) rc = p->run.final_rows(h, 0, ntr, p->num.as<u32>(), 0, (u32)h->nrow, nullptr, nullptr);
    pfd_seg_end(h, 1);
    if (rc == PFD_OK) {
      const u32 n_own = (u32)(h->own_rows * h->ncol);
      const u32 *num = p->num.as<u32>() + (size_t)h->halo_top * ncol;
      const u32 g = cdiv_u32(n_own, 256);
      pfd_seg_begin(h, "labels_out");
      switch (p->id_size) {
        case 1: k_labels_out_block<u8><<<g, 256, 0, h->stream>>>(num, p->ids.as<u8>(), (const u64 *)hl.dev, ncol, n_own, (u8 *)p->out.dev); break;
        case 2: k_labels_out_block<uint16_t><<<g, 256, 0, h->stream>>>(num, p->ids.as<uint16_t>(), (const u64 *)hl.dev, ncol, n_own, (uint16_t *)p->out.dev); break;
        default: k_labels_out_block<u64><<<g, 256, 0, h->stream>>>(num, p->ids.as<u64>(), (const u64 *)hl.dev, ncol, n_own, (u64 *)p->out.dev); break;
      }
      pfd_seg_end(h, 1);
      if (hipGetLastError() != hipSuccess) rc = PFD_EHIP;
    }
    if (rc == PFD_OK) rc = p->run.unsaturated(h, &bad);
  }
  if (rc == PFD_OK && bad) *complete = 0;
  if (rc == PFD_OK) rc = p->out.finish(h->stream);
  else (void)hipStreamSynchronize(h->stream);
  pfd_free_pending_basins(h);
  return rc;
}

// path queries over a derived code raster (same shape as the handle's), for exact.hip
int pfd_path_rank(pfd_raster *h, const u8 *codes, u32 *out_dev, int *complete) {
  return run_paths<MODE_RANK>(h, nullptr, out_dev, complete, nullptr, codes);
}
// ranks of the handle's own raster + the largest of them (order64.hip: any raster size)
int pfd_path_rank_max(pfd_raster *h, u32 *out_dev, int *complete, u32 *maxrank) {
  return run_paths<MODE_RANK>(h, nullptr, out_dev, complete, maxrank);
}
// hops to the end of the path AND the end itself (linear index + 1 of the pit; 0 on cells that are no path cells) in one query
int pfd_path_rank_tails(pfd_raster *h, const u8 *codes, u32 *hops_dev, u32 *tails_dev, int *complete) {
  return run_paths<MODE_RANK>(h, nullptr, hops_dev, complete, nullptr, codes, nullptr, nullptr, tails_dev);
}
int pfd_path_labels(pfd_raster *h, const u8 *codes, const u32 *seed_dev, u32 *out_dev, int *complete) {
  return run_paths<MODE_LABEL>(h, seed_dev, out_dev, complete, nullptr, codes);
}

// ---------------------------------------------------------------------------------------------
// cell ordering from ranks: keys = rank, stable radix sort of the cell indices by key => cells
// grouped by rank, ascending index inside a rank (level 0 = the pits in ascending order)
// ---------------------------------------------------------------------------------------------
__global__ void k_iota(u32 *__restrict__ v, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ void k_clamp_keys(u32 *__restrict__ k, u32 n, u32 inval) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && k[i] > inval) k[i] = inval;
}
__global__ void k_level_offsets(const u32 *__restrict__ keys, u32 nseq, i64 *__restrict__ lvl_off) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseq) return;
  const u32 k = keys[i];
  if (i == 0 || keys[i - 1] != k) lvl_off[k] = (i64)i;  // ranks are contiguous 0..max
}

// returns *ok = 1 if the level structure (h->seq, h->lvl_off) was built this way
int pfd_order_cells_by_rank(pfd_raster *h, int *ok) {
  *ok = 0;
  const u32 n = h->geo.n;
  DevBuf keys, keys2;
  PFDCHK(keys.alloc((size_t)n * sizeof(u32)));
  int complete = 0;
  u32 maxrank = 0;
  PFDCHK(run_paths<MODE_RANK>(h, nullptr, keys.as<u32>(), &complete, &maxrank));
  h->acyclic = complete ? 1 : -1;  // (too large for the slot ids also lands here: the level engine serves it)
  if (!complete) return PFD_OK;  // cycles (or raster too large for the slot ids): breadth-first build instead
  PFDCHK(keys2.alloc((size_t)n * sizeof(u32)));
  if (h->seq) {  // sorted values need n entries (nodata cells sort to the tail)
    pfd_dfree(h->seq);
    h->bytes_held -= (size_t)h->n_valid * sizeof(u32);
    h->seq = nullptr;
  }
  PFDCHK(pfd_dmalloc((void **)&h->seq, (size_t)n * sizeof(u32)));
  h->bytes_held += (size_t)n * sizeof(u32);
  // nodata keys (0xFFFFFFFF) become maxrank+1, so that sorting on the low `bits` bits is enough; the clamp and the
  // cell numbers ride the sort's first pass as iterators (two streaming kernels less: 1.9 of 24.6 ms at 30000^2)
  int bits = 1;
  while ((1ull << bits) <= (u64)maxrank + 1) ++bits;
  const u32 inval = maxrank + 1;
  auto keys_in = rocprim::make_transform_iterator(keys.as<u32>(), [inval] __device__(u32 k) { return k > inval ? inval : k; });
  rocprim::counting_iterator<u32> cells(0u);
  size_t tmp_bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys2.as<u32>(), cells, h->seq, (size_t)n, 0u, (unsigned)bits,
                                   h->stream));
  DevBuf tmp;
  PFDCHK(tmp.alloc(tmp_bytes));
  HIPCHK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in, keys2.as<u32>(), cells, h->seq, (size_t)n, 0u, (unsigned)bits,
                                   h->stream));
  const i64 nlev = (i64)maxrank + 1;
  DevBuf lo;
  PFDCHK(lo.alloc((size_t)(nlev + 1) * sizeof(i64)));
  k_level_offsets<<<cdiv_u32((u64)h->n_valid, 256), 256, 0, h->stream>>>(keys2.as<u32>(), (u32)h->n_valid, lo.as<i64>());
  KCHK();
  h->lvl_off.assign((size_t)nlev + 1, 0);
  HIPCHK(hipMemcpyAsync(h->lvl_off.data(), lo.p, (size_t)nlev * sizeof(i64), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->lvl_off[nlev] = h->n_valid;
  h->n_levels = nlev;
  h->n_seq = h->n_valid;
  h->ordered = true;
  h->aux_ready = false;
  *ok = 1;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// basins through the tiled label query; *ok = 0 -> caller uses the level engine
// ---------------------------------------------------------------------------------------------
__global__ void k_seed_numbers(const i64 *__restrict__ idx, u32 k, u32 *__restrict__ seed) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) seed[idx[t]] = t + 1;
}
// sparse seeding: flag the 64 x 64 tiles that hold an outlet, zero the seeds of those tiles only
__global__ void k_flag_seed_tiles(const i64 *__restrict__ idx, u32 k, u32 ncol, u32 ntc, u8 *__restrict__ tflag) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const u64 x = (u64)idx[t];
  const u32 r = (u32)(x / ncol), c = (u32)(x - (u64)r * ncol);
  tflag[(size_t)(r >> 6) * ntc + (c >> 6)] = 1;
}
__global__ void __launch_bounds__(256) k_zero_seed_tiles(const u8 *__restrict__ tflag, u32 nrow, u32 ncol, u32 ntc,
                                                         u32 *__restrict__ seed) {
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_, tr = by_;
  if (!tflag[(size_t)tr * ntc + tc]) return;
  for (u32 i = threadIdx.x; i < TS * TS / 4; i += 256u) {  // quads of the tile
    const u32 r = tr * TS + (i >> 4), c = tc * TS + 4u * (i & 15u);
    if (r >= nrow) break;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (c + b < ncol) seed[(size_t)r * ncol + c + b] = 0;
  }
}
template <class L>
__global__ void k_labels_out(const u32 *__restrict__ num, const L *__restrict__ ids, u64 n, L *__restrict__ out) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {  // (n may exceed 2^32)
    const u32 v = num[i];
    out[i] = v ? ids[v - 1] : (L)0;
  }
}

int pfd_basins_tiled(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 k, int id_size, void *out_dev, int *ok) {
  *ok = 0;
  if (h->halo_top || h->halo_bot) return PFD_OK;
  const size_t n = (size_t)h->n;  // (any size: the query addresses tiles and slots, the labels are 32-bit VALUES)
  const bool wide = pfd_wide_cells(h);
  DevBuf seed, num;
  PFDCHK(seed.alloc((size_t)n * sizeof(u32) + 64));  // + slack: quads are loaded 16 bytes at a time
  const bool direct = id_size == 4;  // 32-bit labels: the final tile pass writes them itself
  if (!direct) PFDCHK(num.alloc((size_t)n * sizeof(u32)));
  // The label query stops at the first outlet, so an outlet on (or downstream of) a cycle would hide the
  // cycle and label cells that never reach a pit — cells the reference never visits (they are not in
  // idxs_seq; found by the randomised stress test).  The tiled path is only taken on rasters known to
  // be acyclic; the rank query answers that once per handle.
  if (h->acyclic == 0) {
    DevBuf rk;
    if (!wide) PFDCHK(rk.alloc((size_t)n * sizeof(u32)));  // (beyond 2^32 cells: the form without the output array)
    int ok_rank = 0;
    u32 mr = 0;
    pfd_seg_begin(h, "tile_rank_check");
    PFDCHK(run_paths<MODE_RANK>(h, nullptr, wide ? nullptr : rk.as<u32>(), &ok_rank, wide ? &mr : nullptr));
    pfd_seg_end(h, 2);
    h->acyclic = ok_rank ? 1 : -1;
  }
  if (h->acyclic < 0) return PFD_OK;
  const u32 ptr_ = cdiv_u32((u64)h->nrow, TS), ptc_ = cdiv_u32((u64)h->ncol, TS);
  DevBuf tflag;
  PFDCHK(tflag.alloc((size_t)ptr_ * ptc_));
  HIPCHK(hipMemsetAsync(tflag.p, 0, (size_t)ptr_ * ptc_, h->stream));
  if (k) {
    k_flag_seed_tiles<<<cdiv_u32(k, 256), 256, 0, h->stream>>>(idx_dev, k, (u32)h->ncol, ptc_, tflag.as<u8>());
    k_zero_seed_tiles<<<dim3(ptc_, ptr_), 256, 0, h->stream>>>(tflag.as<u8>(), (u32)h->nrow, (u32)h->ncol, ptc_, seed.as<u32>());
    k_seed_numbers<<<cdiv_u32(k, 256), 256, 0, h->stream>>>(idx_dev, k, seed.as<u32>());
    KCHK();
  }
  int complete = 0;
  pfd_seg_begin(h, "tile_labels");
  PFDCHK(run_paths<MODE_LABEL>(h, seed.as<u32>(), direct ? (u32 *)out_dev : num.as<u32>(), &complete, nullptr, nullptr,
                               (direct && k) ? (const u32 *)ids_dev : nullptr, tflag.as<u8>()));
  pfd_seg_end(h, 2);
  if (!complete) return PFD_OK;
  if (direct) {
    *ok = 1;
    return PFD_OK;
  }
  const u32 grid = (u32)std::min<u64>(cdiv_u32((u64)n, 256), 1u << 22);
  switch (id_size) {
    case 1: k_labels_out<u8><<<grid, 256, 0, h->stream>>>(num.as<u32>(), (const u8 *)ids_dev, n, (u8 *)out_dev); break;
    case 2: k_labels_out<uint16_t><<<grid, 256, 0, h->stream>>>(num.as<u32>(), (const uint16_t *)ids_dev, n, (uint16_t *)out_dev); break;
    case 4: k_labels_out<u32><<<grid, 256, 0, h->stream>>>(num.as<u32>(), (const u32 *)ids_dev, n, (u32 *)out_dev); break;
    default: k_labels_out<u64><<<grid, 256, 0, h->stream>>>(num.as<u32>(), (const u64 *)ids_dev, n, (u64 *)out_dev); break;
  }
  KCHK();
  *ok = 1;
  return PFD_OK;
}


// ---------------------------------------------------------------------------------------------
// graph statistics a benchmark has to report next to every number: longest flow path
// (max rank; the tiled rank query without its output array, so it also serves rasters beyond 2^32
// cells) and the in-degree histogram of the valid cells (reference core.upstream_count, core.py:50-61).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_indeg_hist(const u8 *__restrict__ ncode, u32 nrow, u32 ncol, u32 row_first,
                                                    u32 row_last, unsigned long long *__restrict__ hist) {
  __shared__ u32 s[9];
  if (threadIdx.x < 9) s[threadIdx.x] = 0;
  __syncthreads();
  const u32 c = blockIdx.x * 64 + (threadIdx.x & 63);
  u32 cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (u32 r = row_first + blockIdx.y * 4 + (threadIdx.x >> 6); r <= row_last && c < ncol; r += gridDim.y * 4) {
    const size_t i = (size_t)r * ncol + c;
    if (ncode[i] == D8_MV) continue;
    u32 d = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 rr = r + (u32)d8_dr(k), cc = c + (u32)d8_dc(k);
      if (rr < nrow && cc < ncol) d += ncode[(size_t)rr * ncol + cc] == (1u << ((k + 4) & 7)) ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) cnt[k] += d == (u32)k ? 1u : 0u;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    u32 v = cnt[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 9 && s[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

extern "C" int pfd_graph_stats(pfd_raster *h, int64_t stats[16]) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "graph_stats"));
  if (!stats) {
    pfd_set_error("pfd_graph_stats: NULL stats");
    return PFD_EINVAL;
  }
  for (int k = 0; k < 16; ++k) stats[k] = 0;
  DevBuf hist;
  PFDCHK(hist.alloc(9 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(hist.p, 0, 9 * sizeof(unsigned long long), h->stream));
  const dim3 grid(cdiv_u32((u64)h->ncol, 64), std::min<u32>(cdiv_u32((u64)h->own_rows, 4), 4096u));
  k_indeg_hist<<<grid, 256, 0, h->stream>>>(h->ncode, (u32)h->nrow, (u32)h->ncol, (u32)h->halo_top,
                                            (u32)(h->halo_top + h->own_rows - 1), hist.as<unsigned long long>());
  KCHK();
  unsigned long long hh[9];
  HIPCHK(hipMemcpyAsync(hh, hist.p, sizeof(hh), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  stats[0] = h->n_valid;
  stats[1] = h->n_pits;
  stats[2] = -1;
  for (int k = 0; k < 9; ++k) stats[3 + k] = (int64_t)hh[k];
  for (int k = 0; k < 4; ++k) stats[12 + k] = h->tile_rounds[k];
  if (!h->halo_top && !h->halo_bot) {  // longest flow path (cells): max rank over the raster
    // (-1: the raster holds cycles; -2: the raster is beyond the slot ids of the tiled query — nothing is known)
    const u32 ntr = cdiv_u32((u64)h->nrow, TS), ntc = cdiv_u32((u64)h->ncol, TS);
    const size_t nslots = (size_t)cdiv_u32(ntr, SG) * cdiv_u32(ntc, SG) * SSL;
    if (nslots >= 0x3FFFFFFFull || ntr > 65535u) {
      stats[2] = -2;
    } else {
      int complete = 0;
      u32 maxrank = 0;
      PFDCHK(run_paths<MODE_RANK>(h, nullptr, nullptr, &complete, &maxrank));
      if (complete) stats[2] = (int64_t)maxrank;
    }
  }
  return PFD_OK;
}
