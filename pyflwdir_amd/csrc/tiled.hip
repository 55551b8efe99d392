// tiled.hip — LDS-tiled fast path for FlwdirRaster.upstream_area(unit="cell")
// (reference pyflwdir/pyflwdir.py:770-801 = core.idxs_seq + streams.accuflux on unit weights).
//
// Integer accumulation is associative (int32 wrap-around included), so the serial ordering of
// the reference is not needed.  The raster is cut into 64x64-cell tiles; one 256-thread
// workgroup owns one tile and keeps its whole state in LDS (160 KB/CU on MI355X):
//
//   phase 1  k_tile<false>   per tile: dependency-driven up-sweep INSIDE the tile in LDS (every
//                            cell is visited once; a thread that delivers the last missing child
//                            of a cell carries on with that cell: one 64-bit LDS atomic per flow
//                            edge, no level barriers).  Emits, per perimeter slot, the local
//                            count of every cell that drains out of the tile ("exit"), the slot
//                            it drains into, and for every perimeter cell that receives flow
//                            from outside ("entry") the exit its in-tile path ends at ("link").
//   phase 2  k_coarse_link / k_coarse_chase   the exits form a forest ~30x smaller than the
//                            raster: exit e -> link(target(e)).  Same dependency-driven sweep
//                            with 64-bit global atomics gives the TOTAL count at every exit
//                            and, summed per target, the inflow at every entry.
//   phase 3  k_tile<true>    per tile: the in-LDS up-sweep again, entries now weighted
//                            1 + inflow; the finished tile is written to HBM once, coalesced.
//
// HBM traffic: 2 x 1 B/cell (codes) + 4 B/cell (result) + ~0.4 B/cell of perimeter records.
// Cells on or upstream of a cycle are never finalised; the run counts finalised cells and
// exits, and pfd_upstream_area_cell falls back to the level engine when a count is short.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

#define TS 64               // tile edge (cells)
#define TCELLS (TS * TS)    // 4096
#define HW (TS + 2)         // halo'd row pitch in LDS
#define PSL 256             // perimeter slots per tile (252 used)
#define NONE32 0xFFFFFFFFu

enum { T_PROC = 8, T_NEXITS = 9, T_XDONE = 10 };  // ctrl slots (u64)

struct TileArgs {
  const u8 *ncode;
  u32 nrow, ncol, ntr, ntc;
  u64 *xtot;      // [ntiles*PSL] coarse state: total<<32 | expected<<16 | arrived
  u64 *xrec;      // [ntiles*PSL] next exit on the path (slot) << 32 | slot the exit drains into;
                  //              low half NONE32 if the slot is no exit, high half NONE32 at a path end
  u32 *elink;     // [ntiles*PSL] perimeter slot (0..251) of the exit an entry's path reaches
  u32 *inflow;    // [ntiles*PSL] sum of the totals of the exits draining into this slot
  u64 *ctrl;
  i32 *out;
  int ablate;  // debugging/profiling knob (env PFD_TILE_ABLATE): bit0 skip sweep, bit1 skip scatter
};

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

#define TSTAMP(slot)                                                                            \
  if (a.ablate & 16) {                                                                          \
    __syncthreads();                                                                            \
    if (tid == 0) {                                                                             \
      const u64 t_ = __builtin_readcyclecounter();                                              \
      atomicAdd((unsigned long long *)&a.ctrl[(FINAL ? 40 : 24) + slot], (unsigned long long)(t_ - tprev)); \
      tprev = t_;                                                                               \
    }                                                                                           \
  }

template <bool FINAL>
__global__ void __launch_bounds__(256) k_tile(TileArgs a) {
  u64 tprev = __builtin_readcyclecounter();
  __shared__ u64 state[TCELLS];  // total<<32 | expected_children<<16 | arrived_children
  __shared__ u8 code[HW * HW];
  __shared__ u32 s_proc, s_exits, s_next;
  const u32 tid = threadIdx.x;
  const u32 tc = blockIdx.x, tr = blockIdx.y;
  const u32 tile = tr * a.ntc + tc;
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  if (tid == 0) s_proc = s_exits = s_next = 0;

  // ---- stage the tile's codes (+1-cell halo) in LDS --------------------------------------
  for (u32 idx = tid; idx < HW * HW; idx += 256) {
    const i64 gr = r0 + (i64)(idx / HW) - 1, gc = c0 + (i64)(idx % HW) - 1;
    u8 v = (u8)D8_MV;
    if (gr >= 0 && gc >= 0 && gr < (i64)a.nrow && gc < (i64)a.ncol) v = (a.ablate & 4) ? (u8)1 : a.ncode[(size_t)gr * a.ncol + (size_t)gc];
    code[idx] = v;
  }
  __syncthreads();
  TSTAMP(0)

  // ---- initial weights: 1 per valid cell (+ inflow from other tiles in the final pass) -----
  u32 anyvalid = 0;
#pragma unroll 4
  for (u32 j = 0; j < TCELLS / 256; ++j) {
    const u32 l = tid + 256 * j;
    const int lr = l >> 6, lc = l & 63;
    const u32 c = code[(lr + 1) * HW + lc + 1];
    u32 w = 0;
    if (c != D8_MV) {
      w = 1;
      anyvalid = 1;
      if (FINAL) {
        const int p = pslot(lr, lc);
        if (p >= 0) w += a.inflow[(size_t)tile * PSL + p];
      }
    }
    state[l] = (u64)w << 32;
  }
  if (!__syncthreads_or((int)anyvalid)) {  // all-nodata tile
    if (FINAL) {
      for (u32 j = 0; j < TCELLS / 256; ++j) {
        const u32 l = tid + 256 * j;
        const i64 gr = r0 + (l >> 6), gc = c0 + (l & 63);
        if (gr < (i64)a.nrow && gc < (i64)a.ncol) a.out[(size_t)gr * a.ncol + (size_t)gc] = -9999;
      }
    } else if (tid < PSL) {
      const size_t s = (size_t)tile * PSL + tid;
      a.xtot[s] = 0;
      a.xrec[s] = ~0ull;
      a.elink[s] = NONE32;
    }
    return;
  }

  TSTAMP(1)
  // ---- expected children per cell: every cell with an in-tile target bumps that target ------
#pragma unroll 4
  for (u32 j = 0; j < ((a.ablate & 2) ? 0u : TCELLS / 256); ++j) {
    const u32 l = tid + 256 * j;
    const int lr = l >> 6, lc = l & 63;
    const u32 c = code[(lr + 1) * HW + lc + 1];
    if (d8_is_dir(c)) {
      const int k = d8_slot(c);
      const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
      if ((unsigned)nr < TS && (unsigned)nc < TS) atomicAdd((unsigned long long *)&state[nr * TS + nc], 1ull << 16);
    }
  }
  __syncthreads();
  TSTAMP(2)

  // ---- dependency-driven up-sweep --------------------------------------------------------------
  // Lanes claim cells in chunks of 4 from a tile-wide counter; a claimed cell without in-tile
  // children starts a chain: deliver the finished total to the downstream cell with ONE 64-bit
  // LDS atomic and, if that was the last missing child, carry on with that cell.  A lane whose
  // chain stops (siblings pending, pit, tile edge) claims the next cell at once, so the wave's
  // time is the longest single chain plus its share of the tile, not the sum of per-cell maxima.
  u32 proc = 0, iters = 0;
  if (!(a.ablate & 1)) {
    u32 cbase = 0, cpos = 4, v = 0, c = 0;
    int lr = 0, lc = 0;
    bool active = false, nomore = false;
    for (;;) {
      if (!active && !nomore) {
        if (cpos == 4) {
          cbase = atomicAdd(&s_next, 4u);
          cpos = 0;
          if (cbase >= TCELLS) nomore = true;
        }
        if (!nomore) {
          const u32 l = cbase + cpos;
          ++cpos;
          const u32 cc = code[((l >> 6) + 1) * HW + (l & 63) + 1];
          if (cc != D8_MV) {
            const u64 s0 = state[l];
            if (((s0 >> 16) & 0xFFFFu) == 0) {  // no in-tile children: a chain starts here
              active = true;
              v = (u32)(s0 >> 32);
              lr = l >> 6;
              lc = l & 63;
              c = cc;
              ++proc;
            }
          }
        }
      }
      if (active) {
        active = false;
        if (d8_is_dir(c)) {
          const int k = d8_slot(c);
          lr += d8_dr(k);
          lc += d8_dc(k);
          if ((unsigned)lr < TS && (unsigned)lc < TS) {  // else: leaves the tile (an exit)
            const u32 cn = code[(lr + 1) * HW + lc + 1];
            const u64 old = atomicAdd((unsigned long long *)&state[lr * TS + lc], ((u64)v << 32) | 1ull);
            if (((old & 0xFFFFu) + 1) == ((old >> 16) & 0xFFFFu)) {  // last missing child
              v += (u32)(old >> 32);
              c = cn;
              active = true;
              ++proc;
            }
          }
        }
      }
      ++iters;
      if (!__any(active || !nomore)) break;
    }
  }
  if ((a.ablate & 16) && (tid & 63) == 0) {
    atomicAdd((unsigned long long *)&a.ctrl[(FINAL ? 40 : 24) + 8], (unsigned long long)iters);
    atomicMax((unsigned long long *)&a.ctrl[(FINAL ? 40 : 24) + 9], (unsigned long long)iters);
  }
  TSTAMP(3)
  // block-reduce the number of finalised cells
  for (int o = 32; o > 0; o >>= 1) proc += __shfl_down(proc, o);
  if ((tid & 63) == 0 && proc) atomicAdd(&s_proc, proc);
  __syncthreads();

  TSTAMP(4)
  if (a.ablate & 8) return;
  if (FINAL) {
    // ---- write the finished tile, one 256-B row segment per wave instruction ---------------
#pragma unroll 4
    for (u32 j = 0; j < TCELLS / 256; ++j) {
      const u32 l = tid + 256 * j;
      const int lr = l >> 6, lc = l & 63;
      const i64 gr = r0 + lr, gc = c0 + lc;
      if (gr < (i64)a.nrow && gc < (i64)a.ncol) {
        const u32 c = code[(lr + 1) * HW + lc + 1];
        a.out[(size_t)gr * a.ncol + (size_t)gc] = (c == D8_MV) ? -9999 : (i32)(u32)(state[l] >> 32);
      }
    }
    if (tid == 0 && s_proc) atomicAdd((unsigned long long *)&a.ctrl[T_PROC], (unsigned long long)s_proc);
    TSTAMP(5)
    return;
  }

  // ---- perimeter records for the coarse graph ------------------------------------------------
  if (tid < 2 * TS + 2 * (TS - 2)) {
    int lr, lc;
    pslot_inv((int)tid, &lr, &lc);
    const size_t slot = (size_t)tile * PSL + tid;
    const u32 c = code[(lr + 1) * HW + lc + 1];
    u64 xt = 0;
    u32 tgt = NONE32, link = NONE32;
    if (c != D8_MV) {
      // exit?
      if (d8_is_dir(c)) {
        const int k = d8_slot(c);
        const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
        if ((unsigned)nr >= TS || (unsigned)nc >= TS) {
          const i64 gr = r0 + nr, gc = c0 + nc;  // inside the raster and valid (normalised codes)
          const u32 ttile = (u32)(gr >> 6) * a.ntc + (u32)(gc >> 6);
          tgt = ttile * PSL + (u32)pslot((int)(gr & 63), (int)(gc & 63));
          xt = (state[lr * TS + lc] >> 32) << 32;
          atomicAdd(&s_exits, 1u);
        }
      }
      // entry?  (a neighbour outside the tile drains into this cell)
      bool entry = false;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
        if (((unsigned)nr >= TS || (unsigned)nc >= TS) && code[(nr + 1) * HW + nc + 1] == (1u << ((k + 4) & 7)))
          entry = true;
      }
      if (entry) {  // follow the in-tile path to its exit
        int wr = lr, wc = lc;
        u32 cc = c;
        for (int step = 0; step < TCELLS; ++step) {
          if (!d8_is_dir(cc)) break;  // ends in a pit inside the tile
          const int k = d8_slot(cc);
          const int nr = wr + d8_dr(k), nc = wc + d8_dc(k);
          if ((unsigned)nr >= TS || (unsigned)nc >= TS) {
            link = (u32)pslot(wr, wc);
            break;
          }
          wr = nr;
          wc = nc;
          cc = code[(wr + 1) * HW + wc + 1];
        }
      }
    }
    a.xtot[slot] = xt;
    a.xrec[slot] = ((u64)NONE32 << 32) | tgt;
    a.elink[slot] = link;
  } else if (tid < PSL) {
    const size_t slot = (size_t)tile * PSL + tid;
    a.xtot[slot] = 0;
    a.xrec[slot] = ~0ull;
    a.elink[slot] = NONE32;
  }
  __syncthreads();
  TSTAMP(5)
  if (tid == 0) {
    if (s_proc) atomicAdd((unsigned long long *)&a.ctrl[T_PROC], (unsigned long long)s_proc);
    if (s_exits) atomicAdd((unsigned long long *)&a.ctrl[T_NEXITS], (unsigned long long)s_exits);
  }
}

// exit e -> exit reached from the cell it drains into; count coarse children per exit
__global__ void __launch_bounds__(256) k_coarse_link(TileArgs a, u32 nslots) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const u32 tgt = (u32)a.xrec[s];
  if (tgt == NONE32) return;
  const u32 l = a.elink[tgt];
  if (l != NONE32) {
    const u32 nx = (tgt & ~(u32)(PSL - 1)) + l;
    atomicAdd((unsigned long long *)&a.xtot[nx], 1ull << 16);
    a.xrec[s] = ((u64)nx << 32) | tgt;
  }
}

// dependency-driven sweep over the exit forest; delivers every final total to the entry slot
// of the neighbouring tile (inflow) on the way.  One dependent memory round trip per hop: the
// record of the next exit is fetched while the returning atomic on its state is in flight.
__global__ void __launch_bounds__(256) k_coarse_chase(TileArgs a, u32 nslots) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  u32 done = 0;
  if (s < nslots) {
    u64 rec = a.xrec[s];
    if ((u32)rec != NONE32) {
      const u64 s0 = a.xtot[s];
      if (((s0 >> 16) & 0xFFFFu) == 0) {
        u32 v = (u32)(s0 >> 32);
        for (;;) {
          ++done;
          atomicAdd(&a.inflow[(u32)rec], v);
          const u32 nx = (u32)(rec >> 32);
          if (nx == NONE32) break;
          const u64 rec_nx = a.xrec[nx];
          const u64 old = atomicAdd((unsigned long long *)&a.xtot[nx], ((u64)v << 32) | 1ull);
          if (((old & 0xFFFFu) + 1) != ((old >> 16) & 0xFFFFu)) break;
          v += (u32)(old >> 32);
          rec = rec_nx;
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) done += __shfl_down(done, o);
  if ((threadIdx.x & 63) == 0 && done) atomicAdd((unsigned long long *)&a.ctrl[T_XDONE], (unsigned long long)done);
}

// returns PFD_OK and *complete = 1 when every valid cell was finalised (no cycles)
int pfd_upstream_area_cell_tiled(pfd_raster *h, i32 *out_dev, int *complete) {
  const u32 ntr = cdiv_u32((u64)h->nrow, TS), ntc = cdiv_u32((u64)h->ncol, TS);
  const size_t nslots = (size_t)ntr * ntc * PSL;
  if (nslots >= 0xFFFFFFFFull || ntr > 65535u) {
    *complete = 0;  // slot ids are 32 bit; such rasters go through the level engine
    return PFD_OK;
  }
  DevBuf xtot, xrec, elink, inflow;
  PFDCHK(xtot.alloc(nslots * sizeof(u64)));
  PFDCHK(xrec.alloc(nslots * sizeof(u64)));
  PFDCHK(elink.alloc(nslots * sizeof(u32)));
  PFDCHK(inflow.alloc(nslots * sizeof(u32)));
  TileArgs a{h->ncode, (u32)h->nrow, (u32)h->ncol, ntr, ntc, xtot.as<u64>(), xrec.as<u64>(),
             elink.as<u32>(), inflow.as<u32>(), h->ctrl, out_dev, 0};
  if (const char *e = getenv("PFD_TILE_ABLATE")) a.ablate = atoi(e);
  HIPCHK(hipMemsetAsync(h->ctrl + 8, 0, 56 * sizeof(u64), h->stream));
  HIPCHK(hipMemsetAsync(inflow.p, 0, nslots * sizeof(u32), h->stream));
  const dim3 grid(ntc, ntr);
  pfd_seg_begin(h, "tile_local");
  k_tile<false><<<grid, 256, 0, h->stream>>>(a);
  KCHK();
  pfd_seg_end(h, 1);
  pfd_seg_begin(h, "tile_exits");
  k_coarse_link<<<cdiv_u32(nslots, 256), 256, 0, h->stream>>>(a, (u32)nslots);
  k_coarse_chase<<<cdiv_u32(nslots, 256), 256, 0, h->stream>>>(a, (u32)nslots);
  KCHK();
  pfd_seg_end(h, 2);
  pfd_seg_begin(h, "tile_final");
  k_tile<true><<<grid, 256, 0, h->stream>>>(a);
  KCHK();
  pfd_seg_end(h, 1);
  u64 c[3];
  HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (a.ablate & 16) {
    u64 t[32];
    HIPCHK(hipMemcpy(t, h->ctrl + 24, sizeof(t), hipMemcpyDeviceToHost));
    const double nt = (double)ntr * ntc;
    for (int ph = 0; ph < 2; ++ph) {
      const u64 *q = t + 16 * ph;
      fprintf(stderr, "[k_tile<%d>] cycles/tile: load %.0f init %.0f scatter %.0f sweep %.0f red %.0f out %.0f | iters/wave avg %.1f max %llu\n",
              ph, q[0] / nt, q[1] / nt, q[2] / nt, q[3] / nt, q[4] / nt, q[5] / nt, q[8] / (nt * 4), (unsigned long long)q[9]);
    }
  }
  // T_PROC counted both tile passes
  *complete = (c[0] == 2ull * (u64)h->n_valid) && (c[1] == c[2]);
  return PFD_OK;
}
