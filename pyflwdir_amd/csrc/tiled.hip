// tiled.hip — LDS-tiled fast path for FlwdirRaster.upstream_area(unit="cell")
// (reference pyflwdir/pyflwdir.py:770-801 = core.idxs_seq + streams.accuflux on unit weights).
//
// Integer accumulation is associative (int32 wrap-around included), so the serial ordering of
// the reference is not needed.  The raster is cut into 64x64-cell tiles; one 256-thread
// workgroup owns one tile and keeps its whole state in LDS (160 KB/CU on MI355X, 5 tiles
// resident per CU).  Inside a tile — and again on the graph of tile exits — subtree sums are
// computed by POINTER DOUBLING instead of a dependent walk:
//
//     A_0(y) = w(y),  J_0(z) = downstream cell of z
//     round k:  A[J_k(z)] += A_k(z)  for every z whose 2^k-th ancestor J_k(z) exists,
//               J_{k+1}(z) = J_k(J_k(z))
//     => A_k(y) = sum of w over the upstream cells of y closer than 2^k   (exact, any order)
//
// log2(longest in-tile path) rounds, every lane busy, no dependent chains: the work per round
// is a handful of LDS reads, one non-returning ds_add_u32 and one ds_write_b16 per cell.
// A pointer that runs off the end of its path saturates at the path's last cell ("root": the
// cell where the flow leaves the tile, or a pit) and is flagged done; the roots of the
// perimeter cells are exactly the links the coarse graph needs.
//
//   phase 1  k_tile<false>   per tile: local counts -> per perimeter slot: the local count of
//                            every cell that drains out of the tile ("exit") and the slot it
//                            drains into; for every perimeter cell that receives flow from
//                            outside ("entry") the exit its in-tile path ends at ("link").
//   phase 2  k_coarse_*      the exits form a forest ~50x smaller than the raster:
//                            exit e -> link(target(e)).  Same doubling with global atomics
//                            (ping-pong buffers, one launch per round) gives the TOTAL count
//                            at every exit and, summed per target, the inflow at every entry.
//   phase 3  k_tile<true>    per tile: the doubling again with entries weighted 1 + inflow;
//                            the finished tile is written to HBM once, coalesced.
//
// HBM traffic: 2 x 1 B/cell (codes) + 4 B/cell (result) + ~1 B/cell of perimeter records.
// Cells on or upstream of a cycle never saturate; the run counts saturated cells and exits, and
// pfd_upstream_area_cell falls back to the level engine when a count is short.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

#define TS 64               // tile edge (cells)
#define TCELLS (TS * TS)    // 4096
#define HW (TS + 2)         // halo'd row pitch in LDS
#define PSL 256             // perimeter slots per tile (252 used)
#define NPERIM (2 * TS + 2 * (TS - 2))
#define NONE32 0xFFFFFFFFu
#define PDONE 0x8000u       // in-tile pointer saturated at its root
#define XDONE 0x80000000u   // coarse pointer saturated
#define MAXROUNDS_TILE 13   // 2^13 > 4096 cells: more rounds mean a cycle
#define CPT (TCELLS / 256)  // cells per thread

enum { T_PROC = 8, T_NEXITS = 9, T_XACTIVE = 10 };  // ctrl slots (u64)

struct TileArgs {
  const u8 *ncode;
  u32 nrow, ncol, ntr, ntc;
  u32 *xid;       // [nslots] dense id of the exit sitting on this perimeter slot, NONE32 if none
  u32 *eT;        // [nexits] local count of an exit (dense exit id)
  u32 *etgt;      // [nexits] global perimeter slot the exit drains into
  u32 *elink;     // [nslots] perimeter slot (0..251) of the exit an entry's path reaches
  u32 *inflow;    // [nslots] sum of the totals of the exits draining into this slot
  u64 *ctrl;
  i32 *out;
  int ablate;     // profiling knob (env PFD_TILE_ABLATE): bit0 skip doubling, bit4 cycle stamps
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

#define TSTAMP(slot)                                                                                        \
  if (a.ablate & 16) {                                                                                      \
    __syncthreads();                                                                                        \
    if (tid == 0) {                                                                                         \
      const u64 t_ = __builtin_readcyclecounter();                                                          \
      atomicAdd((unsigned long long *)&a.ctrl[(FINAL ? 40 : 24) + slot], (unsigned long long)(t_ - tprev)); \
      tprev = t_;                                                                                           \
    }                                                                                                       \
  }

template <bool FINAL>
__global__ void __launch_bounds__(256) k_tile(TileArgs a) {
  __shared__ u32 A[TCELLS];       // running subtree count of the cell
  __shared__ uint16_t P[TCELLS];  // 2^k-th ancestor (local index) | PDONE once saturated
  __shared__ u8 code[HW * HW];    // normalised codes with a 1-cell halo
  __shared__ u32 s_proc, s_exits, s_xbase;
  u64 tprev = __builtin_readcyclecounter();
  const u32 tid = threadIdx.x;
  const u32 tc = blockIdx.x, tr = blockIdx.y;
  const u32 tile = tr * a.ntc + tc;
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  if (tid == 0) s_proc = s_exits = 0;

  // ---- stage the tile's codes (+1-cell halo): all loads in flight before the first store -----
  {
    u8 v[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const u32 idx = tid + 256u * k;
      const i64 gr = r0 + (i64)(idx / HW) - 1, gc = c0 + (i64)(idx % HW) - 1;
      // unconditional load from a clamped address (a load inside a branch would be waited for
      // on the spot and the 18 loads would serialise), nodata selected afterwards
      const bool inside = idx < HW * HW && gr >= 0 && gc >= 0 && gr < (i64)a.nrow && gc < (i64)a.ncol;
      const i64 cr = gr < 0 ? 0 : (gr >= (i64)a.nrow ? (i64)a.nrow - 1 : gr);
      const i64 cc = gc < 0 ? 0 : (gc >= (i64)a.ncol ? (i64)a.ncol - 1 : gc);
      const u8 ld = a.ncode[(size_t)cr * a.ncol + (size_t)cc];
      v[k] = inside ? ld : (u8)D8_MV;
    }
    u32 inf = 0;
    if (FINAL) inf = a.inflow[(size_t)tile * PSL + tid];  // 256 slots per tile: always in bounds
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const u32 idx = tid + 256u * k;
      if (idx < HW * HW) code[idx] = v[k];
    }
    __syncthreads();
    TSTAMP(0)

    // ---- initial weights and downstream pointers -------------------------------------------
    // thread owns cells l = tid + 256*j (a wave = one 64-cell row segment: conflict-free LDS)
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const u32 l = tid + 256u * j;
      const int lr = l >> 6, lc = l & 63;
      const u32 c = code[(lr + 1) * HW + lc + 1];
      u32 p = l | PDONE;  // nodata, pit, or flow leaves the tile: the cell is its own root
      if (d8_is_dir(c)) {
        const int k = d8_slot(c);
        const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
        if ((unsigned)nr < TS && (unsigned)nc < TS) p = (u32)(nr * TS + nc);
      }
      A[l] = (c != D8_MV) ? 1u : 0u;
      P[l] = (uint16_t)p;
    }
    if (FINAL) {
      __syncthreads();
      if (tid < NPERIM && inf) {  // flow entering the tile from its neighbours
        int lr, lc;
        pslot_inv((int)tid, &lr, &lc);
        A[lr * TS + lc] += inf;
      }
    }
  }
  __syncthreads();
  TSTAMP(1)

  // ---- pointer doubling ------------------------------------------------------------------------
  u32 y[CPT];        // current 2^k-th ancestor of own cell j (valid while its bit in `live` is set)
  u32 live = 0;      // bit j: own cell j still has an unsaturated pointer
  u32 nvalid = 0;
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const u32 l = tid + 256u * j;
    const u32 p = P[l];
    y[j] = p & 0xFFFu;
    if (!(p & PDONE)) live |= 1u << j;
    nvalid += code[((l >> 6) + 1) * HW + (l & 63) + 1] != D8_MV;
  }
  if (!(a.ablate & 1)) {
    for (int round = 0; round < MAXROUNDS_TILE; ++round) {
      u32 av[CPT], q[CPT];
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        if (live & (1u << j)) {
          av[j] = A[tid + 256u * j];
          q[j] = P[y[j]];
        }
      }
      __syncthreads();  // every read of this round precedes every write of this round
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        if (live & (1u << j)) {
          atomicAdd(&A[y[j]], av[j]);
          P[tid + 256u * j] = (uint16_t)q[j];
          y[j] = q[j] & 0xFFFu;
          if (q[j] & PDONE) live &= ~(1u << j);
        }
      }
      if (!__syncthreads_or((int)live)) break;
    }
  }
  TSTAMP(2)
  // saturated valid cells (a cell on or upstream of a cycle never saturates)
  u32 proc = nvalid - (u32)__popc(live);
  for (int o = 32; o > 0; o >>= 1) proc += __shfl_down(proc, o);
  if ((tid & 63) == 0 && proc) atomicAdd(&s_proc, proc);
  __syncthreads();

  if (FINAL) {
    // ---- write the finished tile, one 256-B row segment per wave instruction ---------------
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const u32 l = tid + 256u * j;
      const int lr = l >> 6, lc = l & 63;
      const i64 gr = r0 + lr, gc = c0 + lc;
      if (gr < (i64)a.nrow && gc < (i64)a.ncol) {
        const u32 c = code[(lr + 1) * HW + lc + 1];
        a.out[(size_t)gr * a.ncol + (size_t)gc] = (c == D8_MV) ? -9999 : (i32)A[l];
      }
    }
    if (tid == 0 && s_proc) atomicAdd((unsigned long long *)&a.ctrl[T_PROC], (unsigned long long)s_proc);
    TSTAMP(3)
    return;
  }

  // ---- perimeter records for the coarse graph ------------------------------------------------
  u32 xt = 0, tgt = NONE32, link = NONE32, xrank = 0;
  if (tid < NPERIM) {
    int lr, lc;
    pslot_inv((int)tid, &lr, &lc);
    const u32 c = code[(lr + 1) * HW + lc + 1];
    if (c != D8_MV) {
      if (d8_is_dir(c)) {  // exit?
        const int k = d8_slot(c);
        const int nr = lr + d8_dr(k), nc = lc + d8_dc(k);
        if ((unsigned)nr >= TS || (unsigned)nc >= TS) {
          const i64 gr = r0 + nr, gc = c0 + nc;  // inside the raster and valid (normalised codes)
          const u32 ttile = (u32)(gr >> 6) * a.ntc + (u32)(gc >> 6);
          tgt = ttile * PSL + (u32)pslot((int)(gr & 63), (int)(gc & 63));
          xt = A[lr * TS + lc];
          xrank = atomicAdd(&s_exits, 1u);
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
      if (entry) {  // the root of the in-tile path is an exit or a pit
        const u32 root = P[lr * TS + lc] & 0xFFFu;
        const int rr = root >> 6, rc = root & 63;
        const u32 cr = code[(rr + 1) * HW + rc + 1];
        if (d8_is_dir(cr)) {
          const int k = d8_slot(cr);
          const int nr = rr + d8_dr(k), nc = rc + d8_dc(k);
          if ((unsigned)nr >= TS || (unsigned)nc >= TS) link = (u32)pslot(rr, rc);
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {  // reserve a dense id range for this tile's exits
    s_xbase = s_exits ? (u32)atomicAdd((unsigned long long *)&a.ctrl[T_NEXITS], (unsigned long long)s_exits) : 0u;
    if (s_proc) atomicAdd((unsigned long long *)&a.ctrl[T_PROC], (unsigned long long)s_proc);
  }
  __syncthreads();
  if (tid < PSL) {
    const size_t slot = (size_t)tile * PSL + tid;
    u32 id = NONE32;
    if (tgt != NONE32) {
      id = s_xbase + xrank;
      a.eT[id] = xt;
      a.etgt[id] = tgt;
    }
    a.xid[slot] = id;
    a.elink[slot] = link;
  }
  TSTAMP(3)
}

// ---------------------------------------------------------------------------------------------
// coarse graph: exit e -> exit reached from the cell it drains into.  Pointer doubling with
// ping-pong buffers (a launch is the round barrier).
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

__global__ void __launch_bounds__(256) k_coarse_link(const u32 *__restrict__ etgt, const u32 *__restrict__ elink,
                                                     const u32 *__restrict__ xid, u32 *__restrict__ J, u32 nexits,
                                                     u64 *ctrl) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nexits) return;
  const u32 tgt = etgt[e];
  const u32 l = elink[tgt];
  u32 j = e | XDONE;
  if (l != NONE32) j = xid[(tgt & ~(u32)(PSL - 1)) + l];
  J[e] = j;
  if (!(j & XDONE)) flag_active(ctrl);
}

// round prologue: Tnew = Told (the adds of the round go on top) and reset the activity flag
__global__ void __launch_bounds__(256) k_coarse_prep(const u32 *__restrict__ Told, u32 *__restrict__ Tnew, u32 nexits,
                                                     u64 *ctrl) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) ctrl[T_XACTIVE] = 0;
  if (e < nexits) Tnew[e] = Told[e];
}

__global__ void __launch_bounds__(256) k_coarse_round(const u32 *__restrict__ Told, u32 *__restrict__ Tnew,
                                                      const u32 *__restrict__ Jold, u32 *__restrict__ Jnew,
                                                      u32 nexits, u64 *ctrl) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nexits) return;
  const u32 j = Jold[e];
  if (j & XDONE) {
    Jnew[e] = j;
    return;
  }
  const u32 q = Jold[j];
  atomicAdd(&Tnew[j], Told[e]);
  Jnew[e] = q;
  if (!(q & XDONE)) flag_active(ctrl);
}

__global__ void __launch_bounds__(256) k_coarse_inflow(const u32 *__restrict__ etgt, const u32 *__restrict__ T,
                                                       u32 *__restrict__ inflow, u32 nexits) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nexits) atomicAdd(&inflow[etgt[e]], T[e]);
}

// returns PFD_OK and *complete = 1 when every valid cell was finalised (no cycles)
int pfd_upstream_area_cell_tiled(pfd_raster *h, i32 *out_dev, int *complete) {
  const u32 ntr = cdiv_u32((u64)h->nrow, TS), ntc = cdiv_u32((u64)h->ncol, TS);
  const size_t nslots = (size_t)ntr * ntc * PSL;
  *complete = 0;
  if (nslots >= 0x7FFFFFFFull || ntr > 65535u) return PFD_OK;  // slot ids are 31 bit: level engine
  const size_t xcap = (size_t)ntr * ntc * NPERIM;  // upper bound of the number of exits
  DevBuf T0, T1, J0, J1, etgt, xid, elink, inflow;
  PFDCHK(T0.alloc(xcap * sizeof(u32)));
  PFDCHK(T1.alloc(xcap * sizeof(u32)));
  PFDCHK(J0.alloc(xcap * sizeof(u32)));
  PFDCHK(J1.alloc(xcap * sizeof(u32)));
  PFDCHK(etgt.alloc(xcap * sizeof(u32)));
  PFDCHK(xid.alloc(nslots * sizeof(u32)));
  PFDCHK(elink.alloc(nslots * sizeof(u32)));
  PFDCHK(inflow.alloc(nslots * sizeof(u32)));
  TileArgs a{h->ncode, (u32)h->nrow, (u32)h->ncol, ntr, ntc, xid.as<u32>(), T0.as<u32>(), etgt.as<u32>(),
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
  u64 c[3];
  HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const u32 nexits = (u32)c[1];
  i64 launches = 0;
  bool coarse_done = true;
  u32 *Tc = T0.as<u32>(), *Tn = T1.as<u32>(), *Jc = J0.as<u32>(), *Jn = J1.as<u32>();
  if (nexits) {
    const u32 egrid = cdiv_u32(nexits, 256);
    k_coarse_link<<<egrid, 256, 0, h->stream>>>(etgt.as<u32>(), elink.as<u32>(), xid.as<u32>(), Jc, nexits, h->ctrl);
    ++launches;
    // rounds are idempotent once every pointer is saturated: issue them in batches and look at
    // the "still active" flag only between batches (one host round trip per batch)
    coarse_done = false;
    int batch = 1;
    for (u32 span = 1; span < ntr + ntc; span <<= 1) ++batch;  // ~log2 of a typical path (in tiles)
    for (int rounds = 0; rounds < 40 && !coarse_done;) {
      for (int b = 0; b < batch; ++b, ++rounds) {
        k_coarse_prep<<<egrid, 256, 0, h->stream>>>(Tc, Tn, nexits, h->ctrl);
        k_coarse_round<<<egrid, 256, 0, h->stream>>>(Tc, Tn, Jc, Jn, nexits, h->ctrl);
        launches += 2;
        std::swap(Tc, Tn);
        std::swap(Jc, Jn);
      }
      u64 active = 0;
      HIPCHK(hipMemcpyAsync(&active, h->ctrl + T_XACTIVE, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      coarse_done = active == 0;
      batch = 2;
    }
    k_coarse_inflow<<<egrid, 256, 0, h->stream>>>(etgt.as<u32>(), Tc, inflow.as<u32>(), nexits);
    ++launches;
    KCHK();
  }
  pfd_seg_end(h, launches);

  pfd_seg_begin(h, "tile_final");
  k_tile<true><<<grid, 256, 0, h->stream>>>(a);
  KCHK();
  pfd_seg_end(h, 1);
  HIPCHK(hipMemcpyAsync(c, h->ctrl + 8, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (a.ablate & 16) {
    u64 t[32];
    HIPCHK(hipMemcpy(t, h->ctrl + 24, sizeof(t), hipMemcpyDeviceToHost));
    const double nt = (double)ntr * ntc;
    for (int ph = 0; ph < 2; ++ph) {
      const u64 *q = t + 16 * ph;
      fprintf(stderr, "[k_tile<%d>] cycles/tile: load %.0f init %.0f doubling %.0f out %.0f\n", ph, q[0] / nt,
              q[1] / nt, q[2] / nt, q[3] / nt);
    }
  }
  // T_PROC counted both tile passes
  *complete = coarse_done && (c[0] == 2ull * (u64)h->n_valid);
  return PFD_OK;
}
