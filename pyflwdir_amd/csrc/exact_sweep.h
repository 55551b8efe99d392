// exact_sweep.h — sweep kernels of the exact-order engine (exact.h), templates over the operation.
// Included by sweeps.hip, where the operations (AccuUp, Strahler, Hand, ...) are defined.
//
// Up-sweeps (values flow downstream):   leaves in LDS per tile  ->  trunk bucket by bucket (pre / scan / scatter)
// Down-sweeps (values flow upstream):   trunk bucket by bucket, highest first  ->  leaves in LDS per tile
//
// What an operation provides for the up direction:
//   LV  tile_init(g, nodata)          value of a cell before any upstream cell is added (tile LDS image)
//   LV  tile_combine(l, kids, val)    value of leaf l from the LDS values of its upstream cells
//   void tile_store(g, LV)
//   Elem pre_real(x, kids, hs)        own payload + the light upstream cells the serial loop adds BEFORE the
//                                     heavy one (slot hs; 8 = none: every upstream cell), from final values
//   Elem pre_post(child)              a light upstream cell added AFTER the heavy one
//   V first(Elem) / V fold(V, Elem, bool post)   the serial fold along a chain;  void store(x, V)
// and for the down direction:
//   DElem dpre(x, code)               everything apply() reads from memory, gathered per trunk slot
//   V droot(DElem) / V dfold(DElem, V pv);  V top(p);  V apply(x, code, root, pv);  void store(x, V)
#pragma once
#include "exact.h"

struct XTileArgs {
  u32 nrow, ncol, ntc;
  const u8 *lh, *kids, *ncode;
  const uint16_t *tord, *toff;
};

// ---- leaves, up ---------------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(256) k_xtile_up(Op op, XTileArgs a) {
  typedef typename Op::LV LV;
  __shared__ LV val[XTC];
  __shared__ u8 K[XTC];
  __shared__ uint16_t ord[XTC];
  __shared__ uint16_t off[XOFF];
  const u32 tid = threadIdx.x;
  const u32 tc = blockIdx.x, tr = blockIdx.y;
  const size_t tile = (size_t)tr * a.ntc + tc;
  const u32 r0 = tr * XT, c0 = tc * XT;
  if (tid < XOFF) off[tid] = a.toff[tile * XOFF + tid];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const u32 l = tid + 256u * j;
    const u32 gr = r0 + (l >> 6), gc = c0 + (l & 63);
    LV v = LV();
    u32 k = 0;
    if (gr < a.nrow && gc < a.ncol) {
      const u32 g = gr * a.ncol + gc;
      k = a.kids[g];
      v = op.tile_init(g, a.lh[g] == XL_NODATA);
    }
    val[l] = v;
    K[l] = (u8)k;
    ord[l] = a.tord[tile * XTC + l];
  }
  __syncthreads();
  const u32 total = off[XOFF - 1];
  for (int s = 1; s < XOFF - 1; ++s) {  // step 0 = headwaters: their value is the initial one
    const u32 b = off[s], e = off[s + 1];
    if (b >= total) break;
    for (u32 j = b + tid; j < e; j += 256u) {
      const u32 x = ord[j];
      val[x] = op.tile_combine(x, (u32)K[x], val);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const u32 l = tid + 256u * j;
    const u32 gr = r0 + (l >> 6), gc = c0 + (l & 63);
    if (gr < a.nrow && gc < a.ncol) op.tile_store(gr * a.ncol + gc, val[l]);
  }
}

// ---- trunk, up ----------------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_pre(Op op, const u32 *__restrict__ scell,
                                                    const uint16_t *__restrict__ sinfo, u32 s0, u32 s1,
                                                    typename Op::Elem *__restrict__ E) {
  const u32 s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  const u32 info = sinfo[s];
  const u32 x = scell[s];
  E[s] = (info & XS_POST) ? op.pre_post(x) : op.pre_real(x, info & 0xFFu, (info >> 8) & 0xFu);
}

// one LANE per chain: the running value never leaves its register; the slots of a chain are contiguous,
// so the loads of the next XU slots are all in flight while the current ones are folded
#define XU 16
template <class Op>
__global__ void __launch_bounds__(64) k_xtrunk_scan(Op op, const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                    u32 c0, u32 c1, const uint16_t *__restrict__ sinfo,
                                                    const typename Op::Elem *__restrict__ E,
                                                    typename Op::V *__restrict__ R) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  const u32 c = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = c < c1;
  const u32 s0 = active ? cstart[c] : 0u;
  const u32 m = active ? clen[c] : 0u;
  V t = V();
  Elem cur[XU], nxt[XU];
  u32 curi[XU], nxti[XU];
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const u32 sl = s0 + ((u32)u < m ? (u32)u : 0u);
    cur[u] = E[sl];
    curi[u] = sinfo[sl];
  }
  for (u32 base = 0; __any((int)(base < m)); base += XU) {
    const u32 nb = base + XU;
#pragma unroll
    for (int u = 0; u < XU; ++u) {  // prefetch the next block (clamped address: always a valid slot of the chain)
      const u32 i = nb + (u32)u;
      const u32 sl = s0 + (i < m ? i : (m ? m - 1u : 0u));
      nxt[u] = E[sl];
      nxti[u] = sinfo[sl];
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const u32 i = base + (u32)u;
      if (i < m) {
        t = i == 0 ? op.first(cur[u]) : op.fold(t, cur[u], (curi[u] & XS_POST) != 0);
        R[s0 + i] = t;
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      cur[u] = nxt[u];
      curi[u] = nxti[u];
    }
  }
}

template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_scatter(Op op, const u32 *__restrict__ scell,
                                                        const uint16_t *__restrict__ sinfo, u32 s0, u32 s1,
                                                        const typename Op::V *__restrict__ R) {
  const u32 s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  const u32 info = sinfo[s];
  if (info & XS_POST) return;
  op.store(scell[s], R[s + ((info >> 12) & 7u)]);  // the cell's value = the running value after its last post slot
}

template <class Op>
static int run_exact_up(pfd_raster *h, const Op &op, const char *name) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  ExactPlan *p = (ExactPlan *)h->xplan;
  pfd_seg_begin(h, name);
  i64 launches = 1;
  XTileArgs a{(u32)h->nrow, (u32)h->ncol, p->ntc, p->lh, p->kids, h->ncode, p->tord, p->toff};
  k_xtile_up<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a);
  KCHK();
  DevBuf E, R;
  PFDCHK(E.alloc(std::max<size_t>((size_t)p->nslot, 1) * sizeof(Elem) + 64));
  PFDCHK(R.alloc(std::max<size_t>((size_t)p->nslot, 1) * sizeof(V) + 64));
  for (int b = 0; b < 32; ++b) {
    const u32 s0 = (u32)p->b_slot[b], s1 = (u32)p->b_slot[b + 1];
    const u32 c0 = (u32)p->b_chain[b], c1 = (u32)p->b_chain[b + 1];
    if (c1 == c0) continue;
    k_xtrunk_pre<Op><<<cdiv_u32(s1 - s0, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, s0, s1, E.as<Elem>());
    k_xtrunk_scan<Op><<<cdiv_u32(c1 - c0, 64), 64, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1, p->sinfo, E.as<Elem>(),
                                                                   R.as<V>());
    k_xtrunk_scatter<Op><<<cdiv_u32(s1 - s0, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, s0, s1, R.as<V>());
    launches += 3;
  }
  KCHK();
  pfd_seg_end(h, launches);
  HIPCHK(hipStreamSynchronize(h->stream));  // E / R are released on return
  return PFD_OK;
}

// ---- trunk, down --------------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_dpre(Op op, const u32 *__restrict__ scell,
                                                     const uint16_t *__restrict__ sinfo, const u8 *__restrict__ ncode,
                                                     u32 s0, u32 s1, typename Op::DElem *__restrict__ E) {
  const u32 s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  if (sinfo[s] & XS_POST) return;
  const u32 x = scell[s];
  E[s] = op.dpre(x, (u32)ncode[x]);
}

template <class Op>
__global__ void __launch_bounds__(64) k_xtrunk_dscan(Op op, const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                     u32 c0, u32 c1, const u32 *__restrict__ scell,
                                                     const uint16_t *__restrict__ sinfo, const u8 *__restrict__ ncode,
                                                     Geo g, const typename Op::DElem *__restrict__ E,
                                                     typename Op::V *__restrict__ R) {
  typedef typename Op::DElem Elem;
  typedef typename Op::V V;
  const u32 c = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = c < c1;
  const u32 s0 = active ? cstart[c] : 0u;
  const u32 m = active ? clen[c] : 0u;
  // the chain is walked from its last slot (the tail cell, possibly followed by its post slots) upstream;
  // position i counts from the end: slot = s0 + m - 1 - i
  V t = V();
  bool started = false;
  Elem cur[XU], nxt[XU];
  u32 curi[XU], nxti[XU];
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const u32 sl = s0 + (m ? m - 1u - ((u32)u < m ? (u32)u : m - 1u) : 0u);
    cur[u] = E[sl];
    curi[u] = sinfo[sl];
  }
  for (u32 base = 0; __any((int)(base < m)); base += XU) {
    const u32 nb = base + XU;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const u32 i = nb + (u32)u;
      const u32 sl = s0 + (m ? m - 1u - (i < m ? i : m - 1u) : 0u);
      nxt[u] = E[sl];
      nxti[u] = sinfo[sl];
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const u32 i = base + (u32)u;
      if (i < m && !(curi[u] & XS_POST)) {
        const u32 sl = s0 + m - 1u - i;
        if (!started) {  // the tail: its downstream cell belongs to a chain of a higher bucket (final), or it is a pit
          started = true;
          const u32 x = scell[sl];
          const u32 code = ncode[x];
          t = d8_is_dir(code) ? op.dfold(cur[u], op.top(d8_down(g, x, code))) : op.droot(cur[u]);
        } else {
          t = op.dfold(cur[u], t);
        }
        R[sl] = t;
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      cur[u] = nxt[u];
      curi[u] = nxti[u];
    }
  }
}

template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_dscatter(Op op, const u32 *__restrict__ scell,
                                                         const uint16_t *__restrict__ sinfo, u32 s0, u32 s1,
                                                         const typename Op::V *__restrict__ R) {
  const u32 s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  if (sinfo[s] & XS_POST) return;
  op.store(scell[s], R[s]);
}

// ---- leaves, down -------------------------------------------------------------------------------
#define XHW (XT + 2)
template <class Op>
__global__ void __launch_bounds__(256) k_xtile_down(Op op, XTileArgs a) {
  typedef typename Op::V V;
  __shared__ V val[XHW * XHW];  // with a 1-cell ring: the downstream cell of a leaf may be a trunk cell next door
  __shared__ u8 C[XTC];
  __shared__ uint16_t ord[XTC];
  __shared__ uint16_t off[XOFF];
  const u32 tid = threadIdx.x;
  const u32 tc = blockIdx.x, tr = blockIdx.y;
  const size_t tile = (size_t)tr * a.ntc + tc;
  const i64 r0 = (i64)tr * XT, c0 = (i64)tc * XT;
  if (tid < XOFF) off[tid] = a.toff[tile * XOFF + tid];
  for (u32 i = tid; i < XHW * XHW; i += 256u) {
    const i64 gr = r0 + (i64)(i / XHW) - 1, gc = c0 + (i64)(i % XHW) - 1;
    V v = V();
    if (gr >= 0 && gr < (i64)a.nrow && gc >= 0 && gc < (i64)a.ncol) v = op.top((u32)(gr * (i64)a.ncol + gc));
    val[i] = v;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const u32 l = tid + 256u * j;
    const i64 gr = r0 + (l >> 6), gc = c0 + (l & 63);
    C[l] = (gr < (i64)a.nrow && gc < (i64)a.ncol) ? a.ncode[(size_t)gr * a.ncol + (size_t)gc] : (u8)D8_MV;
    ord[l] = a.tord[tile * XTC + l];
  }
  __syncthreads();
  const u32 total = off[XOFF - 1];
  int last = 0;
  for (int s = 1; s < XOFF - 1; ++s) last = off[s] < total ? s : last;
  for (int s = last; s >= 0; --s) {
    const u32 b = off[s], e = off[s + 1];
    for (u32 j = b + tid; j < e; j += 256u) {
      const u32 x = ord[j];
      const int lr = x >> 6, lc = x & 63;
      const u32 code = C[x];
      const bool root = !d8_is_dir(code);
      int pr = lr, pc = lc;
      if (!root) {
        const int k = d8_slot(code);
        pr += d8_dr(k);
        pc += d8_dc(k);
      }
      const V pv = val[(pr + 1) * XHW + pc + 1];
      const u32 g = (u32)((r0 + lr) * (i64)a.ncol + c0 + lc);
      val[(lr + 1) * XHW + lc + 1] = op.apply(g, code, root, pv);
    }
    __syncthreads();
  }
  for (u32 j = tid; j < total; j += 256u) {  // only the leaves changed
    const u32 x = ord[j];
    const int lr = x >> 6, lc = x & 63;
    op.store((u32)((r0 + lr) * (i64)a.ncol + c0 + lc), val[(lr + 1) * XHW + lc + 1]);
  }
}

template <class Op>
static int run_exact_down(pfd_raster *h, const Op &op, const char *name) {
  typedef typename Op::DElem Elem;
  typedef typename Op::V V;
  ExactPlan *p = (ExactPlan *)h->xplan;
  pfd_seg_begin(h, name);
  i64 launches = 1;
  DevBuf E, R;
  PFDCHK(E.alloc(std::max<size_t>((size_t)p->nslot, 1) * sizeof(Elem) + 64));
  PFDCHK(R.alloc(std::max<size_t>((size_t)p->nslot, 1) * sizeof(V) + 64));
  for (int b = 31; b >= 0; --b) {
    const u32 s0 = (u32)p->b_slot[b], s1 = (u32)p->b_slot[b + 1];
    const u32 c0 = (u32)p->b_chain[b], c1 = (u32)p->b_chain[b + 1];
    if (c1 == c0) continue;
    k_xtrunk_dpre<Op><<<cdiv_u32(s1 - s0, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, h->ncode, s0, s1,
                                                                     E.as<Elem>());
    k_xtrunk_dscan<Op><<<cdiv_u32(c1 - c0, 64), 64, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1, p->scell, p->sinfo,
                                                                    h->ncode, h->geo, E.as<Elem>(), R.as<V>());
    k_xtrunk_dscatter<Op><<<cdiv_u32(s1 - s0, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, s0, s1, R.as<V>());
    launches += 3;
  }
  XTileArgs a{(u32)h->nrow, (u32)h->ncol, p->ntc, p->lh, p->kids, h->ncode, p->tord, p->toff};
  k_xtile_down<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a);
  KCHK();
  pfd_seg_end(h, launches);
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}
