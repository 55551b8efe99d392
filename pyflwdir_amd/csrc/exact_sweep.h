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
//   DElem dpre(x, code)               everything apply() reads from memory, gathered per cell
//   V droot(DElem) / V dfold(DElem, V pv);  V top(p);  void store(x, V);  V dnodata(x) value of a nodata cell
#pragma once
#include <type_traits>
#include <typeinfo>

#include "exact.h"

// An operation may WATCH what the down tile pass stores (k_xtile_down writes every cell of the raster exactly once):
// members watch_cnt / watch_list / watch_cap and a predicate watched(code, value).  Row-block HAND lists the cells whose
// height is still unknown that way — two scans of 9 bytes per cell (count, then collect) were a fifth of a block's sweep.
template <class Op, class = void>
struct XWatch : std::false_type {};
template <class Op>
struct XWatch<Op, std::void_t<decltype(std::declval<const Op &>().watch_cnt)>> : std::true_type {};

struct XTileArgs {
  u32 nrow, ncol, ntc;
  const u8 *lh, *kids, *ncode;
  const uint16_t *tord, *toff;
  // down-sweeps: the values of the trunk cells stay in chain order (R, indexed through cslot) — the tile pass picks
  // them up from there and writes every cell of the raster once; no scatter pass per round
  const u32 *cslot;
  const void *R;
  // the trunk cells of a tile as a dense list (ExactPlan::tlist): what the raster-order passes over the trunk walk
  const uint2 *tlist = nullptr;
  const u32 *tl_off = nullptr;
};

// ---- leaves, up ---------------------------------------------------------------------------------
// A thread owns 4 quads of 4 consecutive cells (one 16-byte global access per quad and array where the
// quad lies inside the raster; cell by cell on the raster's last columns / rows).
template <class Op>
__global__ void __launch_bounds__(256) k_xtile_up(Op op, XTileArgs a) {
  typedef typename Op::LV LV;
  __shared__ __attribute__((aligned(16))) LV val[XTC];
  __shared__ __attribute__((aligned(16))) u8 K[XTC];
  __shared__ __attribute__((aligned(16))) uint16_t ord[XTC];
  __shared__ uint16_t off[XOFF];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_, tr = by_;
  const size_t tile = (size_t)tr * a.ntc + tc;
  const u32 r0 = tr * XT, c0 = tc * XT;
  if (tid < XOFF) off[tid] = a.toff[tile * XOFF + tid];
  if (r0 + XT <= a.nrow && c0 + XT <= a.ncol) {
    // a tile inside the raster: the loads of all four quads are issued before the first one is used (quad by quad the
    // compiler waits for each quad's three loads in turn: four round trips instead of one)
    u32 k4s[4], nds[4] = {0, 0, 0, 0};
    uint2 o4s[4];
    LV vs[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 l0 = 4u * tid + 1024u * j;
      const u32 g0 = (r0 + (l0 >> 6)) * a.ncol + c0 + (l0 & 63);
      __builtin_memcpy(&k4s[j], a.kids + g0, 4);
      __builtin_memcpy(&o4s[j], a.tord + tile * XTC + l0, 8);
      if (Op::NEEDS_NODATA) {
        u32 l4;
        __builtin_memcpy(&l4, a.lh + g0, 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) nds[j] |= (((l4 >> (8 * b)) & 0xFFu) == XL_NODATA) ? 1u << b : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 l0 = 4u * tid + 1024u * j;
      op.tile_init4((r0 + (l0 >> 6)) * a.ncol + c0 + (l0 & 63), nds[j], vs[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 l0 = 4u * tid + 1024u * j;
#pragma unroll
      for (int b = 0; b < 4; ++b) val[l0 + b] = vs[j][b];
      *(u32 *)&K[l0] = k4s[j];
      *(uint2 *)&ord[l0] = o4s[j];
    }
  } else {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 gr = r0 + (l0 >> 6), gc = c0 + (l0 & 63);
    LV v[4] = {LV(), LV(), LV(), LV()};
    u32 k4 = 0;
    if (gr < a.nrow && gc + 3 < a.ncol) {
      const u32 g0 = gr * a.ncol + gc;
      __builtin_memcpy(&k4, a.kids + g0, 4);
      u32 nd = 0;
      if (Op::NEEDS_NODATA) {
        u32 l4;
        __builtin_memcpy(&l4, a.lh + g0, 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) nd |= (((l4 >> (8 * b)) & 0xFFu) == XL_NODATA) ? 1u << b : 0u;
      }
      op.tile_init4(g0, nd, v);
    } else if (gr < a.nrow) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (gc + b < a.ncol) {
          const u32 g = gr * a.ncol + gc + b;
          k4 |= (u32)a.kids[g] << (8 * b);
          v[b] = op.tile_init(g, a.lh[g] == XL_NODATA);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) val[l0 + b] = v[b];
    *(u32 *)&K[l0] = k4;
    uint2 o4;
    __builtin_memcpy(&o4, a.tord + tile * XTC + l0, 8);
    *(uint2 *)&ord[l0] = o4;
  }
  }
  __syncthreads();
  const u32 total = off[XOFF - 1];
  for (int s = 1; s < XOFF - 1; ++s) {  // step 0 = headwaters: their value is the initial one
    const u32 b = off[s], e = off[s + 1];
    if (b >= total) break;
    for (u32 j = b + tid; j < e; j += 256u) {
      const u32 x = ord[j] & 0xFFFu;  // (the upper bits serve the down-sweep)
      val[x] = op.tile_combine(x, (u32)K[x], val);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 gr = r0 + (l0 >> 6), gc = c0 + (l0 & 63);
    LV v[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = val[l0 + b];
    if (gr < a.nrow && gc + 3 < a.ncol) {
      op.tile_store4(gr * a.ncol + gc, v);
    } else if (gr < a.nrow) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gc + b < a.ncol) op.tile_store(gr * a.ncol + gc + b, v[b]);
    }
  }
}

// ---- trunk, up ----------------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_pre(Op op, const u32 *__restrict__ scell,
                                                    const uint16_t *__restrict__ sinfo, u32 s0, u32 s1,
                                                    typename Op::Elem *__restrict__ E) {
  const u32 s = s0 + pfd_block_1d() * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  const u32 info = sinfo[s];
  const u32 x = scell[s];
  E[s] = (info & XS_POST) ? op.pre_post(x) : op.pre_real(x, info & 0xFFu, (info >> 8) & 0xFu);
}

// ---- the serial fold: one LANE per chain ---------------------------------------------------------
// The running value never leaves its register.  A chain's slots are contiguous and start on a multiple of
// 4, so a lane moves 4 slots per load / store instruction; two register buffers of XB slots are used
// alternately (no copies between them: a copy of a loaded value would wait for the whole prefetch), and the
// post flags come from a bit array (2 loads per block).  In the last rounds only a few lanes of the chip
// are busy and the time is the INSTRUCTION count per slot of one wave: the body is branch-free, slots past
// the end of a chain are folded into a value nobody reads (the padding belongs to the chain).
template <class T>
struct alignas(16) XVec4 {
  T v[4];
};
template <class E>
struct XBlk {  // slots per register buffer: 16, or 8 for 16-byte elements
  static constexpr int G = sizeof(E) <= 8 ? 4 : 2;  // groups of 4 slots
};
__device__ __forceinline__ u32 xpost_bits(const u32 *__restrict__ spost, u32 s) {  // flags of slots s .. s+31 (low bits first)
  const u32 w0 = spost[s >> 5], w1 = spost[(s >> 5) + 1u];
  return (u32)((((u64)w1 << 32) | (u64)w0) >> (s & 31u));
}

// the fold of ONE long chain by one wave (lane = 0..63), see k_xtrunk_scan; sE / sR / sB: 64 entries each; sync: the
// workgroup barrier where the wave is the workgroup, a wave-level fence where other waves of the workgroup are elsewhere
template <class Op, class Sync, class Finish>
__device__ __forceinline__ void xscan_long(const Op &op, u32 cc, u32 lane, const u32 *__restrict__ cstart,
                                           const u32 *__restrict__ clen, const u32 *__restrict__ spost,
                                           const typename Op::Elem *__restrict__ E, typename Op::V *__restrict__ R,
                                           XVec4<typename Op::Elem> *sE, XVec4<typename Op::V> *sR, u32 *sB, Sync sync,
                                           Finish finish) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  {
    const u32 s0 = cstart[cc];
    const u32 ng = ((clen[cc] & XC_LEN) + 3u) >> 2;
    const XVec4<Elem> *E4 = (const XVec4<Elem> *)E + (s0 >> 2);
    XVec4<V> *R4 = (XVec4<V> *)R + (s0 >> 2);
    auto gload = [&](u32 gi, XVec4<Elem> &e, u32 &bits) {
      const u32 gg = gi < ng ? gi : ng - 1u;
      e = E4[gg];
      bits = xpost_bits(spost, s0 + 4u * gg) & 0xFu;
    };
    XVec4<Elem> cur;
    u32 cb;
    gload(lane, cur, cb);
    V t = V();
    for (u32 g0 = 0; g0 < ng; g0 += 64u) {
      sE[lane] = cur;
      sB[lane] = cb;
      gload(g0 + 64u + lane, cur, cb);
      sync();
      const u32 cnt = ng - g0 < 64u ? ng - g0 : 64u;
      // exact fold of the groups [q0, cnt) of the block by lane 0, from the running value tt
      auto exact = [&](u32 q0, V tt) {
        for (u32 gq = q0; gq < cnt; ++gq) {
          const XVec4<Elem> e = sE[gq];
          const u32 bits = sB[gq];
          XVec4<V> r;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const V f = op.fold(tt, e.v[j], ((bits >> j) & 1u) != 0);
            tt = (j == 0 && g0 + gq == 0) ? op.first(e.v[j]) : f;
            r.v[j] = tt;
          }
          sR[gq] = r;
        }
        return tt;
      };
      if (!Op::FAST) {
        if (lane == 0) t = exact(0, t);
      } else {
        // speculative: lane 0 folds without the operation's special cases (one dependent instruction per slot),
        // then every lane checks its own group; a block with a special operand is redone exactly
        V tt = t;
        const u32 q0 = g0 == 0 ? 1u : 0u;  // (the chain's first group holds the head: always exact)
        if (lane == 0) {
          if (q0) {
            const XVec4<Elem> e = sE[0];
            const u32 bits = sB[0];
            XVec4<V> r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const V f = op.fold(tt, e.v[j], ((bits >> j) & 1u) != 0);
              tt = j == 0 ? op.first(e.v[j]) : f;
              r.v[j] = tt;
            }
            sR[0] = r;
          }
          if (!Op::FAST_CONST) {
#pragma unroll 4
            for (u32 gq = q0; gq < cnt; ++gq) {
              const XVec4<Elem> e = sE[gq];
              XVec4<V> r;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                tt = op.fold_fast(tt, e.v[j]);
                r.v[j] = tt;
              }
              sR[gq] = r;
            }
          }
        }
        bool bad = false;
        if constexpr (Op::FAST_CONST) {
          // the speculative fold leaves the running value as it is (Strahler along a main stem): no serial loop at all —
          // every lane fills its own group with the value that entered the block and checks its four operands
          static_assert(!Op::FAST_CONST || sizeof(V) == 4, "broadcast of the running value");
          u32 tb;
          __builtin_memcpy(&tb, &tt, 4);
          tb = (u32)__shfl((int)tb, 0);
          __builtin_memcpy(&tt, &tb, 4);
          if (lane >= q0 && lane < cnt) {
            const XVec4<Elem> e = sE[lane];
            XVec4<V> r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r.v[j] = tt;
            sR[lane] = r;
            bad = (int)op.special(tt, e.v[0]) | (int)op.special(tt, e.v[1]) | (int)op.special(tt, e.v[2]) |
                  (int)op.special(tt, e.v[3]);
          }
        } else {
        sync();
        if (lane >= q0 && lane < cnt) {
          const XVec4<Elem> e = sE[lane];
          const XVec4<V> r = sR[lane];
          V prev = t;  // (lane 0: the value that entered the block)
          if (lane) prev = sR[lane - 1u].v[3];
          bad = (int)op.special(prev, e.v[0]) | (int)op.special(r.v[0], e.v[1]) | (int)op.special(r.v[1], e.v[2]) |
                (int)op.special(r.v[2], e.v[3]);
        }
        }
        if (__any((int)bad)) {
          sync();
          if (lane == 0) tt = exact(0, t);
        }
        t = tt;
      }
      sync();
      if (g0 + lane < ng) R4[g0 + lane] = sR[lane];
      if (lane == 0 && g0 + 64u >= ng) {  // the block that holds the chain's last slot
        const u32 last = (clen[cc] & XC_LEN) - 1u;
        finish(cc, s0, clen[cc], sR[(last >> 2) - g0].v[last & 3u]);
      }
    }
  }
}
struct XSyncWG {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
struct XSyncWave {  // one wave works alone on its LDS words: program order + a compiler / LDS fence
  __device__ __forceinline__ void operator()() const {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
};
// The first `nlong` workgroups take one LONG chain each (ids in longc): the wave streams the chain through LDS,
// 64 groups of 4 slots per block (the next block's loads are in flight during the fold), lane 0 folds the block
// from LDS and the wave stores the results.  A lane of its own would run at one HBM round trip per 48 slots.
template <class Op>
__global__ void __launch_bounds__(64) k_xtrunk_scan(Op op, const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                    u32 c0, u32 c1, const u32 *__restrict__ longc, u32 nlong,
                                                    const u32 *__restrict__ spost,
                                                    const typename Op::Elem *__restrict__ E,
                                                    typename Op::V *__restrict__ R, const u32 *__restrict__ scell,
                                                    u8 *__restrict__ dirty = nullptr, const u32 *__restrict__ dchain = nullptr) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  constexpr int G = XBlk<Elem>::G;
  // (dirty != nullptr: an incremental re-sweep of a row block — only the chains marked there are folded again)
  // The END of a chain is the only trunk cell a later round reads from the raster (as a light upstream cell): whoever
  // folds the chain stores it — its value follows its post slots, so it is the value of the chain's last slot — and,
  // in a re-sweep, marks the chain that end drains into (always a chain of a later round).
  auto finish = [&](u32 c, u32 s0, u32 cl, V endv) {
    op.store(scell[s0 + (cl & XC_LEN) - 1u - (cl >> 29)], endv);
    if (dchain) {
      const u32 d = dchain[c];
      if (d != 0xFFFFFFFFu) dirty[d] = 1;
    }
  };
  if (blockIdx.x < nlong) {
    __shared__ XVec4<Elem> sE[64];
    __shared__ XVec4<V> sR[64];
    __shared__ u32 sB[64];
    const u32 cc = longc[blockIdx.x];
    if (dirty && !dirty[cc]) return;
    xscan_long(op, cc, threadIdx.x, cstart, clen, spost, E, R, sE, sR, sB, XSyncWG(), finish);
    return;
  }
  const u32 c = c0 + (blockIdx.x - nlong) * blockDim.x + threadIdx.x;
  const bool active = c < c1 && (!dirty || dirty[c]);
  const u32 s0 = active ? cstart[c] : 0u;
  u32 m = active ? (clen[c] & XC_LEN) : 0u;
  if (m >= XLONG) m = 0;  // folded by a wave of its own (above)
  const u32 ng = (m + 3u) >> 2;  // groups of the chain
  const XVec4<Elem> *E4 = (const XVec4<Elem> *)E + (s0 >> 2);
  XVec4<V> *R4 = (XVec4<V> *)R + (s0 >> 2);
  V t = V();
  XVec4<Elem> ea[G], eb[G], ec[G], ed[G];
  u32 ba, bb, bc, bd;
  auto load = [&](u32 g0, XVec4<Elem>(&e)[G], u32 &bits) {  // groups g0 .. g0+G-1 (clamped: always a group of the chain)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const u32 gi = g0 + (u32)g;
      e[g] = E4[gi < ng ? gi : (ng ? ng - 1u : 0u)];
    }
    bits = xpost_bits(spost, s0 + 4u * (g0 < ng ? g0 : 0u));
  };
  auto fold = [&](u32 g0, const XVec4<Elem>(&e)[G], u32 bits) {
    if (Op::FAST_SHORT) {
      // speculative block: fold without the operation's special cases (accuflux: the nodata rule) and
      // check afterwards that no operand of the block was special — one dependent instruction per slot
      // instead of six.  A block with a special operand in ANY lane of the wave is redone exactly.
      V tt = t;
      bool bad = false;
      XVec4<V> r[G];
      const bool head = g0 == 0;
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const Elem x = e[g].v[j];
          if (g == 0 && j == 0) {
            bad |= !head && op.special(tt, x);
            const V f = op.fold_fast(tt, x);
            tt = head ? op.first(x) : f;
          } else {
            bad |= op.special(tt, x);
            tt = op.fold_fast(tt, x);
          }
          r[g].v[j] = tt;
        }
      }
      if (!__any((int)bad)) {
        t = tt;
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (g0 + (u32)g < ng) R4[g0 + (u32)g] = r[g];
        return;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const u32 gi = g0 + (u32)g;
      XVec4<V> r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const V f = op.fold(t, e[g].v[j], ((bits >> (4 * g + j)) & 1u) != 0);
        t = (g == 0 && j == 0 && g0 == 0) ? op.first(e[g].v[j]) : f;
        r.v[j] = t;
      }
      if (gi < ng) R4[gi] = r;
    }
  };
  // four buffers in a ring: the loads run 3 blocks (48 slots) ahead of the fold — one HBM round trip
  load(0, ea, ba);
  load(G, eb, bb);
  load(2 * G, ec, bc);
  for (u32 g0 = 0; __any((int)(g0 < ng)); g0 += 4 * G) {
    load(g0 + 3 * G, ed, bd);
    fold(g0, ea, ba);
    load(g0 + 4 * G, ea, ba);
    fold(g0 + G, eb, bb);
    load(g0 + 5 * G, eb, bb);
    fold(g0 + 2 * G, ec, bc);
    load(g0 + 6 * G, ec, bc);
    fold(g0 + 3 * G, ed, bd);
  }
  if (m) finish(c, s0, clen[c], R[s0 + m - 1u]);  // (the lane reads back what it stored itself)
}

// ---- gather + fold of the SHORT chains in one kernel (round 6) --------------------------------------------------------
// k_xtrunk_pre is bandwidth (scattered sectors), the lane-per-chain fold of k_xtrunk_scan is latency (a chain of 8 slots
// costs its lane ~28 memory instructions, most of them clamped repeats, and a 16-byte access per lane and trip), and the
// two ran one after the other with the elements making a round trip through HBM between them.  Here a workgroup takes
// 256 consecutive chains — a contiguous run of slots, chains being laid out back to back — and moves that run through LDS
// in chunks of XFuse::CAP slots: (1) every thread gathers the elements of the chunk's slots tid, tid + 256, ... (coalesced
// slot arrays, the same hooks as k_xtrunk_pre) into LDS, the post flags as ballots; (2) the lane that owns a chain folds
// the part of it that lies in the chunk from LDS, 4 slots per trip, running value in a register across chunks,
// speculative form first (a group with a special operand is redone exactly, per lane); (3) the workgroup stores the
// chunk's values with 16-byte accesses.  The gathers of one workgroup run under the folds of the others on its CU.
// Chains of XLONG slots or more keep their own path (k_xtrunk_pre_long + the wave-per-chain part of k_xtrunk_scan): a
// chunk never covers their slots — it ends where the next long chain of the workgroup starts.
#ifndef XF_ABLATE
#define XF_ABLATE 0  // timing experiments only: 1 = no fold, 2 = no gather (wrong results)
#endif
// chains a round must hold to take the workgroup-per-256-chains kernels (k_xtrunk_prescan, k_xtrunk_dscan_lds);
// PFD_TEST_FUSE_MIN: the tests run rasters of a few million cells through them
static u32 xfuse_min_chains() {
  const char *e = pfd_knob("PFD_TEST_FUSE_MIN");
  return e ? (u32)atoll(e) : (1u << 20);
}
template <class Op>
struct XFuse {
  static constexpr u32 B = (u32)(sizeof(typename Op::Elem) + sizeof(typename Op::V));
#ifndef XF_CAP4
#define XF_CAP4 2048u
#endif
  static constexpr u32 CAP = B <= 8 ? XF_CAP4 : (B <= 16 ? 1024u : 512u);  // slots per chunk: <= 16 KB of elements + values
};
__device__ __forceinline__ u32 xwg_min(u32 v, u32 *s_red) {  // minimum over a 256-thread workgroup (s_red: 4 words, reusable after return)
  for (int o = 32; o > 0; o >>= 1) v = min(v, (u32)__shfl_xor((int)v, o));
  __syncthreads();  // (the previous call's readers are done)
  if ((threadIdx.x & 63u) == 0u) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
}
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_prescan(Op op, const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                        u32 c0, u32 c1, const u32 *__restrict__ longc, u32 nlong,
                                                        const u32 *__restrict__ spost, const u32 *__restrict__ scell,
                                                        const uint16_t *__restrict__ sinfo,
                                                        const typename Op::Elem *__restrict__ E,
                                                        typename Op::V *__restrict__ R) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  constexpr u32 CAP = XFuse<Op>::CAP;
  __shared__ XVec4<Elem> sE4[CAP / 4];
  __shared__ XVec4<V> sR4[CAP / 4];
  __shared__ u32 sP[CAP / 32];
  __shared__ u32 sB[64];
  __shared__ u32 s_red[4];
  const u32 tid = threadIdx.x;
  if (blockIdx.x < nlong) {  // a LONG chain of the round (its elements: k_xtrunk_pre_long): one wave, beside the others' chunks
    if (tid >= 64u) return;
    const u32 cc = longc[blockIdx.x];
    auto finish = [&](u32, u32 s0, u32 cl, V endv) { op.store(scell[s0 + (cl & XC_LEN) - 1u - (cl >> 29)], endv); };
    xscan_long(op, cc, tid, cstart, clen, spost, E, R, sE4, sR4, sB, XSyncWave(), finish);
    return;
  }
  Elem *sE = reinterpret_cast<Elem *>(sE4);
  const V *sR = reinterpret_cast<const V *>(sR4);
  // (XCD-aware order of the short-chain workgroups only: the long ones come first in the grid)
  u32 L = blockIdx.x - nlong;
  if (PFD_XCD_ORDER) {
    const u32 n = gridDim.x - nlong, q = n >> 3, r = n & 7u, k = L & 7u;
    L = k * q + min(k, r) + (L >> 3);
  }
  const u32 c = c0 + L * 256u + tid;
  const bool act = c < c1;
  const u32 a = act ? cstart[c] : 0xFFFFFFFFu;
  const u32 cl = act ? clen[c] : 0u;
  u32 m = cl & XC_LEN;
  const bool islong = m >= XLONG;
  if (islong) m = 0;
  const u32 endp = a + ((m + 3u) & ~3u);  // (padding belongs to the chain)
  u32 cur = m ? a : 0xFFFFFFFFu;          // next slot of the own chain to fold; nothing left: NONE
  // end of the last short chain of the workgroup (maximum = NOT of the minimum of the complements) and whether a
  // long chain lies between them
  const u32 lastend = ~xwg_min(m ? ~endp : 0xFFFFFFFFu, s_red);
  const bool anylong = xwg_min(islong ? 0u : 1u, s_red) == 0u;
  V t = V();
  for (;;) {
    const u32 g = xwg_min(cur, s_red);
    if (g == 0xFFFFFFFFu) break;
    // the chunk [g, e): up to CAP slots, not into a long chain, not beyond the chains of this workgroup
    u32 e = min(g + CAP, lastend);
    if (anylong) e = min(e, xwg_min((islong && a >= g) ? a : 0xFFFFFFFFu, s_red));
    const u32 cnt = e - g;
    // (1) gather, four slots per thread in flight
    for (u32 i0 = 0; i0 < cnt; i0 += 1024u) {
      u32 info[4], x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 i = i0 + 256u * (u32)k + tid;
        const u32 sl = g + (i < cnt ? i : cnt - 1u);
        info[k] = sinfo[sl];
        x[k] = scell[sl];
      }
      Elem ev[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#if XF_ABLATE == 2
        ev[k] = Elem();
#else
        ev[k] = (info[k] & XS_POST) ? op.pre_post(x[k]) : op.pre_real(x[k], info[k] & 0xFFu, (info[k] >> 8) & 0xFu);
#endif
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 i = i0 + 256u * (u32)k + tid;
        const bool in = i < cnt;
        const unsigned long long pb = __ballot((int)(in && (info[k] & XS_POST)));
        if (in) sE[i] = ev[k];
        if ((tid & 63u) == 0u && i0 + 256u * (u32)k < cnt) {
          sP[i >> 5] = (u32)pb;  // (the wave's 64 slots: two words; words past the chunk are never read)
          sP[(i >> 5) + 1u] = (u32)(pb >> 32);
        }
      }
    }
    __syncthreads();
    // (2) fold
#if XF_ABLATE == 1
    if (cur != 0xFFFFFFFFu && cur < e) cur = endp <= e ? 0xFFFFFFFFu : e;
#endif
    if (cur != 0xFFFFFFFFu && cur < e) {
      const u32 hi = min(endp, e) - g;
      u32 q = cur - g;
      XVec4<Elem> evn = sE4[q >> 2];
      for (; q < hi; q += 4u) {
        const XVec4<Elem> ev = evn;
        if (q + 4u < hi) evn = sE4[(q + 4u) >> 2];  // (the next group is on its way during the fold)
        const u32 bits = (sP[q >> 5] >> (q & 31u)) & 0xFu;
        const bool head = g + q == a;
        XVec4<V> r;
        V tt = t;
        bool bad = !Op::FAST_SHORT;
        if (Op::FAST_SHORT) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const Elem xe = ev.v[j];
            if (j == 0) {
              bad |= !head && op.special(tt, xe);
              const V f = op.fold_fast(tt, xe);
              tt = head ? op.first(xe) : f;
            } else {
              bad |= op.special(tt, xe);
              tt = op.fold_fast(tt, xe);
            }
            r.v[j] = tt;
          }
        }
        if (bad) {
          tt = t;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const V f = op.fold(tt, ev.v[j], ((bits >> j) & 1u) != 0);
            tt = (j == 0 && head) ? op.first(ev.v[j]) : f;
            r.v[j] = tt;
          }
        }
        t = tt;
        sR4[q >> 2] = r;
      }
      if (endp <= e) {  // the chain ends in this chunk: its END is the only trunk cell a later round reads from the raster
        const u32 last = a + m - 1u;
        op.store(scell[last - (cl >> 29)], sR[last - g]);
        cur = 0xFFFFFFFFu;
      } else {
        cur = e;
      }
    }
    __syncthreads();
    // (3) store
    XVec4<V> *R4 = reinterpret_cast<XVec4<V> *>(R) + (g >> 2);
    for (u32 q = tid; q < (cnt >> 2); q += 256u) R4[q] = sR4[q];
  }
}
// elements of the LONG chains of a round (blockIdx.x = the chain's number in longc, blockIdx.y = a piece of 1024 slots)
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_pre_long(Op op, const u32 *__restrict__ longc, const u32 *__restrict__ cstart,
                                                         const u32 *__restrict__ clen, const u32 *__restrict__ scell,
                                                         const uint16_t *__restrict__ sinfo,
                                                         typename Op::Elem *__restrict__ E) {
  const u32 cc = longc[blockIdx.x];
  const u32 s0 = cstart[cc], n4 = ((clen[cc] & XC_LEN) + 3u) & ~3u;
  const u32 i1 = min(n4, (blockIdx.y + 1u) * 1024u);
  for (u32 i = blockIdx.y * 1024u + threadIdx.x; i < i1; i += 256u) {
    const u32 s = s0 + i;
    const u32 info = sinfo[s];
    const u32 x = scell[s];
    E[s] = (info & XS_POST) ? op.pre_post(x) : op.pre_real(x, info & 0xFFu, (info >> 8) & 0xFu);
  }
}

template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_scatter(Op op, const u32 *__restrict__ scell,
                                                        const uint16_t *__restrict__ sinfo, u32 s0, u32 s1,
                                                        const typename Op::V *__restrict__ R) {
  const u32 s = s0 + pfd_block_1d() * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  const u32 info = sinfo[s];
  if (info & XS_POST) return;
  op.store(scell[s], R[s + ((info >> 12) & 7u)]);  // the cell's value = the running value after its last post slot
}

// Between the rounds only the END of a chain is read from the raster (as a light upstream cell of a later round's
// slot): the scan stores it (k_xtrunk_scan, finish).  Every other trunk cell reaches the raster in one pass in RASTER
// order at the end (k_xtrunk_unscatter): coalesced reads of the marks and slot numbers, partial but sector-local writes —
// the per-slot scatter in chain order paid a whole sector per 4-byte value.
// (One workgroup per 64 x 64 TILE, not per strip of a raster row: a chain crosses a tile in a run of ~64 consecutive
//  slots, so the workgroup's scattered accesses in chain order fall into a few hundred bytes per chain and the L2
//  serves all but the first touch of a sector — a row strip meets every chain once and pays a sector per value.)
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_unscatter(Op op, XTileArgs a, const typename Op::V *__restrict__ R,
                                                          u32 s_limit = 0xFFFFFFFFu) {  // only the cells of slots below s_limit
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 r0 = by_ * XT, c0 = bx_ * XT;
  u32 l4s[4], x0s[4];
  uint4 c4s[4];
  bool full[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 gr = r0 + (l0 >> 6), gc = c0 + (l0 & 63);
    full[j] = gr < a.nrow && gc + 3 < a.ncol;
    x0s[j] = gr * a.ncol + gc;
    l4s[j] = XL_NODATA * 0x01010101u;
    if (full[j]) {
      __builtin_memcpy(&l4s[j], a.lh + x0s[j], 4);
    } else if (gr < a.nrow) {
      for (u32 b = 0; b < 4u && gc + b < a.ncol; ++b) l4s[j] = (l4s[j] & ~(0xFFu << (8 * b))) | ((u32)a.lh[x0s[j] + b] << (8 * b));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32 tm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) tm |= xl_trunk((l4s[j] >> (8 * b)) & 0xFFu) ? 1u << b : 0u;
    c4s[j] = make_uint4(0u, 0u, 0u, 0u);
    if (tm) {
      if (full[j]) {
        __builtin_memcpy(&c4s[j], a.cslot + x0s[j], 16);
      } else {
        u32 t[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if ((tm >> b) & 1u) t[b] = a.cslot[x0s[j] + b];
        c4s[j] = make_uint4(t[0], t[1], t[2], t[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l4 = l4s[j];
    u32 tm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) tm |= xl_trunk((l4 >> (8 * b)) & 0xFFu) ? 1u << b : 0u;
    if (!tm) continue;
    const u32 cs[4] = {c4s[j].x, c4s[j].y, c4s[j].z, c4s[j].w};
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (cs[b] >= s_limit) tm &= ~(1u << b);  // (the chains of the last rounds are scattered when they are done)
    typename Op::V v[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)  // (the mark carries the number of post slots: the cell's value sits behind them)
      v[b] = R[(tm >> b) & 1u ? cs[b] + ((l4 >> (8 * b)) & 7u) : 0u];
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if ((tm >> b) & 1u) op.store(x0s[j] + b, v[b]);
  }
}

// The same pass over the tile's dense trunk list (round 5): 8 contiguous bytes per trunk cell — slot, local index, post
// slots — instead of the marks of all 4096 cells and a 16-byte quad of cslot wherever a quad holds a trunk cell (along a
// river that is one useful word per 64-byte sector).
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_unscatter_list(Op op, XTileArgs a, const typename Op::V *__restrict__ R,
                                                               u32 s_limit = 0xFFFFFFFFu) {
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tile = by_ * a.ntc + bx_;
  const u32 b = a.tl_off[tile], e = a.tl_off[tile + 1];
  const u32 r0 = by_ * XT, c0 = bx_ * XT;
  for (u32 i0 = b; i0 < e; i0 += 1024u) {  // four entries per thread in flight
    uint2 en[4];
    typename Op::V v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 i = i0 + threadIdx.x + 256u * (u32)k;
      en[k] = a.tlist[i < e ? i : b];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = R[en[k].x + ((en[k].y >> 12) & 7u)];  // (the value sits behind the cell's post slots)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 i = i0 + threadIdx.x + 256u * (u32)k;
      const u32 l = en[k].y & 0xFFFu;
      if (i < e && en[k].x < s_limit) op.store((r0 + (l >> 6)) * a.ncol + c0 + (l & 63u), v[k]);
    }
  }
}

// ---- incremental re-sweep of a row block (ExactPlan::schain ...): the same steps for the dirty chains only ----
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_pre_inc(Op op, const u32 *__restrict__ scell, const uint16_t *__restrict__ sinfo,
                                                        const u32 *__restrict__ schain, const u8 *__restrict__ dirty, u32 s0,
                                                        u32 s1, typename Op::Elem *__restrict__ E) {
  const u32 s = s0 + pfd_block_1d() * blockDim.x + threadIdx.x;
  if (s >= s1 || !dirty[schain[s]]) return;
  const u32 info = sinfo[s];
  const u32 x = scell[s];
  E[s] = (info & XS_POST) ? op.pre_post(x) : op.pre_real(x, info & 0xFFu, (info >> 8) & 0xFu);
}
// the cells of the dirty chains, in chain order (a sector per value — but only below the seeds that changed)
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_scatter_inc(Op op, const u32 *__restrict__ scell, const uint16_t *__restrict__ sinfo,
                                                            const u32 *__restrict__ schain, const u8 *__restrict__ dirty, u32 nslot,
                                                            const typename Op::V *__restrict__ R) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslot || !dirty[schain[s]]) return;
  const u32 info = sinfo[s];
  if (info & XS_POST) return;  // (padding slots are post slots)
  op.store(scell[s], R[s + ((info >> 12) & 7u)]);
}
// A sweep that forks work onto the handle's second stream frees its element / value buffers on return: on an error path
// between fork and join the second stream may still read them — whoever leaves the function waits for it first.
struct XStream2Guard {
  pfd_raster *h;
  ~XStream2Guard() {
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  }
};
// (A/B knob: PFD_XLIST_OFF=1 walks the marks of every cell as rounds 3-4 did)
static inline bool xlist_on() {
  static const bool on = pfd_knob("PFD_XLIST_OFF") == nullptr;
  return on;
}
// an update request (pfd_set_block_update(h, 2)) can be served: the kept sweep is this operation's, into this buffer
static inline bool xinc_applies(pfd_raster *h, const void *out_dev, size_t tag) {
  const ExactPlan *p = (const ExactPlan *)h->xplan;
  return h->block_update == 2 && h->xplan_state == 1 && p && p->xinc_ready && p->inc_valid && p->inc_tag == tag && p->inc_out == out_dev;
}
template <class Op>
static int run_exact_up_inc(pfd_raster *h, const Op &op) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  ExactPlan *p = (ExactPlan *)h->xplan;
  pfd_seg_begin(h, "exact_up_block_update");
  PFDCHK(pfd_xinc_mark(h, h->xseed, h->xseed_elem));
  i64 launches = 2;
  Elem *E = (Elem *)p->incE;
  V *R = (V *)p->incR;
  for (int b = 0; b < 32; ++b) {
    const u32 s0 = (u32)p->b_slot[b], s1 = (u32)p->b_slot[b + 1];
    const u32 c0 = (u32)p->b_chain[b], c1 = (u32)p->b_chain[b + 1];
    if (c1 == c0) continue;
    k_xtrunk_pre_inc<Op><<<cdiv_u32(s1 - s0, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, p->schain, p->dirty, s0, s1, E);
    const u32 nl = (u32)(p->b_long[b + 1] - p->b_long[b]);
    k_xtrunk_scan<Op><<<nl + cdiv_u32(c1 - c0, 64), 64, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1, p->longc + p->b_long[b], nl,
                                                                        p->spost, E, R, p->scell, p->dirty, p->dchain);
    launches += 2;
  }
  if (p->nslot) {
    k_xtrunk_scatter_inc<Op><<<cdiv_u32((u64)p->nslot, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, p->schain, p->dirty,
                                                                                 (u32)p->nslot, R);
    ++launches;
  }
  KCHK();
  pfd_seg_end(h, launches);
  return PFD_OK;
}

// keep: 0 = a sweep that leaves nothing behind; 1 = row block: keep the element / value arrays for updates;
// 2 = row block: `out` holds the result of the kept sweep for other halo seeds — fold only what the changed seeds reach
// (a full sweep, kept, when there is nothing to update from)
template <class Op>
static int run_exact_up(pfd_raster *h, const Op &op, const char *name, int keep = 0) {
  typedef typename Op::Elem Elem;
  typedef typename Op::V V;
  ExactPlan *p = (ExactPlan *)h->xplan;
  const size_t tag = typeid(Op).hash_code();
  if (!(h->xseed && h->xseed_out && h->halo_raw)) keep = 0;
  // (the caller has put the seeds into the halo rows of `out`)
  if (keep == 2 && xinc_applies(h, h->xseed_out, tag)) return run_exact_up_inc(h, op);
  pfd_xinc_drop(h);
  if (keep && pfd_xinc_prepare(h) != PFD_OK) keep = 0;
  pfd_seg_begin(h, name);
  i64 launches = 1;
  XTileArgs a{(u32)h->nrow, (u32)h->ncol, p->ntc, p->lh, p->kids, h->ncode, p->tord, p->toff, nullptr, nullptr};
  k_xtile_up<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a);
  KCHK();
  XDBG(h, "tile_up");
  if (h->xseed && h->xseed_out) {  // row block: the tile pass wrote the halo cells like any other; their values are given
    const size_t rowb = (size_t)h->ncol * h->xseed_elem;
    if (h->halo_top)
      HIPCHK(hipMemcpyAsync((char *)h->xseed_out + (size_t)(h->halo_top - 1) * rowb, h->xseed, rowb, hipMemcpyDeviceToDevice, h->stream));
    if (h->halo_bot)
      HIPCHK(hipMemcpyAsync((char *)h->xseed_out + (size_t)(h->halo_top + h->own_rows) * rowb, (const char *)h->xseed + rowb, rowb,
                            hipMemcpyDeviceToDevice, h->stream));
  }
  DevBuf Eb, Rb;
  const size_t ebytes = std::max<size_t>((size_t)p->nslot, 1) * sizeof(Elem) + 64, rbytes = std::max<size_t>((size_t)p->nslot, 1) * sizeof(V) + 64;
  const size_t sbytes = 2 * (size_t)h->ncol * h->xseed_elem;
  if (keep) {
    int rc;
    if ((rc = pfd_dmalloc(&p->incE, ebytes)) != PFD_OK || (rc = pfd_dmalloc(&p->incR, rbytes)) != PFD_OK ||
        (rc = pfd_dmalloc(&p->incSeed, sbytes)) != PFD_OK) {
      pfd_xinc_drop(h);
      return rc;
    }
    p->inc_bytes = ebytes + rbytes + sbytes;
    h->bytes_held += p->inc_bytes;
  } else {
    PFDCHK(Eb.alloc(ebytes));
    PFDCHK(Rb.alloc(rbytes));
  }
  Elem *E = keep ? (Elem *)p->incE : Eb.as<Elem>();
  V *R = keep ? (V *)p->incR : Rb.as<V>();
  // The last two rounds hold the main stems and their largest tributaries: a few thousand chains, the longest as long
  // as the longest flow path, folded serially — 0.5 ms each at 30000 x 30000 with most of the chip idle.  The raster-order
  // pass that writes the trunk cells (k_xtrunk_unscatter, bandwidth) therefore starts BESIDE them, on the handle's
  // second stream, for every chain of the earlier rounds; the chains of the last two rounds — a few per cent of the
  // slots — are scattered in chain order when they are done.
  int bsplit = xplan_tail_split(p);
  if (bsplit >= 0 && pfd_aux_stream(h) != PFD_OK) bsplit = -1;
  XStream2Guard guard2{h};  // (declared after E / R: runs before they are released)
  const u32 s_split = bsplit >= 0 ? (u32)p->b_slot[bsplit] : 0xFFFFFFFFu;
  // (a kept sweep needs the element array of every slot: the short chains' elements never leave LDS in the fused form)
  const bool fused = Op::FUSE_UP && !keep && !pfd_knob("PFD_SCAN_UNFUSED");
  const u32 fuse_min = xfuse_min_chains();
  for (int b = 0; b < 32; ++b) {
    const u32 s0 = (u32)p->b_slot[b], s1 = (u32)p->b_slot[b + 1];
    const u32 c0 = (u32)p->b_chain[b], c1 = (u32)p->b_chain[b + 1];
    if (c1 == c0) continue;
    if (b == bsplit) {
      a.cslot = p->cslot;
      HIPCHK(hipEventRecord(h->ev_fork, h->stream));
      HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
      a.tlist = p->tlist, a.tl_off = p->tl_off;
      if (xlist_on())
        k_xtrunk_unscatter_list<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream2>>>(op, a, R, s_split);
      else
        k_xtrunk_unscatter<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream2>>>(op, a, R, s_split);
      HIPCHK(hipEventRecord(h->ev_join, h->stream2));
      ++launches;
    }
    const u32 nl = (u32)(p->b_long[b + 1] - p->b_long[b]);
    // (a round that is mostly long chains — the main stems' — keeps the two-kernel form: nothing to fuse there)
    // and so do the rounds that run beside the raster-order pass on the second stream: the long chains' own gather
    // kernel — a few thousand half-empty workgroups — waits behind that pass's workgroups (1.9 ms measured)
    // and a round of fewer than 2^20 chains: 4096 workgroups are two generations of the chip's workgroup slots — below
    // that the workgroup with the longest run of chunks is the round (10000^2: accuflux 1.79 -> 2.02 ms when every round fused)
    if (fused && c1 - c0 >= fuse_min && (u64)nl * 16u <= (u64)(c1 - c0) && (bsplit < 0 || b < bsplit)) {
      if (nl) {
        k_xtrunk_pre_long<Op><<<dim3(nl, cdiv_u32(p->b_maxlen[b] + 3u, 1024u)), 256, 0, h->stream>>>(
            op, p->longc + p->b_long[b], p->cstart, p->clen, p->scell, p->sinfo, E);
        ++launches;
      }
      k_xtrunk_prescan<Op><<<nl + cdiv_u32(c1 - c0, 256), 256, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1,
                                                                               p->longc + p->b_long[b], nl, p->spost, p->scell,
                                                                               p->sinfo, E, R);
      ++launches;
      XDBG(h, "prescan");
      continue;
    }
    k_xtrunk_pre<Op><<<cdiv_u32(s1 - s0, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, s0, s1, E);
    XDBG(h, "pre");
    k_xtrunk_scan<Op><<<nl + cdiv_u32(c1 - c0, 64), 64, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1,
                                                                        p->longc + p->b_long[b], nl, p->spost, E, R, p->scell);
    XDBG(h, "scan");
    launches += 2;
  }
  if (bsplit >= 0) {
    k_xtrunk_scatter<Op><<<cdiv_u32((u32)p->nslot - s_split, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, s_split,
                                                                                     (u32)p->nslot, R);
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    ++launches;
  } else if (p->nslot) {
    a.cslot = p->cslot;
    a.tlist = p->tlist, a.tl_off = p->tl_off;
    if (xlist_on())
      k_xtrunk_unscatter_list<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a, R);
    else
      k_xtrunk_unscatter<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a, R);
    XDBG(h, "unscatter");
    ++launches;
  }
  KCHK();
  pfd_seg_end(h, launches);
  if (keep) {  // what an update starts from: this sweep's arrays, seeds and result buffer
    HIPCHK(hipMemcpyAsync(p->incSeed, h->xseed, sbytes, hipMemcpyDeviceToDevice, h->stream));
    p->inc_valid = true, p->inc_tag = tag, p->inc_out = h->xseed_out;
  }
  HIPCHK(hipStreamSynchronize(h->stream));  // E / R are released on return (unless kept)
  return PFD_OK;
}

// ---- trunk, down --------------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_dpre(Op op, const u32 *__restrict__ scell,
                                                     const uint16_t *__restrict__ sinfo, const u8 *__restrict__ ncode,
                                                     u32 s0, u32 s1, typename Op::DElem *__restrict__ E) {
  const u32 s = s0 + pfd_block_1d() * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  const u32 info = sinfo[s];
  const u32 x = scell[s];  // (a post slot names the light upstream cell: a valid cell as well)
  if (info & XS_POST) return;
  E[s] = op.dpre(x, (u32)ncode[x]);
}

// The same gather for ALL rounds at once, one workgroup per TILE (see k_xtrunk_unscatter): the operation's loads
// run in raster order, the stores into chain order fall into the few runs of slots that cross the tile.
template <class Op, bool LIMIT = false>
__global__ void __launch_bounds__(256) k_xtrunk_demit(Op op, XTileArgs a, typename Op::DElem *__restrict__ E,
                                                      u32 s_limit = 0xFFFFFFFFu) {  // LIMIT: only the slots below s_limit
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 r0 = by_ * XT, c0 = bx_ * XT;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const u32 gr = r0 + (l0 >> 6), gc = c0 + (l0 & 63);
    if (gr >= a.nrow) continue;
    const u32 x0 = gr * a.ncol + gc;
    u32 l4 = XL_NODATA * 0x01010101u, c4 = 0;
    if (gc + 3 < a.ncol) {
      __builtin_memcpy(&l4, a.lh + x0, 4);
      __builtin_memcpy(&c4, a.ncode + x0, 4);
    } else {
      for (u32 b = 0; b < 4u && gc + b < a.ncol; ++b) {
        l4 = (l4 & ~(0xFFu << (8 * b))) | ((u32)a.lh[x0 + b] << (8 * b));
        c4 |= (u32)a.ncode[x0 + b] << (8 * b);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (xl_trunk((l4 >> (8 * b)) & 0xFFu)) {
        // (the element first, whatever the slot: its loads must not wait for the slot number)
        const typename Op::DElem e = op.dpre(x0 + (u32)b, (c4 >> (8 * b)) & 0xFFu);
        const u32 sl = a.cslot[x0 + b];
        if (!LIMIT || sl < s_limit) E[sl] = e;
      }
    }
  }
}

// ... and over the tile's dense trunk list (see k_xtrunk_unscatter_list)
template <class Op, bool LIMIT = false>
__global__ void __launch_bounds__(256) k_xtrunk_demit_list(Op op, XTileArgs a, typename Op::DElem *__restrict__ E,
                                                           u32 s_limit = 0xFFFFFFFFu) {
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tile = by_ * a.ntc + bx_;
  const u32 b = a.tl_off[tile], e = a.tl_off[tile + 1];
  const u32 r0 = by_ * XT, c0 = bx_ * XT;
  for (u32 i0 = b; i0 < e; i0 += 512u) {  // two entries per thread in flight
    uint2 en[2];
    u32 x[2], cd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32 i = i0 + threadIdx.x + 256u * (u32)k;
      en[k] = a.tlist[i < e ? i : b];
      const u32 l = en[k].y & 0xFFFu;
      x[k] = (r0 + (l >> 6)) * a.ncol + c0 + (l & 63u);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) cd[k] = a.ncode[x[k]];
    typename Op::DElem el[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) el[k] = op.dpre(x[k], cd[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32 i = i0 + threadIdx.x + 256u * (u32)k;
      if (i < e && (!LIMIT || en[k].x < s_limit)) E[en[k].x] = el[k];
    }
  }
}

// the down-fold of ONE long chain by one wave (lane = 0..63), see k_xtrunk_dscan; sE / sR / sB: 64 entries each, sT: the
// running value that enters a block; sync: XSyncWG where the wave is the workgroup, XSyncWave otherwise
template <class Op, class Sync>
__device__ __forceinline__ void xdscan_long(const Op &op, u32 cc, u32 lane, const u32 *__restrict__ cstart,
                                            const u32 *__restrict__ clen, const u32 *__restrict__ scell,
                                            const u32 *__restrict__ spost, const u8 *__restrict__ ncode, const Geo &g,
                                            const u8 *__restrict__ lh, const u32 *__restrict__ cslot,
                                            const typename Op::DElem *__restrict__ E, typename Op::V *R,
                                            XVec4<typename Op::DElem> *sE, XVec4<typename Op::V> *sR, u32 *sB,
                                            typename Op::V *sT, Sync sync) {
  typedef typename Op::DElem Elem;
  typedef typename Op::V V;
  auto top_of = [&](u32 pc) -> V { return xl_trunk(lh[pc]) ? R[cslot[pc]] : op.top(pc); };
  {
    const u32 s0 = cstart[cc], cl = clen[cc];
    const u32 m = cl & XC_LEN;
    const u32 tail = m - 1u - (cl >> 29);
    const u32 ng = (tail >> 2) + 1u;
    const XVec4<Elem> *E4 = (const XVec4<Elem> *)E + (s0 >> 2);
    XVec4<V> *R4 = (XVec4<V> *)R + (s0 >> 2);
    V t = V();
    if (lane == 0) {
      const u32 x = scell[s0 + tail];
      const u32 code = ncode[x];
      const Elem e = E[s0 + tail];
      t = d8_is_dir(code) ? op.dfold(e, top_of(d8_down(g, x, code))) : op.droot(e);
    }
    auto gload = [&](i32 gi, XVec4<Elem> &e, u32 &bits) {
      const u32 gg = gi > 0 ? (u32)gi : 0u;
      e = E4[gg];
      bits = xpost_bits(spost, s0 + 4u * gg) & 0xFu;
    };
    XVec4<Elem> cur;
    u32 cb;
    i32 lo = (i32)ng - 64;
    gload(lo + (i32)lane, cur, cb);
    for (; lo > -64; lo -= 64) {
      {
        // post slots and the slots from the tail on do not change the running value
        const i32 fb = (i32)tail - 4 * (lo + (i32)lane);  // group-relative slot of the tail
        const u32 skip = fb < 4 ? (cb | (fb <= 0 ? 0xFu : (~((1u << fb) - 1u) & 0xFu))) : cb;
        sB[lane] = skip;
        if constexpr (Op::FAST) {  // the speculative fold runs over every slot: skipped ones hold the neutral element
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((skip >> j) & 1u) cur.v[j] = op.dneutral();
        }
      }
      sE[lane] = cur;
      gload(lo - 64 + (i32)lane, cur, cb);
      if (lane == 0) *sT = t;
      sync();
      const int lmin = lo < 0 ? -lo : 0;
      auto exact = [&](V tt) {
        for (int l = 63; l >= lmin; --l) {
          const XVec4<Elem> e = sE[l];
          const u32 skip = sB[l];
          XVec4<V> r;
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            const V f = op.dfold(e.v[j], tt);
            tt = ((skip >> j) & 1u) ? tt : f;
            r.v[j] = tt;
          }
          sR[l] = r;
        }
        return tt;
      };
      if constexpr (!Op::FAST) {
        if (lane == 0) t = exact(t);
      } else {
        V tt = t;
        if (lane == 0) {
#pragma unroll 4
          for (int l = 63; l >= lmin; --l) {
            const XVec4<Elem> e = sE[l];
            XVec4<V> r;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
              tt = op.dfold_fast(e.v[j], tt);
              r.v[j] = tt;
            }
            sR[l] = r;
          }
        }
        sync();
        bool bad = false;
        if ((int)lane >= lmin) {
          const XVec4<Elem> e = sE[lane];
          const XVec4<V> r = sR[lane];
          const u32 skip = sB[lane];
          const V prev = lane < 63u ? sR[lane + 1u].v[0] : *sT;
          bad = (int)(!(skip & 8u) && op.dspecial(e.v[3], prev)) | (int)(!(skip & 4u) && op.dspecial(e.v[2], r.v[3])) |
                (int)(!(skip & 2u) && op.dspecial(e.v[1], r.v[2])) | (int)(!(skip & 1u) && op.dspecial(e.v[0], r.v[1]));
        }
        if (__any((int)bad)) {
          sync();
          if (lane == 0) tt = exact(t);
        }
        t = tt;
      }
      sync();
      const i32 gi = lo + (i32)lane;
      if (gi >= 0) R4[gi] = sR[lane];
    }
  }
}

// the chain is walked from its last cell (slot `tail`, possibly followed by its post slots) upstream;
// post slots are skipped.  The value of the tail comes from its downstream cell, which belongs to a chain
// of a later round (final), or the tail is a pit.
template <class Op>
__global__ void __launch_bounds__(64) k_xtrunk_dscan(Op op, const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                     u32 c0, u32 c1, const u32 *__restrict__ longc, u32 nlong,
                                                     const u32 *__restrict__ scell,
                                                     const u32 *__restrict__ spost, const u8 *__restrict__ ncode, Geo g,
                                                     const u8 *__restrict__ lh, const u32 *__restrict__ cslot,
                                                     const typename Op::DElem *__restrict__ E,
                                                     typename Op::V *R) {
  typedef typename Op::DElem Elem;
  typedef typename Op::V V;
  constexpr int G = XBlk<Elem>::G;
  // the value of a chain's downstream cell: a trunk cell of a later round — final, in chain order — or (row blocks) a
  // halo cell, whose value is given in the raster
  auto top_of = [&](u32 pc) -> V { return xl_trunk(lh[pc]) ? R[cslot[pc]] : op.top(pc); };
  if (blockIdx.x < nlong) {  // a long chain: the whole wave (see k_xtrunk_scan), blocks of 64 groups from the top
    __shared__ XVec4<Elem> sE[64];
    __shared__ XVec4<V> sR[64];
    __shared__ u32 sB[64];
    __shared__ V sT;  // the running value that enters the block (lane 0 holds it)
    xdscan_long(op, longc[blockIdx.x], threadIdx.x, cstart, clen, scell, spost, ncode, g, lh, cslot, E, R, sE, sR, sB, &sT,
                XSyncWG());
    return;
  }
  const u32 c = c0 + (blockIdx.x - nlong) * blockDim.x + threadIdx.x;
  const bool active = c < c1;
  const u32 s0 = active ? cstart[c] : 0u;
  const u32 cl = active ? clen[c] : 0u;
  const u32 m = (cl & XC_LEN) >= XLONG ? 0u : (cl & XC_LEN);
  const u32 tail = m ? m - 1u - (cl >> 29) : 0u;  // slot of the last cell
  const u32 ng = m ? (tail >> 2) + 1u : 0u;        // groups 0 .. ng-1 hold the slots 0 .. tail
  const XVec4<Elem> *E4 = (const XVec4<Elem> *)E + (s0 >> 2);
  XVec4<V> *R4 = (XVec4<V> *)R + (s0 >> 2);
  V t = V();
  if (active && m) {
    const u32 x = scell[s0 + tail];
    const u32 code = ncode[x];
    const Elem e = E[s0 + tail];
    t = d8_is_dir(code) ? op.dfold(e, top_of(d8_down(g, x, code))) : op.droot(e);
  }
  // blocks of G groups, counted from the top: block q holds the groups [lo, lo + G), lo = ng - (q+1) G
  // (groups below 0 do not exist: the top block of a chain may be partial)
  XVec4<Elem> ea[G], eb[G], ec[G], ed[G];
  u32 ba, bb, bc, bd;
  auto load = [&](u32 q, XVec4<Elem>(&e)[G], u32 &bits) {
    const i32 lo = (i32)ng - (i32)((q + 1u) * (u32)G);
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const i32 gi = lo + k;
      e[k] = E4[gi > 0 ? gi : 0];
    }
    bits = xpost_bits(spost, s0 + 4u * (u32)(lo > 0 ? lo : 0));
  };
  auto fold = [&](u32 q, const XVec4<Elem>(&e)[G], u32 bits) {
    const i32 lo = (i32)ng - (i32)((q + 1u) * (u32)G);
    // skip mask of the block's 4 G slots (bit 4 k + j = slot j of group lo + k): post slots, the slots
    // from the tail on (the tail keeps the value computed above) and the slots of groups below 0 do not
    // change the running value
    u32 skip;
    {
      const i32 b0 = lo > 0 ? lo : 0;
      const u32 sh = (u32)(b0 - lo) * 4u;                     // slots of missing groups
      skip = (sh >= 32u ? 0xFFFFFFFFu : ((bits << sh) | ((1u << sh) - 1u)));
      const i32 first_bad = (i32)tail - 4 * lo;               // block-relative slot of the tail
      if (first_bad < 4 * G) skip |= first_bad <= 0 ? 0xFFFFFFFFu : ~((1u << first_bad) - 1u);
    }
    if (Op::FAST) {
      V tt = t;
      bool bad = false;
      XVec4<V> r[G];
#pragma unroll
      for (int k = G - 1; k >= 0; --k) {
#pragma unroll
        for (int j = 3; j >= 0; --j) {
          const Elem x = e[k].v[j];
          const bool sk = ((skip >> (4 * k + j)) & 1u) != 0;
          bad |= !sk && op.dspecial(x, tt);
          const V f = op.dfold_fast(x, tt);
          tt = sk ? tt : f;
          r[k].v[j] = tt;
        }
      }
      if (!__any((int)bad)) {
        t = tt;
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const i32 gi = lo + k;
          if (gi >= 0 && gi < (i32)ng) R4[gi] = r[k];
        }
        return;
      }
    }
#pragma unroll
    for (int k = G - 1; k >= 0; --k) {
      const i32 gi = lo + k;
      XVec4<V> r;
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        const V f = op.dfold(e[k].v[j], t);
        t = ((skip >> (4 * k + j)) & 1u) ? t : f;
        r.v[j] = t;
      }
      if (gi >= 0 && gi < (i32)ng) R4[gi] = r;
    }
  };
  load(0, ea, ba);
  load(1, eb, bb);
  load(2, ec, bc);
  for (u32 q = 0; __any((int)(q * (u32)G < ng)); q += 4) {
    load(q + 3, ed, bd);
    fold(q, ea, ba);
    load(q + 4, ea, ba);
    fold(q + 1, eb, bb);
    load(q + 5, eb, bb);
    fold(q + 2, ec, bc);
    load(q + 6, ec, bc);
    fold(q + 3, ed, bd);
  }
}

// ---- the down-fold of the SHORT chains through LDS (round 6) ----------------------------------------------------------
// The lane-per-chain fold of k_xtrunk_dscan reads its chain with one 16-byte access per lane and trip, strided by the
// chain lengths, clamped repeats included (a chain of 8 slots: ~28 memory instructions).  Here a workgroup takes 256
// consecutive chains — one contiguous run of slots — and moves the run through LDS in chunks of XFuseD::CAP slots, from
// its upper end down (a chain is folded from its last cell upstream): coalesced 16-byte loads of the elements and of the
// post-flag words, the lane that owns a chain folds the part of it that lies in the chunk from LDS (running value in a
// register across chunks), coalesced 16-byte stores of the values.  Long chains: one wave each, in the same launch.
template <class Op>
struct XFuseD {
  static constexpr u32 B = (u32)(sizeof(typename Op::DElem) + sizeof(typename Op::V));
  static constexpr u32 CAP = B <= 8 ? 2048u : (B <= 16 ? 1024u : 512u);
};
template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_dscan_lds(Op op, const u32 *__restrict__ cstart, const u32 *__restrict__ clen,
                                                          u32 c0, u32 c1, const u32 *__restrict__ longc, u32 nlong,
                                                          const u32 *__restrict__ scell, const u32 *__restrict__ spost,
                                                          const u8 *__restrict__ ncode, Geo g, const u8 *__restrict__ lh,
                                                          const u32 *__restrict__ cslot,
                                                          const typename Op::DElem *__restrict__ E, typename Op::V *R) {
  typedef typename Op::DElem Elem;
  typedef typename Op::V V;
  constexpr u32 CAP = XFuseD<Op>::CAP;
  __shared__ XVec4<Elem> sE4[CAP / 4];
  __shared__ XVec4<V> sR4[CAP / 4];
  __shared__ u32 sP[CAP / 32 + 2];
  __shared__ u32 sB[64];
  __shared__ V sT;
  __shared__ u32 s_red[4];
  const u32 tid = threadIdx.x;
  if (blockIdx.x < nlong) {
    if (tid >= 64u) return;
    xdscan_long(op, longc[blockIdx.x], tid, cstart, clen, scell, spost, ncode, g, lh, cslot, E, R, sE4, sR4, sB, &sT, XSyncWave());
    return;
  }
  u32 L = blockIdx.x - nlong;
  if (PFD_XCD_ORDER) {
    const u32 n = gridDim.x - nlong, q = n >> 3, r = n & 7u, k = L & 7u;
    L = k * q + min(k, r) + (L >> 3);
  }
  const u32 c = c0 + L * 256u + tid;
  const bool act = c < c1;
  const u32 a = act ? cstart[c] : 0u;
  const u32 cl = act ? clen[c] : 0u;
  const u32 mfull = cl & XC_LEN;
  const bool islong = mfull >= XLONG;
  const u32 m = islong ? 0u : mfull;
  const u32 pend = a + ((mfull + 3u) & ~3u);       // end of the chain's padded run (long ones too: what a chunk must not enter)
  const u32 tail = m ? a + m - 1u - (cl >> 29) : 0u;  // SLOT of the chain's last cell
  u32 cur = m ? pend : 0u;                        // the part [a, cur) is still to fold; 0: nothing left
  V t = V();
#if XF_ABLATE != 3
  if (m)
#else
  if (m && a == 12345u)
#endif
  {  // the value of the last cell: from its downstream cell — a trunk cell of a later round, final — or a pit's own
    const u32 x = scell[tail];
    const u32 code = ncode[x];
    const Elem e = E[tail];
    if (d8_is_dir(code)) {
      const u32 pc = d8_down(g, x, code);
      t = op.dfold(e, xl_trunk(lh[pc]) ? R[cslot[pc]] : op.top(pc));
    } else {
      t = op.droot(e);
    }
  }
  const u32 first = xwg_min(m ? a : 0xFFFFFFFFu, s_red);  // start of the first short chain of the workgroup
  const bool anylong = xwg_min(islong ? 0u : 1u, s_red) == 0u;
  for (;;) {
    const u32 top = ~xwg_min(~cur, s_red);  // maximum
    if (top == 0u) break;
    // the chunk [lo, top): up to CAP slots, not into a long chain, not below the chains of this workgroup
    u32 lo = max(top > CAP ? top - CAP : 0u, first);
    if (anylong) lo = max(lo, ~xwg_min((islong && pend <= top) ? ~pend : 0xFFFFFFFFu, s_red));
    const u32 cnt = top - lo, w0 = lo >> 5;
    {
      const XVec4<Elem> *E4 = reinterpret_cast<const XVec4<Elem> *>(E) + (lo >> 2);
      for (u32 q = tid; q < (cnt >> 2); q += 256u) sE4[q] = E4[q];
      if (tid <= ((top - 1u) >> 5) - w0) sP[tid] = spost[w0 + tid];
    }
    __syncthreads();
#if XF_ABLATE == 4
    if (cur > lo && cur <= top && cur != 0u) cur = max(a, lo) == a ? 0u : max(a, lo);
#endif
    if (cur > lo && cur <= top && cur != 0u) {
      const u32 stop = max(a, lo);
      for (u32 sg = cur - 4u;; sg -= 4u) {  // group of the slots sg .. sg + 3
        const u32 q = sg - lo;
        const XVec4<Elem> ev = sE4[q >> 2];
        u32 skip = (sP[(sg >> 5) - w0] >> (sg & 31u)) & 0xFu;  // post slots; and the slots from the tail on keep the value
        if (sg + 3u >= tail) skip |= sg >= tail ? 0xFu : (~((1u << (tail - sg)) - 1u) & 0xFu);
        XVec4<V> r;
        V tt = t;
        bool bad = !Op::FAST;
        if (Op::FAST) {
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            const bool sk = ((skip >> j) & 1u) != 0;
            bad |= !sk && op.dspecial(ev.v[j], tt);
            const V f = op.dfold_fast(ev.v[j], tt);
            tt = sk ? tt : f;
            r.v[j] = tt;
          }
        }
        if (bad) {
          tt = t;
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            const V f = op.dfold(ev.v[j], tt);
            tt = ((skip >> j) & 1u) ? tt : f;
            r.v[j] = tt;
          }
        }
        t = tt;
        sR4[q >> 2] = r;
        if (sg == stop) break;
      }
      cur = stop == a ? 0u : stop;
    }
    __syncthreads();
    XVec4<V> *R4 = reinterpret_cast<XVec4<V> *>(R) + (lo >> 2);
    for (u32 q = tid; q < (cnt >> 2); q += 256u) R4[q] = sR4[q];
  }
}

template <class Op>
__global__ void __launch_bounds__(256) k_xtrunk_dscatter(Op op, const u32 *__restrict__ scell,
                                                         const uint16_t *__restrict__ sinfo, u32 s0, u32 s1,
                                                         const typename Op::V *__restrict__ R) {
  const u32 s = s0 + pfd_block_1d() * blockDim.x + threadIdx.x;
  if (s >= s1) return;
  const u32 info = sinfo[s];  // (three independent loads, then the decision)
  const u32 x = scell[s];
  const typename Op::V v = R[s];
  if (!(info & XS_POST)) op.store(x, v);
}

// ---- leaves, down -------------------------------------------------------------------------------
// LDS image: the values with a 1-cell ring (the downstream cell of a leaf may be a trunk cell next door:
// trunk values are final when this kernel runs) and, per own cell, the element of the operation (what
// apply() would read from memory: gathered up front, quad by quad).  Writes every own cell once.
// (LDS per workgroup decides how many tiles a CU overlaps, and the step loop is latency: keep it small.)
#define XHW (XT + 2)
#ifndef XORD_REGS
#define XORD_REGS 0  // k_xtile_down, 1: the leaf list in registers instead of a 16 KB LDS copy with precomputed ring indices —
                     // measured twice (round 4: per-lane range tests; round 6: uniform slices behind scalar branches,
                     // profiles/r06_ab_ord_regs.txt) and slower both times although a tile more fits per CU: off
#endif
// cells whose final value is in place before the tile kernel runs: trunk cells, and the halo cells of a row block
__device__ __forceinline__ bool xl_given(u32 m) { return xl_trunk(m) || m == XL_HALO; }
template <class Op>
__global__ void __launch_bounds__(256) k_xtile_down(Op op, XTileArgs a) {
  typedef typename Op::V V;
  typedef typename Op::DTile Elem;
  // INPL: the tile image of a leaf has the type of the result and waits in the leaf's own word of val until the
  // leaf is computed (trunk cells and the ring hold final values from the start) — no second array, one more
  // workgroup per CU
  constexpr bool INPL = std::is_same<Elem, V>::value;
  __shared__ __attribute__((aligned(16))) V val[XHW * XHW];
  __shared__ __attribute__((aligned(16))) Elem De[INPL ? 4 : XTC];
  // per leaf, in step order: own cell (12 bits) | ring index of its downstream cell << 12 (13 bits) | pit << 25
  // (looked up once per leaf here instead of once per step through the cell's code)
#if !XORD_REGS
  __shared__ __attribute__((aligned(16))) u32 ord[XTC];
#endif
  __shared__ u32 F[XTC / 32];  // one flag bit per cell, for operations whose element needs one (HAND: drain)
  __shared__ uint16_t off[XOFF];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_, tr = by_;
  const size_t tile = (size_t)tr * a.ntc + tc;
  const i64 r0 = (i64)tr * XT, c0 = (i64)tc * XT;
  if (tid < XOFF) off[tid] = a.toff[tile * XOFF + tid];
  if (tid < XTC / 32) F[tid] = 0;
  // a value that is in place before this kernel runs: a trunk cell's in chain order (R through cslot: the rounds do
  // not scatter), a halo cell's (row blocks) in the raster.  Loads are unconditional, the mark selects.
  const V *__restrict__ Rv = (const V *)a.R;
  auto given = [&](u32 x) -> V {
    const bool tk = xl_trunk(a.lh[x]);
    const u32 cs = a.cslot[x];
    const V rv = Rv[tk ? cs : 0u];
    const V ov = op.top(x);
    return tk ? rv : ov;
  };
  // the ring: 2 x 66 + 2 x 64 cells of the neighbouring tiles (trunk cells there are final)
  for (u32 i = tid; i < 4u * XT + 4u; i += 256u) {
    int rr, cc;
    if (i < (u32)XHW) {
      rr = -1, cc = (int)i - 1;
    } else if (i < 2u * XHW) {
      rr = XT, cc = (int)(i - XHW) - 1;
    } else if (i < 2u * XHW + XT) {
      rr = (int)(i - 2u * XHW), cc = -1;
    } else {
      rr = (int)(i - 2u * XHW - XT), cc = XT;
    }
    const i64 gr = r0 + rr, gc = c0 + cc;
    V v = V();
    if (gr >= 0 && gr < (i64)a.nrow && gc >= 0 && gc < (i64)a.ncol) v = given((u32)(gr * (i64)a.ncol + gc));
    val[(rr + 1) * XHW + cc + 1] = v;
  }
  __syncthreads();  // (F is cleared)
  u32 mycodes[4];
  auto general_init = [&](u32 (&mycodes)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const int lr = l0 >> 6, lc = l0 & 63;
    const i64 gr = r0 + lr, gc = c0 + lc;
    u32 c4 = D8_MV * 0x01010101u;
    u32 l4 = XL_NODATA * 0x01010101u;  // the marks of the quad: leaf steps, trunk, halo (cells off the raster: nothing)
    V v[4] = {V(), V(), V(), V()};
    if (gr < (i64)a.nrow && gc + 3 < (i64)a.ncol) {
      const u32 g0 = (u32)(gr * (i64)a.ncol + gc);
      __builtin_memcpy(&c4, a.ncode + g0, 4);
      __builtin_memcpy(&l4, a.lh + g0, 4);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (xl_given((l4 >> (8 * b)) & 0xFFu)) v[b] = given(g0 + (u32)b);
    } else if (gr < (i64)a.nrow) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (gc + b < (i64)a.ncol) {
          const u32 g = (u32)(gr * (i64)a.ncol + gc + b);
          c4 = (c4 & ~(0xFFu << (8 * b))) | ((u32)a.ncode[g] << (8 * b));
          l4 = (l4 & ~(0xFFu << (8 * b))) | ((u32)a.lh[g] << (8 * b));
          if (xl_given((u32)a.lh[g])) v[b] = given(g);
        }
      }
    }
    mycodes[j] = c4;
    u32 fl = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u32 code = (c4 >> (8 * b)) & 0xFFu;
      const bool leaf = ((l4 >> (8 * b)) & 0xFFu) <= (u32)XCAP;
      Elem e = Elem();
      bool f = false;
      if (code != D8_MV && (!INPL || leaf)) e = op.dtile((u32)(gr * (i64)a.ncol + gc + b), code, f);
      fl |= f ? 1u << b : 0u;
      if (INPL) {
        if (leaf) __builtin_memcpy(&v[b], &e, sizeof(V));  // (Elem is V)
      } else {
        De[l0 + b] = e;
      }
      val[(lr + 1) * XHW + lc + b + 1] = v[b];
    }
    if (Op::DTILE_FLAG && fl) atomicOr(&F[l0 >> 5], fl << (l0 & 31u));
  }
  };
  if constexpr (Op::DTILE4) {
    // quads inside the raster: three load phases (codes + leaf steps; everything that depends on them, unconditionally;
    // the final values of quads holding a trunk cell), each with the loads of all four quads in flight together
    if (r0 + XT <= (i64)a.nrow && c0 + XT <= (i64)a.ncol) {
      u32 c4s[4], l4s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 l0 = 4u * tid + 1024u * j;
        const u32 g0 = (u32)((r0 + (l0 >> 6)) * (i64)a.ncol + c0 + (l0 & 63));
        __builtin_memcpy(&c4s[j], a.ncode + g0, 4);
        __builtin_memcpy(&l4s[j], a.lh + g0, 4);
      }
      typename Op::DQuad dq[4];
      V vq[4][4];
      uint4 cs4[4];
      u32 tmask[4], hmask[4];  // trunk cells / halo cells of the quad
      const bool by_list = a.tlist != nullptr;  // (uniform) the trunk values come in through the tile's dense list below
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 l0 = 4u * tid + 1024u * j;
        const u32 g0 = (u32)((r0 + (l0 >> 6)) * (i64)a.ncol + c0 + (l0 & 63));
        op.dtile4_load(g0, c4s[j], dq[j]);
        tmask[j] = hmask[j] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const u32 mk = (l4s[j] >> (8 * b)) & 0xFFu;
          tmask[j] |= xl_trunk(mk) ? 1u << b : 0u;
          hmask[j] |= mk == XL_HALO ? 1u << b : 0u;
        }
        cs4[j] = make_uint4(0u, 0u, 0u, 0u);
        if (tmask[j] && !by_list) __builtin_memcpy(&cs4[j], a.cslot + g0, 16);  // (slot numbers of the quad's trunk cells)
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 l0 = 4u * tid + 1024u * j;
        const u32 g0 = (u32)((r0 + (l0 >> 6)) * (i64)a.ncol + c0 + (l0 & 63));
#pragma unroll
        for (int b = 0; b < 4; ++b) vq[j][b] = V();
        if (tmask[j] && !by_list) {  // trunk values from chain order: four loads, the marks select
          const u32 cs[4] = {cs4[j].x, cs4[j].y, cs4[j].z, cs4[j].w};
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const V rv = Rv[(tmask[j] >> b) & 1u ? cs[b] : 0u];
            if ((tmask[j] >> b) & 1u) vq[j][b] = rv;
          }
        }
        if (hmask[j]) {  // (row blocks: the given values of halo cells are in the raster)
          V hv[4];
          op.top4(g0, hv);
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if ((hmask[j] >> b) & 1u) vq[j][b] = hv[b];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 l0 = 4u * tid + 1024u * j;
        const int lr = l0 >> 6, lc = l0 & 63;
        const u32 c4 = c4s[j], l4 = l4s[j];
        mycodes[j] = c4;
        u32 fl = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const u32 code = (c4 >> (8 * b)) & 0xFFu;
          const bool leaf = ((l4 >> (8 * b)) & 0xFFu) <= (u32)XCAP;
          Elem e = Elem();
          bool f = false;
          if (code != D8_MV && (!INPL || leaf)) e = op.dtile4_get(dq[j], b, f);
          fl |= f ? 1u << b : 0u;
          V v = vq[j][b];
          if (INPL) {
            if (leaf) __builtin_memcpy(&v, &e, sizeof(V));  // (Elem is V)
          } else {
            De[l0 + b] = e;
          }
          if (!(by_list && ((tmask[j] >> b) & 1u))) val[(lr + 1) * XHW + lc + b + 1] = v;  // (a trunk cell's word: the list loop's)
        }
        if (Op::DTILE_FLAG && fl) atomicOr(&F[l0 >> 5], fl << (l0 & 31u));
      }
      if (by_list) {
        // the values of the tile's trunk cells, final in chain order: 8 contiguous bytes of the list per trunk cell instead
        // of a 16-byte quad of cslot wherever a quad holds one (one useful word per sector along a river)
        const u32 lb = a.tl_off[tile], le = a.tl_off[tile + 1];
        for (u32 i0 = lb; i0 < le; i0 += 1024u) {
          uint2 en[4];
          V rv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u32 i = i0 + tid + 256u * (u32)k;
            en[k] = a.tlist[i < le ? i : lb];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) rv[k] = Rv[en[k].x];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u32 i = i0 + tid + 256u * (u32)k;
            const u32 l = en[k].y & 0xFFFu;
            if (i < le) val[((l >> 6) + 1u) * XHW + (l & 63u) + 1u] = rv[k];
          }
        }
      }
    } else {
      general_init(mycodes);
    }
  } else {
    general_init(mycodes);
  }
  const u32 total = off[XOFF - 1];
#if XORD_REGS
  // The leaf list stays in REGISTERS: list position p belongs to thread p mod 256, so a thread holds positions tid + 256 m,
  // m = 0 .. 15, as sixteen u16 entries in eight registers — loaded once, coalesced — and the positions of a step
  // [off[s], off[s + 1]) lie in the slices m = off[s] >> 8 .. (off[s + 1] - 1) >> 8: a range that is UNIFORM over the
  // workgroup, so the sixteen copies of the step body sit behind scalar branches (round 4 walked all sixteen with a per-lane
  // range test and lost 3 %).  16 KB of LDS less: HAND runs four tiles per CU instead of three, float32 down-sweeps six
  // instead of four.  The ring index of a leaf's downstream cell is recomputed from its 3-bit direction per use.
  // RESULT (same box, 30000^2 / C5 shape): accuflux down 9.6 -> 10.4 / 21.9 -> 23.4 ms, HAND 15.4 -> 15.4 / 34.0 -> 34.9 ms.
  u32 en[8];
#pragma unroll
  for (int m = 0; m < 16; m += 2) {
    const u32 e0 = a.tord[tile * XTC + tid + 256u * (u32)m], e1 = a.tord[tile * XTC + tid + 256u * (u32)(m + 1)];  // (zeros past `total`)
    en[m >> 1] = e0 | (e1 << 16);
  }
  __syncthreads();
  int last = 0;
  for (int s = 1; s < XOFF - 1; ++s) last = off[s] < total ? s : last;
  for (int s = last; s >= 0; --s) {
    const u32 b = (u32)__builtin_amdgcn_readfirstlane((int)off[s]), e = (u32)__builtin_amdgcn_readfirstlane((int)off[s + 1]);
    if (e > b) {
      const u32 mlo = b >> 8, mhi = (e - 1u) >> 8;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if ((u32)m < mlo || (u32)m > mhi) continue;  // (scalar: the whole workgroup skips the slice)
        const u32 j = tid + 256u * (u32)m;
        if (j >= b && j < e) {
          const u32 w = (en[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
          const u32 x = w & 0xFFFu, root = w >> 15;
          const int k = (int)((w >> 12) & 7u);
          const int dr = root ? 0 : (int)((0x101A9u >> (2 * k)) & 3u) - 1, dc = root ? 0 : (int)((0x1901Au >> (2 * k)) & 3u) - 1;
          const u32 own = ((x >> 6) + 1) * XHW + (x & 63u) + 1;
          const V pv = val[(u32)((int)own + dr * (int)XHW + dc)];
          Elem el;
          if (INPL)
            __builtin_memcpy(&el, &val[own], sizeof(Elem));
          else
            el = De[x];
          const bool f = Op::DTILE_FLAG ? ((F[x >> 5] >> (x & 31u)) & 1u) != 0 : false;
          val[own] = root ? op.dtroot(el, f) : op.dtfold(el, f, pv);
        }
      }
    }
    __syncthreads();
  }
#else
  {
    uint2 o4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_memcpy(&o4[j], a.tord + tile * XTC + 4u * tid + 1024u * j, 8);  // (zeros past `total`)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 e4[4] = {o4[j].x & 0xFFFFu, o4[j].x >> 16, o4[j].y & 0xFFFFu, o4[j].y >> 16};
      u32 w4[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const u32 x = e4[b] & 0xFFFu, root = e4[b] >> 15;
        const int k = (int)((e4[b] >> 12) & 7u);
        const int pr = (int)(x >> 6) + (root ? 0 : d8_dr(k)), pc = (int)(x & 63u) + (root ? 0 : d8_dc(k));
        w4[b] = x | ((u32)((pr + 1) * XHW + pc + 1) << 12) | (root << 25);
      }
      *(uint4 *)&ord[4u * tid + 1024u * j] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
  }
  __syncthreads();
  int last = 0;
  for (int s = 1; s < XOFF - 1; ++s) last = off[s] < total ? s : last;
  for (int s = last; s >= 0; --s) {
    const u32 b = off[s], e = off[s + 1];
    for (u32 j = b + tid; j < e; j += 256u) {
      const u32 w = ord[j];
      const u32 x = w & 0xFFFu;
      const V pv = val[(w >> 12) & 0x1FFFu];
      const u32 own = ((x >> 6) + 1) * XHW + (x & 63u) + 1;
      Elem el;
      if (INPL)
        __builtin_memcpy(&el, &val[own], sizeof(Elem));
      else
        el = De[x];
      const bool f = Op::DTILE_FLAG ? ((F[x >> 5] >> (x & 31u)) & 1u) != 0 : false;
      val[own] = (w >> 25) ? op.dtroot(el, f) : op.dtfold(el, f, pv);
    }
    __syncthreads();
  }
#endif
  u32 wmask = 0;  // (XWatch) bit 4 j + b: the value stored for that cell is one the operation watches
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const int lr = l0 >> 6, lc = l0 & 63;
    const i64 gr = r0 + lr, gc = c0 + lc;
    if (gr >= (i64)a.nrow) continue;
    const u32 c4 = mycodes[j];
    V v[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      v[b] = val[(lr + 1) * XHW + lc + b + 1];
      if (((c4 >> (8 * b)) & 0xFFu) == D8_MV && gc + b < (i64)a.ncol) v[b] = op.dnodata((u32)(gr * (i64)a.ncol + gc + b));
      if constexpr (XWatch<Op>::value)
        wmask |= (gc + b < (i64)a.ncol && op.watched((c4 >> (8 * b)) & 0xFFu, v[b])) ? 1u << (4 * j + b) : 0u;
    }
    if (gc + 3 < (i64)a.ncol) {
      op.dstore4((u32)(gr * (i64)a.ncol + gc), v);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gc + b < (i64)a.ncol) op.store((u32)(gr * (i64)a.ncol + gc + b), v[b]);
    }
  }
  if constexpr (XWatch<Op>::value) {
    if (op.watch_cnt) {  // (uniform) the watched cells of the tile join the list: ONE global atomic per tile that holds any
      __shared__ u32 s_wt[4], s_wbase;
      const u32 lane = tid & 63u, wave = tid >> 6;
      const u32 c = (u32)__popc(wmask);
      u32 incl = c;
      for (int o = 1; o < 64; o <<= 1) {
        const u32 y = (u32)__shfl_up((int)incl, o);
        if (lane >= (u32)o) incl += y;
      }
      if (lane == 63u) s_wt[wave] = incl;
      __syncthreads();
      if (tid == 0) {
        const u32 tot = s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
        s_wbase = tot ? (u32)atomicAdd(op.watch_cnt, (unsigned long long)tot) : 0u;  // (the count is exact beyond the capacity too)
      }
      __syncthreads();
      if (c) {
        u32 pos = s_wbase + incl - c;
        for (u32 w = 0; w < wave; ++w) pos += s_wt[w];
        u32 m = wmask;
        while (m) {
          const u32 bit = (u32)__ffs((int)m) - 1u;
          m &= m - 1u;
          const u32 l0 = 4u * tid + 1024u * (bit >> 2);
          const u32 cell = (u32)((r0 + (l0 >> 6)) * (i64)a.ncol + c0 + (l0 & 63) + (bit & 3u));
          if (pos < op.watch_cap) op.watch_list[pos] = cell;
          ++pos;
        }
      }
    }
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
  XTileArgs a{(u32)h->nrow, (u32)h->ncol, p->ntc, p->lh, p->kids, h->ncode, p->tord, p->toff, p->cslot, R.p};
  a.tlist = p->tlist, a.tl_off = p->tl_off;
  if (!xlist_on()) a.tlist = nullptr, a.tl_off = nullptr;
  // The rounds start with the main stems and their largest tributaries (the last two rounds of the layout): a few
  // thousand long chains, folded serially — 0.7 ms each for HAND at 30000 x 30000 with most of the chip idle.  Their
  // operands are gathered in chain order first (a few per cent of the slots); the raster-order gather of everything
  // else (k_xtrunk_demit: bandwidth) runs BESIDE those two rounds on the handle's second stream.
  int bsplit = xplan_tail_split(p);
  if (bsplit >= 0 && pfd_aux_stream(h) != PFD_OK) bsplit = -1;
  XStream2Guard guard2{h};  // (declared after E / R: runs before they are released)
  if (bsplit >= 0) {
    const u32 s_split = (u32)p->b_slot[bsplit];
    HIPCHK(hipEventRecord(h->ev_fork, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    if (xlist_on())
      k_xtrunk_demit_list<Op, true><<<dim3(p->ntc, p->ntr), 256, 0, h->stream2>>>(op, a, E.as<Elem>(), s_split);
    else
      k_xtrunk_demit<Op, true><<<dim3(p->ntc, p->ntr), 256, 0, h->stream2>>>(op, a, E.as<Elem>(), s_split);
    HIPCHK(hipEventRecord(h->ev_join, h->stream2));
    k_xtrunk_dpre<Op><<<cdiv_u32((u32)p->nslot - s_split, 256), 256, 0, h->stream>>>(op, p->scell, p->sinfo, h->ncode, s_split,
                                                                                  (u32)p->nslot, E.as<Elem>());
    launches += 2;
  } else if (p->nslot) {  // what the folds read from memory, for every trunk cell at once (the rounds only fold)
    if (xlist_on())
      k_xtrunk_demit_list<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a, E.as<Elem>());
    else
      k_xtrunk_demit<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a, E.as<Elem>());
    ++launches;
  }
  bool joined = bsplit < 0;
  // (PFD_DSCAN_LDS=1: the LDS form for every operation — the tests run all of them through it)
  const bool lds_scan = !pfd_knob("PFD_DSCAN_GLOBAL") && (Op::DSCAN_LDS || pfd_knob("PFD_DSCAN_LDS"));
  const u32 fuse_min = xfuse_min_chains();  // (small rounds keep the lane-per-chain kernel: HAND at 10000^2 2.24 -> 2.71 ms otherwise)
  for (int b = 31; b >= 0; --b) {
    const u32 c0 = (u32)p->b_chain[b], c1 = (u32)p->b_chain[b + 1];
    if (c1 == c0) continue;
    if (!joined && b < bsplit) {  // (from here on the rounds need the raster-order gather)
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
      joined = true;
    }
    const u32 nl = (u32)(p->b_long[b + 1] - p->b_long[b]);
    // the short chains through LDS (k_xtrunk_dscan_lds) where a round is mostly short chains and does not run beside the
    // raster-order gather; PFD_DSCAN_GLOBAL: the lane-per-chain fold from global memory everywhere
    if (lds_scan && c1 - c0 >= fuse_min && (u64)nl * 16u <= (u64)(c1 - c0) && (bsplit < 0 || b < bsplit)) {
      k_xtrunk_dscan_lds<Op><<<nl + cdiv_u32(c1 - c0, 256), 256, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1,
                                                                                p->longc + p->b_long[b], nl, p->scell, p->spost,
                                                                                h->ncode, h->geo, p->lh, p->cslot, E.as<Elem>(),
                                                                                R.as<V>());
      ++launches;
      continue;
    }
    k_xtrunk_dscan<Op><<<nl + cdiv_u32(c1 - c0, 64), 64, 0, h->stream>>>(op, p->cstart, p->clen, c0, c1,
                                                                         p->longc + p->b_long[b], nl, p->scell, p->spost,
                                                                         h->ncode, h->geo, p->lh, p->cslot, E.as<Elem>(),
                                                                         R.as<V>());
    ++launches;  // (no gather, no scatter: the rounds and the tile pass read the trunk values in chain order)
  }
  if (!joined) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  k_xtile_down<Op><<<dim3(p->ntc, p->ntr), 256, 0, h->stream>>>(op, a);
  KCHK();
  pfd_seg_end(h, launches);
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}
