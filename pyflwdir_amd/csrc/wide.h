// wide.h — ORDER-FREE upstream area in 64-bit fixed point (included at the end of tiled.hip).
//
// FlwdirRaster.upstream_area(unit != "cell") on a lat/lon grid is a float64 accumulation of cell areas (reference
// pyflwdir/pyflwdir.py:770-801 + streams.accuflux, streams.py:15-41).  Float addition is not associative, so the exact
// (default) form folds every sum in the order of the reference's serial loop (exact.hip: a plan of 44 ms at 30000^2
// and a sweep of 17 ms).  This file is the OPT-IN alternative (`exact=False` in the front end): the areas are
// quantised once to 64-bit fixed point — area[r] * 2^s, s the largest power that keeps the sum over the whole raster
// below 2^64; a cell gets the integer part plus its share of the fraction (w_cell: Bresenham along the row, so that
// any run of consecutive cells of a row is off by less than one unit) — and accumulated as INTEGERS on the LDS-tiled pointer-doubling engine of this file's
// includer.  Integer addition is associative: the result is the same in any execution order (run to run, tile order,
// and it would be across row blocks), every cell's value is the exact sum of the quantised areas of its upstream
// cells, and what separates it from the real-number sum is the quantisation alone:
//     |result - sum| < (row runs in the upstream set) * 2^-s  <=  N_up(x) * 2^-s,   2^-s <= (sum of all areas) / 2^63
// i.e. relative <= n_cells / 2^63 * mean / min area: 1.3e-10 at 30000^2, 6e-10 at 2^32 cells, reached at cells with a
// handful of upstream cells (one quantum against one cell's area); a large basin is off by far less (measured:
// see profiles/r06f_wide_probe.txt) plus one float64 rounding of the final value.  The reference's own serial float64 sum carries up to N * 2^-53
// relative error (1e-7 worst case at 9e8 terms, ~3e-12 typical) — the two agree far inside the north star's 1e-6.
//
// The pass reuses everything of the count pass that does not depend on the values:
//   1. TiledRun::phase_a() — the u32 count's local tile pass and exit-graph solve, unchanged: normalises a deferred
//      handle, detects cycles, and leaves the STRUCTURE: per slot the record word, the exit lists of the supertiles
//      with every exit's root (R2L) and the ids / first hops of the super-exits (sxidL, sx_slot, sx_n1);
//   2. k_wtile_local   — per tile, the sum of the quantised areas per exit (pointer jumping again, u64 LDS counters);
//   3. k_wsuper_up     — no doubling: an exit's root inside its supertile is known, its sum goes there;
//   4. k_wlink3 / k_wround x R / k_wsx_totals — the super-exits as ONE flat forest in global memory (u64 values);
//   5. k_wsuper_down   — the value-carrying doubling of a supertile's exits in LDS (u64: 80 KB, one per CU);
//   6. k_wtile_final   — the value-carrying doubling of a tile (u64: 41 KB, three per CU, 512 threads), float64 result.
// Anything the u64 forms cannot hold (a supertile with more than SCAP exits, a raster with cycles, row blocks) is
// reported as "not taken" and the front end runs the exact form.
#pragma once

struct WideArgs {
  const u8 *ncode;
  u32 nrow, ncol, ntr, ntc, nstc;
  const u64 *wrow;   // [nrow] quantised area of a cell of row r: the integer part ...
  const u32 *wfrac;  // [nrow] ... and the fraction (x 2^32) the columns share out between them (w_cell)
  const u32 *xrec;   // [nslots] records of the count pass
  u64 *xT;           // [nslots] tile-local sum of the exit on the slot
  const u64 *xtot;   // [nslots] (final) total of the exit on the slot
  double *out;
  double inv;        // 2^-s
  u32 tr_lo = 0, tr_hi = 0, tc_lo = 0, tc_hi = 0;  // the interior rectangle of the count pass (k_wtile_local<true>: the frame around it)
};

// the branch-free initial pointer of the cell in register slot s of quad (lr, lc0) — tile_body's general form
__device__ __forceinline__ u32 w_init_ptr(u32 c, u32 l, int lr, int lc) {
  const int k = (int)__builtin_ctz(c | 0x100u);
  const int dr = (int)((0x101A9u >> (2 * k)) & 3u) - 1;
  const int dc = (int)((0x1901Au >> (2 * k)) & 3u) - 1;
  const int nr = lr + dr, nc = lc + dc;
  const bool go = d8_is_dir(c) && (unsigned)nr < TS && (unsigned)nc < TS;
  return go ? PHYS((u32)(nr * TS + nc)) << 1 : ((l << 1) | PDONE);
}

// FRAME: only the tiles of the frame around the interior rectangle (1-D grid) — the interior tiles' sums came out of the
// count pass's own local kernel (k_tile_local_fast<.., WIDE>)
template <bool FRAME>
__global__ void __launch_bounds__(256) k_wtile_local(WideArgs a) {
  __shared__ __attribute__((aligned(16))) u64 A[PSL * 4];  // 4 replicas per perimeter slot, picked by lane
  __shared__ __attribute__((aligned(16))) uint16_t P[TCELLS];
  __shared__ __attribute__((aligned(16))) u8 code[HW * CP];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][4];
  const u32 tid = threadIdx.x;
  u32 tc, tr;
  if (FRAME) frame_tile(blockIdx.x, a.ntr, a.ntc, a.tr_lo, a.tr_hi, a.tc_lo, a.tc_hi, &tr, &tc);
  else pfd_tile_of_block(&tc, &tr);
  const u32 sbase = sslot_base(tr, tc, a.nstc);
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  {
    u32 v[5];
    stage_load_auto(a.ncode, a.nrow, a.ncol, r0, c0, tid, v);
    stage_store(code, tid, v);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) A[tid + 256u * k] = 0;
  const u32 rec = a.xrec[sbase + tid];
  __syncthreads();
  const u32 qs = (tid >> 3) & 3u;
  u32 pc[QPT * 4];
  u32 live = 0;
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const int lr = l0 >> 6, lc0 = l0 & 63;
    const u32 c4 = *(const u32 *)&CODE(lr, lc0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32 b = (u32)s ^ qs;
      pc[4 * j + s] = w_init_ptr((c4 >> (8 * b)) & 0xFFu, l0 + (u32)s, lr, lc0 + (int)b);
    }
    *(uint2 *)&P[l0] = make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
    if (!(pc[4 * j + 0] & pc[4 * j + 1] & pc[4 * j + 2] & pc[4 * j + 3] & PDONE)) live |= 1u << j;
  }
  __syncthreads();
  // gather-only pointer jumping, two jumps per round (a root's word names the root: jumping from it stays there)
  for (int round = 0; round < MAXROUNDS_TILE; ++round) {
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      if (live & (1u << j)) {
        u32 q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = *(const uint16_t *)((const u8 *)P + (pc[4 * j + b] & 0x1FFEu));
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = *(const uint16_t *)((const u8 *)P + (q[b] & 0x1FFEu));
#pragma unroll
        for (int b = 0; b < 4; ++b) pc[4 * j + b] = q[b];
        if (q[0] & q[1] & q[2] & q[3] & PDONE) live &= ~(1u << j);
        *(uint2 *)&P[4u * tid + 1024u * j] = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
      }
    }
    if (!fx_vote(s_flag, round, tid, live != 0u)) break;
  }
  // every valid cell adds the weight of its row to the counter of its exit (cells of a quad share their row and
  // mostly their exit: combined in registers first)
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const u32 l0 = 4u * tid + 1024u * j;
    const int lr = l0 >> 6, lc0 = l0 & 63;
    const u32 c4 = *(const u32 *)&CODE(lr, lc0);
    const u32 grow = min((u32)(r0 + lr), a.nrow - 1u);
    const u64 wr = a.wrow[grow];
    const u32 wf = a.wfrac[grow];
    u64 w[4];
    u32 r[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32 b = (u32)s ^ qs;
      const u32 c = (c4 >> (8 * b)) & 0xFFu;
      w[s] = c != D8_MV ? w_cell(wr, wf, (u32)c0 + (u32)lc0 + b) : 0ull;
      r[s] = pc[4 * j + s];
    }
    {
      const bool e10 = r[1] == r[0], e20 = r[2] == r[0], e21 = r[2] == r[1], e30 = r[3] == r[0], e31 = r[3] == r[1], e32 = r[3] == r[2];
      w[0] += (e10 ? w[1] : 0ull) + (e20 ? w[2] : 0ull) + (e30 ? w[3] : 0ull);
      w[1] = e10 ? 0ull : w[1] + ((!e20 && e21) ? w[2] : 0ull) + ((!e30 && e31) ? w[3] : 0ull);
      w[2] = (e20 || e21) ? 0ull : w[2] + ((!e30 && !e31 && e32) ? w[3] : 0ull);
      w[3] = (e30 || e31 || e32) ? 0ull : w[3];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (!w[s] || r[s] < PDONE) continue;  // (unsaturated: on or upstream of a cycle — the count pass has noticed)
      const u32 L = PHYS((r[s] & 0x1FFEu) >> 1);  // logical index of the root; its perimeter slot as in tile_body
      const u32 rr = L >> 6, rc = L & 63u;
      const bool tb = ((rr + 1u) & 62u) == 0u, lrc = ((rc + 1u) & 62u) == 0u;
      const u32 s_tb = rc + ((rr + 1u) & 64u);
      const u32 s_lr = 127u + rr + (((rc + 1u) & 64u) - (((rc + 1u) >> 5) & 2u));
      if (tb || lrc) atomicAdd((unsigned long long *)&A[(tb ? s_tb : s_lr) * 4u + (tid & 3u)], (unsigned long long)w[s]);
    }
  }
  __syncthreads();
  const bool isexit = tid < NPERIM && (rec & 0xFFu) != XR_NONE;
  a.xT[sbase + tid] = isexit ? A[4u * tid] + A[4u * tid + 1] + A[4u * tid + 2] + A[4u * tid + 3] : 0ull;
}

// the sum of every exit goes to the super-exit its path leaves the supertile through (roots: R2L of the count pass).
// One workgroup per supertile, the sums per root in LDS (global atomics: 19 M of them at 30000^2 cost 1.5 ms), then one
// plain store per super-exit — every super-exit belongs to exactly one supertile.
template <u32 CAP, u32 NT>
__global__ void __launch_bounds__(NT) k_wsuper_up(SuperArgs s, const u64 *__restrict__ xT64, u64 *__restrict__ T3w) {
  constexpr int DPT = CAP / NT;
  __shared__ u64 T[CAP];
  const u32 tid = threadIdx.x, st = blockIdx.x, base = st << SSHIFT;
  const u32 n = min(s.scount[st], CAP);
  u32 root[DPT];
  u64 v[DPT];
#pragma unroll 4
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    T[e] = 0;
    root[k] = NONE32, v[k] = 0;
    if (e < n) {
      root[k] = s.R2L[base + e];
      v[k] = xT64[base + ((u32)s.xl_slot[base + e] & (SSL - 1))];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < DPT; ++k)
    if (v[k]) atomicAdd((unsigned long long *)&T[root[k]], (unsigned long long)v[k]);
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    if (e < n && root[k] == e) {  // a root: a super-exit, or an exit whose path ends inside the supertile
      const u32 id = s.sxidL[base + e];
      if (id != NONE32) T3w[id] = T[e];
    }
  }
}

// level 3, one flat forest: super-exit -> the super-exit its flow leaves the next supertile through
__global__ void __launch_bounds__(256) k_wlink3(SuperArgs s, u32 n3, u32 *__restrict__ J, u64 *__restrict__ Tnext) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n3) return;
  Tnext[k] = 0;
  u32 j = k | XDONE;
  if (sx_active(s, k)) {
    const u32 pos = s.sx_n1[k];  // (a list position since k_link3 of the count pass)
    if (pos != NONE32) {
      const u32 id = s.sxidL[(pos & ~(u32)(SSL - 1)) + s.R2L[pos]];
      if (id != NONE32) j = id;
    }
  }
  J[k] = j;
}
// one doubling round over the flat forest (k_coarse_round with 64-bit values)
__global__ void __launch_bounds__(256) k_wround(const u64 *__restrict__ Told, u64 *__restrict__ Tnew, u64 *__restrict__ Tzero,
                                                const u32 *__restrict__ Jold, u32 *__restrict__ Jnew, u32 n) {
  __shared__ u32 hk[512];
  __shared__ u64 hv[512];
  const u32 tid = threadIdx.x;
  for (u32 base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    hk[tid] = NONE32, hk[tid + 256u] = NONE32;
    hv[tid] = 0, hv[tid + 256u] = 0;
    const u32 e = base + tid;
    __syncthreads();
    if (e < n) {
      const u32 j = Jold[e];
      const u64 t = Told[e];
      Tzero[e] = 0;
      if (t) atomicAdd((unsigned long long *)&Tnew[e], (unsigned long long)t);
      if (j & XDONE) {
        Jnew[e] = j;
      } else {
        Jnew[e] = Jold[j];
        if (t) {  // pushes to one target are combined per workgroup first (rivers: thousands of nodes, one ancestor)
          u32 slot = (j * 2654435761u) >> 23;
          for (;;) {
            const u32 prev = atomicCAS(&hk[slot], NONE32, j);
            if (prev == NONE32 || prev == j) {
              atomicAdd((unsigned long long *)&hv[slot], (unsigned long long)t);
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
      const u32 key = hk[tid + 256u * k];
      const u64 val = hv[tid + 256u * k];
      if (key != NONE32 && val) atomicAdd((unsigned long long *)&Tnew[key], (unsigned long long)val);
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_wsx_totals(SuperArgs s, u32 n3, const u64 *__restrict__ T, u64 *__restrict__ xtot64) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n3 && sx_active(s, k)) xtot64[s.sx_slot[k]] = T[k];
}

// the totals of all exits of a supertile: value-carrying doubling over its exit list (super_solve<true> with u64)
template <u32 CAP, u32 NT>
__global__ void __launch_bounds__(NT) k_wsuper_down(SuperArgs s, const u64 *__restrict__ xT64, u64 *__restrict__ xtot64) {
  constexpr int NW = NT / 64;
  constexpr int DPT = CAP / NT;
  constexpr int NSB = SBN / NT;
  __shared__ u64 T[CAP];
  __shared__ uint16_t P[CAP];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][NW];
  const u32 tid = threadIdx.x, st = blockIdx.x, base = st << SSHIFT;
  const u32 n = min(s.scount[st], CAP);
  u32 sbr[NSB];
#pragma unroll
  for (int hh = 0; hh < NSB; ++hh) sbr[hh] = s.sb[(size_t)st * SBN + NT * hh + tid];
#pragma unroll 4
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    if (e >= n) continue;
    const u32 w = s.xl_slot[base + e];
    T[e] = xT64[base + (w & (SSL - 1))];
    P[e] = (uint16_t)((w & XL_SX) ? (e | SDONE) : (u32)s.xl_next[base + e]);
  }
  __syncthreads();
  {  // flow entering the supertile: totals of the super-exits that drain into its boundary cells (k_wsx_totals)
    const u32 str = st / s.nstc, stc = st % s.nstc;
#pragma unroll
    for (int hh = 0; hh < NSB; ++hh) {
      const u32 r = sbr[hh];
      if (!(r & SB_VALID)) continue;
      u32 R, C;
      sb_cell(tid + NT * hh, &R, &C);
      u32 m = (r >> 16) & 0xFFu;
      u64 v = 0;
      while (m) {
        const int k = __ffs((int)m) - 1;
        m &= m - 1u;
        v += xtot64[nbr_slot(str * SG + (R >> 6), stc * SG + (C >> 6), (int)(R & 63u), (int)(C & 63u), k, s.nstc)];
      }
      atomicAdd((unsigned long long *)&T[r & (SSL - 1)], (unsigned long long)v);
    }
  }
  __syncthreads();
  u32 y[DPT], live = 0;
#pragma unroll
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    u32 p = P[e < CAP ? e : 0u];
    p = e < n ? p : SDONE;
    y[k] = p & (SDONE - 1u);
    live |= (p & SDONE) ? 0u : 1u << k;
  }
#pragma nounroll
  for (int round = 0; round < MAXROUNDS_SUPER; ++round) {
    u64 av[DPT];
    u32 q[DPT];
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
        atomicAdd((unsigned long long *)&T[y[k]], (unsigned long long)av[k]);
        P[tid + NT * k] = (uint16_t)q[k];
        y[k] = q[k] & (SDONE - 1u);
        if (q[k] & SDONE) live &= ~(1u << k);
      }
    }
    if (!wg_vote<NW>(s_flag, round, tid, live != 0u)) break;
  }
#pragma unroll 4
  for (int k = 0; k < DPT; ++k) {
    const u32 e = tid + NT * k;
    if (e < n) xtot64[base + ((u32)s.xl_slot[base + e] & (SSL - 1))] = T[e];
  }
}

// final pass of a tile: the doubling with 64-bit values, entries start with their own weight + the inflow they pull
template <int NT>
__global__ void __launch_bounds__(NT) k_wtile_final(WideArgs a) {
  constexpr int QF = TCELLS / 4 / NT;  // quads per thread (4 with 256 threads, 2 with 512)
  constexpr u32 QSTR = 4u * NT;        // cells between a thread's quads
  // (32 KB + 8 KB + the vote's flags: three tiles per CU — a fourth would need the image to be EXACTLY a quarter of the LDS,
  //  and any vote costs a few bytes; no sink words — a saturated cell issues no atomic)
  __shared__ __attribute__((aligned(16))) u64 A[TCELLS];
  __shared__ __attribute__((aligned(16))) uint16_t P[TCELLS];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][8];
  const u32 tid = threadIdx.x;
  u32 tc, tr;
  pfd_tile_of_block(&tc, &tr);
  const u32 sbase = sslot_base(tr, tc, a.nstc);
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  u32 cq[QF];
#pragma unroll
  for (int j = 0; j < QF; ++j) {  // own quads straight from HBM (clamped address, masked afterwards)
    const u32 l0 = 4u * tid + QSTR * j;
    const i64 gr = r0 + (l0 >> 6), gc0 = c0 + (l0 & 63);
    const i64 crr = gr >= (i64)a.nrow ? (i64)a.nrow - 1 : gr;
    const i64 ccs = gc0 >= (i64)a.ncol ? (i64)a.ncol - 1 : gc0;
    u32 w;
    __builtin_memcpy(&w, a.ncode + (size_t)crr * a.ncol + (size_t)ccs, 4);  // (ncode carries slack)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (gr >= (i64)a.nrow || gc0 + b >= (i64)a.ncol) w = (w & ~(0xFFu << (8 * b))) | (D8_MV << (8 * b));
    cq[j] = w;
  }
  u64 inf = 0;
  int plr = 0, plc = 0;
  if (tid < NPERIM) {
    pslot_inv((int)tid, &plr, &plc);
    u32 m = (a.xrec[sbase + tid] >> 16) & 0xFFu;
    while (m) {
      const int k = __ffs((int)m) - 1;
      m &= m - 1u;
      inf += a.xtot[nbr_slot(tr, tc, plr, plc, k, a.nstc)];
    }
  }
  const u32 qs = (tid >> 3) & 3u;
  u32 pc[QF * 4];
  u32 live = 0;
#pragma unroll
  for (int j = 0; j < QF; ++j) {
    const u32 l0 = 4u * tid + QSTR * j;
    const int lr = l0 >> 6, lc0 = l0 & 63;
    const u32 grow = min((u32)(r0 + lr), a.nrow - 1u);
    const u64 wr = a.wrow[grow];
    const u32 wf = a.wfrac[grow];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32 b = (u32)s ^ qs;
      const u32 c = (cq[j] >> (8 * b)) & 0xFFu;
      pc[4 * j + s] = w_init_ptr(c, l0 + (u32)s, lr, lc0 + (int)b);
      A[l0 + (u32)s] = c != D8_MV ? w_cell(wr, wf, (u32)c0 + (u32)lc0 + b) : 0ull;
    }
    *(uint2 *)&P[l0] = make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
    if (!(pc[4 * j + 0] & pc[4 * j + 1] & pc[4 * j + 2] & pc[4 * j + 3] & PDONE)) live |= 1u << j;
  }
  __syncthreads();
  if (inf) A[PHYS((u32)(plr * TS + plc))] += inf;  // (one slot per perimeter cell: no two threads share a word)
  __syncthreads();
  for (int round = 0; round < MAXROUNDS_TILE; ++round) {
    u64 av[QF * 4];
    u32 q[QF * 4];
#pragma unroll
    for (int j = 0; j < QF; ++j) {
      if (live & (1u << j)) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          av[4 * j + b] = A[4u * tid + QSTR * j + (u32)b];
          q[4 * j + b] = *(const uint16_t *)((const u8 *)P + (pc[4 * j + b] & 0x1FFEu));
        }
      }
    }
    __syncthreads();  // every read of this round precedes every write of this round
#pragma unroll
    for (int j = 0; j < QF; ++j) {
      if (live & (1u << j)) {
        // the four cells of a quad are neighbours in a row and, after a few rounds, mostly share their target: combined
        // in registers (same-address LDS atomics are served one lane after the other); a saturated cell has delivered
        u32 t[4];
        u64 w[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          t[b] = pc[4 * j + b];
          w[b] = t[b] >= PDONE ? 0ull : av[4 * j + b];
        }
        const bool e10 = t[1] == t[0], e20 = t[2] == t[0], e21 = t[2] == t[1], e30 = t[3] == t[0], e31 = t[3] == t[1], e32 = t[3] == t[2];
        w[0] += (e10 ? w[1] : 0ull) + (e20 ? w[2] : 0ull) + (e30 ? w[3] : 0ull);
        w[1] = e10 ? 0ull : w[1] + ((!e20 && e21) ? w[2] : 0ull) + ((!e30 && e31) ? w[3] : 0ull);
        w[2] = (e20 || e21) ? 0ull : w[2] + ((!e30 && !e31 && e32) ? w[3] : 0ull);
        w[3] = (e30 || e31 || e32) ? 0ull : w[3];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (w[b]) atomicAdd((unsigned long long *)&A[(t[b] & 0x1FFEu) >> 1], (unsigned long long)w[b]);
          pc[4 * j + b] = q[4 * j + b];
        }
        if (pc[4 * j + 0] & pc[4 * j + 1] & pc[4 * j + 2] & pc[4 * j + 3] & PDONE) live &= ~(1u << j);
        *(uint2 *)&P[4u * tid + QSTR * j] =
            make_uint2(pc[4 * j + 0] | (pc[4 * j + 1] << 16), pc[4 * j + 2] | (pc[4 * j + 3] << 16));
      }
    }
    if (!fx_vote_n<NT / 64>(s_flag, round, tid, live != 0u)) break;  // (one barrier; __syncthreads_or is three)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < QF; ++j) {
    const u32 l0 = 4u * tid + QSTR * j;
    const i64 gr = r0 + (l0 >> 6), gc0 = c0 + (l0 & 63);
    if (gr >= (i64)a.nrow || gc0 >= (i64)a.ncol) continue;
    double o4[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {  // logical cell b of the quad sits in slot b ^ qs
      const u32 c = (cq[j] >> (8 * b)) & 0xFFu;
      o4[b] = c == D8_MV ? -9999.0 : (double)A[l0 + ((u32)b ^ qs)] * a.inv;
    }
    double *dst = a.out + (size_t)gr * a.ncol + (size_t)gc0;
    if (gc0 + 3 < (i64)a.ncol && (((size_t)dst) & 15) == 0) {
      *(double2 *)dst = make_double2(o4[0], o4[1]);
      *(double2 *)(dst + 2) = make_double2(o4[2], o4[3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gc0 + b < (i64)a.ncol) dst[b] = o4[b];
    }
  }
}

// *complete = 1: `out_dev` holds every cell's upstream sum of wrow (scaled by inv), -9999 on nodata cells
int pfd_upstream_area_wide_tiled(pfd_raster *h, const u64 *wrow_dev, const u32 *wfrac_dev, double inv, double *out_dev, int *complete) {
  *complete = 0;
  if (h->halo_top || h->halo_bot) return PFD_OK;
  TiledRun run;
  PFDCHK(run.init(h, nullptr));
  if (!run.supported) return PFD_OK;
  // the interior tiles' sums per exit come out of the count pass's own local kernel (TileArgs::xT64)
  const size_t nslots = run.nslots;
  DevBuf slots64, l3w;
  PFDCHK(slots64.alloc(2 * nslots * sizeof(u64)));
  u64 *xT64 = slots64.as<u64>(), *xtot64 = xT64 + nslots;
  const char *uf = pfd_knob("PFD_WIDE_UNFUSED");  // (test knob: the separate local pass over every tile)
  const bool fused = !(uf && atoi(uf) != 0) && !run.a.weights;
  if (fused) run.a.wrow = wrow_dev, run.a.wfrac = wfrac_dev, run.a.xT64 = xT64;
  // ---- structure: the count pass without its final tile pass ----
  int ok = 0;
  for (int tries = 0; tries < 4; ++tries) {
    PFDCHK(run.phase_a());
    u64 c0[48];
    HIPCHK(hipMemcpyAsync(c0, h->ctrl, sizeof(c0), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    PFDCHK(run.phase_b_collect(c0, &ok));
    if (run.overflowed && tries < 3) {
      run.force_flat = true;
      continue;
    }
    if (run.short_of_rounds && tries < 2) {
      run.extra_rounds += 8;
      continue;
    }
    break;
  }
  if (!ok) return PFD_OK;  // cycles (or an exit graph that would not saturate): the exact form keeps the reference's semantics
  h->acyclic = 1;
  u32 nflag = 0;
  HIPCHK(hipMemcpyAsync(&nflag, run.sa.nflag, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (nflag) return PFD_OK;  // a supertile with more exits than the u64 LDS form holds (contrived rasters)
  const SuperArgs &sa = run.sa;
  if (run.flat_nosync) {  // (the count of super-exits stayed on the device)
    u64 ns = 0;
    HIPCHK(hipMemcpyAsync(&ns, h->ctrl + T_NSUPER, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    run.nsuper = (u32)ns;
  }
  const u32 n3 = sa.hmode ? (u32)((size_t)run.nht * HCAP) : run.nsuper;
  PFDCHK(l3w.alloc((size_t)std::max(n3, 1u) * (3 * sizeof(u64) + 2 * sizeof(u32))));
  u64 *T[3] = {l3w.as<u64>(), l3w.as<u64>() + n3, l3w.as<u64>() + 2 * (size_t)n3};
  u32 *J[2] = {(u32 *)(l3w.as<u64>() + 3 * (size_t)n3), (u32 *)(l3w.as<u64>() + 3 * (size_t)n3) + n3};
  WideArgs wa{h->ncode, (u32)h->nrow, (u32)h->ncol, run.ntr, run.ntc, run.nstc, wrow_dev, wfrac_dev, run.xrec, xT64, xtot64, out_dev, inv};
  const dim3 grid(run.ntc, run.ntr);
  pfd_seg_begin(h, "wide_tile_local");
  if (fused) {
    wa.tr_lo = run.a.tr_lo, wa.tr_hi = run.a.tr_hi, wa.tc_lo = run.a.tc_lo, wa.tc_hi = run.a.tc_hi;
    k_wtile_local<true><<<frame_tiles(run.ntr, run.ntc, wa.tr_lo, wa.tr_hi, wa.tc_lo, wa.tc_hi), 256, 0, h->stream>>>(wa);
  } else {
    k_wtile_local<false><<<grid, 256, 0, h->stream>>>(wa);
  }
  KCHK();
  pfd_seg_end(h, 1);
  pfd_seg_begin(h, "wide_exit_graph");
  i64 launches = 0;
  if (n3) {
    HIPCHK(hipMemsetAsync(T[0], 0, (size_t)n3 * sizeof(u64), h->stream));
    k_wsuper_up<SCAP, 1024u><<<run.nst, 1024, 0, h->stream>>>(sa, xT64, T[0]);
    const u32 g3 = cdiv_u32(n3, 256);
    k_wlink3<<<g3, 256, 0, h->stream>>>(sa, n3, J[0], T[1]);
    launches += 3;
    int batch = 3;
    for (u32 span = 1; span < (run.ntr + run.ntc) / SG + 2; span <<= 1) ++batch;  // ~log2 of a path in supertiles
    const u32 gr = std::min(g3, 4096u);
    bool done = false;
    for (int rounds = 0; rounds < 64 && !done;) {
      for (int b = 0; b < batch; ++b, ++rounds) {
        k_wround<<<gr, 256, 0, h->stream>>>(T[0], T[1], T[2], J[0], J[1], n3);
        u64 *t0 = T[0];
        T[0] = T[1], T[1] = T[2], T[2] = t0;
        std::swap(J[0], J[1]);
      }
      launches += batch;
      HIPCHK(hipMemsetAsync(h->ctrl + T_XACTIVE, 0, sizeof(u64), h->stream));
      k_check_saturated<<<gr, 256, 0, h->stream>>>(J[0], n3, h->ctrl, nullptr, 1u);
      KCHK();
      u64 act = 0;
      HIPCHK(hipMemcpyAsync(&act, h->ctrl + T_XACTIVE, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      done = act == 0;
      batch = 2;
    }
    if (!done) {
      pfd_seg_end(h, launches);
      return PFD_OK;
    }
    k_wsx_totals<<<g3, 256, 0, h->stream>>>(sa, n3, T[0], xtot64);
    ++launches;
  }
  k_wsuper_down<SCAP, 1024u><<<run.nst, 1024, 0, h->stream>>>(sa, xT64, xtot64);
  KCHK();
  pfd_seg_end(h, launches + 1);
  pfd_seg_begin(h, "wide_tile_final");
  // (512 threads: the 40 KB image fixes four tiles per CU either way, and eight waves per tile overlap more of the LDS
  //  round trips than four — profiles/r06f_wide_probe.txt; PFD_WIDE_NT=256 is the other form)
  const char *nt = pfd_knob("PFD_WIDE_NT");
  if (nt && atoi(nt) == 256) k_wtile_final<256><<<grid, 256, 0, h->stream>>>(wa);
  else k_wtile_final<512><<<grid, 512, 0, h->stream>>>(wa);
  KCHK();
  pfd_seg_end(h, 1);
  *complete = 1;
  return PFD_OK;
}

// C-ABI: upstream sums of one float64 value per raster row, accumulated in 64-bit fixed point (see the header of this
// file).  *used = 0: not taken (cycles, row block, values that are not finite and positive, ...) — `out` is undefined
// and the caller runs pfd_accuflux_rows.
extern "C" int pfd_upstream_area_rows_fixed(pfd_raster *h, const double *row_values, double *out, int memspace, int *used,
                                            double *quantum) {
  PFDCHK(pfd_check_handle_lazy(h));
  if (!row_values || !out || !used) {
    pfd_set_error("pfd_upstream_area_rows_fixed: NULL argument");
    return PFD_EINVAL;
  }
  *used = 0;
  if (quantum) *quantum = 0.0;
  if (h->gen || h->halo_top || h->halo_bot) return PFD_OK;
  double tot = 0.0;
  for (i64 r = 0; r < h->nrow; ++r) {
    const double v = row_values[r];
    if (!(v > 0.0) || !(v < 1e300)) return PFD_OK;  // (also NaN)
    tot += v;
  }
  tot *= (double)h->ncol;
  if (!(tot > 0.0) || !(tot < 1e300)) return PFD_OK;
  int ex = 0;
  (void)frexp(tot, &ex);          // tot = f * 2^ex, f in [0.5, 1)  =>  tot * 2^(64 - ex) in [2^63, 2^64): unsigned sums cannot wrap
  const int s = 64 - ex;          // (a row of cells sums to ncol * base + floor(ncol * fraction) <= ncol * area * 2^s)
  const double scale = ldexp(1.0, s), inv = ldexp(1.0, -s);
  // per row: floor(area * 2^s) and the fraction left over, as a 32-bit fixed-point number the columns share out (w_cell)
  std::vector<u64> q((size_t)h->nrow);
  std::vector<u32> qf((size_t)h->nrow);
  for (i64 r = 0; r < h->nrow; ++r) {
    const double x = row_values[r] * scale, fl = floor(x);  // (exact: a power-of-two scale)
    if (!(fl >= 1.0)) return PFD_OK;  // an area below the quantum: the raster spans too many orders of magnitude
    q[(size_t)r] = (u64)fl;
    qf[(size_t)r] = (u32)std::min((x - fl) * 4294967296.0, 4294967295.0);
  }
  pfd_seg_clear(h);
  DevBuf wq;
  PFDCHK(wq.alloc(q.size() * (sizeof(u64) + sizeof(u32))));
  HIPCHK(hipMemcpyAsync(wq.p, q.data(), q.size() * sizeof(u64), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(wq.as<u64>() + q.size(), qf.data(), qf.size() * sizeof(u32), hipMemcpyHostToDevice, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(double), memspace));
  int complete = 0;
  PFDCHK(pfd_upstream_area_wide_tiled(h, wq.as<u64>(), (const u32 *)(wq.as<u64>() + q.size()), inv, (double *)o.dev, &complete));
  if (!complete) {
    HIPCHK(hipStreamSynchronize(h->stream));
    return PFD_OK;
  }
  *used = 1;
  if (quantum) *quantum = inv;
  return o.finish(h->stream);
}
