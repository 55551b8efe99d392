// tile_patch.h — local pass of an interior tile with REGISTER-LEVEL CONTRACTION before the LDS doubling
// (included by tiled.hip behind tile_fast.h, selected by TiledRun::use_patch).
//
// k_tile_local_fast gives a thread four quads 16 rows apart and runs the pointer jumping over all 4096 cells in LDS:
// ~8 random u16 gathers per cell, 70 % LDS-pipe busy, 41 % of it bank conflicts.  Here a thread owns a compact 4 x 4
// PATCH and resolves the hops that stay inside it in registers:
//   * the in-patch successor of a cell is a nibble-sized index kept as a BYTE, 16 of them in 4 VGPRs; one pointer jump
//     of all 16 cells is J <- J o J = four 16-entry byte-table lookups of 4 selectors each, and v_perm_b32 IS an
//     8-entry byte-table lookup of 4 selectors (two perms + one v_bfi per 4 cells): 4 jumps = 2^4 hops cover any path
//     inside 16 cells in ~100 VALU instructions per thread, no LDS, no barrier;
//   * what is left for the LDS doubling are the PATCH ROOTS — cells whose flow leaves the patch (or ends): a third of the
//     cells, on paths ~3.5 x shorter (a hop crosses a patch).  Every cell's word first names the root of its OWN patch
//     (a static forward); a root then replaces its word by the word it finds at the cell it flows into — the root of the
//     next patch — so that from then on every pointer names a root, and root words are the ones that advance;
//   * afterwards a cell's tile root is the final pointer of its patch root — a table lookup in the thread's own
//     registers (two byte planes, again v_perm_b32), no gather back through LDS.
// Everything else (staging, normalisation of a deferred handle, count per exit, perimeter records) is k_tile_local_fast's.
#pragma once

#define PX_S4_LO 0x03040501u  // in-patch index steps (row pitch 4)  E +1, SE +5, S +4, SW +3
#define PX_S4_HI 0xFDFCFBFFu  //                                      W -1, NW -5, N -4, NE -3

// four lookups into a 16-entry byte table (entry e = byte e & 3 of t[e >> 2]); selector bytes 0..15
__device__ __forceinline__ u32 px_lookup16(const u32 (&t)[4], u32 sel) {
  const u32 s7 = sel & 0x07070707u;
  const u32 lo = __builtin_amdgcn_perm(t[1], t[0], s7);
  const u32 hi = __builtin_amdgcn_perm(t[3], t[2], s7);
  const u32 m = ((sel >> 3) & 0x01010101u) * 0xFFu;
  return (hi & m) | (lo & ~m);
}

#define PX_NOTROOT 0x4000u  // q[]: the cell is no patch root (bit 14; its pointer register then holds "nobody" | this)

template <bool RAW>
__global__ void __launch_bounds__(256, 8) k_tile_local_patch(TileArgs a) {
  __shared__ __attribute__((aligned(16))) u32 A[PSL * FXP];      // count words per perimeter slot
  __shared__ __attribute__((aligned(16))) uint16_t P[FX_PN];     // pointer words: cells (row-major), slots, nobody
  __shared__ __attribute__((aligned(16))) u8 code[HW * CP];
  __shared__ __attribute__((aligned(16))) u32 s_flag[2][4];
  __shared__ u64 s_cnt[4];
  const u32 tid = threadIdx.x;
  u32 bx_, by_;
  pfd_tile_of_block(&bx_, &by_);
  const u32 tc = bx_ + a.tc_lo, tr = by_ + a.tr_lo;
  const u32 sbase = sslot_base(tr, tc, a.nstc);
  const i64 r0 = (i64)tr * TS, c0 = (i64)tc * TS;
  bool mvq = false;
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
  P[TCELLS + tid] = (uint16_t)(FX_SLOT0 + 2u * tid);  // a root word points at itself
  if (tid == 0) P[TCELLS + PSL] = (uint16_t)FX_NOBODY;
  const bool tile_mv = RAW ? (__syncthreads_or(mvq ? 1 : 0) != 0) : (__syncthreads(), false);

  // ---- decode (+ normalise) the thread's 4 x 4 patch: in-patch successor bytes; a patch root's tile-level target ------
  const u32 rbase = 4u * (tid >> 4), cbase = 4u * (tid & 15u);
  const u32 Lb = 64u * rbase + cbase;  // tile index of the patch's first cell
  u32 Nb[4], q[16];
  u32 ndir = 0, npit = 0, nbad = 0;
  auto decode = [&](auto chk) {
    constexpr bool CHKMV = decltype(chk)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32 lr = rbase + (u32)i;
      const u32 ca0 = (lr + 1u) * CP + cbase + 4u;  // byte offset of CODE(lr, cbase)
      const u32 c4 = *(const u32 *)&code[ca0];
      const u32 rm = (i == 0 || i == 3) ? fx_rowmask(lr) : 0u;
      u32 nb4 = 0, n4 = 0, badq = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 x = 4u * i + (u32)k;
        const u32 c = (c4 >> (8 * k)) & 0xFFu;
        const u32 kk = fx_ffbl(c);
        const u32 pcnt = __popc(c);
        bool isdir = pcnt == 1u;
        if (RAW) {
          if (CHKMV) {
            const u32 t = code[(u32)((int)(ca0 + (u32)k) + fx_sext8(__builtin_amdgcn_perm(FX_TC_HI, FX_TC_LO, kk)))];
            isdir = isdir && t != D8_MV;
          }
          const u32 t0 = __builtin_amdgcn_perm(FX_T0_HI, FX_T0_LO, pcnt);
          const u32 n = isdir ? c : t0;
          badq |= t0 & ~c;
          n4 |= n << (8 * k);
          ndir += isdir ? 1u : 0u;
          npit += n == 0u ? 1u : 0u;
        }
        const u32 pm = (i == 0 ? 0xE0u : 0u) | (i == 3 ? 0x0Eu : 0u) | (k == 0 ? 0x38u : 0u) | (k == 3 ? 0x83u : 0u);
        const bool in_patch = isdir && (c & pm) == 0u;
        const u32 nbx = in_patch ? (u32)((int)x + fx_sext8(__builtin_amdgcn_perm(PX_S4_HI, PX_S4_LO, kk))) : x;
        nb4 |= nbx << (8 * k);
        // a patch root: the cell it flows into, the word of its perimeter slot when that lies outside the tile, or nobody
        const u32 lc = cbase + (u32)k;
        u32 tg = 2u * (u32)((int)(Lb + 64u * i + k) + fx_sext8(__builtin_amdgcn_perm(FX_TP_HI, FX_TP_LO, kk)));
        u32 cm = 0;
        if (k == 0 || k == 3) {
          cm = fx_colmask(lc);
          tg = (c & cm) ? FX_SLOT0 + 2u * (127u + lr + (lc ? 62u : 0u)) : tg;        // columns 0 / 63: slot 127 + lr, 189 + lr
        }
        // rows 0 / 63: slot lc, 64 + lc — whichever way the flow leaves the tile (a corner cell's slot is its row's)
        if (i == 0 || i == 3) tg = (rm && (c & (rm | cm))) ? FX_SLOT0 + 2u * (lc + (lr ? 64u : 0u)) : tg;
        q[x] = in_patch ? (FX_NOBODY | PX_NOTROOT) : (isdir ? tg : FX_NOBODY);
      }
      Nb[i] = nb4;
      if (RAW) {
        if (badq) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u32 c = (c4 >> (8 * k)) & 0xFFu;
            nbad += (__builtin_amdgcn_perm(FX_T0_HI, FX_T0_LO, __popc(c)) & ~c) ? 1u : 0u;
          }
        }
        *(u32 *)&code[ca0] = n4;
        __builtin_memcpy(a.ncode_w + (size_t)(r0 + lr) * a.ncol + (size_t)(c0 + cbase), &n4, 4);
      }
    }
  };
  if (RAW && tile_mv) decode(std::true_type{});
  else decode(std::false_type{});
  if (RAW) {
    u64 pk = (u64)(ndir + npit) | ((u64)npit << 16) | ((u64)nbad << 32);
    for (int o = 32; o > 0; o >>= 1) pk += __shfl_down(pk, o);
    if ((tid & 63u) == 0) s_cnt[tid >> 6] = pk;
  }

  // ---- in-register pointer jumping over the patch: J <- J o J until nothing moves (2 jumps for paths of <= 4 cells;
  //      5 = more than 16 hops: a cycle inside the patch) ------------------------------------------------------------
  bool pcycle = true;
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    const u32 n0 = px_lookup16(Nb, Nb[0]), n1 = px_lookup16(Nb, Nb[1]), n2 = px_lookup16(Nb, Nb[2]), n3 = px_lookup16(Nb, Nb[3]);
    const bool moved = ((n0 ^ Nb[0]) | (n1 ^ Nb[1]) | (n2 ^ Nb[2]) | (n3 ^ Nb[3])) != 0u;
    Nb[0] = n0, Nb[1] = n1, Nb[2] = n2, Nb[3] = n3;
    if (__ballot(moved) == 0ull) {
      pcycle = false;
      break;
    }
    pcycle = moved;
  }
  // every cell's word: the root of its own patch (root byte rb -> tile offset 64 (rb >> 2) + (rb & 3) behind the patch)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32 off = (Nb[i] & 0x03030303u) | ((Nb[i] & 0x0C0C0C0Cu) << 4);
    const u32 p01 = __builtin_amdgcn_perm(0u, off, 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, off, 0x0c030c02u);
    const u32 b2 = (2u * Lb) * 0x10001u;
    *(uint2 *)&P[Lb + 64u * i] = make_uint2((p01 << 1) + b2, (p23 << 1) + b2);
  }
  __syncthreads();
  if (RAW && tid == 0) a.tcnt[(size_t)tr * a.ntc + tc] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];

  // ---- the patch roots: the word found at the cell the root flows into names the root of the next patch ---------------
  // (unconditional gathers, all 16 in flight; whoever has nothing to ask reads "nobody": one address, a broadcast.  A word
  //  may be read before or after its owner has replaced "myself" by "the next root": both are ancestors)
  {
    u32 t[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) t[x] = *(const uint16_t *)((const u8 *)P + ((q[x] & (FX_SLOT0 | PX_NOTROOT)) ? FX_NOBODY : q[x]));
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      q[x] = (q[x] & (FX_SLOT0 | PX_NOTROOT)) ? q[x] : t[x];
      if (!(q[x] & PX_NOTROOT)) P[Lb + 64u * (x >> 2) + (x & 3)] = (uint16_t)q[x];
    }
  }
  // (one thread per perimeter slot: the target of the exit sitting there — what the records below carry)
  u32 xt12 = XR_NONE;
  int plr = 0, plc = 0;
  if (tid < NPERIM) {
    pslot_inv((int)tid, &plr, &plc);
    const u32 c = CODE(plr, plc);
    if (d8_is_dir(c)) {
      const int k = d8_slot(c);
      const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
      if ((unsigned)nr >= TS || (unsigned)nc >= TS) xt12 = xr_t12(nr, nc);
    }
  }

  // ---- LDS doubling over the patch roots only, two jumps per round, one barrier per round --------------------------
  // (a pointer that has arrived carries bit 13 — slot words and "nobody" lie behind the cells —: no flag of its own.  A
  //  word still saying "myself" because its owner is late just returns the asker's own pointer: no progress, no harm)
  int round = 0;
#pragma nounroll
  for (; round < MAXROUNDS_TILE; ++round) {
    u32 t[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) t[x] = *(const uint16_t *)((const u8 *)P + ((q[x] & FX_SLOT0) ? FX_NOBODY : q[x]));
#pragma unroll
    for (int x = 0; x < 16; ++x) t[x] = *(const uint16_t *)((const u8 *)P + t[x]);
    u32 moving = 0;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      if (!(q[x] & FX_SLOT0)) {
        q[x] = t[x];
        P[Lb + 64u * (x >> 2) + (x & 3)] = (uint16_t)t[x];
        moving |= ~t[x] & FX_SLOT0;
      }
    }
    if (!fx_vote(s_flag, round, tid, moving != 0u)) break;
  }
  u32 live = pcycle ? 1u : 0u;
#pragma unroll
  for (int x = 0; x < 16; ++x) live += (q[x] & FX_SLOT0) ? 0u : 1u;
  if ((a.ablate & 32) && tid == 0) {
    const unsigned long long r = (unsigned long long)min(round + 1, MAXROUNDS_TILE);
    const u32 w = (tr * a.ntc + tc) & 255u;
    atomicMax((unsigned long long *)&a.rcnt[w], r);
    atomicAdd((unsigned long long *)&a.rcnt[256 + w], r);
  }
  if (live) atomicAdd((unsigned long long *)&a.ctrl[T_UNSAT], (unsigned long long)live);

  // ---- every cell: the final pointer of its patch root (own registers), its count to that exit's counter -----------
  u32 ql[4], qh[4];  // the 16 pointers as two byte planes
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ql[i] = __builtin_amdgcn_perm(q[4 * i + 1], q[4 * i], 0x0c0c0400u) | __builtin_amdgcn_perm(q[4 * i + 3], q[4 * i + 2], 0x04000c0cu);
    qh[i] = __builtin_amdgcn_perm(q[4 * i + 1], q[4 * i], 0x0c0c0501u) | __builtin_amdgcn_perm(q[4 * i + 3], q[4 * i + 2], 0x05010c0cu);
  }
  const u32 rep = 4u * (tid & (FXP - 1u));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32 fl = px_lookup16(ql, Nb[i]), fh = px_lookup16(qh, Nb[i]) & 0x3F3F3F3Fu;  // (bit 14: a marker, no address)
    const u32 f01 = __builtin_amdgcn_perm(fh, fl, 0x05010400u), f23 = __builtin_amdgcn_perm(fh, fl, 0x07030602u);  // u16 pairs
    *(uint2 *)&P[Lb + 64u * i] = make_uint2(f01, f23);  // (the records below ask)
    u32 x[4], w[4];
    x[0] = (f01 & 0xFFFFu) - FX_SLOT0, x[1] = (f01 >> 16) - FX_SLOT0, x[2] = (f23 & 0xFFFFu) - FX_SLOT0, x[3] = (f23 >> 16) - FX_SLOT0;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = x[k] < 2u * PSL ? 1u : 0u;  // 2 x slot; >= 2 * PSL: nobody asks
    if (FX_COMBINE) {
      const bool e10 = x[1] == x[0], e20 = x[2] == x[0], e21 = x[2] == x[1], e30 = x[3] == x[0], e31 = x[3] == x[1], e32 = x[3] == x[2];
      w[0] += (e10 ? w[1] : 0u) + (e20 ? w[2] : 0u) + (e30 ? w[3] : 0u);
      w[1] = e10 ? 0u : w[1] + ((!e20 && e21) ? w[2] : 0u) + ((!e30 && e31) ? w[3] : 0u);
      w[2] = (e20 || e21) ? 0u : w[2] + ((!e30 && !e31 && e32) ? w[3] : 0u);
      w[3] = (e30 || e31 || e32) ? 0u : w[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (w[k]) atomicAdd((u32 *)((u8 *)A + (x[k] << (FXP == 8 ? 4 : 3)) + rep), w[k]);
  }
  __syncthreads();

  // ---- perimeter records for the exit graph (as k_tile_local_fast) ----------------------------------------------------
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
    if (c != D8_MV) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int nr = plr + d8_dr(k), nc = plc + d8_dc(k);
        if (((unsigned)nr >= TS || (unsigned)nc >= TS) && CODE(nr, nc) == (1u << ((k + 4) & 7))) inmask |= 1u << k;
      }
    }
    if (inmask) {
      const u32 x = (u32)P[(u32)(plr * TS + plc)] - FX_SLOT0;
      if (x < 2u * PSL) link = x >> 1;
    }
  }
  a.xT[sbase + tid] = xt;
  a.xrec[sbase + tid] = xr_pack(xt12 & 0xFFu, xt12 >> 8, link, inmask);
  const u64 xm = __ballot(xt12 != XR_NONE);
  if ((tid & 63u) == 0u) a.xmask[(sbase + tid) >> 6] = xm;
}
