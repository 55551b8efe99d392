// exact.h — the exact-order engine (exact.hip): plan of a raster + the sweep drivers sweeps.hip calls.
//
// Order-sensitive operations (float accuflux, HAND, ...) must combine values in the order of the
// reference's serial loop, so their critical path is the longest flow path: one dependent update per
// cell.  The level engine pays one kernel launch per 1-2 cells of that path.  This engine splits the
// raster so that the dependent chain runs at register speed instead:
//
//   LEAVES  cells whose whole upstream subtree lies inside their 64 x 64 tile and is at most XCAP steps
//           high: a tile kernel resolves them in LDS, step by step, in a precomputed per-tile order;
//   TRUNK   every other cell.  The trunk is cut into heavy chains (heavy child = upstream TRUNK cell with
//           the largest upstream area); chains are laid out contiguously, upstream end first, grouped by
//           ROUND = 31 - (light cells between the chain's last cell and the pit; at most log2(n) of them) — a
//           chain only depends on chains of earlier rounds, and all main stems share the last round.  Per
//           round: a parallel pre-pass gathers, per layout slot, everything that does not depend on the
//           chain itself (own payload + light upstream cells, already final); one
//           LANE per chain then folds the slots serially (running value in a register, loads software-
//           pipelined) — except for the few chains of XLONG slots or more (main stems), which get a WAVE
//           each: 64 lanes stream the slots through LDS, one lane folds them from there, so that the
//           fold runs at LDS instead of HBM latency; a parallel scatter writes the results back.
//
// An upstream cell that the serial loop adds AFTER the heavy one gets a slot of its own behind its
// parent's ("post" slot), so that the fold stays a flat left-to-right scan with the reference's exact
// operand order.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

#define XT 64            // tile edge
#define XTC (XT * XT)    // 4096 cells
#ifndef XCAP
#define XCAP 30          // highest leaf step; toff holds XCAP + 2 = 32 u16 per tile
#endif
#define XOFF (XCAP + 2)
#define XL_TRUNK 240u    // lh[] marks: trunk cell = XL_TRUNK + the number of post slots behind its slot (0..7), see xl_trunk()
#define XL_NODATA 254u
#define XL_HALO 253u     // cell of a halo row of a row block: its value is GIVEN (the neighbouring block's), never computed
// sinfo (u16 per slot): bits 0-7 child mask, 8-11 slot of the heavy child (8 = none: chain head),
// 12-14 number of post slots that follow, 15 = this is a post slot (scell = the upstream cell it carries)
#define XS_POST 0x8000u
#define XC_LEN 0x1FFFFFFFu  // clen: length bits
#ifndef XLONG
#define XLONG 512u          // a chain of at least this many slots is folded by a whole wave (see k_xtrunk_scan)
#endif

__host__ __device__ inline bool xl_trunk(u32 m) { return (m & 0xF8u) == XL_TRUNK; }

struct ExactPlan {
  u32 ntr = 0, ntc = 0;
  u8 *lh = nullptr;       // [n] leaf step (0..XCAP), XL_TRUNK + post slots, XL_NODATA, XL_HALO
  u8 *kids = nullptr;     // [n] mask of the neighbour slots draining into the cell
  uint16_t *tord = nullptr;  // [ntiles * 4096] leaf cells of the tile, ordered by step: local index | downstream slot << 12 | pit << 15
  uint16_t *toff = nullptr;  // [ntiles * XOFF] start of step s in tord; entries past the last step = total
  u32 *cslot = nullptr;   // [n] slot of a trunk cell (its real slot; post slots follow it); undefined elsewhere
  // The trunk cells of every 64 x 64 tile as a dense list (round 5): {slot, local index | post slots << 12}.  The passes
  // that visit the trunk cells in raster order (k_xtrunk_demit, k_xtrunk_unscatter) read 8 contiguous bytes per trunk
  // cell instead of the marks of every cell plus 4 scattered bytes of cslot per trunk cell — a quarter of the cells, one
  // or two per 64-byte sector along a river.
  uint2 *tlist = nullptr;  // [ntrunk]
  u32 *tl_off = nullptr;   // [ntiles + 1] first entry of the tile
  u32 *scell = nullptr;   // [nslot]
  uint16_t *sinfo = nullptr;  // [nslot]
  u32 *spost = nullptr;   // [nslot / 32 + 4] bit s = slot s is a post slot (what the serial fold needs of sinfo)
  u32 *cstart = nullptr;  // [nchain] first slot of the chain (a multiple of 4: chains are padded)
  u32 *clen = nullptr;    // [nchain] slots in the chain | post slots behind its last cell << 29
  u32 *longc = nullptr;   // [nlong] ids of the chains of >= XLONG slots, ascending (= by round)
  i64 nlong = 0;
  i64 b_long[33] = {0};   // long chains of round b = longc[b_long[b] .. b_long[b+1])
  u32 b_maxlen[32] = {0}; // slots of the longest chain of round b (0: no chain of XLONG slots or more)
  // The rounds at the END of the layout hold the main stems: where their longest chain has this many slots the round is
  // latency (one wave folds it serially), and the sweeps run a bandwidth pass beside them (xplan_tail_split)
  i64 nslot = 0, nchain = 0, ntrunk = 0;
  i64 b_chain[33] = {0};  // chains of round b = [b_chain[b], b_chain[b+1])
  i64 b_slot[33] = {0};   // slots  of round b = [b_slot[b],  b_slot[b+1])
  size_t bytes = 0;
  // ---- incremental re-sweeps of a row block (run_exact_up, pfd_set_block_update) ------------------------------
  // A block's up-sweep is repeated with other halo seeds until the boundary rows settle; from the second sweep on
  // only the chains below a CHANGED seed are folded again: the element / value arrays of the last sweep are kept,
  // a chain is dirty when a halo cell with another seed drains into it, or when the chain that ends in it was dirty.
  u32 *schain = nullptr;  // [nslot] chain of a slot (built on first use)
  u32 *dchain = nullptr;  // [nchain] chain of the cell the chain's last cell drains into (0xFFFFFFFF: none)
  u32 *hfeed = nullptr;   // [2 * ncol] chain of the own cell a halo cell drains into (0xFFFFFFFF: none)
  u8 *dirty = nullptr;    // [nchain]
  bool xinc_ready = false;    // the four maps above are built AND checked (pfd_xinc_prepare)
  bool xinc_refused = false;  // they cannot be: a halo cell drains into a non-trunk cell of this plan
  size_t xinc_map_bytes = 0;
  void *incE = nullptr, *incR = nullptr, *incSeed = nullptr;
  size_t inc_tag = 0, inc_bytes = 0;  // operation of the kept sweep (type hash); bytes of incE + incR + incSeed
  const void *inc_out = nullptr;      // its result buffer
  bool inc_valid = false;
};
#define XTAIL_LONG 16384u
// first of the (at most two) last rounds that are latency-bound and hold at most an eighth of the slots, or -1
inline int xplan_tail_split(const ExactPlan *p) {
  if (pfd_knob("PFD_TAIL_SPLIT_OFF")) return -1;
  int nb = 0, rounds[32];
  for (int b = 0; b < 32; ++b)
    if (p->b_chain[b + 1] > p->b_chain[b]) rounds[nb++] = b;
  if (nb < 4) return -1;
  const int b2 = rounds[nb - 2], b1 = rounds[nb - 1];
  if (p->b_maxlen[b1] < XTAIL_LONG && p->b_maxlen[b2] < XTAIL_LONG) return -1;
  if ((p->nslot - p->b_slot[b2]) * 8 > p->nslot) return -1;
  return b2;
}
int pfd_xinc_prepare(pfd_raster *h);  // builds schain / dchain / hfeed / dirty (once per plan)
void pfd_xinc_drop(pfd_raster *h);    // releases the kept sweep (incE / incR / incSeed)
int pfd_xinc_mark(pfd_raster *h, const void *seed_dev, size_t elem);  // dirty <- chains below a changed seed; keeps the seeds

// builds the plan on first use; h->xplan_state: 0 not built, 1 ready, -1 not available (cycles, row
// block, more than 2^32 - 2 cells)
int pfd_ensure_xplan(pfd_raster *h, bool allow_block = false);
void pfd_free_xplan(pfd_raster *h);

// debugging aid (builds with DEVTOOLS=1 only, env PFD_XDEBUG): synchronise after a step and name it, so that a GPU
// memory fault points at its kernel
#ifdef PFD_DEVTOOLS
#define XDBG(h, what)                                                          \
  do {                                                                         \
    const char *xd_ = getenv("PFD_XDEBUG");                                    \
    if (xd_ && (xd_[0] == '1' || (xd_[0] == 'p') == (what[0] == 'k'))) {       \
      const hipError_t e_ = hipStreamSynchronize((h)->stream);                 \
      fprintf(stderr, "[xdbg] %s: %s\n", what, hipGetErrorString(e_));          \
      fflush(stderr);                                                          \
    }                                                                          \
  } while (0)
#else
#define XDBG(h, what) ((void)0)
#endif
