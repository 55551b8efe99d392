// tiled.h — shared declarations of the LDS-tiled accumulation engine (tiled.hip) and its
// multi-GPU driver (dist.hip).
#pragma once
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
// "the path ends on a halo sink" encoding (row blocks): bit 31 | side << 30 | column
#define ENC_SINK 0x80000000u
#define ENC_SIDE1 0x40000000u
#define ENC_COL 0x3FFFFFFFu

enum { T_PROC = 8, T_NEXITS = 9, T_XACTIVE = 10 };  // ctrl slots (u64)

struct TileArgs {
  const u8 *ncode;
  u32 nrow, ncol, ntr, ntc;
  u32 row_first, row_last;  // owned rows (inclusive) of the device raster; the rest are halo rows
  u32 *xid;        // [nslots] dense id of the exit sitting on this perimeter slot, NONE32 if none
  u32 *eT;         // [nexits] local count of an exit (dense exit id)
  u32 *etgt;       // [nexits] global perimeter slot the exit drains into
  u32 *elink;      // [nslots] dense id of the exit an entry's in-tile path reaches, NONE32 if none
  u32 *inflow;     // [nslots] sum of the totals of the exits draining into this slot
  u32 *esink;      // [2*ntc*PSL] first/last tile row: halo sink an entry's in-tile path ends on
  u32 *brow_first; // [2*ncol] boundary rows: where the in-tile path of the cell ends (exit id / sink)
  u32 *haloA;      // [2*ncol] flow that reached a halo sink inside its tile
  u32 *brow_inflow;// [2*ncol] flow entering the boundary rows from the neighbouring row blocks
  u64 *ctrl;
  i32 *out;
  int ablate;      // profiling knob (env PFD_TILE_ABLATE): bit0 skip doubling, bit4 cycle stamps
};

struct TiledRun {
  pfd_raster *h = nullptr;
  u32 ntr = 0, ntc = 0, nexits = 0;
  size_t nslots = 0;
  bool supported = false, is_block = false, coarse_done = false;
  DevBuf T0, T1, J0, J1, Jlink, etgt, xid, elink, inflow, esink, bnd;
  u32 *brow_first = nullptr, *haloA = nullptr, *haloL = nullptr, *brow_sink = nullptr, *brow_inflow = nullptr;
  u32 *Tc = nullptr, *Tn = nullptr, *Jc = nullptr, *Jn = nullptr;
  TileArgs a{};
  int init(pfd_raster *hh, i32 *out_dev);
  int phase_a();
  int phase_b(int *complete);
};

int pfd_doubling_rounds(pfd_raster *h, u32 **Tc, u32 **Tn, u32 **Jc, u32 **Jn, u32 n, int first_batch, bool *done,
                        i64 *launches);
