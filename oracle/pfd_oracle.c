/* TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.
 *
 * Single-threaded CPU restatement ("oracle") of the reference's D8 flow-accumulation hot
 * path (Deltares/pyflwdir v0.5.12).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker / the
 * timed CPU baseline — never from pyflwdir_amd/.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against outputs of
 * the reference itself (imported in the build container in its own interpreted test mode,
 * see oracle/gen_golden.py) on the committed fixtures under tests/golden/.
 *
 * Reference citations are file:line into the reference checkout.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_D8_MV 247u /* core_d8._mv (pyflwdir/core_d8.py:17) */

static double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* core_d8.drdc (pyflwdir/core_d8.py:22-39).  The reference decodes with two range tests
 * and log2; values that are not one of the eight direction codes fall through exactly as
 * they do there: dd<=8 branch: dd>=2 -> dr=1, dc=int8(2-log2(dd)); else dr=0, dc=dd;
 * dd<=128 branch: 16 -> W, otherwise dr=-1, dc=int8(log2(dd)-6); dd>128 (incl. 255) ->
 * (0,0).  int8() of a double truncates toward zero. */
static void orc_d8_drdc(uint8_t dd, int *dr, int *dc) {
  static const double LOG2_TAB[9] = {0, 0.0, 1.0, 1.5849625007211562, 2.0, 2.321928094887362,
                                     2.584962500721156, 2.807354922057604, 3.0};
  *dr = 0;
  *dc = 0;
  if (dd <= 8) {
    if (dd >= 2) {
      *dr = 1;
      *dc = (int)(2.0 - LOG2_TAB[dd]);
    } else {
      *dc = (int)dd;
    }
  } else if (dd <= 128) {
    if (dd == 16) {
      *dc = -1;
    } else {
      /* log2 of 9..128 excluding 16; exact for the powers of two the codec uses */
      double l = 0.0;
      switch (dd) {
        case 32: l = 5.0; break;
        case 64: l = 6.0; break;
        case 128: l = 7.0; break;
        default: {
          /* non-code value: the reference still evaluates int8(log2(dd) - 6) */
          double x = (double)dd;
          int e = 0;
          while (x >= 2.0) { x *= 0.5; ++e; }
          /* log2(dd) in [e, e+1): only the truncated difference matters */
          l = (double)e + (x > 1.0 ? 0.5 : 0.0);
        }
      }
      *dr = -1;
      *dc = (int)(l - 6.0);
    }
  }
}

/* exported for the codec test */
void orc_drdc(uint8_t dd, int8_t *dr, int8_t *dc) {
  int a, b;
  orc_d8_drdc(dd, &a, &b);
  *dr = (int8_t)a;
  *dc = (int8_t)b;
}

#define IDX int32_t
#define SFX i32
#include "pfd_oracle_idx.inc"
#undef IDX
#undef SFX

#define IDX uint32_t
#define SFX u32
#include "pfd_oracle_idx.inc"
#undef IDX
#undef SFX

#define IDX int64_t
#define SFX i64
#include "pfd_oracle_idx.inc"
#undef IDX
#undef SFX

/* ---------------------------------------------------------------------------------
 * Synthetic D8 raster generator (ours).  Host twin of the device
 * generator in pyflwdir_amd/csrc/synth.hip — the two must agree bit-for-bit (tested).
 *
 * Integer pseudo-elevation z(r,c) = tilt*(nrow-1-r) + four octaves of bilinear lattice
 * value-noise (cell sizes 8, 32, 128, 512) + per-cell white noise, all from a 64-bit mix
 * hash of (seed, octave, lattice row, lattice col).  A cell drains to the LOWEST of the
 * neighbours that are strictly lower under the total order (z, linear index), first one in
 * the order E,SE,S,SW,W,NW,N,NE on ties — acyclic by construction; cells with no lower
 * neighbour get pit code 0.  With `nodata_pct` > 0 a low-frequency octave carves nodata
 * (247) regions; valid cells next to them may point INTO nodata (the decode rule then
 * makes them pits, pyflwdir/core_d8.py:57-63).
 * --------------------------------------------------------------------------------- */
static inline uint64_t orc_mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
static inline uint32_t orc_lat(uint64_t seed, uint32_t oct, int64_t R, int64_t C) {
  const uint64_t key = seed * 0x9E3779B97F4A7C15ULL + (uint64_t)oct * 0xD1B54A32D192ED03ULL +
                       (((uint64_t)(uint32_t)R) << 32 | (uint64_t)(uint32_t)C);
  return (uint32_t)(orc_mix64(key) & 0xFFFFu);
}
/* bilinear value noise with cell size 2^k, returns value*2^k in [0, 65535*2^k] */
static inline int64_t orc_octave(uint64_t seed, uint32_t k, int64_t r, int64_t c) {
  const int64_t S = (int64_t)1 << k;
  const int64_t R = r >> k, C = c >> k, fr = r & (S - 1), fc = c & (S - 1);
  const int64_t v00 = orc_lat(seed, k, R, C), v01 = orc_lat(seed, k, R, C + 1);
  const int64_t v10 = orc_lat(seed, k, R + 1, C), v11 = orc_lat(seed, k, R + 1, C + 1);
  const int64_t top = v00 * (S - fc) + v01 * fc;
  const int64_t bot = v10 * (S - fc) + v11 * fc;
  return (top * (S - fr) + bot * fr) >> k;
}
typedef struct {
  uint64_t seed;
  int64_t nrow, ncol;
  int64_t tilt;       /* elevation drop per row towards row nrow-1 ... see orc_synth_z */
  int64_t white;      /* amplitude multiplier of the per-cell white-noise term */
  int32_t nodata_pct; /* 0..90: approximate percentage of nodata cells */
} orc_synth_t;

int64_t orc_synth_z(const orc_synth_t *p, int64_t r, int64_t c) {
  int64_t z = p->tilt * (p->nrow - 1 - r);
  z += orc_octave(p->seed, 3, r, c);
  z += orc_octave(p->seed, 5, r, c);
  z += orc_octave(p->seed, 7, r, c);
  z += orc_octave(p->seed, 9, r, c);
  z += (int64_t)orc_lat(p->seed, 0, r, c) * p->white;
  return z;
}
int orc_synth_isnodata(const orc_synth_t *p, int64_t r, int64_t c) {
  if (p->nodata_pct <= 0) return 0;
  /* low-frequency mask octave (cell size 256), value in [0, 65535] */
  const int64_t v = orc_octave(p->seed ^ 0xA5A5A5A5ULL, 8, r, c) >> 8;
  return v * 100 < (int64_t)p->nodata_pct * 65536;
}

/* neighbour visiting order and D8 codes (core_d8._ds, pyflwdir/core_d8.py:15) */
static const int ORC_NB_DR[8] = {0, 1, 1, 1, 0, -1, -1, -1};
static const int ORC_NB_DC[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const uint8_t ORC_NB_CODE[8] = {1, 2, 4, 8, 16, 32, 64, 128};

uint8_t orc_synth_code(const orc_synth_t *p, int64_t r, int64_t c) {
  if (orc_synth_isnodata(p, r, c)) return (uint8_t)ORC_D8_MV;
  const int64_t z0 = orc_synth_z(p, r, c);
  int64_t best = -1;
  uint8_t code = 0;
  for (int k = 0; k < 8; ++k) {
    const int64_t rr = r + ORC_NB_DR[k], cc = c + ORC_NB_DC[k];
    if (rr < 0 || cc < 0 || rr >= p->nrow || cc >= p->ncol) continue;
    int64_t score;
    if (orc_synth_isnodata(p, rr, cc)) {
      /* the sea: always lower, drop measured against elevation 0 */
      score = z0 + 1;
    } else {
      const int64_t zn = orc_synth_z(p, rr, cc);
      const int lower = (zn < z0) || (zn == z0 && (rr * p->ncol + cc) < (r * p->ncol + c));
      if (!lower) continue;
      score = z0 - zn;
    }
    if (score > best) {
      best = score;
      code = ORC_NB_CODE[k];
    }
  }
  return code;
}

void orc_synth_d8(uint64_t seed, int64_t nrow, int64_t ncol, int64_t tilt, int64_t white,
                  int32_t nodata_pct, int64_t row0, int64_t nrows_out, uint8_t *out) {
  orc_synth_t p = {seed, nrow, ncol, tilt, white, nodata_pct};
  for (int64_t r = 0; r < nrows_out; ++r)
    for (int64_t c = 0; c < ncol; ++c) out[r * ncol + c] = orc_synth_code(&p, row0 + r, c);
}

/* float32 pseudo-elevation in "metres" for HAND tests/benches: z / 65536 */
void orc_synth_elev_f32(uint64_t seed, int64_t nrow, int64_t ncol, int64_t tilt, int64_t white,
                        int32_t nodata_pct, int64_t row0, int64_t nrows_out, float *out) {
  orc_synth_t p = {seed, nrow, ncol, tilt, white, nodata_pct};
  for (int64_t r = 0; r < nrows_out; ++r)
    for (int64_t c = 0; c < ncol; ++c)
      out[r * ncol + c] = (float)((double)orc_synth_z(&p, row0 + r, c) * (1.0 / 65536.0));
}

/* deterministic float32 weights U[0,1) from (seed, linear index) — config C3 */
void orc_synth_weights_f32(uint64_t seed, int64_t i0, int64_t n, float *out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = orc_mix64(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(i0 + i) + 0x632BE59BD9B4E019ULL);
    out[i] = (float)(h >> 40) * (1.0f / 16777216.0f);
  }
}
