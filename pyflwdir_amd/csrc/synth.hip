// synth.hip — device twin of the synthetic D8 generator in oracle/pfd_oracle.c
// (orc_synth_*).  Rasters for the large configurations are generated directly
// in HBM (a 90000 x 90000 raster is 8.1 GB and is never shipped over PCIe).  The integer
// arithmetic is identical to the host twin; tests compare the two byte for byte.
#include "common.h"

struct SynthP {
  u64 seed;
  i64 nrow, ncol, tilt, white;
  i32 nodata_pct;
};

__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
__device__ __forceinline__ u32 lat(u64 seed, u32 oct, i64 R, i64 C) {
  const u64 key = seed * 0x9E3779B97F4A7C15ULL + (u64)oct * 0xD1B54A32D192ED03ULL +
                  (((u64)(u32)R) << 32 | (u64)(u32)C);
  return (u32)(mix64(key) & 0xFFFFu);
}
__device__ __forceinline__ i64 octave(u64 seed, u32 k, i64 r, i64 c) {
  const i64 S = (i64)1 << k;
  const i64 R = r >> k, C = c >> k, fr = r & (S - 1), fc = c & (S - 1);
  const i64 v00 = lat(seed, k, R, C), v01 = lat(seed, k, R, C + 1);
  const i64 v10 = lat(seed, k, R + 1, C), v11 = lat(seed, k, R + 1, C + 1);
  const i64 top = v00 * (S - fc) + v01 * fc;
  const i64 bot = v10 * (S - fc) + v11 * fc;
  return (top * (S - fr) + bot * fr) >> k;
}
__device__ __forceinline__ i64 synth_z(const SynthP &p, i64 r, i64 c) {
  i64 z = p.tilt * (p.nrow - 1 - r);
  z += octave(p.seed, 3, r, c);
  z += octave(p.seed, 5, r, c);
  z += octave(p.seed, 7, r, c);
  z += octave(p.seed, 9, r, c);
  z += (i64)lat(p.seed, 0, r, c) * p.white;
  return z;
}
__device__ __forceinline__ bool synth_isnodata(const SynthP &p, i64 r, i64 c) {
  if (p.nodata_pct <= 0) return false;
  const i64 v = octave(p.seed ^ 0xA5A5A5A5ULL, 8, r, c) >> 8;
  return v * 100 < (i64)p.nodata_pct * 65536;
}

__global__ void __launch_bounds__(256) k_synth_d8(SynthP p, i64 row0, i64 nrows, u8 *__restrict__ out) {
  const i64 c = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const i64 rl = (i64)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (rl >= nrows || c >= p.ncol) return;
  const i64 r = row0 + rl;
  u8 code = 0;
  if (synth_isnodata(p, r, c)) {
    code = (u8)D8_MV;
  } else {
    const i64 z0 = synth_z(p, r, c);
    i64 best = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const i64 rr = r + d8_dr(k), cc = c + d8_dc(k);
      if (rr < 0 || cc < 0 || rr >= p.nrow || cc >= p.ncol) continue;
      i64 score;
      if (synth_isnodata(p, rr, cc)) {
        score = z0 + 1;
      } else {
        const i64 zn = synth_z(p, rr, cc);
        const bool lower = (zn < z0) || (zn == z0 && (rr * p.ncol + cc) < (r * p.ncol + c));
        if (!lower) continue;
        score = z0 - zn;
      }
      if (score > best) {
        best = score;
        code = (u8)(1u << k);
      }
    }
  }
  out[rl * p.ncol + c] = code;
}

__global__ void __launch_bounds__(256) k_synth_elev(SynthP p, i64 row0, i64 nrows, float *__restrict__ out) {
  const i64 c = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const i64 rl = (i64)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (rl >= nrows || c >= p.ncol) return;
  out[rl * p.ncol + c] = (float)((double)synth_z(p, row0 + rl, c) * (1.0 / 65536.0));
}

__global__ void k_synth_weights(u64 seed, i64 i0, i64 n, float *__restrict__ out) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 hsh = mix64(seed * 0x9E3779B97F4A7C15ULL + (u64)(i0 + i) + 0x632BE59BD9B4E019ULL);
  out[i] = (float)(hsh >> 40) * (1.0f / 16777216.0f);
}

static int synth_common(int device, i64 nrow, i64 ncol, i64 row0, i64 nrows, const void *out) {
  if (!out || nrow <= 0 || ncol <= 0 || row0 < 0 || nrows <= 0 || row0 + nrows > nrow) {
    pfd_set_error("pfd_synth: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  return PFD_OK;
}

extern "C" int pfd_synth_d8(int device, uint64_t seed, int64_t nrow, int64_t ncol, int64_t tilt, int64_t white,
                            int32_t nodata_pct, int64_t row0, int64_t nrows, uint8_t *out_dev) {
  PFDCHK(synth_common(device, nrow, ncol, row0, nrows, out_dev));
  SynthP p{seed, nrow, ncol, tilt, white, nodata_pct};
  // grid.y is limited to 65535 blocks of 4 rows: loop over row slabs
  const i64 SLAB = 65535LL * 4;
  for (i64 r0 = 0; r0 < nrows; r0 += SLAB) {
    const i64 nr = std::min(SLAB, nrows - r0);
    dim3 grid(cdiv_u32((u64)ncol, 64), cdiv_u32((u64)nr, 4));
    k_synth_d8<<<grid, 256>>>(p, row0 + r0, nr, out_dev + r0 * ncol);
  }
  KCHK();
  HIPCHK(hipDeviceSynchronize());
  return PFD_OK;
}

// a small base raster tiled over nrow x ncol cells, every copy inside a one-cell nodata frame (flow that left the
// base raster stays an outlet — the pit rule — instead of entering the neighbouring copy): the realistic regime of
// bench.py at sizes that are never shipped over PCIe
__global__ void __launch_bounds__(256) k_synth_mosaic(const u8 *__restrict__ base, i64 brow, i64 bcol, i64 row0, i64 nrows,
                                                      i64 ncol, u8 *__restrict__ out) {
  const i64 c = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const i64 rl = (i64)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (rl >= nrows || c >= ncol) return;
  const i64 rr = (row0 + rl) % (brow + 2), cc = c % (bcol + 2);
  const bool frame = rr == 0 || rr == brow + 1 || cc == 0 || cc == bcol + 1;
  out[rl * ncol + c] = frame ? (u8)D8_MV : base[(rr - 1) * bcol + (cc - 1)];
}
extern "C" int pfd_synth_mosaic(int device, const uint8_t *base_host, int64_t brow, int64_t bcol, int64_t nrow, int64_t ncol,
                                uint8_t *out_dev) {
  if (!base_host || !out_dev || brow <= 0 || bcol <= 0 || nrow <= 0 || ncol <= 0) {
    pfd_set_error("pfd_synth_mosaic: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  DevBuf base;
  PFDCHK(base.alloc((size_t)(brow * bcol)));
  HIPCHK(hipMemcpy(base.p, base_host, (size_t)(brow * bcol), hipMemcpyHostToDevice));
  const i64 SLAB = 65535LL * 4;
  for (i64 r0 = 0; r0 < nrow; r0 += SLAB) {
    const i64 nr = std::min(SLAB, nrow - r0);
    dim3 grid(cdiv_u32((u64)ncol, 64), cdiv_u32((u64)nr, 4));
    k_synth_mosaic<<<grid, 256>>>(base.as<u8>(), brow, bcol, r0, nr, ncol, out_dev + r0 * ncol);
  }
  KCHK();
  HIPCHK(hipDeviceSynchronize());
  return PFD_OK;
}

extern "C" int pfd_synth_elev_f32(int device, uint64_t seed, int64_t nrow, int64_t ncol, int64_t tilt,
                                  int64_t white, int32_t nodata_pct, int64_t row0, int64_t nrows, float *out_dev) {
  PFDCHK(synth_common(device, nrow, ncol, row0, nrows, out_dev));
  SynthP p{seed, nrow, ncol, tilt, white, nodata_pct};
  const i64 SLAB = 65535LL * 4;
  for (i64 r0 = 0; r0 < nrows; r0 += SLAB) {
    const i64 nr = std::min(SLAB, nrows - r0);
    dim3 grid(cdiv_u32((u64)ncol, 64), cdiv_u32((u64)nr, 4));
    k_synth_elev<<<grid, 256>>>(p, row0 + r0, nr, out_dev + r0 * ncol);
  }
  KCHK();
  HIPCHK(hipDeviceSynchronize());
  return PFD_OK;
}

extern "C" int pfd_synth_weights_f32(int device, uint64_t seed, int64_t i0, int64_t n, float *out_dev) {
  if (!out_dev || n <= 0) {
    pfd_set_error("pfd_synth_weights_f32: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  k_synth_weights<<<(unsigned)((n + 255) / 256), 256>>>(seed, i0, n, out_dev);
  KCHK();
  HIPCHK(hipDeviceSynchronize());
  return PFD_OK;
}
