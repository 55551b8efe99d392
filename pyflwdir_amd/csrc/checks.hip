// checks.hip — full-size verification of an upstream_area("cell") result by its LOCAL equations.
//
// On a raster without cycles the system   upa(x) = 1 + sum of upa over the cells draining into x
// (valid x),   upa(x) = -9999 (nodata x)   has exactly one solution — the result of the reference's
// streams.accuflux over ones (pyflwdir/streams.py:15-41, pyflwdir/pyflwdir.py:770-801).  Checking
// every cell's equation is one streaming pass that shares nothing with the engines that produced the
// result (no tiles, no ordering), needs no oracle and works at any size: it is how bench.py and the
// large-size tests certify the 8.1-Gcell result (SURVEY.md 8d, C4 checks ii/iii), together with the
// reference's own invariant "the upstream areas of the pits add up to the number of valid cells"
// (tests/test_streams_basins.py:24-27).
#include <algorithm>

#include "common.h"

__global__ void __launch_bounds__(256) k_verify_upa(const u8 *__restrict__ ncode, const i32 *__restrict__ upa, u32 nrow,
                                                    u32 ncol, unsigned long long *__restrict__ res) {
  __shared__ unsigned long long s[6];
  if (threadIdx.x < 6) s[threadIdx.x] = 0;
  __syncthreads();
  const u32 c = blockIdx.x * 64 + (threadIdx.x & 63);
  unsigned long long bad = 0, badmv = 0, pitsum = 0, npit = 0, csum = 0, nvalid = 0;
  for (u32 r = blockIdx.y * 4 + (threadIdx.x >> 6); r < nrow && c < ncol; r += gridDim.y * 4) {
    const size_t i = (size_t)r * ncol + c;
    const u32 code = ncode[i];
    const i32 v = upa[i];
    csum += (unsigned long long)(long long)v;
    if (code == D8_MV) {
      badmv += v != -9999;
      continue;
    }
    ++nvalid;
    u32 acc = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 rr = r + (u32)d8_dr(k), cc = c + (u32)d8_dc(k);
      if (rr < nrow && cc < ncol) {
        const size_t j = (size_t)rr * ncol + cc;
        if (ncode[j] == (1u << ((k + 4) & 7))) acc += (u32)upa[j];
      }
    }
    bad += acc != (u32)v;
    if (code == 0) {
      ++npit;
      pitsum += (unsigned long long)(u32)v;
    }
  }
  unsigned long long vals[6] = {bad, badmv, pitsum, npit, csum, nvalid};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned long long v = vals[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 6 && s[threadIdx.x]) atomicAdd(&res[threadIdx.x], s[threadIdx.x]);
}

extern "C" int pfd_verify_upstream_area_cell(pfd_raster *h, const int32_t *upa, int memspace, int64_t res[8]) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "verify_upstream_area_cell"));
  if (!upa || !res) {
    pfd_set_error("pfd_verify_upstream_area_cell: bad arguments");
    return PFD_EINVAL;
  }
  if (h->halo_top || h->halo_bot) {
    pfd_set_error("pfd_verify_upstream_area_cell: whole rasters only");
    return PFD_EUNSUPPORTED;
  }
  InArg in;
  PFDCHK(in.bind(upa, (size_t)h->n * sizeof(i32), memspace, h->stream));
  DevBuf acc;
  PFDCHK(acc.alloc(8 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(acc.p, 0, 8 * sizeof(unsigned long long), h->stream));
  const dim3 grid(cdiv_u32((u64)h->ncol, 64), std::min<u32>(cdiv_u32((u64)h->nrow, 4), 8192u));
  k_verify_upa<<<grid, 256, 0, h->stream>>>(h->ncode, (const i32 *)in.dev, (u32)h->nrow, (u32)h->ncol,
                                            acc.as<unsigned long long>());
  KCHK();
  unsigned long long r[8];
  HIPCHK(hipMemcpyAsync(r, acc.p, sizeof(r), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 8; ++k) res[k] = (int64_t)r[k];
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// basins labels and HAND by their local equations (twins of k_verify_upa): every cell is checked against its
// downstream cell only, so the kernels share nothing with the engines (no tiles, no ordering, no path queries).
// On a raster without cycles each system has exactly one solution, the reference's result:
//   labels (basins.basins + core.fillnodata_upstream, pyflwdir/basins.py:12-18, core.py:120-146):
//       seeded cell -> its seed;  other valid cell -> the label of its downstream cell, 0 for a pit;  nodata -> 0
//   HAND (dem.height_above_nearest_drain, pyflwdir/dem.py:299-330):
//       nodata -> -9999;  drain cell -> 0;  other cell -> hand[ds] + (double)(elevtn[x] - elevtn[ds]), the
//       difference in the elevation dtype (a pit is its own downstream cell: 0 + 0), compared bit for bit
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t chk_down(size_t i, u32 ncol, u32 code) {
  if (!d8_is_dir(code)) return i;
  const int k = d8_slot(code);
  return (size_t)((long long)i + (long long)d8_dr(k) * (long long)ncol + d8_dc(k));
}
__device__ __forceinline__ void chk_reduce(const unsigned long long (&vals)[4], unsigned long long *s,
                                           unsigned long long *__restrict__ res) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned long long v = vals[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s[threadIdx.x]) atomicAdd(&res[threadIdx.x], s[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_scatter_seed(const i64 *__restrict__ idx, const u32 *__restrict__ ids, u32 k,
                                                      u32 *__restrict__ seed) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) seed[idx[t]] = ids[t];
}
__global__ void __launch_bounds__(256) k_verify_labels(const u8 *__restrict__ ncode, const u32 *__restrict__ seed,
                                                       const u32 *__restrict__ lab, u32 nrow, u32 ncol,
                                                       unsigned long long *__restrict__ res) {
  __shared__ unsigned long long s[4];
  if (threadIdx.x < 4) s[threadIdx.x] = 0;
  __syncthreads();
  const u32 c = blockIdx.x * 64 + (threadIdx.x & 63);
  unsigned long long bad = 0, badmv = 0, csum = 0, nlab = 0;
  for (u32 r = blockIdx.y * 4 + (threadIdx.x >> 6); r < nrow && c < ncol; r += gridDim.y * 4) {
    const size_t i = (size_t)r * ncol + c;
    const u32 code = ncode[i], v = lab[i];
    csum += v;
    if (code == D8_MV) {
      badmv += v != 0u;
      continue;
    }
    nlab += v != 0u;
    const u32 sd = seed[i];
    u32 exp = sd;
    if (!sd) exp = d8_is_dir(code) ? lab[chk_down(i, ncol, code)] : 0u;
    bad += exp != v;
  }
  const unsigned long long vals[4] = {bad, badmv, csum, nlab};
  chk_reduce(vals, s, res);
}
template <class E>
__global__ void __launch_bounds__(256) k_verify_hand(const u8 *__restrict__ ncode, const u8 *__restrict__ drain,
                                                     const E *__restrict__ elev, const double *__restrict__ hand,
                                                     u32 nrow, u32 ncol, unsigned long long *__restrict__ res) {
  __shared__ unsigned long long s[4];
  if (threadIdx.x < 4) s[threadIdx.x] = 0;
  __syncthreads();
  const u32 c = blockIdx.x * 64 + (threadIdx.x & 63);
  unsigned long long bad = 0, badmv = 0, csum = 0, ndrain = 0;
  for (u32 r = blockIdx.y * 4 + (threadIdx.x >> 6); r < nrow && c < ncol; r += gridDim.y * 4) {
    const size_t i = (size_t)r * ncol + c;
    const u32 code = ncode[i];
    const double v = hand[i];
    csum += (unsigned long long)__double_as_longlong(v);
    if (code == D8_MV) {
      badmv += v != -9999.0;
      continue;
    }
    double exp = 0.0;
    if (drain[i] == 1) {
      ++ndrain;
    } else {
      const size_t p = chk_down(i, ncol, code);
      const E dz = elev[i] - elev[p];
      exp = (p == i ? 0.0 : hand[p]) + (double)dz;
    }
    bad += !(__double_as_longlong(exp) == __double_as_longlong(v) || (exp != exp && v != v));
  }
  const unsigned long long vals[4] = {bad, badmv, csum, ndrain};
  chk_reduce(vals, s, res);
}

static int chk_common(pfd_raster *h, const char *what, const void *a, const void *b) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, what));
  if (!a || !b) {
    pfd_set_error("%s: bad arguments", what);
    return PFD_EINVAL;
  }
  if (h->halo_top || h->halo_bot) {
    pfd_set_error("%s: whole rasters only", what);
    return PFD_EUNSUPPORTED;
  }
  return PFD_OK;
}
static int chk_finish(pfd_raster *h, DevBuf &acc, int64_t res[4]) {
  KCHK();
  unsigned long long r[4];
  HIPCHK(hipMemcpyAsync(r, acc.p, sizeof(r), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 4; ++k) res[k] = (int64_t)r[k];
  return PFD_OK;
}

extern "C" int pfd_verify_basins(pfd_raster *h, const int64_t *outlets, const uint32_t *ids, int64_t k,
                                 const uint32_t *labels, int memspace, int64_t res[4]) {
  PFDCHK(chk_common(h, "pfd_verify_basins", labels, res));
  if (k < 0 || (k > 0 && (!outlets || !ids))) {
    pfd_set_error("pfd_verify_basins: bad arguments");
    return PFD_EINVAL;
  }
  for (i64 j = 0; j < k; ++j)
    if (outlets[j] < 0 || outlets[j] >= h->n || ids[j] == 0) {
      pfd_set_error("pfd_verify_basins: outlet %lld outside the raster or id 0", (long long)j);
      return PFD_EINVAL;
    }
  InArg in, di, dl;
  PFDCHK(in.bind(labels, (size_t)h->n * sizeof(u32), memspace, h->stream));
  PFDCHK(di.bind(k ? outlets : nullptr, (size_t)k * sizeof(i64), PFD_HOST, h->stream));
  PFDCHK(dl.bind(k ? ids : nullptr, (size_t)k * sizeof(u32), PFD_HOST, h->stream));
  DevBuf seed, acc;
  PFDCHK(seed.alloc((size_t)h->n * sizeof(u32)));
  PFDCHK(acc.alloc(4 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(seed.p, 0, (size_t)h->n * sizeof(u32), h->stream));
  HIPCHK(hipMemsetAsync(acc.p, 0, 4 * sizeof(unsigned long long), h->stream));
  // (the outlets must be distinct: a repeated index would make the scatter's winner arbitrary)
  if (k) k_scatter_seed<<<cdiv_u32((u64)k, 256), 256, 0, h->stream>>>((const i64 *)di.dev, (const u32 *)dl.dev, (u32)k, seed.as<u32>());
  const dim3 grid(cdiv_u32((u64)h->ncol, 64), std::min<u32>(cdiv_u32((u64)h->nrow, 4), 8192u));
  k_verify_labels<<<grid, 256, 0, h->stream>>>(h->ncode, seed.as<u32>(), (const u32 *)in.dev, (u32)h->nrow, (u32)h->ncol,
                                              acc.as<unsigned long long>());
  return chk_finish(h, acc, res);
}

extern "C" int pfd_verify_hand(pfd_raster *h, const uint8_t *drain, int elev_dtype, const void *elevtn,
                               const double *hand, int memspace, int64_t res[4]) {
  PFDCHK(chk_common(h, "pfd_verify_hand", hand, res));
  if (!drain || !elevtn || (elev_dtype != PFD_F32 && elev_dtype != PFD_F64)) {
    pfd_set_error("pfd_verify_hand: bad arguments (elevation dtype code %d)", elev_dtype);
    return PFD_EINVAL;
  }
  InArg dr, el, ha;
  PFDCHK(dr.bind(drain, (size_t)h->n, memspace, h->stream));
  PFDCHK(el.bind(elevtn, (size_t)h->n * (elev_dtype == PFD_F32 ? 4 : 8), memspace, h->stream));
  PFDCHK(ha.bind(hand, (size_t)h->n * sizeof(double), memspace, h->stream));
  DevBuf acc;
  PFDCHK(acc.alloc(4 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(acc.p, 0, 4 * sizeof(unsigned long long), h->stream));
  const dim3 grid(cdiv_u32((u64)h->ncol, 64), std::min<u32>(cdiv_u32((u64)h->nrow, 4), 8192u));
  if (elev_dtype == PFD_F32)
    k_verify_hand<float><<<grid, 256, 0, h->stream>>>(h->ncode, (const u8 *)dr.dev, (const float *)el.dev,
                                                     (const double *)ha.dev, (u32)h->nrow, (u32)h->ncol,
                                                     acc.as<unsigned long long>());
  else
    k_verify_hand<double><<<grid, 256, 0, h->stream>>>(h->ncode, (const u8 *)dr.dev, (const double *)el.dev,
                                                      (const double *)ha.dev, (u32)h->nrow, (u32)h->ncol,
                                                      acc.as<unsigned long long>());
  return chk_finish(h, acc, res);
}

// sum of n int32 values (two's complement, 64 bit): the checksum the N-block runs compare with the 1-GPU run
__global__ void __launch_bounds__(256) k_checksum_i32(const i32 *__restrict__ v, u64 n, unsigned long long *__restrict__ res) {
  unsigned long long s = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) s += (unsigned long long)(long long)v[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  __shared__ unsigned long long sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(res, sh[0] + sh[1] + sh[2] + sh[3]);
}
extern "C" int pfd_checksum_i32(int device, const int32_t *dev_ptr, int64_t n, int64_t *sum) {
  if (!dev_ptr || n < 0 || !sum) {
    pfd_set_error("pfd_checksum_i32: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  unsigned long long host = 0;
  DevBuf acc;  // (the caching allocator: a raw hipMalloc here was a driver call per checksum; released on every way out)
  PFDCHK(acc.alloc(8));
  HIPCHK(hipMemsetAsync(acc.p, 0, 8, nullptr));
  if (n) k_checksum_i32<<<4096, 256>>>(dev_ptr, (u64)n, acc.as<unsigned long long>());
  KCHK();
  HIPCHK(hipMemcpy(&host, acc.p, 8, hipMemcpyDeviceToHost));
  *sum = (int64_t)host;
  return PFD_OK;
}

// number of NaN / +-inf values in a device array of float32 (dtype PFD_F32) or float64: what a caller with
// device-resident elevations asks before the row-block HAND, whose "-inf = not known yet" marker a non-finite
// elevation difference could imitate (host inputs are checked with numpy)
template <class T>
__global__ void __launch_bounds__(256) k_count_nonfinite(const T *__restrict__ v, u64 n, unsigned long long *res) {
  unsigned long long c = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) c += isfinite(v[i]) ? 0u : 1u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(res, c);
}
extern "C" int pfd_count_nonfinite(int device, int dtype, const void *dev_ptr, int64_t n, int64_t *count) {
  if (!dev_ptr || n < 0 || !count || (dtype != PFD_F32 && dtype != PFD_F64)) {
    pfd_set_error("pfd_count_nonfinite: bad arguments (dtype code %d)", dtype);
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  unsigned long long host = 0;
  DevBuf acc;  // (caching allocator, released on every way out; the scan is ordered behind the caller's work by the
               //  null stream, which waits for nothing here: the handles' streams are non-blocking, so the caller must
               //  have synchronised whatever produced dev_ptr — every API call of this library does before it returns)
  PFDCHK(acc.alloc(8));
  HIPCHK(hipMemsetAsync(acc.p, 0, 8, nullptr));
  if (n && dtype == PFD_F32) k_count_nonfinite<float><<<4096, 256>>>((const float *)dev_ptr, (u64)n, acc.as<unsigned long long>());
  if (n && dtype == PFD_F64) k_count_nonfinite<double><<<4096, 256>>>((const double *)dev_ptr, (u64)n, acc.as<unsigned long long>());
  KCHK();
  HIPCHK(hipMemcpy(&host, acc.p, 8, hipMemcpyDeviceToHost));
  *count = (int64_t)host;
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// core.snap, downstream direction, cell units (reference pyflwdir/core.py:440-480 with core._trace
// :316-366; Flwdir.snap flwdir.py:404-463): per start cell the first cell on its downstream path (the start
// itself included) where `mask` is set, or the pit the path ends in; dist = hops walked.  One thread per start
// cell; this is a bounded walk over k cells (k = number of outlets a user passes), not a raster sweep.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_snap_down(const u8 *__restrict__ ncode, u64 ncol, const u8 *__restrict__ mask,
                                                  const i64 *__restrict__ idx0, u32 k, i64 max_hops,
                                                  i64 *__restrict__ out, float *__restrict__ dist) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  u64 x = (u64)idx0[t];  // (64-bit cell indices: a walk over k cells serves a raster of any size)
  i64 d = 0;
  while (!mask[x]) {
    const u32 c = ncode[x];
    if (!d8_is_dir(c)) break;  // pit (or nodata: `idx1 == mv`)
    if (max_hops >= 0 && d + 1 > max_hops) break;
    const int s = d8_slot(c);
    x = (u64)((i64)x + (i64)d8_dr(s) * (i64)ncol + d8_dc(s));
    ++d;
  }
  out[t] = (i64)x;
  dist[t] = (float)d;
}
// The general form: downstream (decoded from the codes) or upstream along the MAIN upstream cells (`nxt_up`, the
// caller's idxs_us_main: int64, -1 = none), in cells or in metres.  The reference's loop (core.py:348-366):
//     while mask is None or not mask[idx0]:  idx1 = nxt[idx0];  if idx1 == idx0 or idx1 == mv: break
//         d = real_length ? distance(idx0, idx1) : 1.0;  if max_length is not None and dist + d > max_length: break
//         dist += d;  idx0 = idx1
// with `dist` a Python float: float64 here, rounded to float32 when stored (core.snap: dists float32).  A step's
// length depends on (row + row of the next cell, kind of step) only: `steps` = [2*nrow - 1][3] float64, evaluated
// by the host in the reference's expression order (gis_utils.distance, gis_utils.py:452-486).
__global__ void __launch_bounds__(64) k_snap(const u8 *__restrict__ ncode, u64 n, u64 ncol, const u8 *__restrict__ mask,
                                             const i64 *__restrict__ nxt_up, const double *__restrict__ steps,
                                             const i64 *__restrict__ idx0, u32 k, double max_length, int has_max,
                                             i64 *__restrict__ out, float *__restrict__ dist) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  u64 x = (u64)idx0[t];
  double d = 0.0;
  for (u64 guard = 0; guard <= n; ++guard) {  // (a cycle in a caller's idxs_us_main must not hang the GPU)
    if (mask && mask[x]) break;
    u64 y;
    if (nxt_up) {
      const i64 v = nxt_up[x];
      if (v < 0 || v == (i64)x || v >= (i64)n) break;
      y = (u64)v;
    } else {
      const u32 c = ncode[x];
      if (!d8_is_dir(c)) break;  // pit (or nodata: `idx1 == mv`)
      const int s = d8_slot(c);
      y = (u64)((i64)x + (i64)d8_dr(s) * (i64)ncol + d8_dc(s));
    }
    double step = 1.0;
    if (steps) {
      const u64 r0 = x / ncol, r1 = y / ncol;
      const u64 c0 = x - r0 * ncol, c1 = y - r1 * ncol;
      const int kind = r0 == r1 ? 1 : (c0 == c1 ? 0 : 2);  // vertical, horizontal, diagonal
      step = steps[(size_t)(r0 + r1) * 3 + kind];
    }
    if (has_max && d + step > max_length) break;
    d += step;
    x = y;
  }
  out[t] = (i64)x;
  dist[t] = (float)d;
}
extern "C" int pfd_snap(pfd_raster *h, const int64_t *idxs, int64_t k, const uint8_t *mask, const int64_t *idxs_us_main,
                        const double *step_lengths, int has_max_length, double max_length, int64_t *idxs_out,
                        float *dist_out) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "snap"));
  PFDCHK(pfd_require_unblocked(h, "snap"));  // (64-bit cell indices: any raster size)
  if (k < 0 || (k > 0 && (!idxs || !idxs_out || !dist_out))) {
    pfd_set_error("pfd_snap: bad arguments");
    return PFD_EINVAL;
  }
  if (k == 0) return PFD_OK;
  for (i64 i = 0; i < k; ++i)
    if (idxs[i] < 0 || idxs[i] >= h->n) {
      pfd_set_error("pfd_snap: index %lld outside the raster", (long long)idxs[i]);
      return PFD_EINVAL;
    }
  InArg di, dm, du, ds;
  PFDCHK(di.bind(idxs, (size_t)k * sizeof(i64), PFD_HOST, h->stream));
  PFDCHK(dm.bind(mask, (size_t)h->n, PFD_HOST, h->stream));
  PFDCHK(du.bind(idxs_us_main, (size_t)h->n * sizeof(i64), PFD_HOST, h->stream));
  PFDCHK(ds.bind(step_lengths, (size_t)(2 * h->nrow - 1) * 3 * sizeof(double), PFD_HOST, h->stream));
  DevBuf o, d;
  PFDCHK(o.alloc((size_t)k * sizeof(i64)));
  PFDCHK(d.alloc((size_t)k * sizeof(float)));
  k_snap<<<cdiv_u32((u64)k, 64), 64, 0, h->stream>>>(h->ncode, (u64)h->n, (u64)h->ncol, (const u8 *)dm.dev, (const i64 *)du.dev,
                                                    (const double *)ds.dev, (const i64 *)di.dev, (u32)k, max_length,
                                                    has_max_length, o.as<i64>(), d.as<float>());
  KCHK();
  HIPCHK(hipMemcpyAsync(idxs_out, o.p, (size_t)k * sizeof(i64), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(dist_out, d.p, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

extern "C" int pfd_snap_downstream(pfd_raster *h, const int64_t *idxs, int64_t k, const uint8_t *mask, int memspace,
                                   int64_t max_hops, int64_t *idxs_out, float *dist_out) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "snap"));
  PFDCHK(pfd_require_unblocked(h, "snap"));
  if (k < 0 || (k > 0 && (!idxs || !idxs_out || !dist_out)) || !mask) {
    pfd_set_error("pfd_snap_downstream: bad arguments");
    return PFD_EINVAL;
  }
  if (k == 0) return PFD_OK;
  for (i64 i = 0; i < k; ++i)
    if (idxs[i] < 0 || idxs[i] >= h->n) {
      pfd_set_error("pfd_snap_downstream: index %lld outside the raster", (long long)idxs[i]);
      return PFD_EINVAL;
    }
  InArg di, dm;
  PFDCHK(di.bind(idxs, (size_t)k * sizeof(i64), PFD_HOST, h->stream));
  PFDCHK(dm.bind(mask, (size_t)h->n, memspace, h->stream));
  DevBuf o, d;
  PFDCHK(o.alloc((size_t)k * sizeof(i64)));
  PFDCHK(d.alloc((size_t)k * sizeof(float)));
  k_snap_down<<<cdiv_u32((u64)k, 64), 64, 0, h->stream>>>(h->ncode, (u64)h->ncol, (const u8 *)dm.dev, (const i64 *)di.dev, (u32)k,
                                                         max_hops, o.as<i64>(), d.as<float>());
  KCHK();
  HIPCHK(hipMemcpyAsync(idxs_out, o.p, (size_t)k * sizeof(i64), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(dist_out, d.p, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}


// ---------------------------------------------------------------------------------------------
// known-bytes streaming kernels: calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters per access width
// (tools/prof_calib.sh -> profiles/pmc_traffic.json "_calibration_widths"; MI355X_MICROARCH.md, HBM: gfx950 tallies a
// 16 B/lane coalesced read at half its bytes — other widths are "calibrate on a known byte count")
// ---------------------------------------------------------------------------------------------
template <class W>
__global__ void __launch_bounds__(256) k_calib_read(const W *__restrict__ p, size_t n, unsigned long long *__restrict__ sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const W v = p[i];
    const unsigned char *b = (const unsigned char *)&v;
    acc += b[0];
  }
  if (acc == 0x7FFFFFFFFFFFFFFFull) *sink = acc;  // (keeps the loads alive; never true)
}
template <class W>
__global__ void __launch_bounds__(256) k_calib_write(W *__restrict__ p, size_t n) {
  W v;
  __builtin_memset(&v, 1, sizeof(W));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// one 4-byte load per 256 bytes: every access its own sector(s) — what a scattered gather costs per useful word
__global__ void __launch_bounds__(256) k_calib_read_strided(const u32 *__restrict__ p, size_t n, unsigned long long *__restrict__ sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i * 64] & 1u;
  if (acc == 0x7FFFFFFFFFFFFFFFull) *sink = acc;
}
extern "C" int pfd_calib_traffic(int device, void *buf_dev, size_t nbytes, int width, int write) {
  if (buf_dev && nbytes >= 256 && width == 256 && !write) {  // strided 4-byte reads, one per 256 bytes
    HIPCHK(hipSetDevice(device));
    DevBuf sk;
    PFDCHK(sk.alloc(8));
    k_calib_read_strided<<<256 * 64, 256>>>((const u32 *)buf_dev, nbytes / 256, sk.as<unsigned long long>());
    KCHK();
    HIPCHK(hipDeviceSynchronize());
    return PFD_OK;
  }
  if (!buf_dev || nbytes < 64 || (width != 1 && width != 4 && width != 8 && width != 16)) {
    pfd_set_error("pfd_calib_traffic: bad arguments");
    return PFD_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  DevBuf sink;
  PFDCHK(sink.alloc(8));
  const unsigned grid = 256 * 64;
  const size_t n = nbytes / (size_t)width;
#define CAL(W)                                                                                     \
  if (write) k_calib_write<W><<<grid, 256>>>((W *)buf_dev, n);                                     \
  else k_calib_read<W><<<grid, 256>>>((const W *)buf_dev, n, sink.as<unsigned long long>());
  if (width == 1) { CAL(u8) } else if (width == 4) { CAL(u32) } else if (width == 8) { CAL(u64) } else { CAL(uint4) }
#undef CAL
  KCHK();
  HIPCHK(hipDeviceSynchronize());
  return PFD_OK;
}
