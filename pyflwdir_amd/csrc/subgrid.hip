// subgrid.hip — unit catchments: subgrid.ucat_area (reference pyflwdir/subgrid.py:51-93;
// FlwdirRaster.ucat_area pyflwdir/pyflwdir.py:1159-1191) = a label flood from the unit-catchment outlets (the
// basins query of paths.hip) + a per-label sum of the cell areas.
//
// The reference accumulates `ucatch_are[label] += area[cell]` while it walks the cells in idxs_seq order, so a
// float sum depends on that order.  Integer areas (unit="cell") are a histogram (atomics, exact in any order);
// float areas are summed in the reference's order: cells of the exact idxs_seq order (order.hip) are stably
// sorted by label, then ONE LANE per label adds its cells from first to last — bit-identical.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "common.h"

int pfd_export_u32(pfd_raster *h, const u32 *src, i64 m, int idx_dtype, void *out, int memspace);  // api.hip

__global__ void k_mark_cells(const i64 *__restrict__ idx, u32 k, u8 *__restrict__ flag) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) flag[idx[t]] = 1;
}
__global__ void __launch_bounds__(256) k_ucat_count(const u32 *__restrict__ lab, const u8 *__restrict__ is_out, u64 n,
                                                    u32 *__restrict__ cnt) {
  // (grid-stride in whole workgroups: n may exceed the 2^32 threads one launch dimension runs)
  for (u64 x0 = (u64)blockIdx.x * blockDim.x; x0 < n; x0 += (u64)gridDim.x * blockDim.x) {
  const u64 x = x0 + threadIdx.x;
  u32 u = 0;
  if (x < n) {
    u = lab[x];
    if (is_out[x]) u = 0;
  }
  // neighbouring cells mostly carry the same label: the first lane of a run of equal labels adds the run's
  // length (a large catchment would otherwise queue up one same-address atomic per cell in L2)
  const u32 lane = threadIdx.x & 63u;
  const u32 prev = (u32)__shfl_up((int)u, 1);
  const bool head = lane == 0 || prev != u;
  const u64 heads = __ballot(head);
  if (head && u) {
    const u64 later = lane == 63u ? 0ull : (heads >> (lane + 1u));
    const u32 len = later ? (u32)__ffsll((long long)later) : 64u - lane;
    atomicAdd(&cnt[u - 1], len);
  }
  }
}
__global__ void __launch_bounds__(256) k_ucat_keys(const u32 *__restrict__ oseq, u32 nseq, const u32 *__restrict__ lab,
                                                   const u8 *__restrict__ is_out, u32 *__restrict__ keys) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nseq) return;
  const u32 x = oseq[j];
  const u32 u = lab[x];
  keys[j] = (u && !is_out[x]) ? u : 0u;
}
__global__ void __launch_bounds__(256) k_seg_bounds(const u32 *__restrict__ keys, u32 m, u32 *__restrict__ first,
                                                    u32 *__restrict__ last) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const u32 k = keys[j];
  if (k == 0) return;
  if (j == 0 || keys[j - 1] != k) first[k - 1] = j;
  if (j + 1 == m || keys[j + 1] != k) last[k - 1] = j + 1;
}
// one lane per label: its cells in idxs_seq order, added one after the other like the serial loop
template <class T>
__global__ void __launch_bounds__(64) k_ucat_sum(const u32 *__restrict__ cells, const u32 *__restrict__ first,
                                                 const u32 *__restrict__ last, u32 k, const T *__restrict__ rows, Geo g,
                                                 T *__restrict__ are) {
  const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= k) return;
  T acc = are[u];
  for (u32 j = first[u]; j < last[u]; ++j) acc = acc + rows[geo_row(g, cells[j])];
  are[u] = acc;
}

template <class T>
static int ucat_float(pfd_raster *h, const u32 *lab, const u8 *is_out, u32 k, const T *rows_dev, T *are_dev) {
  DevBuf oseq, keys, keys2, cells, bounds, tmp;
  PFDCHK(pfd_exact_seq_dev(h, oseq));
  const u32 m = (u32)h->n_seq;
  if (!m) return PFD_OK;
  PFDCHK(keys.alloc((size_t)m * sizeof(u32)));
  PFDCHK(keys2.alloc((size_t)m * sizeof(u32)));
  PFDCHK(cells.alloc((size_t)m * sizeof(u32)));
  PFDCHK(bounds.alloc(2 * (size_t)k * sizeof(u32)));
  HIPCHK(hipMemsetAsync(bounds.p, 0, 2 * (size_t)k * sizeof(u32), h->stream));  // first = last = 0: empty segment
  k_ucat_keys<<<cdiv_u32(m, 256), 256, 0, h->stream>>>(oseq.as<u32>(), m, lab, is_out, keys.as<u32>());
  int bits = 1;
  while ((1ull << bits) <= (u64)k) ++bits;
  size_t tb = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, keys.as<u32>(), keys2.as<u32>(), oseq.as<u32>(), cells.as<u32>(), (size_t)m,
                                   0u, (unsigned)bits, h->stream));
  PFDCHK(tmp.alloc(std::max<size_t>(tb, 16)));
  HIPCHK(rocprim::radix_sort_pairs(tmp.p, tb, keys.as<u32>(), keys2.as<u32>(), oseq.as<u32>(), cells.as<u32>(), (size_t)m, 0u,
                                   (unsigned)bits, h->stream));
  k_seg_bounds<<<cdiv_u32(m, 256), 256, 0, h->stream>>>(keys2.as<u32>(), m, bounds.as<u32>(), bounds.as<u32>() + k);
  k_ucat_sum<T><<<cdiv_u32(k, 64), 64, 0, h->stream>>>(cells.as<u32>(), bounds.as<u32>(), bounds.as<u32>() + k, k, rows_dev,
                                                     h->geo, are_dev);
  KCHK();
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// beyond 2^32 - 2 cells: the same sum over the 64-bit sequence (order64.hip), in pieces of the sequence — per piece the
// (label, area) pairs of its cells are stably sorted by label, then ONE WAVE per label adds the label's values of the
// piece to the label's running sum, one after the other: lane 0's chain of adds is the reference's loop, the other
// lanes only fetch (64 consecutive values per load, the next 64 in flight while the chain runs)
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_ucat_pairs64(const u64 *__restrict__ q, u64 m, const u32 *__restrict__ lab,
                                                      const u8 *__restrict__ is_out, const T *__restrict__ rows, u64 ncol,
                                                      u32 *__restrict__ keys, T *__restrict__ vals) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const u64 x = q[j];
  const u32 u = lab[x];
  keys[j] = (u && !is_out[x]) ? u : 0u;
  vals[j] = rows[x / ncol];
}
template <class T>
__global__ void __launch_bounds__(64) k_ucat_sum_wave(const T *__restrict__ vals, const u32 *__restrict__ first,
                                                      const u32 *__restrict__ last, u32 k, T *__restrict__ are) {
  const u32 u = blockIdx.x, lane = threadIdx.x;
  const u32 b = first[u], e = last[u];
  if (b >= e) return;
  T acc = are[u];
  T v = b + lane < e ? vals[b + lane] : (T)0;
  for (u32 j = b; j < e; j += 64u) {
    const T cur = v;
    const u32 nx = j + 64u + lane;
    v = nx < e ? vals[nx] : (T)0;  // (in flight while the chain below runs)
    const u32 cnt = min(64u, e - j);
    if (cnt == 64u) {
#pragma unroll
      for (int i = 0; i < 64; ++i) acc = acc + __shfl(cur, i);
    } else {
      for (u32 i = 0; i < cnt; ++i) acc = acc + __shfl(cur, (int)i);
    }
  }
  if (lane == 0) are[u] = acc;
}

template <class T>
static int ucat_float_wide(pfd_raster *h, const u32 *lab, const u8 *is_out, u32 k, const T *rows_dev, T *are_dev) {
  DevBuf q;
  u64 nseq = 0;
  PFDCHK(pfd_wide_seq_dev(h, q, &nseq));
  if (!nseq) return PFD_OK;
  u64 piece = 1ull << 28;
  if (const char *e = pfd_knob("PFD_UCAT_PIECE")) piece = std::max<u64>(64, (u64)atoll(e));  // (tests: several pieces of a small raster)
  piece = std::min(piece, nseq);
  DevBuf keys, keys2, vals, vals2, bounds, tmp;
  PFDCHK(keys.alloc((size_t)piece * sizeof(u32)));
  PFDCHK(keys2.alloc((size_t)piece * sizeof(u32)));
  PFDCHK(vals.alloc((size_t)piece * sizeof(T)));
  PFDCHK(vals2.alloc((size_t)piece * sizeof(T)));
  PFDCHK(bounds.alloc(2 * (size_t)k * sizeof(u32)));
  int bits = 1;
  while ((1ull << bits) <= (u64)k) ++bits;
  size_t tb = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, keys.as<u32>(), keys2.as<u32>(), vals.as<T>(), vals2.as<T>(), (size_t)piece, 0u,
                                   (unsigned)bits, h->stream));
  PFDCHK(tmp.alloc(std::max<size_t>(tb, 16)));
  pfd_seg_begin(h, "ucat_sums");
  i64 launches = 0;
  for (u64 p0 = 0; p0 < nseq; p0 += piece) {
    const u64 m = std::min(piece, nseq - p0);
    k_ucat_pairs64<T><<<cdiv_u32(m, 256), 256, 0, h->stream>>>(q.as<u64>() + p0, m, lab, is_out, rows_dev, (u64)h->ncol, keys.as<u32>(),
                                                              vals.as<T>());
    size_t tb2 = tb;
    HIPCHK(rocprim::radix_sort_pairs(tmp.p, tb2, keys.as<u32>(), keys2.as<u32>(), vals.as<T>(), vals2.as<T>(), (size_t)m, 0u,
                                     (unsigned)bits, h->stream));
    HIPCHK(hipMemsetAsync(bounds.p, 0, 2 * (size_t)k * sizeof(u32), h->stream));  // first = last = 0: empty segment
    k_seg_bounds<<<cdiv_u32(m, 256), 256, 0, h->stream>>>(keys2.as<u32>(), (u32)m, bounds.as<u32>(), bounds.as<u32>() + k);
    k_ucat_sum_wave<T><<<k, 64, 0, h->stream>>>(vals2.as<T>(), bounds.as<u32>(), bounds.as<u32>() + k, k, are_dev);
    launches += 5;
  }
  KCHK();
  pfd_seg_end(h, launches);
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

extern "C" int pfd_ucat_area(pfd_raster *h, const int64_t *idxs_out, int64_t k, int map_dtype, void *map_out, int memspace,
                             int area_dtype, const void *area_rows, void *area_out) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_reject_general(h, "ucat_area"));
  PFDCHK(pfd_require_unblocked(h, "ucat_area"));
  const bool wide = pfd_wide_cells(h);  // (64-bit cell indices: the label query runs at any size, the float sums walk order64.hip's sequence)
  if (!idxs_out || k < 0 || k >= 0xFFFFFFFFll || !map_out || !area_out ||
      (area_dtype != PFD_I32 && area_dtype != PFD_F32 && area_dtype != PFD_F64) || (area_dtype != PFD_I32 && !area_rows)) {
    pfd_set_error("pfd_ucat_area: bad arguments");
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  const u64 n = (u64)h->n;
  // outlets: a missing value (< 0) is skipped; of a repeated cell the LAST entry owns the label
  // (`ucatch_map[idx0] = i + 1` in a loop over i); every valid entry starts with its own cell's area
  std::vector<i64> uidx, all_valid;
  std::vector<u32> uid;
  {
    std::unordered_map<i64, size_t> pos;
    for (i64 i = 0; i < k; ++i) {
      const i64 c = idxs_out[i];
      if (c < 0) continue;
      if (c >= (i64)n) {
        pfd_set_error("pfd_ucat_area: outlet index %lld outside the raster", (long long)c);
        return PFD_EINVAL;
      }
      all_valid.push_back(c);
      auto it = pos.find(c);
      if (it == pos.end()) {
        pos[c] = uidx.size();
        uidx.push_back(c);
        uid.push_back((u32)(i + 1));
      } else {
        uid[it->second] = (u32)(i + 1);
      }
    }
  }
  const u32 ku = (u32)uidx.size();
  InArg di, dl;
  PFDCHK(di.bind(ku ? uidx.data() : nullptr, (size_t)ku * sizeof(i64), PFD_HOST, h->stream));
  PFDCHK(dl.bind(ku ? uid.data() : nullptr, (size_t)ku * sizeof(u32), PFD_HOST, h->stream));
  DevBuf lab, is_out;
  PFDCHK(lab.alloc((size_t)n * sizeof(u32) + 64));
  PFDCHK(is_out.alloc((size_t)n));
  PFDCHK(pfd_basins_dev(h, (const i64 *)di.dev, dl.dev, ku, 4, lab.p));
  HIPCHK(hipMemsetAsync(is_out.p, 0, (size_t)n, h->stream));
  if (ku) k_mark_cells<<<cdiv_u32(ku, 256), 256, 0, h->stream>>>((const i64 *)di.dev, ku, is_out.as<u8>());
  KCHK();
  const size_t esz = area_dtype == PFD_F64 ? 8 : 4;
  std::vector<unsigned char> are((size_t)std::max<i64>(k, 1) * esz);
  auto area_of = [&](i64 cell, unsigned char *dst) {
    const i64 r = cell / h->ncol;
    if (area_dtype == PFD_I32) {
      const i32 one = 1;
      memcpy(dst, &one, 4);
    } else {
      memcpy(dst, (const unsigned char *)area_rows + (size_t)r * esz, esz);
    }
  };
  for (i64 i = 0; i < k; ++i) {
    unsigned char *dst = are.data() + (size_t)i * esz;
    if (idxs_out[i] < 0) {
      if (area_dtype == PFD_I32) {
        const i32 v = -9999;
        memcpy(dst, &v, 4);
      } else if (area_dtype == PFD_F32) {
        const float v = -9999.f;
        memcpy(dst, &v, 4);
      } else {
        const double v = -9999.;
        memcpy(dst, &v, 8);
      }
    } else {
      area_of(idxs_out[i], dst);
    }
  }
  if (k) {
    DevBuf are_dev;
    PFDCHK(are_dev.alloc((size_t)k * esz));
    HIPCHK(hipMemcpyAsync(are_dev.p, are.data(), (size_t)k * esz, hipMemcpyHostToDevice, h->stream));
    if (area_dtype == PFD_I32) {
      // (int32 adds commute: the counts are added to the start values, wrapping like the reference's int32)
      k_ucat_count<<<(u32)std::min<u64>(cdiv_u32(n, 256), 1u << 22), 256, 0, h->stream>>>(lab.as<u32>(), is_out.as<u8>(), n, are_dev.as<u32>());
      KCHK();
    } else {
      InArg rows;
      PFDCHK(rows.bind(area_rows, (size_t)h->nrow * esz, PFD_HOST, h->stream));
      if (wide && area_dtype == PFD_F32)
        PFDCHK(ucat_float_wide<float>(h, lab.as<u32>(), is_out.as<u8>(), (u32)k, (const float *)rows.dev, are_dev.as<float>()));
      else if (wide)
        PFDCHK(ucat_float_wide<double>(h, lab.as<u32>(), is_out.as<u8>(), (u32)k, (const double *)rows.dev, are_dev.as<double>()));
      else if (area_dtype == PFD_F32)
        PFDCHK(ucat_float<float>(h, lab.as<u32>(), is_out.as<u8>(), (u32)k, (const float *)rows.dev, are_dev.as<float>()));
      else
        PFDCHK(ucat_float<double>(h, lab.as<u32>(), is_out.as<u8>(), (u32)k, (const double *)rows.dev, are_dev.as<double>()));
    }
    HIPCHK(hipMemcpyAsync(area_out, are_dev.p, (size_t)k * esz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return pfd_export_u32(h, lab.as<u32>(), (i64)n, map_dtype, map_out, memspace);
}
