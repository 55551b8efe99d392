// general.hip — the general `idxs_ds` engine: flow graphs whose links are NOT restricted to the 8 neighbours
// (NEXTXY rasters of CaMa-Flood, reference pyflwdir/core_nextxy.py:41-68; upscaled networks built with
// FlwdirRaster(idxs_ds=...), reference pyflwdir/pyflwdir.py:1079-1085).  SURVEY.md App. A: "keep a general
// idxs_ds-CSR fallback path".
//
// Such graphs are small next to the D8 rasters they are derived from (an upscaled cell stands for 10^2..10^4
// fine cells), so this engine is the plain level-synchronous formulation the north star describes: the
// downstream index per cell, an upstream CSR (stable radix sort of the edges by target: upstream cells in
// ascending index), the exact breadth-first order of core.idxs_seq (core.py:87-117) built level by level
// (count / scan / scatter), and pull sweeps with one launch per level — children combined in descending
// index, the serial loop's order, so float results are bit-identical here too.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <unordered_set>
#include <vector>

#include "common.h"

int pfd_export_u32(pfd_raster *h, const u32 *src, i64 m, int idx_dtype, void *out, int memspace);  // api.hip

#define GNONE 0xFFFFFFFFu

struct GenGraph {
  u32 *ds = nullptr;    // [n] downstream cell; self = pit; GNONE = nodata
  u32 *coff = nullptr;  // [n + 1] CSR offsets of the upstream cells
  u32 *cidx = nullptr;  // [edges] upstream cells, ascending per cell
  bool csr_ready = false;
  u32 *seq = nullptr;   // [n_seq] exact core.idxs_seq order (or the order installed by pfd_set_idxs_seq)
  u32 *pos = nullptr;   // [n] position in an installed order (then the CSR lists upstream cells by ascending position)
  std::vector<i64> lvl_off;
  bool ordered = false;
};
static GenGraph *G(pfd_raster *h) { return (GenGraph *)h->gen; }

void pfd_free_general(pfd_raster *h) {
  GenGraph *g = G(h);
  if (!g) return;
  pfd_dfree(g->ds);
  pfd_dfree(g->coff);
  pfd_dfree(g->cidx);
  pfd_dfree(g->seq);
  pfd_dfree(g->pos);
  delete g;
  h->gen = nullptr;
}

// ---- construction --------------------------------------------------------------------------------------
template <class I>
__global__ void __launch_bounds__(256) k_gen_import(const I *__restrict__ src, u32 n, u32 *__restrict__ ds,
                                                    u8 *__restrict__ ncode, unsigned long long *__restrict__ cnt) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = false, pit = false, bad = false;
  if (x < n) {
    const I v = src[x];
    u32 d = GNONE, code = D8_MV;
    if (v != (I)-1) {  // (-1 cast to the index dtype = the reference's missing value)
      const unsigned long long t = (unsigned long long)v;
      if (t >= (unsigned long long)n) {
        bad = true;
      } else {
        d = (u32)t;
        code = d == x ? 0u : 1u;  // the byte raster only tells nodata / pit / other apart on this engine
        valid = true;
        pit = d == x;
      }
    }
    ds[x] = d;
    ncode[x] = (u8)code;
  }
  // one atomic per wave and counter (one per cell would queue up n same-address atomics in L2)
  const u32 nv = (u32)__popcll(__ballot(valid)), np = (u32)__popcll(__ballot(pit)), nb = (u32)__popcll(__ballot(bad));
  if ((threadIdx.x & 63u) == 0) {
    if (nv) atomicAdd(&cnt[0], (unsigned long long)nv);
    if (np) atomicAdd(&cnt[1], (unsigned long long)np);
    if (nb) atomicAdd(&cnt[2], (unsigned long long)nb);
  }
}
__global__ void __launch_bounds__(256) k_gen_check_targets(const u32 *__restrict__ ds, u32 n, unsigned long long *__restrict__ cnt) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const u32 d = ds[x];
  if (d != GNONE && ds[d] == GNONE) atomicAdd(&cnt[3], 1ull);  // drains into a nodata cell
}
// slot j of the edge list holds cell x = order[j] (nullptr: x = j): the stable sort by target then lists the
// upstream cells of a cell in that order — ascending index, or ascending position of an installed sequence
__global__ void __launch_bounds__(256) k_gen_edges(const u32 *__restrict__ ds, u32 n, const u32 *__restrict__ order,
                                                   u32 *__restrict__ keys, u32 *__restrict__ vals,
                                                   u32 *__restrict__ indeg) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u32 x = order ? order[j] : j;
  const u32 d = ds[x];
  const bool edge = d != GNONE && d != x;
  keys[j] = edge ? d : n;  // non-edges sort behind the last cell
  vals[j] = x;
  if (edge) atomicAdd(&indeg[d], 1u);
}
__global__ void __launch_bounds__(256) k_gen_pos(const u32 *__restrict__ seq, u32 m, u32 *__restrict__ pos) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) pos[seq[j]] = j;
}
// all n cells: the installed sequence first, then every other cell in ascending index (positions m .. n-1)
__global__ void __launch_bounds__(256) k_gen_rest(const u32 *__restrict__ pos, u32 n, u32 m, const u32 *__restrict__ rest_rank,
                                                  u32 *__restrict__ order) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  if (pos[x] == GNONE) order[m + rest_rank[x]] = x;
  else order[pos[x]] = x;
}
__global__ void __launch_bounds__(256) k_gen_notin(const u32 *__restrict__ pos, u32 n, u32 *__restrict__ flag) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x <= n) flag[x] = (x < n && pos[x] == GNONE) ? 1u : 0u;
}
// the sequence must be rank-ordered AND hold every cell once (pos[seq[j]] == j fails for a repeated cell; m = the
// number of cells that reach a pit, so "no repeats" + "every entry has a rank" = a permutation of those cells)
__global__ void __launch_bounds__(256) k_gen_check_seq(const u32 *__restrict__ useq, u32 m, const i32 *__restrict__ rank,
                                                       const u32 *__restrict__ pos, unsigned long long *__restrict__ bad) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const i32 r = rank[useq[j]];
  if (r < 0 || (j > 0 && rank[useq[j - 1]] > r) || pos[useq[j]] != j) atomicAdd(bad, 1ull);
}

static int gen_build_csr(pfd_raster *h) {
  GenGraph *g = G(h);
  if (g->csr_ready) return PFD_OK;
  const u32 n = h->geo.n;
  pfd_dfree(g->coff);
  pfd_dfree(g->cidx);
  g->coff = g->cidx = nullptr;
  PFDCHK(pfd_dmalloc((void **)&g->coff, ((size_t)n + 1) * sizeof(u32)));
  PFDCHK(pfd_dmalloc((void **)&g->cidx, (size_t)n * sizeof(u32)));
  DevBuf keys, keys2, vals, tmp;
  PFDCHK(keys.alloc((size_t)n * sizeof(u32)));
  PFDCHK(keys2.alloc((size_t)n * sizeof(u32)));
  PFDCHK(vals.alloc((size_t)n * sizeof(u32)));
  HIPCHK(hipMemsetAsync(g->coff, 0, ((size_t)n + 1) * sizeof(u32), h->stream));
  DevBuf order, flag, tmp0;
  if (g->pos && (h->n_seq < 0 || h->n_seq > (i64)n)) {  // (cannot happen: every path that invalidates n_seq frees pos)
    pfd_set_error("internal: an installed cell order without its length");
    return PFD_EINVAL;
  }
  if (g->pos) {  // an installed sequence decides the order of the upstream cells
    PFDCHK(order.alloc((size_t)n * sizeof(u32)));
    PFDCHK(flag.alloc(((size_t)n + 1) * sizeof(u32)));
    k_gen_notin<<<cdiv_u32(n + 1, 256), 256, 0, h->stream>>>(g->pos, n, flag.as<u32>());
    size_t tb0 = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tb0, flag.as<u32>(), flag.as<u32>(), 0u, (size_t)n + 1, rocprim::plus<u32>(), h->stream));
    PFDCHK(tmp0.alloc(std::max<size_t>(tb0, 16)));
    HIPCHK(rocprim::exclusive_scan(tmp0.p, tb0, flag.as<u32>(), flag.as<u32>(), 0u, (size_t)n + 1, rocprim::plus<u32>(), h->stream));
    k_gen_rest<<<cdiv_u32(n, 256), 256, 0, h->stream>>>(g->pos, n, (u32)h->n_seq, flag.as<u32>(), order.as<u32>());
  }
  k_gen_edges<<<cdiv_u32(n, 256), 256, 0, h->stream>>>(g->ds, n, g->pos ? order.as<u32>() : nullptr, keys.as<u32>(),
                                                       vals.as<u32>(), g->coff);
  KCHK();
  int bits = 1;
  while ((1ull << bits) <= (u64)n) ++bits;
  size_t tb = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, keys.as<u32>(), keys2.as<u32>(), vals.as<u32>(), g->cidx, (size_t)n, 0u,
                                   (unsigned)bits, h->stream));
  PFDCHK(tmp.alloc(std::max<size_t>(tb, 16)));
  HIPCHK(rocprim::radix_sort_pairs(tmp.p, tb, keys.as<u32>(), keys2.as<u32>(), vals.as<u32>(), g->cidx, (size_t)n, 0u,
                                   (unsigned)bits, h->stream));
  HIPCHK(rocprim::exclusive_scan(nullptr, tb, g->coff, g->coff, 0u, (size_t)n + 1, rocprim::plus<u32>(), h->stream));
  PFDCHK(tmp.alloc(std::max<size_t>(tb, 16)));
  HIPCHK(rocprim::exclusive_scan(tmp.p, tb, g->coff, g->coff, 0u, (size_t)n + 1, rocprim::plus<u32>(), h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  g->csr_ready = true;
  return PFD_OK;
}

extern "C" int pfd_raster_create_general(const void *idxs_ds, int idx_dtype, int64_t nrow, int64_t ncol, int memspace,
                                         int device, pfd_raster **out) {
  if (!out) {
    pfd_set_error("pfd_raster_create_general: NULL out");
    return PFD_EINVAL;
  }
  *out = nullptr;
  const size_t es = idx_dtype == PFD_I64 ? 8 : ((idx_dtype == PFD_I32 || idx_dtype == PFD_U32) ? 4 : 0);
  if (!idxs_ds || nrow <= 0 || ncol <= 0 || !es || (unsigned __int128)nrow * (unsigned __int128)ncol > 4294967294ull) {
    pfd_set_error("pfd_raster_create_general: invalid arguments (shape %lld x %lld, index dtype code %d)", (long long)nrow,
                  (long long)ncol, idx_dtype);
    return PFD_EINVAL;
  }
  pfd_raster *h = nullptr;
  PFDCHK(pfd_handle_alloc(nrow, ncol, device, &h));
  GenGraph *g = new GenGraph();
  h->gen = g;
  const u32 n = h->geo.n;
  int rc = PFD_OK;
  do {
    InArg in;
    if ((rc = in.bind(idxs_ds, (size_t)n * es, memspace, h->stream)) != PFD_OK) break;
    if ((rc = pfd_dmalloc((void **)&g->ds, (size_t)n * sizeof(u32))) != PFD_OK) break;
    if (hipMemsetAsync(h->ctrl, 0, 64 * sizeof(u64), h->stream) != hipSuccess) {
      rc = PFD_EHIP;
      break;
    }
    const u32 grid = cdiv_u32(n, 256);
    unsigned long long *cnt = (unsigned long long *)h->ctrl;
    if (idx_dtype == PFD_I32)
      k_gen_import<i32><<<grid, 256, 0, h->stream>>>((const i32 *)in.dev, n, g->ds, h->ncode, cnt);
    else if (idx_dtype == PFD_U32)
      k_gen_import<u32><<<grid, 256, 0, h->stream>>>((const u32 *)in.dev, n, g->ds, h->ncode, cnt);
    else
      k_gen_import<i64><<<grid, 256, 0, h->stream>>>((const i64 *)in.dev, n, g->ds, h->ncode, cnt);
    k_gen_check_targets<<<grid, 256, 0, h->stream>>>(g->ds, n, cnt);
    u64 c[4];
    if (hipMemcpyAsync(c, h->ctrl, sizeof(c), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
      rc = PFD_EHIP;
      break;
    }
    if (c[2] || c[3]) {
      pfd_set_error("invalid idxs_ds: %llu index(es) outside the raster, %llu cell(s) draining into a nodata cell",
                    (unsigned long long)c[2], (unsigned long long)c[3]);
      rc = PFD_EINVAL;
      break;
    }
    h->n_valid = (i64)c[0];
    h->n_pits = (i64)c[1];
    if (h->n_pits == 0) {
      pfd_set_error("Invalid FlwdirRaster: no pits found");
      rc = PFD_ENOPITS;
      break;
    }
  } while (0);
  if (rc != PFD_OK) {
    pfd_raster_destroy(h);
    return rc;
  }
  *out = h;
  return PFD_OK;
}

// ---- ordering: the exact breadth-first order ----------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gen_deg(const u32 *__restrict__ seq, u32 begin, u32 end,
                                                 const u32 *__restrict__ coff, u32 *__restrict__ deg) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j > end) return;
  u32 d = 0;
  if (j < end) {
    const u32 x = seq[j];
    d = coff[x + 1] - coff[x];
  }
  deg[j - begin] = d;  // (one extra zero: the exclusive scan then ends with the total)
}
__global__ void __launch_bounds__(256) k_gen_expand(u32 *__restrict__ seq, u32 begin, u32 end, const u32 *__restrict__ coff,
                                                    const u32 *__restrict__ cidx, const u32 *__restrict__ offs) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j];
  u32 pos = end + offs[j - begin];
  for (u32 e = coff[x]; e < coff[x + 1]; ++e) seq[pos++] = cidx[e];
}

static int gen_order(pfd_raster *h) {
  GenGraph *g = G(h);
  if (g->ordered) return PFD_OK;
  PFDCHK(gen_build_csr(h));
  PFDCHK(pfd_ensure_pits(h));
  pfd_dfree(g->seq);
  g->seq = nullptr;
  PFDCHK(pfd_dmalloc((void **)&g->seq, (size_t)std::max<i64>(h->n_valid, 1) * sizeof(u32)));
  HIPCHK(hipMemcpyAsync(g->seq, h->pits, (size_t)h->n_pits * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
  std::vector<i64> off{0, h->n_pits};
  DevBuf deg, tmp;
  size_t cap = 0;
  pfd_seg_begin(h, "order_cells_general");
  i64 launches = 0;
  for (;;) {
    const u32 begin = (u32)off[off.size() - 2], end = (u32)off.back();
    const u32 m = end - begin;
    if (m == 0) {
      off.pop_back();
      break;
    }
    if ((size_t)m + 1 > cap) {
      cap = (size_t)(m + 1) * 2;
      PFDCHK(deg.alloc(cap * sizeof(u32)));
    }
    k_gen_deg<<<cdiv_u32(m + 1, 256), 256, 0, h->stream>>>(g->seq, begin, end, g->coff, deg.as<u32>());
    size_t tb = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tb, deg.as<u32>(), deg.as<u32>(), 0u, (size_t)m + 1, rocprim::plus<u32>(), h->stream));
    PFDCHK(tmp.alloc(std::max<size_t>(tb, 16)));
    HIPCHK(rocprim::exclusive_scan(tmp.p, tb, deg.as<u32>(), deg.as<u32>(), 0u, (size_t)m + 1, rocprim::plus<u32>(), h->stream));
    u32 total = 0;
    HIPCHK(hipMemcpyAsync(&total, deg.as<u32>() + m, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if ((i64)end + total > h->n_valid) {  // (cannot happen on a forest; guards the buffer)
      pfd_set_error("general ordering: inconsistent graph");
      return PFD_EINVAL;
    }
    if (total) k_gen_expand<<<cdiv_u32(m, 256), 256, 0, h->stream>>>(g->seq, begin, end, g->coff, g->cidx, deg.as<u32>());
    launches += 3;
    off.push_back((i64)end + total);
  }
  KCHK();
  pfd_seg_end(h, launches);
  g->lvl_off = off;
  h->lvl_off = off;
  h->n_levels = (i64)off.size() - 1;
  h->n_seq = off.back();
  h->ordered = true;
  g->ordered = true;
  h->acyclic = h->n_seq == h->n_valid ? 1 : -1;
  return PFD_OK;
}

// one launch per level; `up`: deepest level first
template <class F>
static int gen_levels(pfd_raster *h, bool up, const char *name, F launch) {
  PFDCHK(gen_order(h));
  GenGraph *g = G(h);
  pfd_seg_begin(h, name);
  const i64 nl = h->n_levels;
  for (i64 t = 0; t < nl; ++t) {
    const i64 l = up ? nl - 1 - t : t;
    const u32 begin = (u32)g->lvl_off[l], end = (u32)g->lvl_off[l + 1];
    if (end > begin) launch(begin, end, l);
  }
  KCHK();
  pfd_seg_end(h, nl);
  return PFD_OK;
}

// ---- payload arithmetic (as in sweeps.hip) ---------------------------------------------------------------
template <class T> struct GNum { static __device__ __forceinline__ T add(T a, T b) { return a + b; } };
template <> struct GNum<i32> { static __device__ __forceinline__ i32 add(i32 a, i32 b) { return (i32)((u32)a + (u32)b); } };
template <> struct GNum<i64> { static __device__ __forceinline__ i64 add(i64 a, i64 b) { return (i64)((u64)a + (u64)b); } };

template <class T>
__global__ void __launch_bounds__(256) k_gen_accu_up(const u32 *__restrict__ seq, u32 begin, u32 end,
                                                     const u32 *__restrict__ coff, const u32 *__restrict__ cidx,
                                                     T *__restrict__ out, T nodata, int has_nodata) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j];
  T acc = out[x];
  const u32 e0 = coff[x];
  for (u32 e = coff[x + 1]; e > e0; --e) {  // descending index: the order of the serial loop (streams.py:36-40)
    const T a = out[cidx[e - 1]];
    if (!has_nodata || (acc != nodata && a != nodata)) acc = GNum<T>::add(acc, a);
  }
  out[x] = acc;
}
template <class T>
__global__ void __launch_bounds__(256) k_gen_accu_down(const u32 *__restrict__ seq, u32 begin, u32 end,
                                                       const u32 *__restrict__ ds, T *__restrict__ out, T nodata,
                                                       int has_nodata) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j], p = ds[x];
  if (p == x) return;
  const T a = out[x], pv = out[p];
  if (!has_nodata || (pv != nodata && a != nodata)) out[x] = GNum<T>::add(a, pv);
}
__global__ void __launch_bounds__(256) k_gen_strahler(const u32 *__restrict__ seq, u32 begin, u32 end,
                                                      const u32 *__restrict__ coff, const u32 *__restrict__ cidx,
                                                      const u8 *__restrict__ mask, u8 *__restrict__ out) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j];
  u32 m = 0, cnt = 0;
  for (u32 e = coff[x]; e < coff[x + 1]; ++e) {
    const u32 c = cidx[e];
    if (mask != nullptr && !mask[c]) continue;
    const u32 v = out[c];
    if (v > m) {
      m = v;
      cnt = 1;
    } else if (v == m) {
      ++cnt;
    }
  }
  u32 r;
  if (cnt == 0) r = (mask == nullptr || mask[x]) ? 1u : 0u;
  else r = cnt >= 2 ? m + 1 : m;
  out[x] = (u8)r;
}
template <class L>
__global__ void __launch_bounds__(256) k_gen_labels(const u32 *__restrict__ seq, u32 begin, u32 end,
                                                    const u32 *__restrict__ ds, L *__restrict__ out) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j], p = ds[x];
  if (out[x] != 0 || p == x) return;
  const L pv = out[p];
  if (pv != 0) out[x] = pv;
}
template <class E>
__global__ void __launch_bounds__(256) k_gen_hand(const u32 *__restrict__ seq, u32 begin, u32 end, const u32 *__restrict__ ds,
                                                  const u8 *__restrict__ drain, const E *__restrict__ elev,
                                                  double *__restrict__ out) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j], p = ds[x];
  if (drain[x] == 1) {
    out[x] = 0.0;
    return;
  }
  const E dz = elev[x] - elev[p];
  out[x] = (p == x ? 0.0 : out[p]) + (double)dz;
}
__global__ void __launch_bounds__(256) k_gen_dist(const u32 *__restrict__ seq, u32 begin, u32 end, const u32 *__restrict__ ds,
                                                  const u8 *__restrict__ mask, const float *__restrict__ steplen,
                                                  void *__restrict__ out) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j], p = ds[x];
  const bool reset = p == x || (mask != nullptr && mask[x]);
  if (steplen != nullptr)
    ((float *)out)[x] = reset ? 0.f : ((float *)out)[p] + steplen[x];
  else
    ((i32 *)out)[x] = reset ? 0 : (i32)((u32)((i32 *)out)[p] + 1u);
}
__global__ void __launch_bounds__(256) k_gen_classic(const u32 *__restrict__ seq, u32 begin, u32 end, const u32 *__restrict__ ds,
                                                     const u8 *__restrict__ flag, const u8 *__restrict__ mask,
                                                     u8 *__restrict__ out) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= end) return;
  const u32 x = seq[j], p = ds[x];
  u32 r = 0;
  if (mask == nullptr || mask[x]) r = p == x ? 1u : (((u32)out[p] + flag[x]) & 0xFFu);
  out[x] = (u8)r;
}
__global__ void __launch_bounds__(256) k_gen_rank(const u32 *__restrict__ seq, u32 begin, u32 end, i32 level, i32 *__restrict__ out) {
  const u32 j = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (j < end) out[seq[j]] = level;
}
template <class T>
__global__ void k_gen_fill(T *__restrict__ out, u32 n, T v) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}
template <class T>
__global__ void k_gen_fill_valid(const u32 *__restrict__ ds, u32 n, T *__restrict__ out, T valid, T invalid) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ds[i] == GNONE ? invalid : valid;
}
template <class T>
__global__ void k_gen_rows(const T *__restrict__ row, Geo g, T *__restrict__ out) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < g.n) out[x] = row[geo_row(g, x)];
}
template <class T>
__global__ void k_gen_mask_invalid(const u32 *__restrict__ ds, u32 n, T *__restrict__ out, T v) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ds[i] == GNONE) out[i] = v;
}
template <class I>
__global__ void k_gen_export_ds(const u32 *__restrict__ ds, u32 n, I *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ds[i] == GNONE ? (I)-1 : (I)ds[i];
}
__global__ void __launch_bounds__(256) k_gen_upcount(const u32 *__restrict__ ds, const u32 *__restrict__ coff,
                                                     const u32 *__restrict__ cidx, const u8 *__restrict__ mask, u32 n,
                                                     int8_t *__restrict__ out) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  if (ds[x] == GNONE) {
    out[x] = -9;
    return;
  }
  int c = 0;
  for (u32 e = coff[x]; e < coff[x + 1]; ++e) c += (mask == nullptr || mask[cidx[e]]) ? 1 : 0;
  out[x] = (int8_t)c;
}
template <class T, class I>
__global__ void __launch_bounds__(256) k_gen_main_upstream(const u32 *__restrict__ ds, const u32 *__restrict__ coff,
                                                           const u32 *__restrict__ cidx, const T *__restrict__ upa,
                                                           T upa_min, u32 n, I *__restrict__ out) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  I best_i = (I)-1;
  if (ds[x] != GNONE) {
    T best = upa_min;
    // core.py:191-219 scans the cells in ascending index and keeps a strictly larger area: the first maximum in
    // ascending index (the CSR may list the upstream cells in another order: pick the smallest index among equals)
    bool any = false;
    u32 arg = 0;
    for (u32 e = coff[x]; e < coff[x + 1]; ++e) {
      const u32 c = cidx[e];
      const T a = upa[c];
      if (a > best || (any && a == best && c < arg)) {
        best = a;
        arg = c;
        any = true;
      }
    }
    if (any) best_i = (I)arg;
  }
  out[x] = best_i;
}
template <class I>
__global__ void __launch_bounds__(256) k_gen_trib_flag(const u32 *__restrict__ ds, const u32 *__restrict__ coff,
                                                       const u32 *__restrict__ cidx, const I *__restrict__ main_us,
                                                       const u8 *__restrict__ mask, u32 n, u8 *__restrict__ flag) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  u8 f = 0;
  const u32 p = ds[x];
  if (p != GNONE && p != x) {
    u32 nup = 0;
    for (u32 e = coff[p]; e < coff[p + 1]; ++e) nup += (mask == nullptr || mask[cidx[e]]) ? 1u : 0u;
    f = (nup > 1 && main_us[p] != (I)x) ? 1 : 0;
  }
  flag[x] = f;
}
template <class L>
__global__ void k_gen_seed(const i64 *__restrict__ idx, const L *__restrict__ ids, u32 k, L *__restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) out[idx[t]] = ids[t];
}
__global__ void k_gen_add_pits(u32 *__restrict__ ds, u8 *__restrict__ ncode, const i64 *__restrict__ idx, u32 k) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const u32 x = (u32)idx[t];
  ds[x] = x;
  ncode[x] = 0;
}

// ---- entry points (called from the C-ABI functions when the handle is a general graph) -------------------------
int pfd_gen_idxs_ds(pfd_raster *h, int idx_dtype, void *out, int memspace) {
  const size_t es = idx_dtype == PFD_I64 ? 8 : 4;
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * es, memspace));
  const u32 n = h->geo.n, grid = cdiv_u32(n, 256);
  if (idx_dtype == PFD_I32) k_gen_export_ds<i32><<<grid, 256, 0, h->stream>>>(G(h)->ds, n, (i32 *)o.dev);
  else if (idx_dtype == PFD_U32) k_gen_export_ds<u32><<<grid, 256, 0, h->stream>>>(G(h)->ds, n, (u32 *)o.dev);
  else k_gen_export_ds<i64><<<grid, 256, 0, h->stream>>>(G(h)->ds, n, (i64 *)o.dev);
  KCHK();
  return o.finish(h->stream);
}
int pfd_gen_order(pfd_raster *h) { return gen_order(h); }
int pfd_gen_idxs_seq(pfd_raster *h, int idx_dtype, void *out, int memspace) {
  PFDCHK(gen_order(h));
  return pfd_export_u32(h, G(h)->seq, h->n_seq, idx_dtype, out, memspace);
}
int pfd_gen_rank(pfd_raster *h, i32 *out, int memspace) {
  PFDCHK(gen_order(h));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(i32), memspace));
  const u32 n = h->geo.n;
  k_gen_fill_valid<i32><<<cdiv_u32(n, 256), 256, 0, h->stream>>>(G(h)->ds, n, (i32 *)o.dev, -1, -9999);
  PFDCHK(gen_levels(h, false, "rank", [&](u32 b, u32 e, i64 l) {
    k_gen_rank<<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(G(h)->seq, b, e, (i32)l, (i32 *)o.dev);
  }));
  return o.finish(h->stream);
}
int pfd_gen_upstream_count(pfd_raster *h, const u8 *mask, int8_t *out, int memspace) {
  PFDCHK(gen_build_csr(h));
  InArg m;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  k_gen_upcount<<<cdiv_u32(h->geo.n, 256), 256, 0, h->stream>>>(G(h)->ds, G(h)->coff, G(h)->cidx, (const u8 *)m.dev, h->geo.n,
                                                               (int8_t *)o.dev);
  KCHK();
  return o.finish(h->stream);
}
template <class T>
static int gen_accuflux_t(pfd_raster *h, const void *data, bool by_row, T nodata, int has_nodata, int direction,
                          int mask_invalid, void *out, int memspace) {
  InArg d;
  if (by_row) PFDCHK(d.bind(data, (size_t)h->nrow * sizeof(T), PFD_HOST, h->stream));
  else PFDCHK(d.bind(data, (size_t)h->n * sizeof(T), memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(T), memspace));
  const u32 n = h->geo.n;
  if (by_row) k_gen_rows<T><<<cdiv_u32(n, 256), 256, 0, h->stream>>>((const T *)d.dev, h->geo, (T *)o.dev);
  else HIPCHK(hipMemcpyAsync(o.dev, d.dev, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
  GenGraph *g = G(h);
  if (direction == PFD_UP) {
    PFDCHK(gen_levels(h, true, "general_accuflux_up", [&](u32 b, u32 e, i64) {
      k_gen_accu_up<T><<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->coff, g->cidx, (T *)o.dev, nodata, has_nodata);
    }));
  } else {
    PFDCHK(gen_levels(h, false, "general_accuflux_down", [&](u32 b, u32 e, i64) {
      k_gen_accu_down<T><<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->ds, (T *)o.dev, nodata, has_nodata);
    }));
  }
  if (mask_invalid) k_gen_mask_invalid<T><<<cdiv_u32(n, 256), 256, 0, h->stream>>>(g->ds, n, (T *)o.dev, nodata);
  KCHK();
  return o.finish(h->stream);
}
int pfd_gen_accuflux(pfd_raster *h, int dtype, const void *data, bool by_row, int64_t nodata_i, double nodata_f,
                     int has_nodata, int direction, int mask_invalid, void *out, int memspace) {
  switch (dtype) {
    case PFD_I32: return gen_accuflux_t<i32>(h, data, by_row, (i32)nodata_i, has_nodata, direction, mask_invalid, out, memspace);
    case PFD_I64: return gen_accuflux_t<i64>(h, data, by_row, (i64)nodata_i, has_nodata, direction, mask_invalid, out, memspace);
    case PFD_F32: return gen_accuflux_t<float>(h, data, by_row, (float)nodata_f, has_nodata, direction, mask_invalid, out, memspace);
    case PFD_F64: return gen_accuflux_t<double>(h, data, by_row, nodata_f, has_nodata, direction, mask_invalid, out, memspace);
    default:
      pfd_set_error("pfd_accuflux: unsupported payload dtype code %d", dtype);
      return PFD_EUNSUPPORTED;
  }
}
int pfd_gen_upstream_area_cell(pfd_raster *h, i32 *out, int memspace) {
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(i32), memspace));
  const u32 n = h->geo.n;
  GenGraph *g = G(h);
  k_gen_fill_valid<i32><<<cdiv_u32(n, 256), 256, 0, h->stream>>>(g->ds, n, (i32 *)o.dev, 1, -9999);
  // (nodata cells are never in seq and never upstream of a valid cell: their -9999 stays untouched)
  PFDCHK(gen_levels(h, true, "general_count_up", [&](u32 b, u32 e, i64) {
    k_gen_accu_up<i32><<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->coff, g->cidx, (i32 *)o.dev, -9999, 1);
  }));
  return o.finish(h->stream);
}
int pfd_gen_strahler(pfd_raster *h, const u8 *mask, u8 *out, int memspace) {
  InArg m;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  HIPCHK(hipMemsetAsync(o.dev, 0, (size_t)h->n, h->stream));
  GenGraph *g = G(h);
  PFDCHK(gen_levels(h, true, "general_strahler", [&](u32 b, u32 e, i64) {
    k_gen_strahler<<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->coff, g->cidx, (const u8 *)m.dev, (u8 *)o.dev);
  }));
  return o.finish(h->stream);
}
int pfd_gen_basins(pfd_raster *h, const i64 *idx_dev, const void *ids_dev, u32 k, int id_size, void *out_dev) {
  HIPCHK(hipMemsetAsync(out_dev, 0, (size_t)h->n * id_size, h->stream));
  GenGraph *g = G(h);
  const u32 sg = cdiv_u32(std::max<u32>(k, 1), 256);
#define GEN_LABELS(L)                                                                                                   \
  do {                                                                                                                  \
    if (k) k_gen_seed<L><<<sg, 256, 0, h->stream>>>(idx_dev, (const L *)ids_dev, k, (L *)out_dev);                          \
    PFDCHK(gen_levels(h, false, "general_labels", [&](u32 b, u32 e, i64) {                                               \
      k_gen_labels<L><<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->ds, (L *)out_dev);                     \
    }));                                                                                                                \
  } while (0)
  switch (id_size) {
    case 1: GEN_LABELS(u8); break;
    case 2: GEN_LABELS(uint16_t); break;
    case 4: GEN_LABELS(u32); break;
    default: GEN_LABELS(u64); break;
  }
#undef GEN_LABELS
  return PFD_OK;
}
int pfd_gen_hand(pfd_raster *h, const u8 *drain, int elev_dtype, const void *elevtn, double *out, int memspace) {
  InArg dr, el;
  PFDCHK(dr.bind(drain, (size_t)h->n, memspace, h->stream));
  PFDCHK(el.bind(elevtn, (size_t)h->n * (elev_dtype == PFD_F32 ? 4 : 8), memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(double), memspace));
  k_gen_fill<double><<<cdiv_u32(h->geo.n, 256), 256, 0, h->stream>>>((double *)o.dev, h->geo.n, -9999.0);
  GenGraph *g = G(h);
  PFDCHK(gen_levels(h, false, "general_hand", [&](u32 b, u32 e, i64) {
    if (elev_dtype == PFD_F32)
      k_gen_hand<float><<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->ds, (const u8 *)dr.dev, (const float *)el.dev, (double *)o.dev);
    else
      k_gen_hand<double><<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->ds, (const u8 *)dr.dev, (const double *)el.dev, (double *)o.dev);
  }));
  return o.finish(h->stream);
}
// real_length: `step_lengths` holds one float32 PER CELL (distance to its downstream cell), n values, same memspace
int pfd_gen_stream_distance(pfd_raster *h, const u8 *mask, int real_length, const float *step_lengths, void *out, int memspace) {
  InArg m, sl;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  if (real_length) PFDCHK(sl.bind(step_lengths, (size_t)h->n * sizeof(float), PFD_HOST, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * 4, memspace));
  const u32 n = h->geo.n;
  if (real_length) k_gen_fill<float><<<cdiv_u32(n, 256), 256, 0, h->stream>>>((float *)o.dev, n, -9999.0f);
  else k_gen_fill<i32><<<cdiv_u32(n, 256), 256, 0, h->stream>>>((i32 *)o.dev, n, -9999);
  GenGraph *g = G(h);
  PFDCHK(gen_levels(h, false, "general_stream_distance", [&](u32 b, u32 e, i64) {
    k_gen_dist<<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->ds, (const u8 *)m.dev,
                                                             real_length ? (const float *)sl.dev : nullptr, o.dev);
  }));
  return o.finish(h->stream);
}
template <class T>
static int gen_main_upstream_t(pfd_raster *h, const void *upa, double upa_min, int idx_dtype, void *out) {
  GenGraph *g = G(h);
  const u32 n = h->geo.n, grid = cdiv_u32(n, 256);
  if (idx_dtype == PFD_I32) k_gen_main_upstream<T, i32><<<grid, 256, 0, h->stream>>>(g->ds, g->coff, g->cidx, (const T *)upa, (T)upa_min, n, (i32 *)out);
  else if (idx_dtype == PFD_U32) k_gen_main_upstream<T, u32><<<grid, 256, 0, h->stream>>>(g->ds, g->coff, g->cidx, (const T *)upa, (T)upa_min, n, (u32 *)out);
  else k_gen_main_upstream<T, i64><<<grid, 256, 0, h->stream>>>(g->ds, g->coff, g->cidx, (const T *)upa, (T)upa_min, n, (i64 *)out);
  KCHK();
  return PFD_OK;
}
int pfd_gen_main_upstream(pfd_raster *h, int dtype, const void *uparea, double upa_min, int idx_dtype, void *out, int memspace) {
  PFDCHK(gen_build_csr(h));
  const size_t es = idx_dtype == PFD_I64 ? 8 : 4, ps = (dtype == PFD_I32 || dtype == PFD_F32) ? 4 : 8;
  InArg a;
  PFDCHK(a.bind(uparea, (size_t)h->n * ps, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * es, memspace));
  int rc;
  switch (dtype) {
    case PFD_I32: rc = gen_main_upstream_t<i32>(h, a.dev, upa_min, idx_dtype, o.dev); break;
    case PFD_I64: rc = gen_main_upstream_t<i64>(h, a.dev, upa_min, idx_dtype, o.dev); break;
    case PFD_F32: rc = gen_main_upstream_t<float>(h, a.dev, upa_min, idx_dtype, o.dev); break;
    default: rc = gen_main_upstream_t<double>(h, a.dev, upa_min, idx_dtype, o.dev); break;
  }
  PFDCHK(rc);
  return o.finish(h->stream);
}
int pfd_gen_classic(pfd_raster *h, int idx_dtype, const void *idxs_us_main, const u8 *mask, u8 *out, int memspace) {
  PFDCHK(gen_build_csr(h));
  const size_t es = idx_dtype == PFD_I64 ? 8 : 4;
  InArg mu, m;
  PFDCHK(mu.bind(idxs_us_main, (size_t)h->n * es, memspace, h->stream));
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  DevBuf flag;
  PFDCHK(flag.alloc((size_t)h->n));
  HIPCHK(hipMemsetAsync(o.dev, 0, (size_t)h->n, h->stream));
  GenGraph *g = G(h);
  const u32 n = h->geo.n, grid = cdiv_u32(n, 256);
  if (idx_dtype == PFD_I32) k_gen_trib_flag<i32><<<grid, 256, 0, h->stream>>>(g->ds, g->coff, g->cidx, (const i32 *)mu.dev, (const u8 *)m.dev, n, flag.as<u8>());
  else if (idx_dtype == PFD_U32) k_gen_trib_flag<u32><<<grid, 256, 0, h->stream>>>(g->ds, g->coff, g->cidx, (const u32 *)mu.dev, (const u8 *)m.dev, n, flag.as<u8>());
  else k_gen_trib_flag<i64><<<grid, 256, 0, h->stream>>>(g->ds, g->coff, g->cidx, (const i64 *)mu.dev, (const u8 *)m.dev, n, flag.as<u8>());
  PFDCHK(gen_levels(h, false, "general_classic_order", [&](u32 b, u32 e, i64) {
    k_gen_classic<<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, g->ds, flag.as<u8>(), (const u8 *)m.dev, (u8 *)o.dev);
  }));
  return o.finish(h->stream);
}
int pfd_gen_add_pits(pfd_raster *h, const i64 *idxs, i64 k) {
  GenGraph *g = G(h);
  std::vector<i64> ok;
  for (i64 i = 0; i < k; ++i) {
    if (idxs[i] < 0 || idxs[i] >= h->n) {
      pfd_set_error("pfd_add_pits: index %lld outside the raster", (long long)idxs[i]);
      return PFD_EINVAL;
    }
    ok.push_back(idxs[i]);
  }
  InArg in;
  PFDCHK(in.bind(ok.data(), ok.size() * sizeof(i64), PFD_HOST, h->stream));
  k_gen_add_pits<<<cdiv_u32((u64)k, 256), 256, 0, h->stream>>>(g->ds, h->ncode, (const i64 *)in.dev, (u32)k);
  KCHK();
  HIPCHK(hipStreamSynchronize(h->stream));
  // an installed sequence (pfd_set_idxs_seq) describes the graph before the edit: drop it with everything derived
  // from it; the caller re-installs its order (FlwdirRaster.add_pits re-runs order_cells("sort") on NEXTXY rasters)
  pfd_dfree(g->pos);
  g->pos = nullptr;
  g->csr_ready = false;
  g->ordered = false;
  h->ordered = false;
  h->pits_ready = false;
  h->n_seq = h->n_levels = -1;
  return PFD_OK;
}

// Flwdir.order_cells("sort") (reference pyflwdir/flwdir.py:231-245: cells sorted by rank with numpy's argsort,
// whose order inside a rank is not the breadth-first one; NEXTXY rasters are always ordered this way,
// pyflwdir.py:292-297).  The serial loops add the upstream cells of a cell in the reverse order of THAT
// sequence, so a float accumulation depends on it: the host hands its sequence over, and the engine walks it.
extern "C" int pfd_set_idxs_seq(pfd_raster *h, int idx_dtype, const void *seq, int64_t n_seq) {
  PFDCHK(pfd_check_handle(h));
  if (!h->gen) {
    pfd_set_error("pfd_set_idxs_seq: only general idxs_ds graphs take an external cell order");
    return PFD_EUNSUPPORTED;
  }
  GenGraph *g = G(h);
  if (g->pos || !g->ordered) {  // (back to the own breadth-first order first)
    pfd_dfree(g->pos);
    g->pos = nullptr;
    g->csr_ready = false;
    g->ordered = false;
  }
  if (!seq && n_seq == 0) return PFD_OK;  // "forget the installed order": the next operation orders breadth-first
  PFDCHK(gen_order(h));  // own breadth-first order: levels + the number of cells in the sequence
  const size_t es = idx_dtype == PFD_I64 ? 8 : 4;
  if (!seq || (idx_dtype != PFD_I32 && idx_dtype != PFD_U32 && idx_dtype != PFD_I64) || n_seq != h->n_seq) {
    pfd_set_error("pfd_set_idxs_seq: the sequence must hold the %lld cells that drain to a pit", (long long)h->n_seq);
    return PFD_EINVAL;
  }
  const u32 n = h->geo.n, m = (u32)n_seq;
  std::vector<u32> useq(m);
  for (u32 j = 0; j < m; ++j) {
    const i64 v = idx_dtype == PFD_I64 ? ((const i64 *)seq)[j] : (idx_dtype == PFD_I32 ? (i64)((const i32 *)seq)[j] : (i64)((const u32 *)seq)[j]);
    if (v < 0 || v >= (i64)n) {
      pfd_set_error("pfd_set_idxs_seq: index %lld outside the raster", (long long)v);
      return PFD_EINVAL;
    }
    useq[j] = (u32)v;
  }
  (void)es;
  DevBuf rank, bad;
  PFDCHK(rank.alloc((size_t)n * sizeof(i32)));
  PFDCHK(bad.alloc(sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), h->stream));
  k_gen_fill<i32><<<cdiv_u32(n, 256), 256, 0, h->stream>>>(rank.as<i32>(), n, -1);
  PFDCHK(gen_levels(h, false, "rank", [&](u32 b, u32 e, i64 l) {
    k_gen_rank<<<cdiv_u32(e - b, 256), 256, 0, h->stream>>>(g->seq, b, e, (i32)l, rank.as<i32>());
  }));
  HIPCHK(hipMemcpyAsync(g->seq, useq.data(), (size_t)m * sizeof(u32), hipMemcpyHostToDevice, h->stream));
  PFDCHK(pfd_dmalloc((void **)&g->pos, (size_t)n * sizeof(u32)));
  HIPCHK(hipMemsetAsync(g->pos, 0xFF, (size_t)n * sizeof(u32), h->stream));
  if (m) {
    k_gen_pos<<<cdiv_u32(m, 256), 256, 0, h->stream>>>(g->seq, m, g->pos);
    k_gen_check_seq<<<cdiv_u32(m, 256), 256, 0, h->stream>>>(g->seq, m, rank.as<i32>(), g->pos, bad.as<unsigned long long>());
  }
  unsigned long long nbad = 0;
  HIPCHK(hipMemcpyAsync(&nbad, bad.p, sizeof(nbad), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (nbad) {  // (cells of a rank form one contiguous run: the level offsets of the own order stay valid)
    pfd_dfree(g->pos);
    g->pos = nullptr;
    g->csr_ready = g->ordered = false;
    h->ordered = false;
    pfd_set_error("pfd_set_idxs_seq: the sequence is not ordered from down- to upstream (by rank) or repeats a cell");
    return PFD_EINVAL;
  }
  g->csr_ready = false;  // upstream cells by ascending position from now on
  return gen_build_csr(h);
}
