// order64.hip — `rank` and the exact `core.idxs_seq` order with 64-BIT cell indices: rasters beyond 2^32 - 2 cells,
// the int64 rung of the reference's index ladder (pyflwdir/pyflwdir.py:105-127).
//
// The level engine (order.hip) addresses cells with 32 bits.  Two things about the reference's order make a 64-bit form
// cheap that needs none of that engine:
//   * a cell's level in the breadth-first queue of core.idxs_seq (core.py:87-117) is its rank — the hops to its pit
//     (core.rank, core.py:17-47) — and the tiled path query (paths.hip) answers the rank of every cell of a raster of any
//     size with 32-bit VALUES, addressing tiles and slots, not cells;
//   * level l + 1 of the queue is level l with every cell replaced by its upstream cells in ascending index order
//     (core.upstream_matrix fills a row in ascending order of the upstream index, core.py:64-84), so the exact order
//     follows from level 0 — the pits, ascending (core_d8.from_array, core_d8.py:42-67) — and the level SIZES, which are
//     a histogram of the ranks: no sort, no per-cell positions.
// Per level: in-degree per queue position + per-chunk sums, scan of the chunk sums, ordered scatter (the same three
// steps as order.hip's k_oseq_*, on u64 queue entries); small levels run the three steps in one workgroup.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "common.h"

int pfd_path_rank_max(pfd_raster *h, u32 *out_dev, int *complete, u32 *maxrank);  // paths.hip

// (PFD_TEST_ORDER64 with PFD_ENABLE_KNOBS=1: tests run the 64-bit form on small rasters against the 32-bit one)
bool pfd_wide_cells(const pfd_raster *h) { return h->n > 4294967294ll || pfd_knob("PFD_TEST_ORDER64") != nullptr; }

namespace {

#define W_CHUNK 1024u   // queue positions per workgroup (256 threads x 4 consecutive)
#define PITC 4096u      // cells per workgroup of the pit compaction (256 threads x 16 consecutive)

struct Shape {
  u64 nrow, ncol;
};

// upstream cells of x in ascending index order: NW N NE W E SW S SE
__device__ __forceinline__ u32 children_of(const u8 *__restrict__ ncode, const Shape &g, u64 x, u64 *kids) {
  const u64 r = x / g.ncol, c = x - r * g.ncol;
  u32 cnt = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = PFD_SLOT_ASC[q];
    const u64 rr = r + (u64)(i64)d8_dr(k), cc = c + (u64)(i64)d8_dc(k);  // wraps beyond nrow / ncol when negative
    if (rr >= g.nrow || cc >= g.ncol) continue;
    const u64 j = rr * g.ncol + cc;
    if (ncode[j] == (1u << ((k + 4) & 7))) {
      if (kids) kids[cnt] = j;
      ++cnt;
    }
  }
  return cnt;
}

// ranks -> the reference's int32 raster (-9999 on nodata), in place; histogram of the ranks
__global__ void __launch_bounds__(256) k_rank_hist(u32 *__restrict__ v, u64 n, u32 nlev, unsigned long long *__restrict__ hist,
                                                   bool convert) {
  for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) {
    const u32 x = v[i];
    if (x == 0xFFFFFFFFu) {
      if (convert) v[i] = (u32)-9999;
    } else if (hist && x < nlev) {
      atomicAdd(&hist[x], 1ull);
    }
  }
}

__global__ void __launch_bounds__(1024) k_scan_u64_1block(u64 *__restrict__ v, u64 m, u64 *__restrict__ total_out = nullptr) {  // in place, exclusive
  __shared__ u64 wsum[16];
  __shared__ u64 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (u64 base = 0; base < m; base += 1024) {
    const u64 i = base + threadIdx.x;
    const u64 x = i < m ? v[i] : 0;
    u64 incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const u64 y = __shfl_up(incl, o);
      if (lane >= (u32)o) incl += y;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    u64 woff = 0, total = 0;
    for (u32 w = 0; w < 16; ++w) {
      if (w < wid) woff += wsum[w];
      total += wsum[w];
    }
    if (i < m) v[i] = carry + woff + incl - x;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (total_out && threadIdx.x == 0) *total_out = carry;
}

// ---- level 0: the pits in ascending index order ------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pit_count64(const u8 *__restrict__ ncode, u64 n, u64 *__restrict__ counts) {
  const u64 base = (u64)blockIdx.x * PITC + threadIdx.x * 16u;
  u32 cnt = 0;
  for (u32 t = 0; t < 16; ++t)
    if (base + t < n && ncode[base + t] == 0) ++cnt;
  __shared__ u32 s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_pit_scatter64(const u8 *__restrict__ ncode, u64 n, const u64 *__restrict__ offs,
                                                       u64 *__restrict__ q) {
  const u64 base = (u64)blockIdx.x * PITC + threadIdx.x * 16u;  // a thread owns 16 CONSECUTIVE cells: linear order
  u32 mask = 0;
  for (u32 t = 0; t < 16; ++t)
    if (base + t < n && ncode[base + t] == 0) mask |= 1u << t;
  const u32 cnt = __popc(mask);
  __shared__ u32 wsum[4];
  const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  u32 incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const u32 y = __shfl_up(incl, o);
    if (lane >= (u32)o) incl += y;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  u32 woff = 0;
  for (u32 w = 0; w < wid; ++w) woff += wsum[w];
  u64 pos = offs[blockIdx.x] + woff + incl - cnt;
  while (mask) {
    const int t = __ffs((int)mask) - 1;
    mask &= mask - 1u;
    q[pos++] = base + (u64)t;
  }
}

// ---- level l -> level l + 1 ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_wseq_count(const u8 *__restrict__ ncode, Shape g, const u64 *__restrict__ q, u64 begin,
                                                    u64 end, u64 *__restrict__ chunk_sums) {
  const u64 j0 = begin + (u64)blockIdx.x * W_CHUNK + threadIdx.x * 4u;
  u32 cnt = 0;
  for (u32 t = 0; t < 4; ++t)
    if (j0 + t < end) cnt += children_of(ncode, g, q[j0 + t], nullptr);
  __shared__ u32 s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s, cnt);
  __syncthreads();
  if (threadIdx.x == 0) chunk_sums[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_wseq_scatter(const u8 *__restrict__ ncode, Shape g, u64 *__restrict__ q, u64 begin, u64 end,
                                                      const u64 *__restrict__ chunk_offs) {
  const u64 j0 = begin + (u64)blockIdx.x * W_CHUNK + threadIdx.x * 4u;
  u32 c4[4], cnt = 0;
  for (u32 t = 0; t < 4; ++t) {
    c4[t] = (j0 + t < end) ? children_of(ncode, g, q[j0 + t], nullptr) : 0;
    cnt += c4[t];
  }
  __shared__ u32 wsum[4];
  const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  u32 incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const u32 y = __shfl_up(incl, o);
    if (lane >= (u32)o) incl += y;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  u32 woff = 0;
  for (u32 w = 0; w < wid; ++w) woff += wsum[w];
  u64 pos = end + chunk_offs[blockIdx.x] + woff + incl - cnt;
  for (u32 t = 0; t < 4; ++t) {
    if (c4[t]) children_of(ncode, g, q[j0 + t], q + pos);
    pos += c4[t];
  }
}
// one workgroup walks a level of at most a few thousand cells: count, scan and scatter without leaving the kernel
__global__ void __launch_bounds__(1024) k_wseq_small(const u8 *__restrict__ ncode, Shape g, u64 *__restrict__ q, u64 begin, u64 end) {
  __shared__ u32 wsum[16];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (u64 base = begin; base < end; base += 1024) {
    const u64 j = base + threadIdx.x;
    const u64 x = j < end ? q[j] : 0;
    const u32 cnt = j < end ? children_of(ncode, g, x, nullptr) : 0;
    u32 incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      const u32 y = __shfl_up(incl, o);
      if (lane >= (u32)o) incl += y;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    u32 woff = 0, total = 0;
    for (u32 w = 0; w < 16; ++w) {
      if (w < wid) woff += wsum[w];
      total += wsum[w];
    }
    if (cnt) children_of(ncode, g, x, q + end + carry + woff + incl - cnt);
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
}

// ranks of every cell (u32, 0xFFFFFFFF on nodata) through the tiled path query
int wide_ranks(pfd_raster *h, DevBuf &keys, u32 *maxrank, const char *what, bool *cyclic) {
  PFDCHK(pfd_reject_general(h, what));
  if (h->halo_top || h->halo_bot) {
    pfd_set_error("%s is not available on a row-block handle", what);
    return PFD_EUNSUPPORTED;
  }
  PFDCHK(keys.alloc((size_t)h->n * sizeof(u32) + 64));
  int complete = 0;
  pfd_seg_begin(h, "tile_rank");
  PFDCHK(pfd_path_rank_max(h, keys.as<u32>(), &complete, maxrank));
  pfd_seg_end(h, 2);
  *cyclic = !complete;  // (cells that never reach a pit, or a raster beyond the slot ids of the tiled query: wide_bfs)
  return PFD_OK;
}

// ranks of a queue level -> the rank raster
__global__ void __launch_bounds__(256) k_wrank_init(const u8 *__restrict__ ncode, u64 n, i32 *__restrict__ rank) {
  for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) rank[i] = ncode[i] == D8_MV ? -9999 : -1;
}
__global__ void __launch_bounds__(256) k_wrank_level(const u64 *__restrict__ q, u64 begin, u64 end, i32 *__restrict__ rank, i32 level) {
  for (u64 j = begin + (u64)blockIdx.x * 256u + threadIdx.x; j < end; j += (u64)gridDim.x * 256u) rank[q[j]] = level;
}

// The reference's own walk for rasters the rank query cannot finish: cells on or upstream of a cycle never reach a pit,
// core.rank marks them -1 and core.idxs_seq leaves them out (pyflwdir/core.py:17-47, :87-117) — which is what a
// breadth-first expansion FROM THE PITS does by construction.  Level sizes are not known up front here (no rank
// histogram), so the host reads one number per level: 4 launches + a synchronisation per level, ~6 s for 10^5 levels —
// the price of a raster that is both cyclic and beyond 32-bit cell indices.  `rank_dev` (optional): -9999 / -1 / level.
int wide_bfs(pfd_raster *h, i32 *rank_dev, DevBuf &q, u64 *nseq) {
  const u64 n = (u64)h->n;
  const Shape g{(u64)h->nrow, (u64)h->ncol};
  const u64 cap = (u64)std::max<i64>(h->n_valid, 1);
  PFDCHK(q.alloc((size_t)cap * sizeof(u64)));
  const u64 npc = (n + PITC - 1) / PITC;
  DevBuf sums;
  PFDCHK(sums.alloc((size_t)(std::max<u64>((cap + W_CHUNK - 1) / W_CHUNK, npc) + 2) * sizeof(u64)));
  pfd_seg_begin(h, "idxs_seq_walk_from_pits");
  if (rank_dev) k_wrank_init<<<4096, 256, 0, h->stream>>>(h->ncode, n, rank_dev);
  k_pit_count64<<<(u32)npc, 256, 0, h->stream>>>(h->ncode, n, sums.as<u64>());
  k_scan_u64_1block<<<1, 1024, 0, h->stream>>>(sums.as<u64>(), npc);
  k_pit_scatter64<<<(u32)npc, 256, 0, h->stream>>>(h->ncode, n, sums.as<u64>(), q.as<u64>());
  KCHK();
  i64 launches = 4;
  u64 begin = 0, end = (u64)h->n_pits;
  for (i32 level = 0; end > begin; ++level) {
    const u64 m = end - begin, nchunk = (m + W_CHUNK - 1) / W_CHUNK;
    if (rank_dev) k_wrank_level<<<(u32)std::min<u64>((m + 255) / 256, 4096), 256, 0, h->stream>>>(q.as<u64>(), begin, end, rank_dev, level);
    k_wseq_count<<<(u32)nchunk, 256, 0, h->stream>>>(h->ncode, g, q.as<u64>(), begin, end, sums.as<u64>());
    k_scan_u64_1block<<<1, 1024, 0, h->stream>>>(sums.as<u64>(), nchunk, sums.as<u64>() + nchunk);
    u64 total = 0;
    HIPCHK(hipMemcpyAsync(&total, sums.as<u64>() + nchunk, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    launches += 3;
    if (end + total > cap) {
      pfd_set_error("idxs_seq: the walk from the pits found more cells than the raster has valid ones");
      return PFD_EHIP;
    }
    if (total) {
      k_wseq_scatter<<<(u32)nchunk, 256, 0, h->stream>>>(h->ncode, g, q.as<u64>(), begin, end, sums.as<u64>());
      ++launches;
    }
    begin = end, end += total;
  }
  KCHK();
  pfd_seg_end(h, launches);
  *nseq = end;
  return PFD_OK;
}

}  // namespace

// core.rank with 64-bit cell addressing: int32 ranks, -9999 on nodata
int pfd_rank_wide(pfd_raster *h, i32 *out, int memspace) {
  DevBuf keys;
  u32 maxrank = 0;
  bool cyclic = false;
  pfd_seg_clear(h);
  PFDCHK(wide_ranks(h, keys, &maxrank, "rank", &cyclic));
  if (cyclic) {  // cells that never reach a pit: the walk from the pits marks them -1 like the reference
    DevBuf q;
    u64 nseq = 0;
    PFDCHK(wide_bfs(h, keys.as<i32>(), q, &nseq));
    h->n_seq = (i64)nseq;
  } else {
    k_rank_hist<<<4096, 256, 0, h->stream>>>(keys.as<u32>(), (u64)h->n, 0u, nullptr, true);
    KCHK();
    h->n_seq = h->n_valid;
  }
  HIPCHK(hipMemcpyAsync(out, keys.p, (size_t)h->n * sizeof(i32), memspace == PFD_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                        h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

// core.idxs_seq with 64-bit cell indices, on the device: q holds *nseq queue entries (h->n_seq is set; a raster with
// cycles: the cells that reach a pit only, like the reference's sequence)
int pfd_wide_seq_dev(pfd_raster *h, DevBuf &q, u64 *nseq_out) {
  u32 maxrank = 0;
  const u64 n = (u64)h->n;
  std::vector<unsigned long long> cnt;
  bool cyclic = false;
  {
    DevBuf keys, hist;
    PFDCHK(wide_ranks(h, keys, &maxrank, "idxs_seq", &cyclic));
    if (cyclic) maxrank = 0;
    const u32 nlev = maxrank + 1u;
    PFDCHK(hist.alloc((size_t)nlev * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(hist.p, 0, (size_t)nlev * sizeof(unsigned long long), h->stream));
    k_rank_hist<<<4096, 256, 0, h->stream>>>(keys.as<u32>(), n, nlev, hist.as<unsigned long long>(), false);
    KCHK();
    cnt.resize(nlev);
    HIPCHK(hipMemcpyAsync(cnt.data(), hist.p, (size_t)nlev * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }  // (the ranks are released: the queue below is twice their size)
  if (cyclic) {  // the reference's sequence leaves the cells that never reach a pit out: n_seq < n_valid entries (h->n_seq)
    u64 nseq = 0;
    PFDCHK(wide_bfs(h, nullptr, q, &nseq));
    h->n_seq = (i64)nseq;
    *nseq_out = nseq;
    return PFD_OK;
  }
  h->n_seq = h->n_valid;
  const u32 nlev = maxrank + 1u;
  std::vector<u64> off(nlev + 1, 0);
  for (u32 l = 0; l < nlev; ++l) off[l + 1] = off[l] + cnt[l];
  if ((i64)off[nlev] != h->n_valid || (i64)cnt[0] != h->n_pits) {
    pfd_set_error("idxs_seq: the rank histogram (%llu cells, %llu pits) does not match the raster (%lld valid cells, %lld pits)",
                  (unsigned long long)off[nlev], cnt[0], (long long)h->n_valid, (long long)h->n_pits);
    return PFD_EHIP;
  }
  const u64 nseq = off[nlev];
  const Shape g{(u64)h->nrow, (u64)h->ncol};
  DevBuf sums;
  PFDCHK(q.alloc(std::max<size_t>((size_t)nseq, 1) * sizeof(u64)));
  u64 maxlev = 0;
  for (u32 l = 0; l < nlev; ++l) maxlev = std::max<u64>(maxlev, cnt[l]);
  const u64 npc = (n + PITC - 1) / PITC;
  PFDCHK(sums.alloc((size_t)std::max<u64>(std::max<u64>((maxlev + W_CHUNK - 1) / W_CHUNK, npc), 1) * sizeof(u64)));
  pfd_seg_begin(h, "idxs_seq_pits");
  k_pit_count64<<<(u32)npc, 256, 0, h->stream>>>(h->ncode, n, sums.as<u64>());
  k_scan_u64_1block<<<1, 1024, 0, h->stream>>>(sums.as<u64>(), npc);
  k_pit_scatter64<<<(u32)npc, 256, 0, h->stream>>>(h->ncode, n, sums.as<u64>(), q.as<u64>());
  KCHK();
  pfd_seg_end(h, 3);
  pfd_seg_begin(h, "idxs_seq_exact_order");
  u64 small = 16384;  // (a level this size: ~16 trips of one workgroup against three launches)
  if (const char *e = pfd_knob("PFD_TEST_ORDER64_SMALL")) small = (u64)atoll(e);
  i64 launches = 0;
  for (u32 l = 0; l + 1 < nlev; ++l) {
    const u64 begin = off[l], end = off[l + 1], m = end - begin;
    if (m <= small) {
      k_wseq_small<<<1, 1024, 0, h->stream>>>(h->ncode, g, q.as<u64>(), begin, end);
      ++launches;
    } else {
      const u64 nchunk = (m + W_CHUNK - 1) / W_CHUNK;
      k_wseq_count<<<(u32)nchunk, 256, 0, h->stream>>>(h->ncode, g, q.as<u64>(), begin, end, sums.as<u64>());
      k_scan_u64_1block<<<1, 1024, 0, h->stream>>>(sums.as<u64>(), nchunk);
      k_wseq_scatter<<<(u32)nchunk, 256, 0, h->stream>>>(h->ncode, g, q.as<u64>(), begin, end, sums.as<u64>());
      launches += 3;
    }
  }
  KCHK();
  pfd_seg_end(h, launches);
  *nseq_out = nseq;
  return PFD_OK;
}

// core.idxs_seq with 64-bit cell indices: h->n_seq entries (n_valid on an acyclic raster)
int pfd_idxs_seq_wide(pfd_raster *h, int idx_dtype, void *out, int memspace) {
  if (idx_dtype != PFD_I64) {
    pfd_set_error("idxs_seq of a raster of %lld cells needs the int64 index dtype (PFD_I64)", (long long)h->n);
    return PFD_EINVAL;
  }
  pfd_seg_clear(h);
  DevBuf q;
  u64 nseq = 0;
  PFDCHK(pfd_wide_seq_dev(h, q, &nseq));
  if (nseq)
    HIPCHK(hipMemcpyAsync(out, q.p, (size_t)nseq * sizeof(u64), memspace == PFD_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}
