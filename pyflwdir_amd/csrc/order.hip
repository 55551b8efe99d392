// order.hip — decode / normalise a D8 raster on the GPU and build the level structure
// (cells grouped by rank) that replaces the reference's serial ordering.
//
// Reference functions replaced (all single-threaded numba loops):
//   core_d8.from_array        pyflwdir/core_d8.py:42-67   -> k_normalise (+ pit compaction)
//   core.upstream_count       pyflwdir/core.py:50-61      -> k_upstream_count
//   core.upstream_matrix      pyflwdir/core.py:67-84      -> never materialised (decoded on the fly)
//   core.idxs_seq             pyflwdir/core.py:87-117     -> k_bfs_level (level sets) and
//                                                            k_oseq_* (exact BFS order on request)
//   core.rank                 pyflwdir/core.py:17-47      -> k_rank_from_levels
#include <stdlib.h>

#include <algorithm>

#include "common.h"

int pfd_export_u32(pfd_raster *h, const u32 *src, i64 m, int idx_dtype, void *out, int memspace);

// ctrl block slots (u64 each)
enum { C_NVALID = 0, C_NPITS = 1, C_BAD = 2, C_TAIL = 3, C_DONE = 4, C_AUX = 5 };

// ---------------------------------------------------------------------------------------------
// normalise: decode + pit rule + validation + counts in one streaming pass.  A thread owns 4
// consecutive cells of a row and works from three unconditional 8-byte window loads (rows r-1,
// r, r+1, columns c-1 .. c+6; clamped addresses, validity applied afterwards) so that no load
// depends on another; the 4 normalised codes leave as one dword store.
// Flags values outside the D8 alphabet (core_d8._all, pyflwdir/core_d8.py:19).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 load_window8(const u8 *__restrict__ p, size_t off, size_t n) {
  u64 w = 0;
  if (off + 8 <= n) {
    __builtin_memcpy(&w, p + off, 8);
  } else {  // last bytes of the raster: stay inside the caller's buffer
    for (int b = 0; b < 8; ++b)
      if (off + b < n) w |= (u64)p[off + b] << (8 * b);
  }
  return w;
}

__global__ void __launch_bounds__(256) k_normalise(const u8 *__restrict__ d8, Geo g, u8 *__restrict__ ncode,
                                                   u64 *__restrict__ ctrl, u32 row_first, u32 row_last) {
  // one block = 256 columns (64 lanes x 4 cells) x 16 rows (4 waves x 4 passes)
  const u32 c0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const size_t n = (size_t)g.nrow * g.ncol;
  u32 valid = 0, pit = 0, bad = 0;
  // (grid.y is capped at 65535: taller rasters loop over bands of 16 rows)
  for (u32 band = blockIdx.y; band * 16u < g.nrow; band += gridDim.y)
  for (u32 pass = 0; pass < 4; ++pass) {
    const u32 r = band * 16 + pass * 4 + (threadIdx.x >> 6);
    if (r >= g.nrow || c0 >= g.ncol) continue;
    // windows: byte k of win[j] = column c0 - 1 + k of row r - 1 + j
    const u32 cstart = c0 ? c0 - 1 : 0;
    u64 win[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const u32 rr = (u32)min(max((int)r - 1 + j, 0), (int)g.nrow - 1);
      u64 w = load_window8(d8, (size_t)rr * g.ncol + cstart, n);
      if (!c0) w <<= 8;  // column -1 does not exist: the window started at column 0
      win[j] = w;
    }
    u32 out4 = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u32 c = c0 + b;
      u32 out = D8_MV;
      if (c < g.ncol) {
        const u32 code = (u32)(win[1] >> (8 * (b + 1))) & 0xFFu;
        out = code;
        if (code == D8_MV) {
          out = D8_MV;
        } else if (r < row_first || r > row_last) {
          // halo row of a row block: the cell belongs to the neighbouring block; here it is a
          // weightless sink that collects the flow leaving this block
          out = ((code & (code - 1)) == 0u || code == 255u) ? D8_HALO : D8_MV;
          if (out == D8_MV) ++bad;
        } else if (code == 0u || code == 255u) {
          out = 0;
          ++valid;
          ++pit;
        } else if ((code & (code - 1)) == 0u) {  // one of the eight direction codes
          ++valid;
          const int k = d8_slot(code);
          const int dr = d8_dr(k), dc = d8_dc(k);
          const u32 rr = r + (u32)dr, cc = c + (u32)dc;
          const u32 tcode = (u32)(win[1 + dr] >> (8 * (b + 1 + dc))) & 0xFFu;
          if (rr >= g.nrow || cc >= g.ncol || tcode == D8_MV) {
            out = 0;  // drains off the raster or into nodata -> pit (core_d8.py:57-63)
            ++pit;
          }
        } else {
          ++bad;
          out = D8_MV;
        }
      }
      out4 |= out << (8 * b);
    }
    u8 *dst = ncode + (size_t)r * g.ncol + c0;
    if (c0 + 3 < g.ncol) {
      __builtin_memcpy(dst, &out4, 4);  // (possibly unaligned) dword store
    } else {
      for (u32 b = 0; b < 4 && c0 + b < g.ncol; ++b) dst[b] = (u8)(out4 >> (8 * b));
    }
  }
  // block reduction -> at most 3 atomics per block, spread over 16 counter copies
  __shared__ u32 s_valid, s_pit, s_bad;
  if (threadIdx.x == 0) s_valid = s_pit = s_bad = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    valid += __shfl_down(valid, o);
    pit += __shfl_down(pit, o);
    bad += __shfl_down(bad, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (valid) atomicAdd(&s_valid, valid);
    if (pit) atomicAdd(&s_pit, pit);
    if (bad) atomicAdd(&s_bad, bad);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 copy = (blockIdx.x + blockIdx.y) & 15u;
    if (s_valid) atomicAdd((unsigned long long *)&ctrl[16 + copy], (unsigned long long)s_valid);
    if (s_pit) atomicAdd((unsigned long long *)&ctrl[32 + copy], (unsigned long long)s_pit);
    if (s_bad) atomicAdd((unsigned long long *)&ctrl[C_BAD], (unsigned long long)s_bad);
  }
}

// ---------------------------------------------------------------------------------------------
// ordered compaction of the pit cells (ascending linear index) into seq[0 .. n_pits):
// per-chunk counts -> single-block exclusive scan of the chunk counts -> ordered scatter.
// ---------------------------------------------------------------------------------------------
#define PIT_CHUNK 4096u  // cells per block (256 threads x 16)

// (halo != 0: row blocks — the halo sinks are roots of the block's level structure like its pits)
__global__ void __launch_bounds__(256) k_pit_count(const u8 *__restrict__ ncode, u32 n, u32 *__restrict__ counts,
                                                   int halo = 0) {
  const u32 base = blockIdx.x * PIT_CHUNK;
  u32 cnt = 0;
  for (u32 t = threadIdx.x; t < PIT_CHUNK; t += 256) {
    const u32 i = base + t;
    if (i < n && (ncode[i] == 0 || (halo && ncode[i] == D8_HALO))) ++cnt;
  }
  __shared__ u32 s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s;
}

// in-place exclusive scan of m u32 values by ONE block of 1024 threads
__global__ void __launch_bounds__(1024) k_scan_u32_1block(u32 *__restrict__ v, u32 m) {
  __shared__ u32 wsum[16];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (u32 base = 0; base < m; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 x = i < m ? v[i] : 0;
    u32 incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const u32 y = __shfl_up(incl, o);
      if (lane >= (u32)o) incl += y;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < wid; ++w) woff += wsum[w];
    u32 total = 0;
    for (u32 w = 0; w < 16; ++w) total += wsum[w];
    const u32 excl = carry + woff + incl - x;
    if (i < m) v[i] = excl;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_pit_scatter(const u8 *__restrict__ ncode, u32 n,
                                                     const u32 *__restrict__ offs, u32 *__restrict__ seq, int halo = 0) {
  // each thread owns 16 CONSECUTIVE cells so that the in-block order is the linear order
  const u32 base = blockIdx.x * PIT_CHUNK + threadIdx.x * 16;
  u32 mask = 0;
  for (u32 t = 0; t < 16; ++t) {
    const u32 i = base + t;
    if (i < n && (ncode[i] == 0 || (halo && ncode[i] == D8_HALO))) mask |= 1u << t;
  }
  const u32 cnt = __popc(mask);
  // block exclusive scan over the 256 thread counts
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
  u32 pos = offs[blockIdx.x] + woff + incl - cnt;
  while (mask) {
    const u32 t = __ffs((int)mask) - 1;
    mask &= mask - 1;
    seq[pos++] = base + t;
  }
}

static int compact_pits(pfd_raster *h) {
  const u32 n = h->geo.n;
  const u32 nchunk = cdiv_u32((u64)n, PIT_CHUNK);
  DevBuf counts;
  PFDCHK(counts.alloc((size_t)nchunk * sizeof(u32)));
  k_pit_count<<<nchunk, 256, 0, h->stream>>>(h->ncode, n, counts.as<u32>());
  KCHK();
  k_scan_u32_1block<<<1, 1024, 0, h->stream>>>(counts.as<u32>(), nchunk);
  KCHK();
  k_pit_scatter<<<nchunk, 256, 0, h->stream>>>(h->ncode, n, counts.as<u32>(), h->pits);
  KCHK();
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

// row blocks: the roots of the block's level structure = its pits + its halo sinks, ascending (kept in h->pits;
// the public pit list stays refused on block handles)
static int block_roots(pfd_raster *h, i64 *nroots) {
  const u32 n = h->geo.n;
  const u32 nchunk = cdiv_u32((u64)n, PIT_CHUNK);
  DevBuf counts;
  PFDCHK(counts.alloc(((size_t)nchunk + 1) * sizeof(u32)));
  HIPCHK(hipMemsetAsync(counts.p, 0, ((size_t)nchunk + 1) * sizeof(u32), h->stream));
  k_pit_count<<<nchunk, 256, 0, h->stream>>>(h->ncode, n, counts.as<u32>(), 1);
  KCHK();
  k_scan_u32_1block<<<1, 1024, 0, h->stream>>>(counts.as<u32>(), nchunk + 1);  // (last entry = the total)
  KCHK();
  u32 total = 0;
  HIPCHK(hipMemcpyAsync(&total, counts.as<u32>() + nchunk, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->pits) pfd_dfree(h->pits);
  h->pits = nullptr;
  PFDCHK(pfd_dmalloc((void **)&h->pits, (size_t)std::max<u32>(total, 1u) * sizeof(u32)));
  k_pit_scatter<<<nchunk, 256, 0, h->stream>>>(h->ncode, n, counts.as<u32>(), h->pits, 1);
  KCHK();
  HIPCHK(hipStreamSynchronize(h->stream));
  *nroots = (i64)total;
  return PFD_OK;
}

static int alloc_pits(pfd_raster *h) {
  if (h->pits) {
    pfd_dfree(h->pits);
    h->pits = nullptr;
  }
  return pfd_dmalloc((void **)&h->pits, (size_t)std::max<i64>(h->n_pits, 1) * sizeof(u32));
}

// the halo rows of a row block as given: the flow ENTERING the block from them is invisible in the normalised codes
// (halo cells are sinks there); the seeded up-sweeps of a row block need it (sweeps.hip, pfd_accuflux_block)
static int save_halo_rows(pfd_raster *h, const u8 *d8_dev) {
  if (!(h->halo_top || h->halo_bot) || !d8_dev) return PFD_OK;
  const size_t ncol = (size_t)h->ncol;
  if (!h->halo_raw) PFDCHK(pfd_dmalloc((void **)&h->halo_raw, 2 * ncol));
  HIPCHK(hipMemsetAsync(h->halo_raw, D8_MV, 2 * ncol, h->stream));
  if (h->halo_top)
    HIPCHK(hipMemcpyAsync(h->halo_raw, d8_dev + (size_t)(h->halo_top - 1) * ncol, ncol, hipMemcpyDeviceToDevice, h->stream));
  if (h->halo_bot)
    HIPCHK(hipMemcpyAsync(h->halo_raw + ncol, d8_dev + (size_t)(h->halo_top + h->own_rows) * ncol, ncol,
                          hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PFD_OK;
}

int pfd_normalise_and_count(pfd_raster *h, const u8 *d8_dev) {
  HIPCHK(hipMemsetAsync(h->ctrl, 0, 64 * sizeof(u64), h->stream));
  const u32 gy = std::min<u32>(cdiv_u32((u64)h->nrow, 16), 65535u);
  dim3 grid(cdiv_u32((u64)h->ncol, 256), gy);
  k_normalise<<<grid, 256, 0, h->stream>>>(d8_dev, h->geo, h->ncode, h->ctrl, (u32)h->halo_top,
                                           (u32)(h->halo_top + h->own_rows - 1));
  KCHK();
  PFDCHK(save_halo_rows(h, d8_dev));
  u64 c[48];
  HIPCHK(hipMemcpyAsync(c, h->ctrl, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return pfd_adopt_counts(h, c);
}

// a deferred handle is normalised by the first entry point that needs the codes
int pfd_ensure_normalised(pfd_raster *h) {
  if (h->normalised) return PFD_OK;
  return pfd_normalise_and_count(h, h->raw);
}

int pfd_adopt_counts(pfd_raster *h, const u64 *c) {
  if (c[C_BAD]) {
    pfd_set_error("raster holds %llu value(s) that are not D8 codes (allowed: 1,2,4,8,16,32,64,128,0,255,247)",
                  (unsigned long long)c[C_BAD]);
    return PFD_EBADCODE;
  }
  h->n_valid = h->n_pits = 0;
  for (int k = 0; k < 16; ++k) {
    h->n_valid += (i64)c[16 + k];
    h->n_pits += (i64)c[32 + k];
  }
  if (h->n_pits == 0 && !(h->halo_top || h->halo_bot)) {
    pfd_set_error("Invalid FlwdirRaster: no pits found");
    return PFD_ENOPITS;
  }
  h->ordered = false;
  h->acyclic = 0;
  pfd_free_xplan(h);
  h->aux_ready = false;
  h->pits_ready = false;  // the ascending pit list is compacted on first use
  h->normalised = true;
  if (!h->halo_raw) PFDCHK(save_halo_rows(h, h->raw));  // (a deferred handle decoded by the tile pass)
  h->raw = nullptr;
  if (h->raw_owned) {
    pfd_dfree(h->raw_owned);
    h->raw_owned = nullptr;
  }
  return PFD_OK;
}

int pfd_require_unblocked(pfd_raster *h, const char *what) {
  if (h->halo_top || h->halo_bot) {
    pfd_set_error("%s is not available on a row-block handle (row blocks run pfd_upstream_area_cell_blocks / _begin / "
                  "_finish / _dist, pfd_basins_begin / _finish and the pfd_*_block sweeps)", what);
    return PFD_EUNSUPPORTED;
  }
  return PFD_OK;
}
int pfd_require_whole(pfd_raster *h, const char *what) {
  PFDCHK(pfd_require_unblocked(h, what));
  if (h->n > 4294967294ll) {
    pfd_set_error("%s needs 32-bit cell indices and is not available for a raster of %lld cells (only "
                  "upstream_area(unit=\"cell\") runs on rasters this large)", what, (long long)h->n);
    return PFD_EUNSUPPORTED;
  }
  return PFD_OK;
}

int pfd_ensure_pits(pfd_raster *h) {
  if (h->pits_ready) return PFD_OK;
  PFDCHK(pfd_require_whole(h, "the pit list"));
  PFDCHK(alloc_pits(h));
  h->bytes_held += (size_t)h->n_pits * sizeof(u32);
  PFDCHK(compact_pits(h));
  h->pits_ready = true;
  return PFD_OK;
}

extern "C" int pfd_idxs_pit(pfd_raster *h, int idx_dtype, void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  PFDCHK(pfd_ensure_pits(h));
  return pfd_export_u32(h, h->pits, h->n_pits, idx_dtype, out, memspace);
}

// ---------------------------------------------------------------------------------------------
// add_pits (reference pyflwdir/flwdir.py:261-279)
// ---------------------------------------------------------------------------------------------
// (64-bit cell indices: also for rasters beyond 2^32 - 2 cells, which only the tiled engine sweeps)
__global__ void k_add_pits(u8 *__restrict__ ncode, const i64 *__restrict__ idxs, u32 k, i64 n, u64 *ctrl) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const i64 i = idxs[t];
  if (i < 0 || i >= n || ncode[i] == D8_MV) {
    atomicAdd((unsigned long long *)&ctrl[C_BAD], 1ull);
    return;
  }
  ncode[i] = 0;
}
__global__ void __launch_bounds__(256) k_count_pits(const u8 *__restrict__ ncode, i64 n, u64 *ctrl) {
  u32 cnt = 0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) cnt += ncode[i] == 0;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd((unsigned long long *)&ctrl[C_NPITS], (unsigned long long)cnt);
}

extern "C" int pfd_add_pits(pfd_raster *h, const int64_t *idxs, int64_t k) {
  PFDCHK(pfd_check_handle(h));
  if (k < 0 || (k > 0 && !idxs)) {
    pfd_set_error("pfd_add_pits: bad arguments");
    return PFD_EINVAL;
  }
  if (k == 0) return PFD_OK;
  if (h->halo_top || h->halo_bot) PFDCHK(pfd_require_whole(h, "add_pits"));  // (row blocks: edit the whole raster)
  if (k > 0x7FFFFFFFll) {
    pfd_set_error("pfd_add_pits: too many indices (%lld)", (long long)k);
    return PFD_EINVAL;
  }
  InArg in;
  HIPCHK(hipMemsetAsync(h->ctrl, 0, 64 * sizeof(u64), h->stream));
  if (h->gen) {
    PFDCHK(pfd_gen_add_pits(h, idxs, k));
  } else {
    PFDCHK(in.bind(idxs, (size_t)k * sizeof(i64), PFD_HOST, h->stream));
    k_add_pits<<<cdiv_u32((u64)k, 256), 256, 0, h->stream>>>(h->ncode, (const i64 *)in.dev, (u32)k, h->n, h->ctrl);
    KCHK();
  }
  k_count_pits<<<1024, 256, 0, h->stream>>>(h->ncode, h->n, h->ctrl);
  KCHK();
  u64 c[3];
  HIPCHK(hipMemcpyAsync(c, h->ctrl, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->n_pits = (i64)c[C_NPITS];
  h->ordered = false;
  h->acyclic = 0;
  pfd_free_xplan(h);
  h->aux_ready = false;
  h->n_seq = h->n_levels = -1;
  h->lvl_off.clear();
  h->pits_ready = false;
  if (c[C_BAD]) {
    pfd_set_error("pfd_add_pits: %llu index(es) outside the raster or on nodata cells were ignored",
                  (unsigned long long)c[C_BAD]);
    return PFD_EINVAL;
  }
  return PFD_OK;
}

// ---------------------------------------------------------------------------------------------
// upstream_count (reference pyflwdir/core.py:50-61): pull form, one thread per cell.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_upstream_count(const u8 *__restrict__ ncode, Geo g,
                                                        const u8 *__restrict__ mask, int8_t *__restrict__ out) {
  const u32 c = blockIdx.x * 64 + (threadIdx.x & 63);
  const u32 r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (r >= g.nrow || c >= g.ncol) return;
  const u32 i = r * g.ncol + c;
  if (ncode[i] == D8_MV) {
    out[i] = -9;
    return;
  }
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    u32 nb;
    if (d8_child(ncode, g, i, r, c, k, &nb) && (mask == nullptr || mask[nb])) ++cnt;
  }
  out[i] = (int8_t)cnt;
}

extern "C" int pfd_upstream_count(pfd_raster *h, const uint8_t *mask, int8_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (h->gen) return pfd_gen_upstream_count(h, mask, out, memspace);
  PFDCHK(pfd_require_whole(h, "upstream_count"));
  if (!out) {
    pfd_set_error("pfd_upstream_count: NULL out");
    return PFD_EINVAL;
  }
  InArg m;
  PFDCHK(m.bind(mask, (size_t)h->n, memspace, h->stream));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n, memspace));
  dim3 grid(cdiv_u32((u64)h->ncol, 64), cdiv_u32((u64)h->nrow, 4));
  k_upstream_count<<<grid, 256, 0, h->stream>>>(h->ncode, h->geo, (const u8 *)m.dev, (int8_t *)o.dev);
  KCHK();
  return o.finish(h->stream);
}

// ---------------------------------------------------------------------------------------------
// level build: top-down breadth-first expansion from the pits.  Level l+1 = all cells draining
// into a cell of level l.  One launch per level; each wave aggregates its children counts and
// reserves output space with ONE atomicAdd (order inside a level is irrelevant to the pull
// sweeps; the exact reference order is produced separately by pfd_idxs_seq).  The last block
// to finish publishes the end offset of the new level, so the next launch needs no host
// round trip.
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_bfs_level(const u8 *__restrict__ ncode, Geo g, u32 *__restrict__ seq,
                                                   u64 *__restrict__ ctrl, i64 *__restrict__ lvl_off, int lvl) {
  const u32 begin = (u32)lvl_off[lvl], end = (u32)lvl_off[lvl + 1];
  const u32 lane = threadIdx.x & 63;
  const u32 nthreads = gridDim.x * blockDim.x;
  // wave-uniform trip count
  for (u32 j0 = begin + (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); j0 < end; j0 += nthreads) {
    const u32 j = j0 + lane;
    u32 kids[8];
    u32 cnt = 0;
    if (j < end) {
      const u32 x = seq[j];
      const u32 r = geo_row(g, x), c = x - r * g.ncol;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        u32 nb;
        if (d8_child(ncode, g, x, r, c, PFD_SLOT_ASC[q], &nb)) kids[cnt++] = nb;
      }
    }
    u32 incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const u32 y = __shfl_up(incl, o);
      if (lane >= (u32)o) incl += y;
    }
    const u32 total = __shfl(incl, 63);
    u32 base = 0;
    if (total) {
      if (lane == 0) base = (u32)atomicAdd((unsigned long long *)&ctrl[C_TAIL], (unsigned long long)total);
      base = __shfl(base, 0);
      u32 pos = base + incl - cnt;
      for (u32 q = 0; q < cnt; ++q) seq[pos + q] = kids[q];
    }
  }
  // last block publishes the end of level lvl+1
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long prev = atomicAdd((unsigned long long *)&ctrl[C_DONE], 1ull);
    if (prev == (unsigned long long)gridDim.x - 1) {
      const unsigned long long tail = atomicAdd((unsigned long long *)&ctrl[C_TAIL], 0ull);
      lvl_off[lvl + 2] = (i64)tail;
      ctrl[C_DONE] = 0;
    }
  }
}

int pfd_order_cells_impl(pfd_raster *h, bool allow_block) {
  if (h->gen) return pfd_gen_order(h);
  const bool block = h->halo_top || h->halo_bot;  // row block: breadth-first from its pits and halo sinks
  // (only the *_block entry points opt in: a whole-raster operation on a row-block handle would return block-local
  //  values — halo cells as roots, nothing entering from the neighbours)
  if (!block || !allow_block) PFDCHK(pfd_require_whole(h, "the cell ordering"));
  if (h->ordered) return PFD_OK;
  if (block && h->n > 4294967294ll) {
    pfd_set_error("the cell ordering needs 32-bit cell indices: a row block of %lld cells is too large", (long long)h->n);
    return PFD_EUNSUPPORTED;
  }
  if (block) PFDCHK(pfd_ensure_normalised(h));
  i64 nroot = h->n_pits;
  if (!block && !pfd_knob("PFD_ORDER_BFS")) {
    // fast path: ranks by LDS-tiled pointer doubling + one radix sort of the cells by rank
    // (paths.hip); rasters with cycles fall through to the breadth-first build below
    int ok = 0;
    pfd_seg_begin(h, "order_cells_by_rank");
    PFDCHK(pfd_order_cells_by_rank(h, &ok));
    pfd_seg_end(h, 4);
    if (ok) return PFD_OK;
  }
  if (block) {
    PFDCHK(block_roots(h, &nroot));
  } else {
    PFDCHK(pfd_ensure_pits(h));
  }
  if (!h->seq) {  // (a buffer left by an abandoned rank-sort attempt holds n >= n_valid entries)
    const size_t cap_seq = block ? (size_t)h->n : (size_t)h->n_valid;  // (block: own cells + halo sinks)
    PFDCHK(pfd_dmalloc((void **)&h->seq, std::max<size_t>(cap_seq, 1) * sizeof(u32)));
    h->bytes_held += cap_seq * sizeof(u32);
  }
  pfd_seg_begin(h, "order_cells");
  HIPCHK(hipMemcpyAsync(h->seq, h->pits, (size_t)nroot * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
  const int BATCH = 128;
  size_t cap = (size_t)(2 * (h->nrow + h->ncol) + 4 * BATCH + 64);
  DevBuf lvl;
  PFDCHK(lvl.alloc(cap * sizeof(i64)));
  HIPCHK(hipMemsetAsync(lvl.p, 0, cap * sizeof(i64), h->stream));
  const i64 first[2] = {0, nroot};
  HIPCHK(hipMemcpyAsync(lvl.p, first, sizeof(first), hipMemcpyHostToDevice, h->stream));
  u64 ctrl0[8] = {0};
  ctrl0[C_TAIL] = (u64)nroot;
  HIPCHK(hipMemcpyAsync(h->ctrl, ctrl0, sizeof(ctrl0), hipMemcpyHostToDevice, h->stream));
  std::vector<i64> off;
  off.push_back(0);
  off.push_back(nroot);
  int lvl_next = 0;  // next level to expand
  i64 launches = 0;
  bool done = false;
  i64 recent_max = nroot;
  std::vector<i64> tmp(BATCH);
  while (!done) {
    if ((size_t)(lvl_next + BATCH + 2) > cap) {  // grow the device offsets array
      const size_t ncap = cap * 2 + BATCH;
      DevBuf bigger;
      PFDCHK(bigger.alloc(ncap * sizeof(i64)));
      HIPCHK(hipMemsetAsync(bigger.p, 0, ncap * sizeof(i64), h->stream));
      HIPCHK(hipMemcpyAsync(bigger.p, lvl.p, cap * sizeof(i64), hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      std::swap(lvl.p, bigger.p);
      cap = ncap;
    }
    // grid sized from the largest recent level (every block ends with one same-address atomic,
    // so an oversized grid costs ~12 ns per block; an undersized one just grid-strides)
    const u32 bfs_grid = (u32)std::min<i64>(2048, std::max<i64>(32, (4 * recent_max + 255) / 256));
    for (int b = 0; b < BATCH; ++b) {
      k_bfs_level<<<bfs_grid, 256, 0, h->stream>>>(h->ncode, h->geo, h->seq, h->ctrl, lvl.as<i64>(), lvl_next + b);
      ++launches;
    }
    KCHK();
    // offsets lvl_next+2 .. lvl_next+BATCH+1 were produced by this batch
    HIPCHK(hipMemcpyAsync(tmp.data(), lvl.as<i64>() + lvl_next + 2, BATCH * sizeof(i64), hipMemcpyDeviceToHost,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    recent_max = 0;
    for (int b = 0; b < BATCH; ++b) {
      if (tmp[b] == off.back()) {  // level lvl_next+b+1 is empty -> finished
        done = true;
        break;
      }
      recent_max = std::max(recent_max, tmp[b] - off.back());
      off.push_back(tmp[b]);
    }
    lvl_next += BATCH;
  }
  h->lvl_off = off;
  h->n_levels = (i64)off.size() - 1;
  h->n_seq = off.back();
  h->ordered = true;
  h->aux_ready = false;
  pfd_seg_end(h, launches);
  return PFD_OK;
}

extern "C" int pfd_order_cells(pfd_raster *h) {
  PFDCHK(pfd_check_handle(h));
  return pfd_order_cells_impl(h);
}

// ---------------------------------------------------------------------------------------------
// rank (reference pyflwdir/core.py:17-47): level id of every ordered cell, -1 for valid cells
// outside the sequence (loops), -9999 on nodata.
// ---------------------------------------------------------------------------------------------
__global__ void k_rank_init(const u8 *__restrict__ ncode, u32 n, i32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ncode[i] == D8_MV ? -9999 : -1;
}
__global__ void k_rank_from_levels(const u32 *__restrict__ seq, const i64 *__restrict__ lvl_off, u32 nlev,
                                   u32 nseq, i32 *__restrict__ out) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nseq) return;
  u32 lo = 0, hi = nlev;  // find l with lvl_off[l] <= j < lvl_off[l+1]
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if ((u32)lvl_off[mid] <= j)
      lo = mid;
    else
      hi = mid;
  }
  out[seq[j]] = (i32)lo;
}

extern "C" int pfd_rank(pfd_raster *h, int32_t *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (!out) {
    pfd_set_error("pfd_rank: NULL out");
    return PFD_EINVAL;
  }
  if (h->gen) return pfd_gen_rank(h, out, memspace);
  if (pfd_wide_cells(h)) return pfd_rank_wide(h, out, memspace);
  PFDCHK(pfd_order_cells_impl(h));
  OutArg o;
  PFDCHK(o.bind(out, (size_t)h->n * sizeof(i32), memspace));
  InArg lo;
  PFDCHK(lo.bind(h->lvl_off.data(), h->lvl_off.size() * sizeof(i64), PFD_HOST, h->stream));
  k_rank_init<<<cdiv_u32((u64)h->n, 256), 256, 0, h->stream>>>(h->ncode, h->geo.n, (i32 *)o.dev);
  KCHK();
  if (h->n_seq > 0) {
    k_rank_from_levels<<<cdiv_u32((u64)h->n_seq, 256), 256, 0, h->stream>>>(h->seq, (const i64 *)lo.dev,
                                                                           (u32)h->n_levels, (u32)h->n_seq, (i32 *)o.dev);
    KCHK();
  }
  return o.finish(h->stream);
}

// ---------------------------------------------------------------------------------------------
// exact core.idxs_seq order (reference pyflwdir/core.py:103-116).  The reference's queue is
// "pits, then for every dequeued cell its upstream cells ascending".  Position of the k-th
// upstream cell of the cell dequeued at position j is  n_pits + S(j) + k  with S the exclusive
// prefix sum of the in-degrees in dequeue order.  The sum is local to a level, so level l+1 is
// laid out from level l with: (A) in-degree per position + per-chunk sums, (B) scan of the chunk
// sums, (C) ordered scatter.  Small levels run A-C fused in a single block.
// ---------------------------------------------------------------------------------------------
#define OSEQ_CHUNK 1024u  // positions per block (256 threads x 4 consecutive)

__device__ __forceinline__ u32 count_children(const u8 *__restrict__ ncode, const Geo &g, u32 x) {
  const u32 r = geo_row(g, x), c = x - r * g.ncol;
  u32 cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    u32 nb;
    cnt += d8_child(ncode, g, x, r, c, k, &nb) ? 1u : 0u;
  }
  return cnt;
}
__device__ __forceinline__ void write_children_asc(const u8 *__restrict__ ncode, const Geo &g, u32 x, u32 *dst) {
  const u32 r = geo_row(g, x), c = x - r * g.ncol;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    u32 nb;
    if (d8_child(ncode, g, x, r, c, PFD_SLOT_ASC[q], &nb)) *dst++ = nb;
  }
}

__global__ void __launch_bounds__(256) k_oseq_count(const u8 *__restrict__ ncode, Geo g, const u32 *__restrict__ oseq,
                                                    u32 begin, u32 end, u32 *__restrict__ chunk_sums) {
  const u32 j0 = begin + blockIdx.x * OSEQ_CHUNK + threadIdx.x * 4;
  u32 cnt = 0;
  for (u32 t = 0; t < 4; ++t)
    if (j0 + t < end) cnt += count_children(ncode, g, oseq[j0 + t]);
  __shared__ u32 s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s, cnt);
  __syncthreads();
  if (threadIdx.x == 0) chunk_sums[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_oseq_scatter(const u8 *__restrict__ ncode, Geo g, u32 *__restrict__ oseq,
                                                      u32 begin, u32 end, const u32 *__restrict__ chunk_offs) {
  const u32 j0 = begin + blockIdx.x * OSEQ_CHUNK + threadIdx.x * 4;
  u32 c4[4], cnt = 0;
  for (u32 t = 0; t < 4; ++t) {
    c4[t] = (j0 + t < end) ? count_children(ncode, g, oseq[j0 + t]) : 0;
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
  u32 pos = end + chunk_offs[blockIdx.x] + woff + incl - cnt;
  for (u32 t = 0; t < 4; ++t) {
    if (c4[t]) write_children_asc(ncode, g, oseq[j0 + t], oseq + pos);
    pos += c4[t];
  }
}

// fused single-block version for levels of at most `OSEQ_SMALL` cells, looping over as many
// consecutive small levels as possible (lvl_sizes known on the host).
__global__ void __launch_bounds__(1024) k_oseq_small(const u8 *__restrict__ ncode, Geo g, u32 *__restrict__ oseq,
                                                     u32 begin, u32 end) {
  __shared__ u32 wsum[16];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (u32 base = begin; base < end; base += 1024) {
    const u32 j = base + threadIdx.x;
    const u32 x = j < end ? oseq[j] : 0;
    const u32 cnt = j < end ? count_children(ncode, g, x) : 0;
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
    if (cnt) write_children_asc(ncode, g, x, oseq + end + carry + woff + incl - cnt);
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
}

// the exact core.idxs_seq order in device memory (oseq: n_seq entries)
int pfd_exact_seq_dev(pfd_raster *h, DevBuf &oseq) {
  PFDCHK(pfd_order_cells_impl(h));
  pfd_seg_begin(h, "idxs_seq_exact_order");
  PFDCHK(oseq.alloc((size_t)std::max<i64>(h->n_seq, 1) * sizeof(u32)));
  HIPCHK(hipMemcpyAsync(oseq.p, h->seq, (size_t)h->n_pits * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
  i64 maxlev = 0;
  for (i64 l = 0; l < h->n_levels; ++l) maxlev = std::max(maxlev, h->lvl_off[l + 1] - h->lvl_off[l]);
  const u32 maxchunks = cdiv_u32((u64)maxlev, OSEQ_CHUNK);
  DevBuf sums;
  PFDCHK(sums.alloc((size_t)std::max<u32>(maxchunks, 1) * sizeof(u32)));
  i64 launches = 0;
  for (i64 l = 0; l + 1 < h->n_levels; ++l) {
    const u32 begin = (u32)h->lvl_off[l], end = (u32)h->lvl_off[l + 1];
    const u32 m = end - begin;
    if (m <= 8192) {
      k_oseq_small<<<1, 1024, 0, h->stream>>>(h->ncode, h->geo, oseq.as<u32>(), begin, end);
      ++launches;
    } else {
      const u32 nchunk = cdiv_u32((u64)m, OSEQ_CHUNK);
      k_oseq_count<<<nchunk, 256, 0, h->stream>>>(h->ncode, h->geo, oseq.as<u32>(), begin, end, sums.as<u32>());
      k_scan_u32_1block<<<1, 1024, 0, h->stream>>>(sums.as<u32>(), nchunk);
      k_oseq_scatter<<<nchunk, 256, 0, h->stream>>>(h->ncode, h->geo, oseq.as<u32>(), begin, end, sums.as<u32>());
      launches += 3;
    }
  }
  KCHK();
  pfd_seg_end(h, launches);
  HIPCHK(hipStreamSynchronize(h->stream));  // (`sums` is released on return)
  return PFD_OK;
}

extern "C" int pfd_idxs_seq(pfd_raster *h, int idx_dtype, void *out, int memspace) {
  PFDCHK(pfd_check_handle(h));
  if (h->gen) return pfd_gen_idxs_seq(h, idx_dtype, out, memspace);
  if (pfd_wide_cells(h)) return pfd_idxs_seq_wide(h, idx_dtype, out, memspace);
  DevBuf oseq;
  PFDCHK(pfd_exact_seq_dev(h, oseq));
  return pfd_export_u32(h, oseq.as<u32>(), h->n_seq, idx_dtype, out, memspace);
}
