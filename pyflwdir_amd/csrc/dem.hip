// dem.hip — dem.fill_depressions (reference pyflwdir/dem.py:17-143): depression filling + local D8 flow
// directions by priority flood (Wang & Liu 2006), the step BEFORE the hot path (DEM -> D8).
//
// HOST code.  The D8 it emits records which neighbour popped first from a heap ordered by
// (float32 elevation, boundary flag, row, column): the pop sequence is a total order over distinct keys, so
// any heap reproduces it, but it is inherently sequential in flats — there is no GPU form with this exact
// tie-breaking.  It runs once per DEM, next to file I/O; the product keeps it native and single-threaded
// like the reference's numba loop.
//
// Restated faithfully, including the arithmetic types of the interpreted reference: the heap key is
// float32(z); dz = float32(z0) - elevtn[r, c] is evaluated in float32 for float32 rasters and in float64
// for float64 / int32 rasters; delv has the raster's dtype (an int32 raster truncates the fill depth);
// the popped cell visits its own position too (structure element includes the centre: an edge cell pops,
// finds itself not "done", and is finalised as a pit with code 0).
#include <math.h>

#include <queue>
#include <vector>

#include "common.h"

namespace {
struct QItem {
  float z;
  uint8_t b;
  uint32_t r, c;
};
struct QGreater {  // std::priority_queue is a max-heap: "greater" gives the smallest tuple first
  bool operator()(const QItem &x, const QItem &y) const {
    if (x.z != y.z) return x.z > y.z;
    if (x.b != y.b) return x.b > y.b;
    if (x.r != y.r) return x.r > y.r;
    return x.c > y.c;
  }
};
template <class T> struct DzType { typedef double type; };
template <> struct DzType<float> { typedef float type; };

// core_d8._us (core_d8.py:16): code of the neighbour at offset (dr, dc) that drains into the centre
static const uint8_t US[3][3] = {{2, 4, 8}, {1, 0, 16}, {128, 64, 32}};

template <class T>
int fill_depressions_t(const T *elevtn, i64 nrow, i64 ncol, double nodata, int nodata_is_nan, double max_depth,
                       int outlets_min, int has_elv_max, double elv_max, const i64 *idxs_pit, i64 npit, int connectivity,
                       T *elev_out, u8 *d8) {
  typedef typename DzType<T>::type P;
  const i64 n = nrow * ncol;
  std::vector<T> delv((size_t)n, (T)0);
  std::vector<unsigned char> done((size_t)n), queued((size_t)n, 0);
  for (i64 i = 0; i < n; ++i) {
    const bool nd = nodata_is_nan ? (elevtn[i] != elevtn[i]) : ((double)elevtn[i] == nodata);
    done[i] = nd;
    d8[i] = nd ? 247 : 0;
  }
  bool st[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) st[a][b] = true;
  if (connectivity == 4) st[0][0] = st[2][2] = st[0][2] = st[2][0] = false;
  // initial queue
  if (!idxs_pit) {  // gis_utils.get_edge (gis_utils.py:118-145): valid cells on the raster border or next to an invalid one
    for (i64 r = 0; r < nrow; ++r)
      for (i64 c = 0; c < ncol; ++c) {
        const i64 i = r * ncol + c;
        if (done[i]) continue;
        bool edge = r == 0 || r == nrow - 1 || c == 0 || c == ncol - 1;
        if (!edge)
          for (int a = 0; a < 3 && !edge; ++a)
            for (int b = 0; b < 3; ++b)
              if (st[a][b] && done[(r + a - 1) * ncol + (c + b - 1)]) {
                edge = true;
                break;
              }
        if (edge && has_elv_max && !((double)elevtn[i] <= elv_max)) edge = false;
        queued[i] = edge;
      }
    if (has_elv_max) {
      bool any = false;
      for (i64 i = 0; i < n && !any; ++i) any = queued[i];
      if (!any) {
        pfd_set_error("No initial outlet cells found.");
        return PFD_EINVAL;
      }
    }
  } else {
    for (i64 k = 0; k < npit; ++k) {
      i64 i = idxs_pit[k];
      if (i < 0) i += n;  // (numpy's negative indexing of .flat)
      if (i < 0 || i >= n) {
        pfd_set_error("fill_depressions: outlet index %lld outside the raster", (long long)idxs_pit[k]);
        return PFD_EINVAL;
      }
      queued[i] = 1;
    }
  }
  std::priority_queue<QItem, std::vector<QItem>, QGreater> q;
  for (i64 i = 0; i < n; ++i)
    if (queued[i]) q.push(QItem{(float)elevtn[i], 1, (uint32_t)(i / ncol), (uint32_t)(i % ncol)});
  if (outlets_min && !q.empty()) {  // restrict the queue to the global edge minimum (single outlet)
    const QItem top = q.top();
    while (!q.empty()) q.pop();
    q.push(top);
    std::fill(queued.begin(), queued.end(), 0);
    queued[(i64)top.r * ncol + top.c] = 1;
  }
  while (!q.empty()) {
    const QItem it = q.top();
    q.pop();
    const float z0 = it.z;
    const i64 r0 = it.r, c0 = it.c;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        if (!st[a][b]) continue;
        const int dr = a - 1, dc = b - 1;
        const i64 r = r0 + dr, c = c0 + dc;
        if (r < 0 || r == nrow || c < 0 || c == ncol || done[r * ncol + c]) continue;
        const i64 i = r * ncol + c;
        P z1 = (P)elevtn[i];
        const P dz = (P)z0 - z1;  // local depression if dz > 0
        if (max_depth >= 0) {     // with a positive max_depth: do not fill when dz >= max_depth
          if ((double)dz >= max_depth) {
            q.push(QItem{(float)z1, 0, (uint32_t)r, (uint32_t)c});
            queued[i] = 1;
            for (int a2 = 0; a2 < 3; ++a2)
              for (int b2 = 0; b2 < 3; ++b2) {  // (re)visit the neighbours; numpy index semantics: -1 wraps, past the end raises
                if (!st[a2][b2]) continue;
                i64 rr = r + a2 - 1, cc = c + b2 - 1;
                if (rr >= nrow || cc >= ncol) {
                  pfd_set_error("index %lld is out of bounds (fill_depressions with max_depth next to the raster's last "
                                "row / column, as in the reference)", (long long)(rr >= nrow ? rr : cc));
                  return PFD_EINVAL;
                }
                if (rr < 0) rr += nrow;
                if (cc < 0) cc += ncol;
                done[rr * ncol + cc] = 0;
              }
            continue;
          } else if (delv[i] > (T)0) {  // reset the cell if previously filled & revisited
            queued[i] = 0;
            delv[i] = (T)0;
          }
        }
        if (dz > (P)0) {
          delv[i] = (T)dz;
          z1 += dz;
        }
        if (!queued[i]) {
          q.push(QItem{(float)z1, 0, (uint32_t)r, (uint32_t)c});
          queued[i] = 1;
        }
        done[i] = 1;
        d8[i] = US[dr + 1][dc + 1];
      }
  }
  for (i64 i = 0; i < n; ++i) elev_out[i] = (T)(elevtn[i] + delv[i]);
  return PFD_OK;
}
}  // namespace

extern "C" int pfd_fill_depressions(int dtype, const void *elevtn, int64_t nrow, int64_t ncol, double nodata,
                                    double max_depth, int outlets_min, int has_elv_max, double elv_max,
                                    const int64_t *idxs_pit, int64_t npit, int connectivity, void *elev_out,
                                    uint8_t *d8_out) {
  if (!elevtn || !elev_out || !d8_out || nrow <= 0 || ncol <= 0 || nrow > 0xFFFFFFFFll || ncol > 0xFFFFFFFFll ||
      (idxs_pit == nullptr && npit > 0)) {
    pfd_set_error("pfd_fill_depressions: bad arguments");
    return PFD_EINVAL;
  }
  if (connectivity != 4 && connectivity != 8) {
    pfd_set_error("\"connectivity\" should either be 4 or 8");
    return PFD_EINVAL;
  }
  const int nan_nd = nodata != nodata;
  const i64 *pits = (idxs_pit || npit == 0) && idxs_pit ? idxs_pit : nullptr;
  switch (dtype) {
    case PFD_F32:
      return fill_depressions_t<float>((const float *)elevtn, nrow, ncol, nodata, nan_nd, max_depth, outlets_min,
                                       has_elv_max, elv_max, pits, npit, connectivity, (float *)elev_out, d8_out);
    case PFD_F64:
      return fill_depressions_t<double>((const double *)elevtn, nrow, ncol, nodata, nan_nd, max_depth, outlets_min,
                                        has_elv_max, elv_max, pits, npit, connectivity, (double *)elev_out, d8_out);
    case PFD_I32:
      return fill_depressions_t<i32>((const i32 *)elevtn, nrow, ncol, nodata, 0, max_depth, outlets_min, has_elv_max,
                                     elv_max, pits, npit, connectivity, (i32 *)elev_out, d8_out);
    default:
      pfd_set_error("pfd_fill_depressions: unsupported elevation dtype code %d (float32, float64, int32)", dtype);
      return PFD_EUNSUPPORTED;
  }
}
