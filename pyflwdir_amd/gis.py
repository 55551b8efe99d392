"""Host-side raster geometry helpers of the hot path (numpy, float64 like the reference).

Only what ``FlwdirRaster.upstream_area(unit != "cell")`` and ``basins(xy=...)`` need is
restated here; the function names and semantics follow the reference's
``pyflwdir/gis_utils.py`` (cell area: :379-412, coordinates: :191-338, :342-359) so that
callers can switch without edits.  These are O(nrow + ncol) or O(k) host computations; the
O(n) weight raster for unit != "cell" is a broadcast of one value per row.
"""
from __future__ import annotations

import numpy as np

from ._affine import get_affine

Affine = get_affine()

_R = 6371e3  # earth radius [m], reference gis_utils.py:10
AREA_FACTORS = {"m2": 1.0, "ha": 1e4, "km2": 1e6, "cell": 1}  # reference gis_utils.py:11
IDENTITY = Affine(1.0, 0.0, 0.0, 0.0, -1.0, 0.0)  # N->S orientation, reference gis_utils.py:13

__all__ = ["AREA_FACTORS", "IDENTITY", "affine_to_coords", "cellarea", "area_grid", "area_rows",
           "reggrid_area", "reggrid_dx", "reggrid_dy", "xy", "rowcol", "idxs_to_coords", "coords_to_idxs"]


def _unit_factor(unit):
    unit = str(unit).lower()
    if unit not in AREA_FACTORS:
        fstr = '", "'.join(AREA_FACTORS.keys())
        raise ValueError(f'Unknown unit: {unit}, select from "{fstr}".')
    return unit, AREA_FACTORS[unit]


def affine_to_coords(affine, shape):
    """Cell-centre coordinate axes (x per column, y per row) — what reference gis_utils.py:342-359 returns: the
    transform applied to the pixel centres of row 0 (for x) and of column 0 (for y)."""
    nrow, ncol = shape
    centre = 0.5
    xs = (affine * (np.arange(ncol) + centre, np.full(ncol, centre)))[0]
    ys = (affine * (np.full(nrow, centre), np.arange(nrow) + centre))[1]
    return xs, ys


def cellarea(lat, xres, yres):
    """Area [m2] of the lat/lon cell(s) centred at ``lat`` (a spherical zone segment: R^2 * dlon * (sin(north
    edge) - sin(south edge)), evaluated in the operand order of reference gis_utils.py:405-412 so that float64
    results agree bit for bit)."""
    dlat_half = np.abs(yres) / 2.0
    south, north = np.radians(lat - dlat_half), np.radians(lat + dlat_half)
    dlon = np.radians(np.abs(xres))
    return _R**2 * dlon * (np.sin(north) - np.sin(south))


def area_grid(transform, shape, latlon=False, unit="m2"):
    """Grid of cell areas with the dtypes of reference gis_utils.py:379-402 (int32 ones for "cell", float64 for
    lat/lon grids, float32 for projected ones).  The area of a regular grid's cell depends on its row only, so the
    grid is ``area_rows`` repeated along the columns — the product never builds it (``pfd_accuflux_rows``)."""
    rows = area_rows(transform, shape, latlon, unit)
    return np.repeat(rows[:, None], shape[1], axis=1)


def area_rows(transform, shape, latlon=False, unit="m2"):
    """Column 0 of ``area_grid`` without building the grid: the cell area of a regular grid depends on
    the row only (same expressions, element for element, as the reference's ``area_grid`` / ``reggrid_area``, gis_utils.py:379-402)."""
    unit, factor = _unit_factor(unit)
    if unit == "cell":
        return np.ones(shape[0], dtype=np.int32)
    if not latlon:  # projected: one float32 value
        return np.full(shape[0], abs(transform[0] * transform[4]) / factor, dtype=np.float32)
    lon, lat = affine_to_coords(transform, shape)
    # (the reference multiplies the float64 column by a float32 matrix of ones before dividing: kept, it is exact)
    xres, yres = np.abs(np.mean(np.diff(lon))), np.abs(np.mean(np.diff(lat)))
    return cellarea(lat, xres, yres) * np.ones(lat.size, dtype=np.float32) / factor


def _mean_step(axis):
    return np.abs(np.mean(np.diff(axis)))


def _per_row_grid(column, ncol, ones_dtype):
    """A per-row quantity as a full grid (the reference's public reggrid_* helpers return grids: column times a matrix
    of ones of the given dtype — kept as a product so that the result dtype follows numpy's promotion as it does there)."""
    column = np.asarray(column)
    return column[:, None] * np.ones((column.size, ncol), dtype=ones_dtype)


def reggrid_area(lats, lons):
    """Cell areas [m2] of a regular lat/lon grid with centres ``lats`` / ``lons`` (reference gis_utils.py:379-385:
    float64 column times float32 ones).  The hot path uses ``area_rows`` and never builds this grid."""
    lats, lons = np.asarray(lats), np.asarray(lons)
    return _per_row_grid(cellarea(lats, _mean_step(lons), _mean_step(lats)), lons.size, np.float32)


def reggrid_dx(lats, lons):
    """Cell widths [m] of a regular lat/lon grid (reference gis_utils.py:363-368)."""
    lats, lons = np.asarray(lats), np.asarray(lons)
    return _per_row_grid(degree_metres_x(lats) * _mean_step(lons), lons.size, lats.dtype)


def reggrid_dy(lats, lons):
    """Cell heights [m] of a regular lat/lon grid (reference gis_utils.py:371-376)."""
    lats, lons = np.asarray(lats), np.asarray(lons)
    return _per_row_grid(degree_metres_y(lats) * _mean_step(lats), lons.size, lats.dtype)


def degree_metres_y(lat):
    """Vertical length of a degree [m] at a latitude; reference gis_utils.py:415-431."""
    radlat = np.radians(lat)
    return 111132.92 + (-559.82 * np.cos(2.0 * radlat)) + (1.175 * np.cos(4.0 * radlat)) + (-0.0023 * np.cos(6.0 * radlat))


def degree_metres_x(lat):
    """Horizontal length of a degree [m] at a latitude; reference gis_utils.py:434-448."""
    radlat = np.radians(lat)
    return (111412.84 * np.cos(radlat)) + (-93.5 * np.cos(3.0 * radlat)) + (0.118 * np.cos(5.0 * radlat))


def step_length_table(nrow, latlon=False, transform=IDENTITY, dtype=np.float32):
    """Length of one D8 step (float32; float64 for core.snap, which adds Python floats) as ``[2*nrow-1, 3]``: (row of the cell + row of its downstream
    cell) x {vertical, horizontal, diagonal}.  ``gis_utils.distance(idx0, idx1, ncol, latlon,
    transform)`` (reference gis_utils.py:452-486) depends on nothing else, so the host evaluates it
    once per row pair — scalar by scalar, in the reference's own expression order (including its
    projected-CRS assignment ``dy = xres; dx = yres``) — and the device only adds.  The float32
    rounding is the reference's ``dist[idx_ds] + d`` (float32 scalar + Python float)."""
    import math

    xres, yres, north = transform[0], transform[4], transform[5]
    tab = np.zeros((max(1, 2 * nrow - 1), 3), dtype)
    for s in range(2 * nrow - 1):
        for kind, (dr, dc) in enumerate(((1, 0), (0, 1), (1, 1))):
            if latlon:
                lat = north + s / 2.0 * yres
                dy = 0.0 if dr == 0 else degree_metres_y(lat) * yres
                dx = 0.0 if dc == 0 else degree_metres_x(lat) * xres
            else:
                dy, dx = xres, yres
            tab[s, kind] = dtype(math.hypot(dy * dr, dx * dc))
    return tab


def cell_step_lengths(idxs_ds, mv, ncol, latlon=False, transform=IDENTITY):
    """float32 length of the step from every cell to its downstream cell for ARBITRARY links (general idxs_ds
    graphs): ``gis_utils.distance`` (reference gis_utils.py:452-486) depends on (r0 + r1, |dr|, |dc|) only, so
    it is evaluated once per distinct triple — scalar by scalar, in the reference's expression order — and
    looked up per cell.  0 for pits / nodata (never read)."""
    import math

    ds = np.asarray(idxs_ds)
    n = ds.size
    idx0 = np.arange(n, dtype=np.int64)
    d64 = np.where(ds == mv, idx0, ds.astype(np.int64))
    r0, r1 = idx0 // ncol, d64 // ncol
    dr, dc = np.abs(r1 - r0), np.abs(d64 % ncol - idx0 % ncol)
    key = ((r0 + r1) << 42) | (dr << 21) | dc
    uk, inv = np.unique(key, return_inverse=True)
    xres, yres, north = transform[0], transform[4], transform[5]
    vals = np.zeros(uk.size, np.float32)
    for i, k in enumerate(uk.tolist()):
        s, kdr, kdc = k >> 42, (k >> 21) & 0x1FFFFF, k & 0x1FFFFF
        if latlon:
            lat = north + s / 2.0 * yres
            dy = 0.0 if kdr == 0 else degree_metres_y(lat) * yres
            dx = 0.0 if kdc == 0 else degree_metres_x(lat) * xres
        else:
            dy, dx = xres, yres
        vals[i] = np.float32(math.hypot(dy * kdr, dx * kdc))
    return vals[inv]


def transform_from_bounds(west, south, east, north, width, height):
    """Affine transform of a raster given its bounds and size; reference gis_utils.py:162-170."""
    from ._affine import get_affine

    A = get_affine()
    return A.translation(west, north) * A.scale((east - west) / width, (south - north) / height)


_OFFSETS = {"center": (0.5, 0.5), "ul": (0, 0), "ur": (1, 0), "ll": (0, 1), "lr": (1, 1)}


def xy(transform, rows, cols, offset="center"):
    """x, y of the pixels (rows, cols) at one of five anchor points; semantics of reference gis_utils.py:191-226."""
    try:
        dcol, drow = _OFFSETS[offset]
    except KeyError:
        raise ValueError("Invalid offset") from None
    anchored = transform * transform.translation(dcol, drow)
    return anchored * (np.asarray(cols), np.asarray(rows))


def rowcol(transform, xs, ys, op=np.floor, precision=None):
    """rows, cols of the pixels containing (x, y); ``precision`` nudges points on a cell edge inwards by
    10^-precision in the direction ``op`` rounds (semantics of reference gis_utils.py:229-261)."""
    nudge = 0.0
    if precision is not None:
        nudge = 10.0**-precision * (1.0 - 2.0 * op(0.1))
    fcols, frows = (~transform) * (np.asarray(xs) + nudge, np.asarray(ys) - nudge)
    return op(frows).astype(int), op(fcols).astype(int)


def idxs_to_coords(idxs, transform, shape, offset="center"):
    """Coordinates of linear cell indices; IndexError outside the raster (reference gis_utils.py:264-298)."""
    idxs = np.asarray(idxs).astype(int)
    if idxs.size and (idxs.min() < 0 or idxs.max() >= shape[0] * shape[1]):
        raise IndexError("idxs coordinates outside domain")
    r, c = np.divmod(idxs, shape[1])
    return xy(transform, r, c, offset=offset)


def coords_to_idxs(xs, ys, transform, shape, op=np.floor, precision=None):
    """Linear cell indices of coordinates; IndexError outside the raster (reference gis_utils.py:301-338)."""
    r, c = rowcol(transform, xs, ys, op=op, precision=precision)
    if np.any((r < 0) | (r >= shape[0]) | (c < 0) | (c >= shape[1])):
        raise IndexError("XY coordinates outside domain")
    return r * shape[1] + c
