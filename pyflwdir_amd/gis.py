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

__all__ = ["AREA_FACTORS", "IDENTITY", "affine_to_coords", "cellarea", "reggrid_area", "area_grid",
           "xy", "rowcol", "idxs_to_coords", "coords_to_idxs"]


def affine_to_coords(affine, shape):
    """Cell-centre coordinate axes (x per column, y per row); reference gis_utils.py:342-359."""
    height, width = shape
    x_coords, _ = affine * (np.arange(width) + 0.5, np.zeros(width) + 0.5)
    _, y_coords = affine * (np.zeros(height) + 0.5, np.arange(height) + 0.5)
    return x_coords, y_coords


def cellarea(lat, xres, yres):
    """Area [m2] of a lat/lon cell centred at ``lat``; reference gis_utils.py:405-412."""
    half = np.abs(yres) / 2.0
    l1 = np.radians(lat - half)
    l2 = np.radians(lat + half)
    dx = np.radians(np.abs(xres))
    return _R**2 * dx * (np.sin(l2) - np.sin(l1))


def reggrid_area(lats, lons):
    """Cell areas [m2] of a regular lat/lon grid; reference gis_utils.py:379-385.

    The result is float64: a float64 column vector times a float32 matrix of ones."""
    xres = np.abs(np.mean(np.diff(lons)))
    yres = np.abs(np.mean(np.diff(lats)))
    ones = np.ones((lats.size, lons.size), dtype=np.float32)
    return cellarea(lats, xres, yres)[:, None] * ones


def area_grid(transform, shape, latlon=False, unit="m2"):
    """Regular grid of cell areas; reference gis_utils.py:388-402 (int32 ones for "cell",
    float64 for lat/lon grids, float32 for projected grids)."""
    unit = str(unit).lower()
    if unit not in AREA_FACTORS:
        fstr = '", "'.join(AREA_FACTORS.keys())
        raise ValueError(f'Unknown unit: {unit}, select from "{fstr}".')
    if unit == "cell":
        return np.ones(shape, dtype=np.int32)
    if latlon:
        lon, lat = affine_to_coords(transform, shape)
        return reggrid_area(lat, lon) / AREA_FACTORS[unit]
    area0 = abs(transform[0] * transform[4]) / AREA_FACTORS[unit]
    return np.full(shape, area0, dtype=np.float32)


def area_rows(transform, shape, latlon=False, unit="m2"):
    """Column 0 of ``area_grid`` without building the grid: the cell area of a regular grid depends on
    the row only (same expressions, element for element, as ``area_grid`` / ``reggrid_area``)."""
    unit = str(unit).lower()
    if unit not in AREA_FACTORS:
        fstr = '", "'.join(AREA_FACTORS.keys())
        raise ValueError(f'Unknown unit: {unit}, select from "{fstr}".')
    if unit == "cell":
        return np.ones(shape[0], dtype=np.int32)
    if latlon:
        lon, lat = affine_to_coords(transform, shape)
        xres = np.abs(np.mean(np.diff(lon)))
        yres = np.abs(np.mean(np.diff(lat)))
        return cellarea(lat, xres, yres) * np.ones(lat.size, dtype=np.float32) / AREA_FACTORS[unit]
    area0 = abs(transform[0] * transform[4]) / AREA_FACTORS[unit]
    return np.full(shape[0], area0, dtype=np.float32)


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
    """x, y of pixels at rows/cols; reference gis_utils.py:191-226."""
    rows, cols = np.asarray(rows), np.asarray(cols)
    if offset not in _OFFSETS:
        raise ValueError("Invalid offset")
    coff, roff = _OFFSETS[offset]
    return transform * transform.translation(coff, roff) * (cols, rows)


def rowcol(transform, xs, ys, op=np.floor, precision=None):
    """rows, cols of the pixels containing (x, y); reference gis_utils.py:229-261."""
    xs, ys = np.asarray(xs), np.asarray(ys)
    eps = 0.0 if precision is None else 10.0**-precision * (1.0 - 2.0 * op(0.1))
    fcols, frows = (~transform) * (xs + eps, ys - eps)
    return op(frows).astype(int), op(fcols).astype(int)


def idxs_to_coords(idxs, transform, shape, offset="center"):
    """Cell coordinates of linear indices; reference gis_utils.py:264-298."""
    idxs = np.asarray(idxs).astype(int)
    size = np.multiply(*shape)
    if np.any(np.logical_or(idxs < 0, idxs >= size)):
        raise IndexError("idxs coordinates outside domain")
    ncol = shape[1]
    return xy(transform, idxs // ncol, idxs % ncol, offset=offset)


def coords_to_idxs(xs, ys, transform, shape, op=np.floor, precision=None):
    """Linear indices of coordinates; raises IndexError outside the raster; reference
    gis_utils.py:301-338."""
    nrow, ncol = shape
    rows, cols = rowcol(transform, xs, ys, op=op, precision=precision)
    inside = np.logical_and(np.logical_and(rows >= 0, rows < nrow), np.logical_and(cols >= 0, cols < ncol))
    if not np.all(inside):
        raise IndexError("XY coordinates outside domain")
    return rows * ncol + cols
