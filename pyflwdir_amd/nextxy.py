"""NEXTXY flow direction type (CaMa-Flood): codec to / from the general downstream index and the binary
reader; reference pyflwdir/core_nextxy.py.  X (column) and Y (row) are one-based; -9 / -10 mark river mouths /
inland pits, -9999 nodata.  The links are arbitrary cells, so NEXTXY rasters run on the general idxs_ds engine
(csrc/general.hip).  Vectorised numpy on the host: a codec pass next to file I/O, like the LDD table."""
from __future__ import annotations

import numpy as np

from . import gis

__all__ = ["read_nextxy"]

MV = np.int32(-9999)
PV = np.array([-9, -10], dtype=np.int32)


def _split(flwdir):
    if not ((isinstance(flwdir, tuple) and len(flwdir) == 2)
            or (isinstance(flwdir, np.ndarray) and flwdir.ndim == 3 and flwdir.shape[0] == 2)):
        raise TypeError("NEXTXY flwdir data not understood")
    nextx, nexty = flwdir
    return np.asarray(nextx), np.asarray(nexty)


def isvalid(flwdir) -> bool:
    """reference pyflwdir/core_nextxy.py:87-103"""
    try:
        nextx, nexty = _split(flwdir)
    except TypeError:
        return False
    mask = (nextx == MV) | np.isin(nextx, PV)
    return bool(nexty.dtype == "int32" and nextx.dtype == "int32" and nexty.shape == nextx.shape
                and np.all(nextx[~mask] >= 0) and np.all(nextx[mask] == nexty[mask]))


def from_array(flwdir, dtype=np.intp):
    """(idxs_ds, idxs_pit, n_valid); reference pyflwdir/core_nextxy.py:24-68: a cell is a pit if its code is a
    pit code, its target lies outside the raster, or its target is nodata."""
    nextx, nexty = _split(flwdir)
    nrow, ncol = nextx.shape[0], nextx.shape[-1]
    nx, ny = nextx.ravel().astype(np.int64), nexty.ravel().astype(np.int64)
    n = nx.size
    valid = nx != MV
    pit = np.isin(nx, PV) | np.isin(ny, PV)
    r_ds, c_ds = ny - 1, nx - 1
    outside = (r_ds >= nrow) | (c_ds >= ncol) | (r_ds < 0) | (c_ds < 0)
    idx_ds = c_ds + r_ds * ncol
    # (the reference reads nextx_flat[idx_ds] before it looks at `pit or outside`: numpy wraps a negative index,
    #  an index past the end raises; reproduce the wrap, treat past-the-end as outside)
    probe = np.where((idx_ds < -n) | (idx_ds >= n), 0, idx_ds)
    tgt_mv = nx[probe] == MV
    is_pit = valid & (pit | outside | tgt_mv)
    idxs_ds = np.full(n, -1, dtype=np.int64)
    cells = np.arange(n, dtype=np.int64)
    idxs_ds[valid] = np.where(is_pit, cells, idx_ds)[valid]
    out = idxs_ds.astype(dtype)  # (-1 -> the unsigned dtype's missing value by wrap-around, like the reference)
    return out, cells[is_pit].astype(dtype), int(valid.sum())


def to_array(idxs_ds, shape, mv=-1):
    """[2, nrow, ncol] int32 NEXTXY raster; reference pyflwdir/core_nextxy.py:36-86 (every pit becomes -9)."""
    ncol = shape[1]
    ds = np.asarray(idxs_ds)
    nextx = np.full(ds.size, MV, dtype=np.int32)
    nexty = np.full(ds.size, MV, dtype=np.int32)
    valid = ds != mv
    pit = valid & (ds == np.arange(ds.size, dtype=ds.dtype))
    link = valid & ~pit
    nextx[pit] = PV[0]
    nexty[pit] = PV[0]
    d64 = ds[link].astype(np.int64)
    nextx[link] = (d64 % ncol + 1).astype(np.int32)
    nexty[link] = (d64 // ncol + 1).astype(np.int32)
    return np.stack([nextx.reshape(shape), nexty.reshape(shape)])


def read_nextxy(fn, nrow: int, ncol: int, bbox):
    """Read NEXTXY data from a CaMa-Flood binary file; reference pyflwdir/core_nextxy.py:122-144.
    Returns (data [2, nrow, ncol] int32, transform)."""
    data = np.fromfile(fn, "i4").reshape(2, nrow, ncol)
    assert len(bbox) == 4, "Bounding box should contain 4 coordinates."
    transform = gis.transform_from_bounds(*bbox, ncol, nrow)
    return data, transform
