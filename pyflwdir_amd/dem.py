"""``pyflwdir.dem`` functions on or next to the hot path (reference pyflwdir/dem.py).

``fill_depressions`` is the step before the path (DEM -> D8): a native host routine behind the C-ABI
(``pfd_fill_depressions``, csrc/dem.hip) with the reference's exact heap order."""
from __future__ import annotations

import numpy as np

from . import _hip

__all__ = ["fill_depressions"]

_DT = {np.dtype(np.float32): _hip.PFD_F32, np.dtype(np.float64): _hip.PFD_F64, np.dtype(np.int32): _hip.PFD_I32}


def fill_depressions(elevtn, outlets="edge", idxs_pit=None, nodata=-9999.0, max_depth=-1.0, elv_max=None,
                     connectivity=8):
    """Fill local depressions in elevation data and derive local D8 flow directions; same arguments and
    return values ``(elevtn_out, d8)`` as the reference (pyflwdir/dem.py:17-143)."""
    elevtn = np.asarray(elevtn)
    if elevtn.ndim != 2:
        raise ValueError("elevtn should be 2 dimensional")
    if connectivity not in [4, 8]:
        raise ValueError('"connectivity" should either be 4 or 8')
    if elevtn.dtype not in _DT:
        if elevtn.dtype.kind == "f":
            elevtn = elevtn.astype(np.float64)
        elif elevtn.dtype.kind in "iu" and elevtn.dtype.itemsize < 4:
            elevtn = elevtn.astype(np.int32)
        else:
            raise NotImplementedError(f"elevation dtype {elevtn.dtype} is not supported (float32, float64, int32)")
    elevtn = np.ascontiguousarray(elevtn)
    nrow, ncol = elevtn.shape
    pits = None if idxs_pit is None else np.ascontiguousarray(idxs_pit, dtype=np.int64).ravel()
    out = np.empty_like(elevtn)
    d8 = np.empty(elevtn.shape, np.uint8)
    _hip.check(_hip.lib().pfd_fill_depressions(_DT[elevtn.dtype], _hip.ptr(elevtn), nrow, ncol, float(nodata), float(max_depth),
                                               int(outlets == "min"), int(elv_max is not None),
                                               0.0 if elv_max is None else float(elv_max), _hip.ptr(pits),
                                               0 if pits is None else pits.size, int(connectivity), _hip.ptr(out), _hip.ptr(d8)))
    return out, d8
