"""TEST INFRASTRUCTURE — ctypes front end of the CPU oracle (oracle/pfd_oracle.c).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  It is the *checker* (and the timed single-thread CPU baseline), never
the product path: nothing under ``pyflwdir_amd/`` imports it.

Function names follow the reference functions they restate:

===========================  =====================================================
``from_array``               pyflwdir/core_d8.py:42-67
``upstream_count``           pyflwdir/core.py:50-61
``idxs_seq``                 pyflwdir/core.py:87-117
``rank``                     pyflwdir/core.py:17-47
``accuflux``                 pyflwdir/streams.py:15-41 and :44-70
``strahler_order``           pyflwdir/streams.py:228-269
``basins``                   pyflwdir/basins.py:12-18 + pyflwdir/core.py:120-146
``height_above_nearest_drain``  pyflwdir/dem.py:299-330
``upstream_area_cell``       pyflwdir/pyflwdir.py:770-801 (unit="cell")
===========================  =====================================================
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_SFX = {np.dtype(np.int32): "i32", np.dtype(np.uint32): "u32", np.dtype(np.int64): "i64"}


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds).  Returns the library path."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pfd_oracle.c", "pfd_oracle_idx.inc")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def get_idxs_dtype(n: int):
    """pyflwdir/pyflwdir.py:105-127"""
    if n < 2147483647:
        return np.int32
    elif n < 4294967294:
        return np.uint32
    return np.int64


def _sfx(idxs_ds):
    return _SFX[np.dtype(idxs_ds.dtype)]


def from_array(d8: np.ndarray, dtype=None):
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    n = d8.size
    dtype = np.dtype(dtype or get_idxs_dtype(n))
    idxs_ds = np.empty(n, dtype)
    pits = np.empty(n, dtype)
    npit = C.c_int64(0)
    f = getattr(lib(), f"orc_d8_from_array_{_SFX[dtype]}")
    f.restype = C.c_int64
    nvalid = f(_p(d8), C.c_int64(nrow), C.c_int64(ncol), _p(idxs_ds), _p(pits), C.byref(npit))
    return idxs_ds, pits[: npit.value].copy(), int(nvalid)


def upstream_count(idxs_ds, mask=None):
    n = idxs_ds.size
    out = np.empty(n, np.int8)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    f = getattr(lib(), f"orc_upstream_count_{_sfx(idxs_ds)}")
    f.restype = None
    f(_p(idxs_ds), C.c_int64(n), _p(m), _p(out))
    return out


def idxs_seq(idxs_ds, idxs_pit):
    n = idxs_ds.size
    seq = np.empty(n, idxs_ds.dtype)
    pits = np.ascontiguousarray(idxs_pit, dtype=idxs_ds.dtype)
    f = getattr(lib(), f"orc_idxs_seq_{_sfx(idxs_ds)}")
    f.restype = C.c_int64
    m = f(_p(idxs_ds), C.c_int64(n), _p(pits), C.c_int64(pits.size), _p(seq))
    if m < 0:
        raise MemoryError("oracle: upstream matrix allocation failed")
    return seq[:m].copy()


def rank(idxs_ds):
    n = idxs_ds.size
    out = np.empty(n, np.int32)
    f = getattr(lib(), f"orc_rank_{_sfx(idxs_ds)}")
    f.restype = C.c_int64
    cnt = f(_p(idxs_ds), C.c_int64(n), _p(out))
    return out, int(cnt)


_ACC = {np.dtype(np.int32): ("i32", C.c_int32), np.dtype(np.int64): ("i64", C.c_int64),
        np.dtype(np.float32): ("f32", C.c_float), np.dtype(np.float64): ("f64", C.c_double)}


def accuflux(idxs_ds, seq, data, nodata=-9999, direction="up"):
    data = np.ascontiguousarray(data)
    view = data
    has_nodata = 1
    if data.dtype == np.uint32:
        view, has_nodata = data.view(np.int32), int(0 <= nodata < 2**32)
        nd = np.uint32(nodata).view(np.int32) if has_nodata else 0
    elif data.dtype == np.uint64:
        view, has_nodata = data.view(np.int64), int(0 <= nodata < 2**64)
        nd = np.uint64(nodata).view(np.int64) if has_nodata else 0
    else:
        nd = nodata
        if data.dtype.kind == "f" and nodata != nodata:
            has_nodata, nd = 0, 0.0
        if data.dtype.kind == "i" and not (np.iinfo(data.dtype).min <= nodata <= np.iinfo(data.dtype).max):
            has_nodata, nd = 0, 0
        if data.dtype.kind == "i" and float(nodata) != int(nodata):
            has_nodata, nd = 0, 0
    sfx, ct = _ACC[np.dtype(view.dtype)]
    out = np.empty_like(view)
    seq = np.ascontiguousarray(seq, dtype=idxs_ds.dtype)
    f = getattr(lib(), f"orc_accuflux_{_sfx(idxs_ds)}_{sfx}")
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, ct, C.c_int, C.c_int, C.c_void_p]
    f(_p(idxs_ds), idxs_ds.size, _p(seq), seq.size, _p(view), ct(nd if sfx[0] == "f" else int(nd)),
      has_nodata, int(direction == "down"), _p(out))
    return out.view(data.dtype)


def strahler_order(idxs_ds, seq, mask=None):
    n = idxs_ds.size
    out = np.empty(n, np.uint8)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=idxs_ds.dtype)
    f = getattr(lib(), f"orc_strahler_{_sfx(idxs_ds)}")
    f.restype = None
    f(_p(idxs_ds), C.c_int64(n), _p(seq), C.c_int64(seq.size), _p(m), _p(out))
    return out


def basins(idxs_ds, idxs_pit, seq, ids=None):
    if ids is None:
        ids = np.arange(1, idxs_pit.size + 1, dtype=np.uint32)
    ids = np.ascontiguousarray(ids)
    out = np.empty(idxs_ds.size, ids.dtype)
    seq = np.ascontiguousarray(seq, dtype=idxs_ds.dtype)
    pits = np.ascontiguousarray(idxs_pit, dtype=idxs_ds.dtype)
    f = getattr(lib(), f"orc_basins_{_sfx(idxs_ds)}")
    f.restype = None
    f(_p(idxs_ds), C.c_int64(idxs_ds.size), _p(seq), C.c_int64(seq.size), _p(pits), _p(ids),
      C.c_int64(pits.size), C.c_int(ids.dtype.itemsize), _p(out))
    return out


def height_above_nearest_drain(idxs_ds, seq, drain, elevtn):
    drain = np.ascontiguousarray(drain).astype(np.uint8)
    elevtn = np.ascontiguousarray(elevtn)
    if elevtn.dtype not in (np.float32, np.float64):
        elevtn = elevtn.astype(np.float64)
    out = np.empty(idxs_ds.size, np.float64)
    seq = np.ascontiguousarray(seq, dtype=idxs_ds.dtype)
    sfx = "f32" if elevtn.dtype == np.float32 else "f64"
    f = getattr(lib(), f"orc_hand_{sfx}_{_sfx(idxs_ds)}")
    f.restype = None
    f(_p(idxs_ds), C.c_int64(idxs_ds.size), _p(seq), C.c_int64(seq.size), _p(drain), _p(elevtn), _p(out))
    return out


def upstream_area_cell(d8, dtype=None):
    """Whole reference pipeline for ``upstream_area(unit="cell")``; returns
    (uparea int32 2-D, timings dict, stats dict)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    dtype = np.dtype(dtype or get_idxs_dtype(d8.size))
    out = np.empty(d8.size, np.int32)
    tim = (C.c_double * 3)()
    st = (C.c_int64 * 3)()
    f = getattr(lib(), f"orc_upstream_area_cell_{_SFX[dtype]}")
    f.restype = C.c_int
    rc = f(_p(d8), C.c_int64(nrow), C.c_int64(ncol), _p(out), tim, st)
    if rc != 0:
        raise MemoryError("oracle: allocation failed")
    return (out.reshape(nrow, ncol),
            {"decode_s": tim[0], "idxs_seq_s": tim[1], "accuflux_s": tim[2]},
            {"n_valid": st[0], "n_pits": st[1], "n_seq": st[2]})


# -- synthetic rasters (host twin of the device generator) ---------------------------------
SYNTH_RIVER = dict(tilt=1 << 26, white=2, nodata_pct=0)   # long rivers, pits on the last row only
SYNTH_ROUGH = dict(tilt=100000, white=2, nodata_pct=0)    # all 8 directions, many interior pits


def synth_d8(nrow, ncol, seed=0, tilt=1 << 26, white=2, nodata_pct=0, row0=0, nrows=None):
    nrows = nrow - row0 if nrows is None else nrows
    out = np.empty((nrows, ncol), np.uint8)
    f = lib().orc_synth_d8
    f.restype = None
    f(C.c_uint64(seed), C.c_int64(nrow), C.c_int64(ncol), C.c_int64(tilt), C.c_int64(white),
      C.c_int32(nodata_pct), C.c_int64(row0), C.c_int64(nrows), _p(out))
    return out


def synth_elev_f32(nrow, ncol, seed=0, tilt=1 << 26, white=2, nodata_pct=0, row0=0, nrows=None):
    nrows = nrow - row0 if nrows is None else nrows
    out = np.empty((nrows, ncol), np.float32)
    f = lib().orc_synth_elev_f32
    f.restype = None
    f(C.c_uint64(seed), C.c_int64(nrow), C.c_int64(ncol), C.c_int64(tilt), C.c_int64(white),
      C.c_int32(nodata_pct), C.c_int64(row0), C.c_int64(nrows), _p(out))
    return out


def synth_weights_f32(n, seed=1, i0=0):
    out = np.empty(n, np.float32)
    f = lib().orc_synth_weights_f32
    f.restype = None
    f(C.c_uint64(seed), C.c_int64(i0), C.c_int64(n), _p(out))
    return out
