"""``FlwdirRaster`` for the MI355X hot path: the call surface of the reference's raster object
for D8 decode -> cell ordering -> accumulation / stream order / basins / HAND, with every
O(n) loop executed by hand-written HIP kernels behind the C-ABI in ``include/pfd.h``.

Mirrors (names, argument meaning, return conventions, error messages):

* ``from_array``                       reference pyflwdir/pyflwdir.py:130-205
* ``FlwdirRaster.__init__``            reference pyflwdir/pyflwdir.py:211-273, pyflwdir/flwdir.py:72-127
* ``upstream_area``                    reference pyflwdir/pyflwdir.py:770-801
* ``accuflux``                         reference pyflwdir/flwdir.py:567-602
* ``stream_order``                     reference pyflwdir/flwdir.py:508-547
* ``basins``                           reference pyflwdir/pyflwdir.py:564-599
* ``hand``                             reference pyflwdir/pyflwdir.py:1485-1511
* ``add_pits`` / ``order_cells``       reference pyflwdir/flwdir.py:231-279, pyflwdir/pyflwdir.py:299-315
* ``_check_data`` / ``_check_idxs_xy`` reference pyflwdir/flwdir.py:782-811, pyflwdir/pyflwdir.py:1548-1566

The host keeps the uint8 D8 raster (1 byte/cell); ``idxs_ds`` / ``idxs_seq`` (4-8 bytes/cell)
are public attributes of the reference and are materialised lazily, on request, by the GPU.
There is no CPU fallback: without the HIP library or a device the methods raise.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from . import _hip
from . import gis
from . import nextxy as core_nextxy
from ._affine import get_affine

Affine = get_affine()

__all__ = ["FlwdirRaster", "from_array", "from_dem", "FTYPES"]

# D8 alphabet, reference pyflwdir/core_d8.py:14-19
D8_DS = np.array([[32, 64, 128], [16, 0, 1], [8, 4, 2]], dtype=np.uint8)
D8_MV = np.uint8(247)
D8_PV = np.array([0, 255], dtype=np.uint8)
D8_ALL = np.array([32, 64, 128, 16, 0, 1, 8, 4, 2, 247, 255], dtype=np.uint8)
FTYPES = ("d8", "ldd", "nextxy")  # "d8" / "ldd": the D8 engines; "nextxy" and non-neighbour idxs_ds: the general engine
# LDD (PCRaster keypad codes, reference pyflwdir/core_ldd.py:11-17) is the same 8-neighbour scheme with
# other labels: a 256-entry table turns it into D8 on the way in (7 8 9 / 4 5 6 / 1 2 3; 5 = pit,
# 255 = nodata) and back on the way out; values outside the alphabet map to an invalid D8 code
LDD_DS = np.array([[7, 8, 9], [4, 5, 6], [1, 2, 3]], dtype=np.uint8)
LDD_MV = np.uint8(255)
LDD_ALL = np.array([7, 8, 9, 4, 5, 6, 1, 2, 3, 255], dtype=np.uint8)
_LDD_TO_D8 = np.full(256, 3, dtype=np.uint8)  # 3 is not a D8 code: rejected by the device like in the reference
_LDD_TO_D8[LDD_DS.ravel()] = D8_DS.ravel()
_LDD_TO_D8[LDD_MV] = D8_MV
_D8_TO_LDD = np.full(256, LDD_MV, dtype=np.uint8)
_D8_TO_LDD[D8_DS.ravel()] = LDD_DS.ravel()

_PAYLOAD = {np.dtype(np.int32): _hip.PFD_I32, np.dtype(np.int64): _hip.PFD_I64,
            np.dtype(np.float32): _hip.PFD_F32, np.dtype(np.float64): _hip.PFD_F64}


_INFER_ON_DEVICE_MIN = 1 << 24  # cells from which ftype="infer" asks the device whether a uint8 raster is D8


def _get_idxs_dtype(n):
    """Smallest index dtype for ``n`` cells; reference pyflwdir/pyflwdir.py:105-127."""
    if n < 2147483647:
        return np.int32
    elif n < 4294967294:
        return np.uint32
    return np.int64


def d8_isvalid(flwdir) -> bool:
    """True if ``flwdir`` is a 2-D uint8 raster of D8 values; reference pyflwdir/core_d8.py:105-122."""
    if not (isinstance(flwdir, np.ndarray) and flwdir.dtype == np.uint8 and flwdir.ndim == 2):
        return False
    # (one pass through a 256-entry table, chunk by chunk: np.unique sorts the raster — 47 s for 4.4e9 cells)
    bad = np.ones(256, dtype=bool)
    bad[D8_ALL] = False
    flat = flwdir.reshape(-1)
    step = 1 << 26
    for i in range(0, flat.size, step):
        if bad[flat[i:i + step]].any():
            return False
    return True


def ldd_isvalid(flwdir) -> bool:
    """True if ``flwdir`` is a 2-D uint8 raster of LDD values; reference pyflwdir/core_ldd.py:100-102."""
    if not (isinstance(flwdir, np.ndarray) and flwdir.dtype == np.uint8 and flwdir.ndim == 2):
        return False
    return bool(np.isin(flwdir, LDD_ALL).all())


def _infer_ftype(flwdir):
    """reference pyflwdir/pyflwdir.py:39-48: the first type whose alphabet holds every value, in
    the reference's order (d8, ldd, nextxy)."""
    if d8_isvalid(flwdir):
        return "d8"
    if ldd_isvalid(flwdir):
        return "ldd"
    if core_nextxy.isvalid(flwdir):
        return "nextxy"
    raise ValueError("The flow direction type could not be inferred.")


def _d8_from_idxs_ds(idxs_ds, shape, mv):
    """D8 codes from downstream indices (the inverse decode, reference core_d8.to_array,
    pyflwdir/core_d8.py:86-102), vectorised; links outside the 8 neighbours raise."""
    nrow, ncol = shape
    idx0 = np.arange(idxs_ds.size, dtype=np.int64)
    ds = idxs_ds.astype(np.int64)
    valid = idxs_ds != mv
    dr = np.where(valid, ds // ncol - idx0 // ncol, 0)
    dc = np.where(valid, ds % ncol - idx0 % ncol, 0)
    if np.any(np.abs(dr) > 1) or np.any(np.abs(dc) > 1):
        raise ValueError("Invalid data downstream index outside 8 neighbors.")
    d8 = D8_DS[dr + 1, dc + 1]
    d8[~valid] = D8_MV
    return d8.reshape(shape)


def _drdc_table():
    """(dr, dc) of every uint8 value exactly as core_d8.drdc computes them (reference pyflwdir/core_d8.py:20-37,
    including the values outside the D8 alphabet that ``check_ftype=False`` lets through: log2 arithmetic
    truncated to int8, everything above 128 a pit)."""
    dr = np.zeros(256, np.int64)
    dc = np.zeros(256, np.int64)
    for dd in range(256):
        if dd <= 8:
            if dd >= 2:
                dr[dd], dc[dd] = 1, int(np.int8(2 - np.log2(dd)))
            else:
                dr[dd], dc[dd] = 0, dd
        elif dd <= 128:
            if dd == 16:
                dr[dd], dc[dd] = 0, -1
            else:
                dr[dd], dc[dd] = -1, int(np.int8(np.log2(dd) - 6))
    return dr, dc


def _from_unchecked_d8(data, **kwargs):
    """``from_array(ftype="d8", check_ftype=False)`` on a raster holding values outside the D8 alphabet: the
    reference decodes them with core_d8.drdc; values that decode to a neighbour are rewritten to that
    neighbour's code (D8 engines), a raster with the -2 column offsets of the values 9..15 becomes a general
    idxs_ds graph."""
    dr, dc = _drdc_table()
    flat = data.ravel()
    present = np.unique(flat[flat != D8_MV])
    if np.all(np.abs(dc[present]) <= 1):
        lut = D8_DS[np.clip(dr + 1, 0, 2), np.clip(dc + 1, 0, 2)].astype(np.uint8)
        lut[D8_MV] = D8_MV
        lut[255] = 255
        flw = FlwdirRaster._from_d8(np.ascontiguousarray(lut[data]), **kwargs)
        pits = flw.idxs_pit  # outlets = pits whose ORIGINAL value is a pit code (pyflwdir.py:193)
        flw.idxs_outlet = pits[np.isin(flat[pits], D8_PV)]
        return flw
    nrow, ncol = data.shape
    n = flat.size
    idx0 = np.arange(n, dtype=np.int64)
    r_ds, c_ds = idx0 // ncol + dr[flat], idx0 % ncol + dc[flat]
    outside = (r_ds >= nrow) | (c_ds >= ncol) | (r_ds < 0) | (c_ds < 0)
    idx_ds = c_ds + r_ds * ncol
    pit = ((dr[flat] == 0) & (dc[flat] == 0)) | outside | (flat[np.where(outside, 0, idx_ds)] == D8_MV)
    dtype = _get_idxs_dtype(n)
    ds = np.where(flat == D8_MV, -1, np.where(pit, idx0, idx_ds)).astype(dtype)
    pits = idx0[(flat != D8_MV) & pit].astype(dtype)
    outl = pits[np.isin(flat[pits], D8_PV)]
    return FlwdirRaster(idxs_ds=ds, shape=data.shape, ftype="d8", idxs_pit=pits, idxs_outlet=outl, **kwargs)


def from_array(data, ftype="infer", check_ftype=True, mask=None, transform=gis.IDENTITY, latlon=False, **kwargs):
    """Flow direction raster parsed to an actionable (device-resident) format.

    Same signature and checks as the reference's ``from_array`` (pyflwdir/pyflwdir.py:130-205).
    ``ftype="d8"`` and ``"ldd"`` (or ``"infer"`` on either) run on the GPU path."""
    if ftype == "infer":
        if (isinstance(data, np.ndarray) and data.dtype == np.uint8 and data.ndim == 2 and mask is None
                and data.size >= _INFER_ON_DEVICE_MIN):
            # large uint8 rasters: "is it D8" is answered by the device pass that builds the graph anyway
            try:
                flw = FlwdirRaster._from_d8(np.ascontiguousarray(data), transform=transform, latlon=latlon, **kwargs)
                flw.ftype = "d8"
                return flw
            except ValueError as exc:
                if "not D8 codes" not in str(exc):
                    raise
        ftype = _infer_ftype(data)
        check_ftype = False
    if ftype not in FTYPES:
        raise ValueError(f'Unknown flow direction type: "{ftype}", select from {", ".join(FTYPES)}')
    if ftype == "nextxy":  # arbitrary links: the general idxs_ds engine (reference pyflwdir/pyflwdir.py:170-205)
        if check_ftype and not core_nextxy.isvalid(data):
            raise ValueError(f'The flow direction data with type "{ftype}" is invalid.')
        nextx, nexty = core_nextxy._split(data)
        if nextx.ndim != 2:
            raise ValueError("The FlwdirRaster should be 2 dimensional")
        if mask is not None:
            if mask.shape != np.asarray(data).shape:
                raise ValueError('"mask" shape does not match with data shape')
            data = np.where(mask != 0, data, core_nextxy.MV)
            nextx, nexty = core_nextxy._split(data)
        shape = nextx.shape
        idxs_ds, idxs_pit, _ = core_nextxy.from_array((nextx, nexty), dtype=_get_idxs_dtype(shape[0] * shape[1]))
        idxs_outlet = idxs_pit[np.isin(nextx.flat[idxs_pit], core_nextxy.PV)]
        return FlwdirRaster(idxs_ds=idxs_ds, idxs_pit=idxs_pit, idxs_outlet=idxs_outlet, shape=shape, ftype=ftype,
                            transform=transform, latlon=latlon, **kwargs)
    data = np.asarray(data)
    if data.ndim != 2:
        raise ValueError("The FlwdirRaster should be 2 dimensional")
    if ftype == "d8" and check_ftype and mask is None and data.dtype == np.uint8 and data.size >= _INFER_ON_DEVICE_MIN:
        # (small rasters keep the host check: argument errors before any device call)  The alphabet check of core_d8.isvalid rides the device's first pass over the raster (k_normalise counts the
        # bytes that are no D8 value) instead of a host pass of its own: 3 s of numpy at 8.1 Gcells
        try:
            flw = FlwdirRaster._from_d8(np.ascontiguousarray(data), transform=transform, latlon=latlon, **kwargs)
        except ValueError as exc:
            if "not D8 codes" in str(exc):
                raise ValueError(f'The flow direction data with type "{ftype}" is invalid.') from None
            raise
        flw.ftype = ftype
        return flw
    if check_ftype and not (d8_isvalid(data) if ftype == "d8" else ldd_isvalid(data)):
        raise ValueError(f'The flow direction data with type "{ftype}" is invalid.')
    if ftype == "ldd":  # same graph, other labels (core_ldd.from_array, reference pyflwdir/core_ldd.py:41-66)
        data = _LDD_TO_D8[np.ascontiguousarray(data, dtype=np.uint8)]
    if mask is not None:
        if mask.shape != data.shape:
            raise ValueError('"mask" shape does not match with data shape')
        data = np.where(mask != 0, data, D8_MV)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if ftype == "d8" and not check_ftype and not d8_isvalid(data):
        flw = _from_unchecked_d8(data, transform=transform, latlon=latlon, **kwargs)
    else:
        flw = FlwdirRaster._from_d8(data, transform=transform, latlon=latlon, **kwargs)
    flw.ftype = ftype
    return flw


def from_dem(data, nodata=-9999.0, max_depth=-1.0, transform=gis.IDENTITY, latlon=False, outlets="edge"):
    """Flow direction raster derived from elevation data by depression filling + steepest local descent
    (priority flood, Wang & Liu 2006); reference pyflwdir/pyflwdir.py:51-102."""
    from .dem import fill_depressions

    d8 = fill_depressions(data, nodata=nodata, max_depth=max_depth, outlets=outlets)[1]
    return from_array(d8, ftype="d8", check_ftype=False, transform=transform, latlon=latlon)



def _fill_where(out, codes, code, value, rows_per=None):
    """``out[codes == code] = value`` for rasters of billions of cells: row pieces on host threads (one numpy pass over
    8.1 Gcells — the comparison, its 8 GB mask, the masked store — took 5 s of ``upstream_area("km2")``)."""
    nrow = out.shape[0]
    rows_per = rows_per or max(1, (1 << 25) // max(1, out.shape[1]))

    def piece(r0):
        o, c = out[r0:r0 + rows_per], codes[r0:r0 + rows_per]
        m = c == code
        if m.any():
            o[m] = value

    starts = range(0, nrow, rows_per)
    if len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(min(16, len(starts), os.cpu_count() or 1)) as ex:
            list(ex.map(piece, starts))
    elif nrow:
        piece(0)

_FLOOD_PIECE = 1 << 25  # cells per host-thread piece of floodplains' input preparation

class FlwdirRaster(object):
    """Flow direction raster parsed to a device-resident graph (see module docstring)."""

    # -- construction -----------------------------------------------------------------------
    def __init__(self, idxs_ds, shape, ftype, idxs_pit=None, idxs_outlet=None, idxs_seq=None, nnodes=None,
                 transform=gis.IDENTITY, latlon=False, cache=True, device=0):
        idxs_ds = np.asarray(idxs_ds)
        if idxs_ds.size <= 1:
            raise ValueError(f"Invalid FlwdirRaster: size {idxs_ds.size}")
        if ftype not in FTYPES:
            raise ValueError(f'Unknown flow direction type: "{ftype}", select from {", ".join(FTYPES)}')
        if np.multiply(*np.array(shape, np.uint64)) != idxs_ds.size:
            raise ValueError(f"Invalid FlwdirRaster: shape {shape} does not match size {idxs_ds.size}")
        mv = self._mv_for(idxs_ds.dtype)
        d8 = None
        if ftype != "nextxy":
            try:
                d8 = _d8_from_idxs_ds(idxs_ds.ravel(), tuple(shape), mv)
            except ValueError:  # links outside the 8 neighbours (e.g. an upscaled network): general engine
                d8 = None
        if d8 is not None:
            self._setup(d8, transform, latlon, cache, device)
        else:
            self._setup_general(idxs_ds.ravel(), tuple(shape), transform, latlon, cache, device)
        self.ftype = ftype
        if self._d8 is None and ftype == "nextxy":
            self.order_cells(method="sort")  # the reference's only ordering of NEXTXY rasters (pyflwdir.py:292-297)
        self._idxs_ds = idxs_ds.ravel()
        self._idx_dtype = idxs_ds.dtype
        self._mv = mv
        if idxs_pit is not None:
            self._pit = np.asarray(idxs_pit)
        # (the reference stores what it is given: None when the object is built without outlets, pyflwdir.py:211-273)
        self.idxs_outlet = None if idxs_outlet is None else np.asarray(idxs_outlet)

    @classmethod
    def _from_d8(cls, d8, transform=gis.IDENTITY, latlon=False, cache=True, device=0, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected keyword arguments: {sorted(kwargs)}")
        if d8.size <= 1:
            raise ValueError(f"Invalid FlwdirRaster: size {d8.size}")
        self = cls.__new__(cls)
        self._setup(d8, transform, latlon, cache, device)
        return self

    @staticmethod
    def _mv_for(dtype):
        # -1 for signed, 4294967295 for uint32, ... ; reference pyflwdir/flwdir.py:112-117
        dtype = np.dtype(dtype)
        if dtype.kind == "u":
            return dtype.type(np.iinfo(dtype).max)
        return np.intp(-1)

    def _setup_general(self, idxs_ds, shape, transform, latlon, cache, device):
        """A graph with arbitrary links: the general idxs_ds engine of the library (csrc/general.hip)."""
        self._d8 = None
        self.shape = tuple(int(s) for s in shape)
        self.size = int(idxs_ds.size)
        self.device = device
        self._idx_dtype = np.dtype(idxs_ds.dtype)
        self._mv = self._mv_for(self._idx_dtype)
        self.cache = cache
        self._cached = dict()
        self._idxs_ds = idxs_ds
        self._pit = None
        self._seq = None
        self._nnodes = None
        self._h = _hip.RasterHandle.general(idxs_ds, self.shape[0], self.shape[1], device=device)
        self.idxs_outlet = self.idxs_pit
        self.set_transform(transform, latlon)
        self._order_nextxy = True  # (see __init__: NEXTXY rasters are ordered by rank before the first sweep)

    def _setup(self, d8, transform, latlon, cache, device):
        self._d8 = d8
        self.shape = tuple(int(s) for s in d8.shape)
        self.size = int(d8.size)
        self.ftype = "d8"
        self.device = device
        self._idx_dtype = np.dtype(_get_idxs_dtype(self.size))
        self._mv = self._mv_for(self._idx_dtype)
        self.cache = cache
        self._cached = dict()
        self._idxs_ds = None
        self._pit = None
        self._seq = None
        self._nnodes = None
        # device graph: raises ValueError("... no pits found") like the reference (flwdir.py:126)
        self._h = _hip.RasterHandle(d8, self.shape[0], self.shape[1], device=device)
        # outlets exclude the pits created at the raster edge / next to nodata (pyflwdir.py:193)
        pits = self.idxs_pit
        self.idxs_outlet = pits[np.isin(self._d8.flat[pits], D8_PV)]
        self.set_transform(transform, latlon)

    # -- representation / io ----------------------------------------------------------------
    @property
    def _dict(self):
        return {"ftype": self.ftype, "shape": self.shape, "nnodes": self.nnodes, "transform": self.transform,
                "latlon": self.latlon, "idxs_ds": self.idxs_ds, "idxs_seq": self._seq, "idxs_pit": self._pit}

    def __getitem__(self, idx):
        return self.idxs_ds[idx]

    def dump(self, fn):
        """Serialize to file with pickle (same dictionary layout as the reference,
        pyflwdir/pyflwdir.py:275-286, pyflwdir/flwdir.py:290-293)."""
        with open(fn, "wb") as handle:
            pickle.dump(self._dict, handle, protocol=-1)

    @staticmethod
    def load(fn):
        with open(fn, "rb") as handle:
            kwargs = pickle.load(handle)
        return FlwdirRaster(**kwargs)

    # -- properties -------------------------------------------------------------------------
    @property
    def idxs_ds(self):
        """Linear indices of the downstream cell (GPU decode, reference core_d8.from_array)."""
        if self._idxs_ds is None:
            if self._row_blocks_needed() > 1:  # int64 rung of the reference's ladder (pyflwdir.py:105-127), slice by slice
                ncol = self.shape[1]

                self._idxs_ds = self._sliced(lambda h, a, e: h.idxs_ds(np.int64), np.int64, -1,
                                             index_offset=True).astype(self._idx_dtype, copy=False)
            else:
                self._idxs_ds = self._h.idxs_ds(self._idx_dtype)
        return self._idxs_ds

    @property
    def idxs_pit(self):
        """Linear indices of pits/outlets, ascending."""
        if self._pit is None:
            if self._row_blocks_needed() > 1:
                ncol, parts = self.shape[1], []
                for r0, r1, a, e, h in self._row_slices():
                    if h is None:
                        continue
                    p = h.idxs_pit(np.int64) + a * ncol
                    parts.append(p[(p >= r0 * ncol) & (p < r1 * ncol)])
                self._pit = (np.concatenate(parts) if parts else np.empty(0, np.int64)).astype(self._idx_dtype, copy=False)
                if self._pit.size == 0:
                    raise ValueError("Invalid FlwdirRaster: no pits found")
            else:
                self._pit = self._h.idxs_pit(self._idx_dtype)
        return self._pit

    @property
    def idxs_seq(self):
        """Linear indices of valid cells ordered from down- to upstream, in the exact order of
        the reference's ``core.idxs_seq`` (pyflwdir/core.py:87-117); NEXTXY rasters are ordered by rank like
        in the reference (pyflwdir/pyflwdir.py:292-297)."""
        if self._seq is None:
            self.order_cells(method="walk" if self.ftype != "nextxy" else "sort")
        return self._seq

    @property
    def nnodes(self):
        if self._nnodes is None:
            if self._wide():  # (64-bit cell indices: no level structure; an acyclic raster orders every valid cell,
                # a cyclic one the cells its rank raster does not mark -1)
                self._nnodes = (self._h.info(counts=True)["n_valid"] if self.isvalid
                                else int(np.count_nonzero(self._h.rank() >= 0)))
            else:
                self._h.order_cells()
                self._nnodes = self._h.info()["n_seq"]
        return self._nnodes

    @property
    def ncells(self):
        return self.nnodes

    @property
    def mask(self):
        """Boolean array of valid cells (flattened like the reference, pyflwdir/flwdir.py:201-204)."""
        if self._d8 is None:
            return self.idxs_ds != self._mv
        return self._d8.ravel() != D8_MV

    @property
    def rank(self):
        """Cell rank: distance to the outlet in cells, -1 off the sequence, -9999 on nodata."""
        if "rank" in self._cached:
            return self._cached["rank"]
        rank = self._h.rank().reshape(self.shape)
        if self.cache:
            self._cached.update(rank=rank)
        return rank

    @property
    def isvalid(self):
        """True if no valid cell is part of (or drains to) a loop."""
        self._cached.pop("rank", None)
        if self._wide():  # (ask the tiled rank query itself: no 32 GB rank raster on the host for a yes / no)
            max_rank = self._h.graph_stats()["max_rank"]
            if max_rank == -2:
                raise NotImplementedError("isvalid: the raster is beyond the tiled rank query (more than 65535 tile rows or "
                                          "~4e9 perimeter slots)")
            return max_rank >= 0
        return bool(np.all(self.rank != -1))

    @property
    def n_upstream(self):
        if self._row_blocks_needed() > 1:
            return self._sliced(lambda h, a, e: h.upstream_count(), np.int8, -9).reshape(self.shape)
        return self._h.upstream_count().reshape(self.shape)

    @property
    def area(self):
        """Cell area [m2]; reference pyflwdir/pyflwdir.py:430-440."""
        if "area" in self._cached:
            return self._cached["area"]
        area = gis.area_grid(self.transform, self.shape, self.latlon, unit="m2")
        if self.cache:
            self._cached.update(area=area)
        return area

    @property
    def bounds(self):
        w, n = self.transform.xoff, self.transform.yoff
        e, s = self.transform * (self.shape[1], self.shape[0])
        return np.array([w, s, e, n], dtype=np.float64)

    # -- set / modify -----------------------------------------------------------------------
    def set_transform(self, transform, latlon=False):
        """reference pyflwdir/pyflwdir.py:317-337"""
        if not (hasattr(transform, "a") and hasattr(transform, "f") and len(transform) >= 6):
            try:
                transform = Affine(*transform)
            except TypeError:
                raise ValueError("Invalid transform.")
        self.transform = transform
        self.latlon = latlon

    def order_cells(self, method="sort"):
        """Order cells from down- to upstream; reference pyflwdir/flwdir.py:231-250 (default "sort" like the
        reference).  "walk" reproduces the reference's breadth-first order exactly on the GPU; "sort" is the
        reference's own numpy expression over the GPU-computed ranks.  Beyond 2**32 - 2 cells (int64 indices,
        pyflwdir.py:105-127) ranks and the breadth-first order come from csrc/order64.hip (cells that never reach a pit:
        rank -1, left out of the sequence — a walk from the pits, like the reference's)."""
        if method == "walk":
            if self._d8 is None:  # general graph: an installed "sort" order must not pose as the breadth-first one
                self._h.clear_idxs_seq()
            self._seq = self._h.idxs_seq(self._idx_dtype)
        elif method == "sort":
            rnk = self._h.rank()
            n = int(np.sum(rnk >= 0))
            self._seq = np.argsort(rnk)[-n:].astype(self._idx_dtype)
            if self._d8 is None:  # general graph: the sweeps follow this sequence (float sums depend on it)
                self._h.set_idxs_seq(self._seq)
        else:
            raise ValueError(f'Invalid method {method}, select from ["walk", "sort"]')
        self._nnodes = self._seq.size

    def add_pits(self, idxs=None, xy=None, streams=None):
        """Add pits to the flow direction raster; reference pyflwdir/pyflwdir.py:299-315,
        pyflwdir/flwdir.py:261-279."""
        idxs1 = self._check_idxs_xy(idxs, xy, streams)
        # validated on the host BEFORE anything is modified, so that host and device state cannot diverge
        i64 = np.asarray(idxs1, dtype=np.int64)
        if np.any(i64 < 0) or np.any(i64 >= self.size):
            raise IndexError("idxs outside domain")
        if np.any(~self.mask[i64]):
            raise ValueError("add_pits: indices must address valid (non-nodata) cells")
        # the reference keeps its pit list as np.unique(concatenate([idxs_pit, idxs1])) (flwdir.py:276): same set as
        # the device's, in the promoted dtype of the two
        pit_after = np.unique(np.concatenate([self.idxs_pit, idxs1]))
        self._h.add_pits(idxs1)
        if self._d8 is None:  # general graph: the host mirror is idxs_ds itself
            self._idxs_ds = self._idxs_ds.copy()
            self._idxs_ds[i64] = i64.astype(self._idx_dtype)
            self._pit = pit_after
            self._seq = None
            self._nnodes = None
            self._cached.clear()
            self.idxs_outlet = self.idxs_pit
            if self.ftype == "nextxy":  # the device forgot the installed order with the edit: the reference re-orders
                self.order_cells(method="sort")  # NEXTXY rasters by rank before the next sweep (pyflwdir.py:292-297)
            return
        self._d8 = self._d8.copy()
        self._d8.flat[idxs1] = 0
        self._idxs_ds = None
        self._pit = pit_after
        self._seq = None
        self._nnodes = None
        self._cached.clear()
        pits = self.idxs_pit
        self.idxs_outlet = pits[np.isin(self._d8.flat[pits], D8_PV)]

    def to_array(self, ftype=None):
        """2-D flow direction raster in D8 or LDD codes; reference pyflwdir/pyflwdir.py:341-359
        (core_d8.to_array / core_ldd.to_array re-encode ``idxs_ds``: every pit is written as the pit code)."""
        if ftype is None:
            ftype = self.ftype
        if ftype not in FTYPES:
            raise ValueError(f'ftype "{ftype}" unknown')
        if ftype == "nextxy":  # core_nextxy.to_array, reference pyflwdir/core_nextxy.py:36-86
            return core_nextxy.to_array(self.idxs_ds, self.shape, mv=self._mv)
        if self._d8 is None:  # raises for links outside the 8 neighbours, like core_d8.to_array (core_d8.py:86-102)
            d8 = _d8_from_idxs_ds(self.idxs_ds, self.shape, self._mv)
            return d8 if ftype == "d8" else _D8_TO_LDD[d8]
        d8 = self._d8.copy()
        d8.flat[self.idxs_pit] = 0
        return d8 if ftype == "d8" else _D8_TO_LDD[d8]

    # -- spatial ----------------------------------------------------------------------------
    def index(self, xs, ys, **kwargs):
        return gis.coords_to_idxs(xs, ys, self.transform, self.shape, **kwargs)

    def xy(self, idxs, **kwargs):
        return gis.idxs_to_coords(idxs, self.transform, self.shape, **kwargs)

    # -- the hot path -------------------------------------------------------------------------
    def upstream_area(self, unit="cell", exact=True):
        """Upstream area map; reference pyflwdir/pyflwdir.py:770-801.  ``unit="cell"`` returns the
        int32 upstream cell count; other units accumulate the cell-area grid (float64 for
        lat/lon grids, float32 for projected ones).  -9999 on nodata cells.

        ``exact`` (not in the reference; default True = bit-identical to its serial sums).  ``exact=False`` is an
        opt-in tolerance mode for the float64 sums of lat/lon grids only: the row areas are quantised to 64-bit fixed
        point and accumulated as integers on the tiled engine of ``unit="cell"`` (csrc/wide.h) — independent of any
        execution order, within n_cells / 2**63 relative of the real-number sum (1.3e-10 at 30000 x 30000), and several
        times faster on a fresh raster because it needs no ordering plan.  It is never offered for float32 sums
        (projected grids): a sequential float32 sum drifts from the real-number sum by more than the 1e-6 this path
        promises, so only the exact order reproduces it.  Whatever the fast form cannot take (cycles, general graphs) is
        answered by the exact form."""
        unit = str(unit).lower()
        if unit not in gis.AREA_FACTORS:
            fstr = '", "'.join(gis.AREA_FACTORS.keys())
            raise ValueError(f'Unknown unit: {unit}, select from "{fstr}".')
        if unit == "cell":
            return self._h.upstream_area_cell().reshape(self.shape)
        # the cell area of a regular grid depends on the row only: hand the device one value per row
        # (element for element the reference's  area / AREA_FACTORS[unit]) instead of an n-element grid
        rows = np.ascontiguousarray(gis.area_rows(self.transform, self.shape, self.latlon, unit="m2")
                                    / gis.AREA_FACTORS[unit])
        nb = self._row_blocks_needed()
        if not exact and rows.dtype == np.float64 and self._d8 is not None:
            # (one handle whatever the size: the tiled engine addresses tiles and slots, not cells — also beyond 2**32 - 2
            #  cells, where the exact form runs in seeded row blocks)
            out, quantum = self._h.upstream_area_rows_fixed(rows)
            if out is not None:
                self._last_quantum = quantum  # (one unit of the fixed-point scale, in the unit asked for)
                return out.reshape(self.shape)
        if nb > 1:  # beyond 32-bit cell indices: seeded row blocks (pyflwdir_amd/dist.py), bit-identical
            from . import dist

            self._refuse_cycles_in_blocks("upstream_area")
            out = dist.accuflux_blocks(self._d8, nb, rows, (-9999, -9999.0, 1), by_row=True)[0]
            _fill_where(out, self._d8, D8_MV, -9999)
            return out
        out = self._h.accuflux_rows(rows, _PAYLOAD[rows.dtype], nodata_i=-9999, nodata_f=-9999.0, has_nodata=1,
                                    direction=_hip.PFD_UP, mask_invalid=1)
        return out.reshape(self.shape)

    def accuflux(self, data, nodata=-9999, direction="up"):
        """Accumulate ``data`` along the flow directions; reference pyflwdir/flwdir.py:567-602."""
        if direction not in ("up", "down"):
            raise ValueError(f'Unknown flow direction: {direction}, select from ["up", "down"].')
        data = np.asarray(data)
        flat = self._check_data(data, "data")
        if flat.dtype in _NARROW_INT:
            return self._accuflux_narrow(flat, nodata, direction).reshape(data.shape)
        view, code, nd_i, nd_f, has_nd = _payload_args(flat, nodata)
        out = self._accuflux_dev(view, code, nd_i, nd_f, has_nd, direction)
        return out.view(flat.dtype).reshape(data.shape)

    def _accuflux_dev(self, view, code, nd_i, nd_f, has_nd, direction):
        """accuflux of a 32- / 64-bit payload on the device: one handle, or row blocks beyond 2**32 - 2 cells."""
        nb = self._row_blocks_needed()
        if nb > 1:
            from . import dist

            self._refuse_cycles_in_blocks("accuflux")
            return dist.accuflux_blocks(self._d8, nb, view, (nd_i, nd_f, has_nd), direction=direction)[0].ravel()
        dirc = _hip.PFD_UP if direction == "up" else _hip.PFD_DOWN
        return self._h.accuflux(view, code, nodata_i=nd_i, nodata_f=nd_f, has_nodata=has_nd, direction=dirc)

    def _accuflux_narrow(self, flat, nodata, direction):
        """int8 / int16 / uint8 / uint16 payloads.  The reference accumulates in the payload's own dtype
        (streams.py:36 ``data.copy()``), wrap-around included; the device kernels exist for 32- and 64-bit integers.
        Accumulating in int32 gives the same values — and the same answers to the running nodata tests — as long
        as no partial sum leaves the narrow range, which is checked: non-negative data accumulate monotonically, so
        the final values bound the partial sums; otherwise the accumulated magnitudes of the cells that take part do
        (the nodata rule only ever leaves operands out, so the magnitudes with the nodata cells set to zero bound every
        partial sum).  A payload that WOULD wrap (the reference then returns wrapped sums) is refused instead of being
        answered differently.  Rasters beyond 2**32 - 2 cells take the row-block path like the wide dtypes."""
        dt = flat.dtype
        info = np.iinfo(dt)
        wide = flat.astype(np.int32)
        try:
            integral = float(nodata) == int(nodata)
        except (OverflowError, ValueError):
            integral = False
        has_nd = 1 if integral and info.min <= int(nodata) <= info.max else 0
        nd = int(nodata) if has_nd else 0
        out = self._accuflux_dev(wide, _hip.PFD_I32, nd, 0.0, has_nd, direction)
        takes_part = wide != nd if has_nd else np.ones(wide.shape, bool)
        if flat.size and int(wide[takes_part].min(initial=0)) >= 0:
            ok = int(out[takes_part].max(initial=0)) <= info.max
        else:
            mag = np.where(takes_part, np.abs(wide), 0).astype(np.int32)
            bound = self._accuflux_dev(mag, _hip.PFD_I32, 0, 0.0, 0, direction)
            ok = int(bound.max()) <= min(info.max, -int(info.min))
        if not ok:
            raise NotImplementedError(f"accuflux: the sums leave the range of the payload dtype {dt} (the reference "
                                      "wraps around there); pass the data as int32 / int64")
        return out.astype(dt)

    def upstream_sum(self, data, mv=-9999):
        """Sum of the values of the cells directly upstream; reference pyflwdir/flwdir.py:412-433,
        pyflwdir/arithmetics.py:147-169 (incl. where its serial loop writes the missing value)."""
        if self._d8 is None:  # (general graphs: the C layer would refuse as well)
            raise NotImplementedError("upstream_sum is not available for flow directions with non-neighbour links")
        data = np.asarray(data)
        flat = self._check_data(data, "data")
        view, code, nd_i, nd_f, has_nd = _payload_args(flat, mv)
        if self._row_blocks_needed() > 1:
            ncol = self.shape[1]
            out = self._sliced(lambda h, a, e: h.upstream_sum(np.ascontiguousarray(view[a * ncol:e * ncol]), code, nodata_i=nd_i,
                                                              nodata_f=nd_f, has_nodata=has_nd), view.dtype, 0)
            return out.view(flat.dtype).reshape(data.shape)
        out = self._h.upstream_sum(view, code, nodata_i=nd_i, nodata_f=nd_f, has_nodata=has_nd)
        return out.view(flat.dtype).reshape(data.shape)

    def stream_order(self, type="strahler", mask=None):
        """Strahler stream order map (uint8); reference pyflwdir/flwdir.py:508-547.  Like the
        reference, the result is cached under "strord" when ``cache=True`` regardless of mask."""
        mask = self._check_data(mask, "mask", optional=True)
        if type.lower() == "strahler":
            if "strord" in self._cached:
                strord = self._cached["strord"]
            else:
                m = None if mask is None else np.ascontiguousarray(mask != 0).view(np.uint8)
                nb = self._row_blocks_needed()
                if nb > 1:
                    from . import dist

                    self._refuse_cycles_in_blocks("stream_order")
                    strord = dist.strahler_blocks(self._d8, nb, m)[0].ravel()
                else:
                    strord = self._h.strahler(m)
                if self.cache:
                    self._cached.update(strord=strord)
        elif type.lower() == "classic":  # reference pyflwdir/flwdir.py:540-543, streams.py:191-225
            m = None if mask is None else np.ascontiguousarray(mask != 0).view(np.uint8)
            nb = self._row_blocks_needed()
            if nb > 1:  # beyond 32-bit cell indices: row blocks, one byte per cell instead of the index array
                from . import dist

                self._refuse_cycles_in_blocks("stream_order")
                strord = dist.classic_blocks(self._d8, nb, self.upstream_area(), m)[0].ravel()
            else:
                strord = self._h.stream_order_classic(np.ascontiguousarray(self.idxs_us_main), m)
        else:
            # the reference falls through to an UnboundLocalError here; be explicit instead
            raise ValueError(f'Unknown stream order type: {type}, select from ["strahler", "classic"].')
        return strord.reshape(self.shape)

    def stream_distance(self, mask=None, unit="cell"):
        """Distance to the outlet or to the next downstream True cell of ``mask``: int32 cell counts
        (``unit="cell"``) or float32 metres (``unit="m"``); reference pyflwdir/pyflwdir.py:837-863."""
        unit = str(unit).lower()
        if unit not in ["m", "cell"]:
            raise ValueError(f'Unknown unit: {unit}, select from "m", "cell"')
        mask = self._check_data(mask, "mask", optional=True)
        m = None if mask is None else np.ascontiguousarray(mask != 0).view(np.uint8)
        if unit == "cell":
            tab = None
        elif self._d8 is None:  # general graph: one step length per cell
            tab = gis.cell_step_lengths(self.idxs_ds, self._mv, self.shape[1], self.latlon, self.transform)
            return self._h.stream_distance(m, tab, per_cell=True).reshape(self.shape)
        else:
            tab = gis.step_length_table(self.shape[0], self.latlon, self.transform)
        nb = self._row_blocks_needed()
        if nb > 1:
            from . import dist

            self._refuse_cycles_in_blocks("stream_distance")
            return dist.stream_distance_blocks(self._d8, nb, m, tab)[0]
        return self._h.stream_distance(m, tab).reshape(self.shape)

    def main_upstream(self, uparea=None):
        """Linear index of the upstream cell with the largest ``uparea`` (default: the upstream cell
        count), the index dtype's missing value at headwaters; reference pyflwdir/flwdir.py:252-258,
        pyflwdir/core.py:191-219."""
        uparea = self._check_data(uparea, "uparea")
        if uparea.dtype not in _PAYLOAD:
            if uparea.dtype.kind not in "iubf":
                raise NotImplementedError(f"uparea dtype {uparea.dtype} is not supported on the HIP path")
            uparea = uparea.astype(np.float64 if uparea.dtype.kind == "f" else np.int64)
        if self._row_blocks_needed() > 1:
            ncol = self.shape[1]

            def one(h, a, e):
                return h.main_upstream(np.ascontiguousarray(uparea[a * ncol:e * ncol]), _PAYLOAD[uparea.dtype], np.int64)
            idxs_us_main = self._sliced(one, np.int64, -1, index_offset=True).astype(self._idx_dtype, copy=False)
        else:
            idxs_us_main = self._h.main_upstream(np.ascontiguousarray(uparea), _PAYLOAD[uparea.dtype], self._idx_dtype)
        if self.cache:
            self._cached.update(idxs_us_main=idxs_us_main)
        return idxs_us_main

    @property
    def idxs_us_main(self):
        """Linear indices of the main upstream cell; reference pyflwdir/flwdir.py:153-161."""
        if "idxs_us_main" in self._cached:
            return self._cached["idxs_us_main"]
        return self.main_upstream()

    def basins(self, idxs=None, xy=None, ids=None, **kwargs):
        """(Sub)basin map with a unique ID per (sub)basin; reference pyflwdir/pyflwdir.py:564-599."""
        if idxs is None and xy is None:
            idxs = self.idxs_pit
        else:
            idxs = self._check_idxs_xy(idxs, xy, **kwargs)
        idxs = np.asarray(idxs)
        if ids is not None:
            ids = np.atleast_1d(ids).ravel()
            if ids.size != idxs.size:
                raise ValueError("IDs size does not match size of idxs.")
            elif np.any(ids == 0):
                raise ValueError("IDs cannot contain a value zero.")
        else:
            ids = np.arange(1, idxs.size + 1, dtype=np.uint32)  # reference basins.py:14-15
        if ids.dtype.kind not in "iu" or ids.dtype.itemsize not in (1, 2, 4, 8):
            raise NotImplementedError(f"basin ids of dtype {ids.dtype} are not supported on the HIP path")
        idxs64 = idxs.astype(np.int64)
        if np.any(idxs64 < 0) or np.any(idxs64 >= self.size):
            raise IndexError("idxs outside domain")
        nb = self._row_blocks_needed()
        if nb > 1:
            # beyond 32-bit cell indices: the tiled label query addresses tiles and slots and runs on the whole raster
            # (csrc/paths.hip pfd_basins_tiled); a raster with cycles needs the level engine's 32-bit order — then the
            # row-block protocol of the multi-GPU path, blocks held by this process
            if self._wide():
                try:
                    return self._h.basins(idxs64, ids).reshape(self.shape)
                except NotImplementedError:
                    pass
            from . import dist

            return dist.basins_blocks(self._d8, nb, idxs64, ids).reshape(self.shape)
        return self._h.basins(idxs64, ids).reshape(self.shape)

    def _row_slices(self):
        """For rasters beyond 32-bit cell indices: (r0, r1, a, e, handle) over row chunks — a PLAIN handle on the rows
        [a, e) = the chunk [r0, r1) plus one context row on every inner side.  What a cell-local export (downstream index,
        pit test, upstream count / sum, main upstream cell) says about a cell of the chunk depends on its 8 neighbours
        only, and those see the same codes as in the whole raster; the context rows' own answers are dropped.  (Cells of a
        context row that point out of the slice become pits THERE — never in the chunk.)"""
        nrow, ncol = self.shape
        per = max(1, min(nrow, (1 << 30) // max(1, ncol)))
        if os.environ.get("PFD_ENABLE_KNOBS") == "1" and os.environ.get("PFD_TEST_BIG_CELLS"):
            per = max(1, int(os.environ["PFD_TEST_BIG_CELLS"]) // max(1, ncol))
        for r0 in range(0, nrow, per):
            r1 = min(nrow, r0 + per)
            a, e = max(0, r0 - 1), min(nrow, r1 + 1)
            try:
                h = _hip.RasterHandle(self._d8[a:e], e - a, ncol, device=self.device)
            except ValueError as exc:  # a slice without any pit (its own rows hold none either)
                if "no pits" not in str(exc):
                    raise
                h = None
            try:
                yield r0, r1, a, e, h
            finally:
                if h is not None:
                    h.close()

    def _sliced(self, fn, dtype, fill, index_offset=False):
        """Assemble a per-cell export of a raster beyond 32-bit cell indices from the row slices: ``fn(handle, a, e)`` ->
        flat array over the slice's rows; the chunk's own rows are copied out.  ``index_offset``: the values are cell
        indices local to the slice (negative: none) and get the slice's first index added on the way.  The copy (and the
        add) runs in a few host threads over pieces of the chunk — numpy releases the GIL inside its loops, and one thread
        moves an 8 GB chunk at a few GB/s: `main_upstream` of 8.1 Gcells spent 100 s here, single-threaded with a masked add."""
        from concurrent.futures import ThreadPoolExecutor

        nrow, ncol = self.shape
        out = np.empty(self.size, dtype)
        nthreads = max(1, min(16, (os.cpu_count() or 1) // 2))
        piece = 1 << 25
        with ThreadPoolExecutor(nthreads) as pool:
            for r0, r1, a, e, h in self._row_slices():
                dst = out[r0 * ncol:r1 * ncol]
                if h is None:
                    dst[:] = fill
                    continue
                src = np.asarray(fn(h, a, e)).reshape(-1)[(r0 - a) * ncol:(r1 - a) * ncol]
                off = a * ncol if index_offset else 0

                def move(i, dst=dst, src=src, off=off):
                    d, q = dst[i:i + piece], src[i:i + piece]
                    if off:
                        np.add(q, off, out=d, casting="unsafe")
                        d[q < 0] = fill
                    else:
                        d[:] = q
                list(pool.map(move, range(0, dst.size, piece)))
        return out

    def _wide(self):
        """The raster's cells need 64-bit indices: rank and idxs_seq come from csrc/order64.hip."""
        return self._d8 is not None and self._h.wide_cells()

    def _row_blocks_needed(self):
        """1, or the number of row blocks a raster beyond 2**32 - 2 cells is cut into for the operations whose engines
        address cells with 32 bits (the reference's index ladder reaches int64, pyflwdir.py:105-127): basins, hand,
        accuflux, upstream_area in area units, the Strahler order and stream_distance then run the row-block protocols of
        pyflwdir_amd/dist.py inside this one process — same kernels, bit-identical results.  (PFD_TEST_BIG_CELLS with PFD_ENABLE_KNOBS=1 lowers the threshold for tests.)"""
        import os

        limit = 4294967294
        if self._d8 is None:
            return 1
        if os.environ.get("PFD_ENABLE_KNOBS") == "1" and os.environ.get("PFD_TEST_BIG_CELLS"):
            limit = int(os.environ["PFD_TEST_BIG_CELLS"])
        if self.size <= limit:
            return 1
        per_block = min(limit, 1 << 31)
        return min(self.shape[0], -(-self.size // per_block))

    def _refuse_cycles_in_blocks(self, what):
        """The seeded up-sweeps over row blocks iterate to a fixpoint, which a cycle through a block edge does not have
        (sums grow for ever; a Strahler order may settle on values the reference never assigns: it leaves cells on or
        above a cycle untouched).  One tiled rank query on the whole raster tells; a raster even beyond that query's slot
        ids (max_rank -2: more than ~4e9 perimeter slots or 65535 tile rows) is left to the iteration bound of the
        fixpoint (``max_iter``: a cycle through a block edge never settles and raises there)."""
        max_rank = self._h.graph_stats()["max_rank"]
        if max_rank == -1:
            raise NotImplementedError(f"{what}: the raster holds a cycle and is too large for one ordering "
                                      "(beyond 2**32 - 2 cells the operation runs in row blocks, which need an acyclic raster)")
        if max_rank == -2 and what == "stream_order":
            # (sums on a cycle grow until max_iter raises; a Strahler order on a cycle can SETTLE on values the reference
            #  never assigns — without the rank query there is no telling, so the order is refused rather than guessed)
            raise NotImplementedError("stream_order: the raster is too large for the cycle check that the row-block "
                                      "Strahler order needs (more than 65535 tile rows or ~4e9 perimeter slots)")

    def hand(self, drain, elevtn):
        """Height above the nearest drain (float64); reference pyflwdir/pyflwdir.py:1485-1511."""
        drain = self._check_data(drain, "drain")
        elevtn = self._check_data(elevtn, "elevtn")
        drain_u8 = np.ascontiguousarray(drain == 1).view(np.uint8)
        if elevtn.dtype == np.float32:
            code = _hip.PFD_F32
        elif elevtn.dtype == np.float64 or elevtn.dtype.kind in "iub":
            code, elevtn = _hip.PFD_F64, elevtn.astype(np.float64, copy=False)
        else:
            raise NotImplementedError(f"elevation dtype {elevtn.dtype} is not supported on the HIP path")
        nb = self._row_blocks_needed()
        if nb > 1:
            from . import dist

            return dist.hand_blocks(self._d8, nb, drain_u8, np.ascontiguousarray(elevtn))[0].reshape(self.shape)
        return self._h.hand(drain_u8, np.ascontiguousarray(elevtn), code).reshape(self.shape)

    def ucat_area(self, idxs_out, unit="cell"):
        """Unit catchment map (high resolution) and area (low resolution); reference
        pyflwdir/pyflwdir.py:1159-1191 + pyflwdir/subgrid.py:51-93."""
        unit = str(unit).lower()
        if unit not in gis.AREA_FACTORS:
            fstr = '", "'.join(gis.AREA_FACTORS.keys())
            raise ValueError(f'Unknown unit: {unit}, select from "{fstr}".')
        idxs_out = np.asarray(idxs_out)
        flat = idxs_out.ravel()
        idx64 = np.where(flat == self._mv, -1, flat.astype(np.int64)) if flat.size else flat.astype(np.int64)
        rows = None
        if unit != "cell":
            rows = np.ascontiguousarray(gis.area_rows(self.transform, self.shape, self.latlon, unit="m2")
                                        / gis.AREA_FACTORS[unit])
        if self._row_blocks_needed() > 1:
            # the library's own form first (label query on the whole raster; float areas summed over the 64-bit sequence
            # on the device, csrc/subgrid.hip ucat_float_wide); a raster with cycles: composed on the host
            ucat_map = None
            if self._wide():
                try:
                    ucat_map, ucat_are = self._h.ucat_area(idx64, self._idx_dtype, rows)
                except NotImplementedError:
                    pass
            if ucat_map is None:
                ucat_map, ucat_are = self._ucat_area_wide(idx64, rows)
        else:
            ucat_map, ucat_are = self._h.ucat_area(idx64, self._idx_dtype, rows)
        return ucat_map.reshape(self.shape), ucat_are.reshape(idxs_out.shape)

    def _ucat_area_wide(self, idx64, rows):
        """subgrid.ucat_area (pyflwdir/subgrid.py:51-93) beyond 2**32 - 2 cells, composed from the operations that run
        there: the map is the label query of ``basins`` with label i + 1 on outlet i (the same fill from down- to upstream;
        a repeated outlet keeps the last label in both), the areas are the reference's own accumulation — every outlet
        starts with the area of its cell, then the cells of the sequence add theirs in sequence order (floats: the order
        is part of the result; ``np.add.at`` is that loop) — over the 64-bit breadth-first order of csrc/order64.hip."""
        k, ncol = idx64.size, self.shape[1]
        sel = np.flatnonzero(idx64 >= 0)
        if np.any(idx64[sel] >= self.size):
            raise IndexError("idxs outside domain")
        ids = (sel + 1).astype(np.uint32)
        if sel.size:
            ucat_map = self.basins(idxs=idx64[sel], ids=ids).ravel().astype(self._idx_dtype)
            ucat_map[idx64[sel]] = ids  # (an outlet on a nodata cell keeps its label, subgrid.py:81-85)
        else:
            ucat_map = np.zeros(self.size, self._idx_dtype)
        if rows is None:  # cells: integer counts, any order
            ucat_are = np.full(k, -9999, np.int32)
            counts = np.zeros(k + 1, np.int64)
            step = 1 << 28
            for i in range(0, ucat_map.size, step):
                counts += np.bincount(ucat_map[i:i + step], minlength=k + 1)
            owner = np.zeros(k, bool)
            owner[sel] = ucat_map[idx64[sel]] == ids
            ucat_are[sel] = 1
            ucat_are[owner] = counts[1:][owner].astype(np.int32)
            return ucat_map, ucat_are
        ucat_are = np.full(k, -9999, rows.dtype)  # (float64 on lat/lon grids, float32 on projected ones: FlwdirRaster.area)
        ucat_are[sel] = rows[idx64[sel] // ncol]
        acc = np.zeros(k + 1, rows.dtype)
        acc[1:] = np.where(idx64 >= 0, ucat_are, 0)
        is_outlet = np.zeros(self.size, bool)
        is_outlet[idx64[sel]] = True
        seq = self.idxs_seq
        step = 1 << 27
        for i in range(0, seq.size, step):
            part = seq[i:i + step]
            w = rows[part // ncol]
            w[is_outlet[part]] = 0.0  # (the reference skips cells that hold a label already: x + 0.0 is x)
            np.add.at(acc, ucat_map[part], w)
        owner = np.zeros(k, bool)
        owner[sel] = ucat_map[idx64[sel]] == ids
        ucat_are[owner] = acc[1:][owner]
        return ucat_map, ucat_are

    def floodplains(self, elevtn, uparea=None, upa_min=1000, b=0.3):
        """Floodplain boundaries from a HAND threshold that scales with upstream area, h ~ A**b (Nardi et al
        2019); reference pyflwdir/pyflwdir.py:1513-1545 + pyflwdir/dem.py:333-379.  int8: 1 floodplain, 0 not,
        -1 off the sequence."""
        elevtn = self._check_data(elevtn, "elevtn")
        uparea = self._check_data(uparea, "uparea", unit="km2")
        # drainh[idx0] = uparea[idx0] ** b, stored as float32: evaluated here, element by element as the reference does
        # (numpy scalar ** python float), only where it is used — in pieces on host threads (numpy releases the GIL; at
        # 8.1 Gcells the one-piece form took 14 of the call's 38 s)
        is_stream = np.empty(self.size, np.uint8)
        hs = np.empty(self.size, np.float32)  # (zeroed by the pieces: pages a thread has touched upload at PCIe rate, the
        #                                        untouched pages of np.zeros at a quarter of it)

        def piece(i0):
            with np.errstate(invalid="ignore"):  # (errstate is per thread)
                u = uparea[i0:i0 + _FLOOD_PIECE]
                m = u >= upa_min
                is_stream[i0:i0 + _FLOOD_PIECE] = m
                h = hs[i0:i0 + _FLOOD_PIECE]
                h[:] = 0
                sel = np.flatnonzero(m)
                if sel.size:
                    h[sel] = (u[sel] ** b).astype(np.float32)

        starts = range(0, self.size, _FLOOD_PIECE)
        if len(starts) > 1:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(min(16, len(starts), os.cpu_count() or 1)) as ex:
                list(ex.map(piece, starts))
        else:
            piece(0)
        if elevtn.dtype == np.float32:
            code = _hip.PFD_F32
        elif elevtn.dtype == np.float64 or elevtn.dtype.kind in "iub":
            code, elevtn = _hip.PFD_F64, elevtn.astype(np.float64, copy=False)
        else:
            raise NotImplementedError(f"elevation dtype {elevtn.dtype} is not supported on the HIP path")
        nb = self._row_blocks_needed()
        if nb > 1:  # beyond 32-bit cell indices: row blocks that exchange the floodplain state of their boundary rows
            from . import dist

            self._refuse_cycles_in_blocks("floodplains")
            out = dist.floodplains_blocks(self._d8, nb, elevtn, is_stream, hs)[0]
            # (cells that reach no pit are off the reference's sequence and keep -1; on an acyclic raster: nodata only)
            return out.reshape(self.shape)
        return self._h.floodplains(np.ascontiguousarray(elevtn), code, is_stream, hs).reshape(self.shape)

    # -- shortcuts ------------------------------------------------------------------------------
    def _check_data(self, data, name, optional=False, flatten=True, **kwargs):
        """Check data shape and size, return the flattened array; reference
        pyflwdir/flwdir.py:782-803 and pyflwdir/pyflwdir.py:1548-1559."""
        if data is None and optional:
            return
        if data is None:
            if name == "uparea":
                data = self.upstream_area(**kwargs)
            elif name == "basins":
                data = self.basins(**kwargs)
            elif name == "strord":
                data = self.stream_order(**kwargs)
        data = np.atleast_1d(data)
        if flatten:
            if data.size == 1:
                data = np.full(self.size, data, dtype=data.dtype)
            elif data.size != self.size:
                raise ValueError(f'"{name}" size does not match.')
            return data.ravel()
        if data.size == 1:
            data = np.full(self.shape, data, dtype=data.dtype)
        elif data.shape != self.shape:
            raise ValueError(f'"{name}" shape does not match.')
        return data

    def _check_idxs_xy(self, idxs=None, xy=None, streams=None):
        """reference pyflwdir/pyflwdir.py:1561-1566 and pyflwdir/flwdir.py:805-811"""
        if (xy is not None and idxs is not None) or (xy is None and idxs is None):
            raise ValueError("Either idxs or xy should be provided.")
        elif xy is not None:
            idxs = self.index(*xy)
        idxs = np.atleast_1d(idxs).ravel()
        streams = self._check_data(streams, "streams", optional=True)
        if streams is not None:  # snap to the first downstream True cell (reference flwdir.py:805-811)
            idxs = self.snap(idxs=idxs, mask=streams)[0]
        return idxs

    def snap(self, idxs=None, xy=None, mask=None, max_length=None, unit="cell", direction="down"):
        """Snap points to the first downstream cell where ``mask`` is True (or the pit of their path);
        reference pyflwdir/pyflwdir.py:500-562, pyflwdir/flwdir.py:404-463, core.snap core.py:440-480.
        One bounded walk per point on the device: downstream over the D8 codes or upstream along the main upstream
        cells, in cells or in metres; returns (idxs, dists)."""
        if self._d8 is None:
            raise NotImplementedError("snap is not available on a general idxs_ds graph on the HIP path")
        unit = str(unit).lower()
        if unit not in ["m", "cell"]:
            raise ValueError(f'Unknown unit: {unit}, select from ["m", "cell"].')
        direction = str(direction).lower()
        if direction not in ["up", "down"]:
            raise ValueError('Unknown flow direction: {direction}, select from ["up", "down"].')
        if (xy is not None and idxs is not None) or (xy is None and idxs is None):
            raise ValueError("Either idxs or xy should be provided.")
        if xy is not None:
            idxs = self.index(*xy)
        idxs = np.atleast_1d(idxs).ravel()
        mask = self._check_data(mask, "mask", optional=True)
        m = None if mask is None else np.ascontiguousarray(mask != 0).view(np.uint8)
        if np.any(idxs < 0) or np.any(idxs >= self.size):
            raise IndexError("idxs outside domain")
        up = None
        if direction == "up":  # along the main upstream cells (reference flwdir.py:551), the missing value = "none"
            us = self.idxs_us_main
            up = us.astype(np.int64)
            up[us == self._mv] = -1
        tab = gis.step_length_table(self.shape[0], self.latlon, self.transform, dtype=np.float64) if unit == "m" else None
        out, dist = self._h.snap(idxs, mask=m, idxs_us_main=up, step_lengths=tab, max_length=max_length)
        return out.astype(idxs.dtype if idxs.dtype.kind in "iu" else np.int64), dist  # (core.snap: dtype of idxs0)


_NARROW_INT = (np.dtype(np.int8), np.dtype(np.int16), np.dtype(np.uint8), np.dtype(np.uint16))


def _payload_args(flat, nodata):
    """Map a flattened payload + Python nodata to the device dtype code and nodata arguments.
    `has_nodata` is False when the reference's comparison ``accu != nodata`` can never be False
    (NaN, out-of-range or non-integral nodata for integer data, negative nodata for unsigned)."""
    dt = flat.dtype
    if dt == np.uint32:
        view, info = flat.view(np.int32), np.iinfo(np.uint32)
    elif dt == np.uint64:
        view, info = flat.view(np.int64), np.iinfo(np.uint64)
    elif dt in _PAYLOAD:
        view, info = flat, (np.iinfo(dt) if dt.kind == "i" else None)
    else:
        raise NotImplementedError(f"payload dtype {dt} is not supported on the HIP path "
                                  "(supported: int8 ... int64, uint8 ... uint64, float32, float64)")
    view = np.ascontiguousarray(view)
    code = _PAYLOAD[view.dtype]
    if dt.kind == "f":
        nd = float(nodata)
        if nd != nd:
            return view, code, 0, 0.0, 0
        return view, code, 0, float(dt.type(nd)), 1
    # integer payloads: equality with a python number holds only for an integral in-range value
    try:
        integral = float(nodata) == int(nodata)
    except (OverflowError, ValueError):
        integral = False
    if not integral or not (info.min <= int(nodata) <= info.max):
        return view, code, 0, 0.0, 0
    nd = int(nodata)
    if dt.kind == "u":  # reinterpret as the signed kernel type
        nd = int(np.array([nd], dtype=dt).view(view.dtype)[0])
    return view, code, nd, 0.0, 1
