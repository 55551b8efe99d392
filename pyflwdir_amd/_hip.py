"""ctypes binding of libpfd_hip.so — the C-ABI declared in include/pfd.h.

There is deliberately NO fallback: if the shared library is missing, or no MI355X/HIP device
is visible, every entry point raises.  The CPU oracle under oracle/ is test infrastructure and
is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpfd_hip.so")

PFD_HOST, PFD_DEVICE = 0, 1
PFD_I32, PFD_U32, PFD_I64, PFD_F32, PFD_F64 = 1, 2, 3, 4, 5
_PAYLOAD_CODE = {np.dtype(np.int32): PFD_I32, np.dtype(np.int64): PFD_I64, np.dtype(np.float32): PFD_F32,
                 np.dtype(np.float64): PFD_F64}
PFD_UP, PFD_DOWN = 0, 1

IDX_CODE = {np.dtype(np.int32): PFD_I32, np.dtype(np.uint32): PFD_U32, np.dtype(np.int64): PFD_I64}

# status code -> Python exception (messages come from pfd_last_error()).  PFD_ENOPITS and
# PFD_EBADCODE map to the ValueErrors the reference raises for the same conditions
# (reference pyflwdir/flwdir.py:126-127, pyflwdir/pyflwdir.py:181-182).
_ERRORS = {-1: ValueError, -2: RuntimeError, -3: RuntimeError, -4: MemoryError, -5: ValueError,
           -6: ValueError, -7: NotImplementedError, -8: RuntimeError}

# every symbol include/pfd.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "pfd_abi_version", "pfd_last_error", "pfd_device_count", "pfd_malloc", "pfd_free", "pfd_memcpy_h2d",
    "pfd_memcpy_d2h", "pfd_device_synchronize", "pfd_trim", "pfd_raster_create", "pfd_raster_create_block",
    "pfd_raster_create_deferred", "pfd_raster_validate",
    "pfd_raster_destroy", "pfd_raster_info", "pfd_upstream_area_cell_blocks", "pfd_comm_unique_id", "pfd_comm_create",
    "pfd_comm_destroy", "pfd_comm_info", "pfd_upstream_area_cell_dist", "pfd_upstream_area_cell_begin", "pfd_upstream_area_cell_finish",
    "pfd_add_pits", "pfd_idxs_ds", "pfd_idxs_pit", "pfd_upstream_count", "pfd_order_cells", "pfd_idxs_seq",
    "pfd_rank", "pfd_upstream_area_cell", "pfd_upstream_area_cell_levels", "pfd_accuflux", "pfd_strahler",
    "pfd_accuflux_rows", "pfd_basins", "pfd_hand", "pfd_main_upstream", "pfd_stream_order_classic", "pfd_stream_distance", "pfd_set_profiling", "pfd_last_timing", "pfd_synth_d8", "pfd_synth_elev_f32",
    "pfd_synth_weights_f32", "pfd_graph_stats", "pfd_verify_upstream_area_cell", "pfd_verify_basins", "pfd_verify_hand", "pfd_hand_block", "pfd_accuflux_block", "pfd_strahler_block", "pfd_stream_distance_block", "pfd_checksum_i32", "pfd_basins_begin", "pfd_basins_finish", "pfd_fill_depressions", "pfd_ucat_area", "pfd_floodplains", "pfd_snap_downstream", "pfd_snap", "pfd_raster_create_general", "pfd_set_idxs_seq", "pfd_upstream_sum",
    "pfd_comm_exchange_rows", "pfd_comm_allgather_host", "pfd_set_block_io", "pfd_synth_mosaic", "pfd_calib_traffic", "pfd_set_block_update",
    "pfd_reserve", "pfd_alloc_stats", "pfd_mem_info", "pfd_transfer_stats", "pfd_count_nonfinite", "pfd_floodplains_block", "pfd_trib_info_block",
    "pfd_stream_order_classic_block", "pfd_upstream_area_rows_fixed", "pfd_floodplains_block_flags",
]

_lib = None


class HipLibraryMissing(ImportError):
    pass


def lib() -> C.CDLL:
    """Load libpfd_hip.so (once).  Raises HipLibraryMissing when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build the HIP extension first (python -m pyflwdir_amd.build, "
                "needs hipcc/ROCm).  pyflwdir_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.pfd_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            if name != "pfd_last_error":
                getattr(L, name).restype = C.c_int
        L.pfd_raster_create.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pfd_raster_create_block.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_void_p)]
        L.pfd_raster_create_deferred.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_void_p)]
        L.pfd_upstream_area_cell_blocks.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int]
        L.pfd_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
        L.pfd_comm_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pfd_comm_destroy.argtypes = [C.c_void_p]
        L.pfd_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pfd_upstream_area_cell_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_upstream_area_cell_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.pfd_upstream_area_cell_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.pfd_raster_destroy.argtypes = [C.c_void_p]
        L.pfd_raster_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.pfd_add_pits.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.pfd_idxs_ds.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.pfd_idxs_pit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.pfd_idxs_seq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.pfd_upstream_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_order_cells.argtypes = [C.c_void_p]
        L.pfd_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_upstream_area_cell.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_upstream_area_cell_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_accuflux.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int]
        L.pfd_accuflux_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int]
        L.pfd_upstream_area_rows_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                                   C.POINTER(C.c_double)]
        L.pfd_floodplains_block_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_strahler.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_basins.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int]
        L.pfd_hand.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_accuflux_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.pfd_stream_distance_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.pfd_strahler_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.POINTER(C.c_int64)]
        L.pfd_hand_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.POINTER(C.c_int64)]
        L.pfd_verify_basins.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.pfd_verify_hand.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.pfd_main_upstream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_int]
        L.pfd_upstream_sum.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_int]
        L.pfd_stream_order_classic.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_stream_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.pfd_verify_upstream_area_cell.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.pfd_checksum_i32.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.pfd_basins_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.pfd_basins_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.pfd_fill_depressions.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.c_double, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.pfd_ucat_area.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.pfd_floodplains.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_snap_downstream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.pfd_snap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.pfd_raster_create_general.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pfd_set_idxs_seq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.pfd_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.pfd_last_timing.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_int)]
        L.pfd_comm_exchange_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pfd_comm_allgather_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.pfd_set_block_io.argtypes = [C.c_void_p, C.c_int]
        L.pfd_set_block_update.argtypes = [C.c_void_p, C.c_int]
        L.pfd_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.pfd_free.argtypes = [C.c_int, C.c_void_p]
        L.pfd_memcpy_h2d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.pfd_memcpy_d2h.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.pfd_device_synchronize.argtypes = [C.c_int]
        L.pfd_trim.argtypes = [C.c_int]
        L.pfd_reserve.argtypes = [C.c_int, C.c_size_t]
        L.pfd_alloc_stats.argtypes = [C.POINTER(C.c_int64)]
        L.pfd_mem_info.argtypes = [C.c_int, C.POINTER(C.c_int64)]
        L.pfd_transfer_stats.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.pfd_count_nonfinite.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.pfd_trib_info_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        L.pfd_device_count.argtypes = [C.POINTER(C.c_int)]
        L.pfd_synth_d8.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                   C.c_int64, C.c_int64, C.c_void_p]
        L.pfd_synth_elev_f32.argtypes = L.pfd_synth_d8.argtypes
        L.pfd_synth_weights_f32.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().pfd_last_error().decode("utf-8", "replace")
        raise _ERRORS.get(rc, RuntimeError)(msg)


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().pfd_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def ptr(a):
    """void* of a numpy array (None -> NULL) or pass-through for raw device addresses."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, DeviceBuffer):
        return C.c_void_p(a.addr)
    return C.c_void_p(int(a))


# state record of the row-block floodplains (csrc/sweeps.hip FloodV): elevation / height threshold of the stream cell that
# started the floodplain, flag (1 floodplain, 0 not, -1 nodata), padding
FLOOD_STATE = np.dtype([("z", np.float32), ("h", np.float32), ("flag", np.int32), ("pad", np.int32)])


_explicit_reserve = False  # the caller sized an arena itself: the default one stands down
_auto_reserved = {}        # device -> bytes of the default arenas of this process
AUTO_RESERVE_MIN = 64 << 20
AUTO_RESERVE_BYTES_PER_CELL = 48


def reserve(nbytes: int, device: int = 0):
    """Reserve ``nbytes`` of HBM as one arena for the library's working buffers (``pfd_reserve``): afterwards a steady
    state never calls hipMalloc, whose latency for multi-GiB blocks is unpredictable (0.2 ms or seconds).  ``nbytes=0``
    releases the arenas that hold no live block.  A process that calls this sizes its arenas itself: the default arena
    of ``ensure_reserved`` is not added."""
    global _explicit_reserve
    check(lib().pfd_reserve(int(device), C.c_size_t(int(nbytes))))
    if nbytes:
        _explicit_reserve = True
    else:
        _auto_reserved.pop(int(device), None)


def mem_info(device: int = 0) -> dict:
    """Free and total HBM of ``device`` in bytes (``pfd_mem_info``)."""
    a = (C.c_int64 * 2)()
    check(lib().pfd_mem_info(int(device), a))
    return dict(free=int(a[0]), total=int(a[1]))


def ensure_reserved(n_cells: int, device: int = 0):
    """The default arena: the first handle of a process reserves
    ``min(0.5 x free HBM, 48 B x n_cells)`` — what the working buffers of the operations on a raster of ``n_cells`` need —
    so that the path a ``FlwdirRaster`` user takes is the path ``bench.py`` times; a later, larger raster adds the
    difference.  ``PFD_RESERVE_GIB`` overrides the size (``0``: no default arena); an explicit ``reserve()`` switches this
    off; arenas stay below half of the device's HBM in total and are released by ``reserve(0)``.  Failure to reserve is
    not an error: the class cache and hipMalloc remain."""
    if _explicit_reserve:
        return
    env = os.environ.get("PFD_RESERVE_GIB")
    have = _auto_reserved.get(int(device), 0)
    try:
        if env is not None:
            want = int(float(env) * 2**30)
            if want <= 0 or have:
                return
        else:
            want = AUTO_RESERVE_BYTES_PER_CELL * int(n_cells)
            if want - have < AUTO_RESERVE_MIN:
                return
            mi = mem_info(device)
            want = min(want - have, mi["free"] // 2, max(0, mi["total"] // 2 - have))
            if want < AUTO_RESERVE_MIN:
                return
        check(lib().pfd_reserve(int(device), C.c_size_t(int(want))))
        _auto_reserved[int(device)] = have + int(want)
    except (RuntimeError, MemoryError, ValueError):
        pass


def transfer_stats(reset: bool = True) -> dict:
    """Host <-> device traffic of this thread's API calls since the last reset (``pfd_transfer_stats``)."""
    a = (C.c_double * 6)()
    check(lib().pfd_transfer_stats(a, 1 if reset else 0))
    keys = ("h2d_bytes", "h2d_ms", "d2h_bytes", "d2h_ms", "prefault_ms", "host_results")
    return {k: float(v) for k, v in zip(keys, a)}


def alloc_stats() -> dict:
    """Allocator counters since the process started (``pfd_alloc_stats``)."""
    a = (C.c_int64 * 8)()
    check(lib().pfd_alloc_stats(a))
    keys = ("hipmalloc_calls", "cache_hits", "near_fit_hits", "arena_blocks", "idle_bytes", "reserved_bytes", "reserved_free", "live_blocks")
    return {k: int(v) for k, v in zip(keys, a)}


class DeviceBuffer:
    """A plain HBM allocation owned by Python (no torch): bench.py keeps the D8 raster and the
    result resident with these and passes them with memspace=PFD_DEVICE."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(lib().pfd_malloc(device, C.c_size_t(self.nbytes), C.byref(p)))
        self.addr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().pfd_memcpy_h2d(self.device, C.c_void_p(self.addr), ptr(arr), C.c_size_t(arr.nbytes)))
        return self

    def download(self, dtype, shape, offset_bytes: int = 0, out=None) -> np.ndarray:
        """``out``: a C-contiguous array of that dtype and size to fill (e.g. a block's rows of the whole result)."""
        if out is None:
            out = np.empty(shape, dtype)
        assert out.dtype == np.dtype(dtype) and out.flags.c_contiguous and out.size == int(np.prod(shape))
        assert out.nbytes + offset_bytes <= self.nbytes
        check(lib().pfd_memcpy_d2h(self.device, ptr(out), C.c_void_p(self.addr + offset_bytes), C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.addr:
            check(lib().pfd_free(self.device, C.c_void_p(self.addr)))
            self.addr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RasterHandle:
    """Owner of one ``pfd_raster`` (device-side graph of one raster on one GPU)."""

    def __init__(self, d8, nrow: int, ncol: int, device: int = 0, memspace: int = PFD_HOST, halo=(0, 0),
                 deferred: bool = False):
        """``nrow`` counts the OWNED rows; with ``halo=(top, bottom)`` (row block of a multi-GPU job) ``d8``
        holds top + nrow + bottom rows.  ``deferred``: decode/validate inside the first operation
        (``pfd_raster_create_deferred``); a device buffer ``d8`` is then referenced until that operation."""
        self._h = C.c_void_p()
        self.nrow, self.ncol, self.n = int(nrow), int(ncol), int(nrow) * int(ncol)
        self.device = device
        self.halo = (int(halo[0]), int(halo[1]))
        self.is_general = False
        if isinstance(d8, np.ndarray):
            d8 = np.ascontiguousarray(d8, dtype=np.uint8)
            assert d8.size == (self.nrow + sum(self.halo)) * self.ncol
        self._d8_ref = d8 if deferred else None  # keep a referenced device buffer alive
        ensure_reserved((self.nrow + sum(self.halo)) * self.ncol, device)
        if deferred:
            check(lib().pfd_raster_create_deferred(ptr(d8), self.nrow, self.ncol, self.halo[0], self.halo[1],
                                                   memspace, device, C.byref(self._h)))
        elif self.halo == (0, 0):
            check(lib().pfd_raster_create(ptr(d8), self.nrow, self.ncol, memspace, device, C.byref(self._h)))
        else:
            check(lib().pfd_raster_create_block(ptr(d8), self.nrow, self.ncol, self.halo[0], self.halo[1], memspace,
                                                device, C.byref(self._h)))

    @classmethod
    def general(cls, idxs_ds: np.ndarray, nrow: int, ncol: int, device: int = 0):
        """Handle of a general ``idxs_ds`` graph (links outside the 8 neighbours; pfd_raster_create_general)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.nrow, self.ncol, self.n = int(nrow), int(ncol), int(nrow) * int(ncol)
        self.device, self.halo, self._d8_ref, self.is_general = device, (0, 0), None, True
        idxs_ds = np.ascontiguousarray(idxs_ds).ravel()
        assert idxs_ds.size == self.n
        ensure_reserved(self.n, device)
        check(lib().pfd_raster_create_general(ptr(idxs_ds), IDX_CODE[idxs_ds.dtype], self.nrow, self.ncol, PFD_HOST, device,
                                              C.byref(self._h)))
        return self

    def close(self):
        if self._h:
            lib().pfd_raster_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- info -------------------------------------------------------------------------------
    def validate(self):
        """Normalise + validate a deferred handle now (counts become available)."""
        check(lib().pfd_raster_validate(self._h))

    def info(self, counts: bool = False) -> dict:
        if counts:
            self.validate()
        a = (C.c_int64 * 8)()
        check(lib().pfd_raster_info(self._h, a))
        keys = ["nrow", "ncol", "n_valid", "n_pits", "n_seq", "n_levels", "device", "bytes_held"]
        return dict(zip(keys, [int(v) for v in a]))

    def graph_stats(self) -> dict:
        """n_valid, n_pits, max_rank (longest flow path; -1 with cycles, -2 unknown: beyond the tiled query) and the in-degree histogram."""
        a = (C.c_int64 * 16)()
        check(lib().pfd_graph_stats(self._h, a))
        ntiles = max(1, -(-self.nrow // 64) * -(-self.ncol // 64))
        return dict(n_valid=int(a[0]), n_pits=int(a[1]), max_rank=int(a[2]), indegree_hist=[int(a[3 + k]) for k in range(9)],
                    tile_rounds=dict(local_max=int(a[12]), local_mean=round(int(a[13]) / ntiles, 2), final_max=int(a[14]),
                                     final_mean=round(int(a[15]) / ntiles, 2)))

    def verify_upstream_area_cell(self, upa, memspace=PFD_HOST) -> dict:
        """Local-equation check of an upstream_area("cell") result (see include/pfd.h)."""
        a = (C.c_int64 * 8)()
        check(lib().pfd_verify_upstream_area_cell(self._h, ptr(upa), memspace, a))
        return dict(bad_cells=int(a[0]), bad_nodata=int(a[1]), pit_sum=int(a[2]), n_pits=int(a[3]), checksum=int(a[4]),
                    n_valid=int(a[5]))

    def set_block_io(self, seed_memspace):
        """Where the ``*_block`` sweeps read their halo seeds from (PFD_HOST / PFD_DEVICE)."""
        check(lib().pfd_set_block_io(self._h, int(seed_memspace)))

    def set_block_update(self, mode):
        """What the next up-sweep of this block keeps / reuses (include/pfd.h pfd_set_block_update): 0 nothing, 1 keep
        the sweep, 2 update the kept sweep's result for the halo seeds that changed."""
        check(lib().pfd_set_block_update(self._h, int(mode)))

    def set_profiling(self, on=True):  # (2: also count the doubling rounds of the tile passes, see graph_stats)
        check(lib().pfd_set_profiling(self._h, int(on)))

    def last_timing(self):
        ms = (C.c_double * 16)()
        ln = (C.c_int64 * 16)()
        names = C.create_string_buffer(512)
        k = C.c_int(0)
        check(lib().pfd_last_timing(self._h, 16, ms, ln, names, 512, C.byref(k)))
        nm = names.value.decode().split(";") if k.value else []
        return [dict(name=nm[i], ms=ms[i], launches=int(ln[i])) for i in range(k.value)]

    # -- graph exports ----------------------------------------------------------------------
    def idxs_ds(self, dtype) -> np.ndarray:
        out = np.empty(self.n, dtype)
        check(lib().pfd_idxs_ds(self._h, IDX_CODE[np.dtype(dtype)], ptr(out), PFD_HOST))
        return out

    def idxs_pit(self, dtype) -> np.ndarray:
        out = np.empty(self.info(counts=True)["n_pits"], dtype)
        check(lib().pfd_idxs_pit(self._h, IDX_CODE[np.dtype(dtype)], ptr(out), PFD_HOST))
        return out

    def order_cells(self):
        check(lib().pfd_order_cells(self._h))

    def wide_cells(self) -> bool:
        """The handle's cells need 64-bit indices (csrc/order64.hip: rank and idxs_seq without the level engine)."""
        return self.n > 4294967294 or (os.environ.get("PFD_ENABLE_KNOBS") == "1" and bool(os.environ.get("PFD_TEST_ORDER64")))

    def idxs_seq(self, dtype) -> np.ndarray:
        if self.wide_cells() and not self.is_general:
            out = np.empty(self.info(counts=True)["n_valid"], np.int64)  # (room for every valid cell; cells that never
            check(lib().pfd_idxs_seq(self._h, PFD_I64, ptr(out), PFD_HOST))  # reach a pit are left out: n_seq entries)
            return out[: self.info()["n_seq"]].astype(dtype, copy=False)
        self.order_cells()
        out = np.empty(self.info()["n_seq"], dtype)
        check(lib().pfd_idxs_seq(self._h, IDX_CODE[np.dtype(dtype)], ptr(out), PFD_HOST))
        return out

    def set_idxs_seq(self, seq: np.ndarray):
        """General graphs: install the host's cell sequence (order_cells("sort")) as the order of the sweeps."""
        seq = np.ascontiguousarray(seq)
        check(lib().pfd_set_idxs_seq(self._h, IDX_CODE[seq.dtype], ptr(seq), seq.size))

    def clear_idxs_seq(self):
        """General graphs: forget an installed sequence (back to the breadth-first order of core.idxs_seq)."""
        check(lib().pfd_set_idxs_seq(self._h, PFD_I32, None, 0))

    def rank(self) -> np.ndarray:
        out = np.empty(self.n, np.int32)
        check(lib().pfd_rank(self._h, ptr(out), PFD_HOST))
        return out

    def upstream_count(self, mask=None) -> np.ndarray:
        out = np.empty(self.n, np.int8)
        m = None if mask is None else np.ascontiguousarray(mask).astype(np.uint8, copy=False).ravel()
        check(lib().pfd_upstream_count(self._h, ptr(m), ptr(out), PFD_HOST))
        return out

    def add_pits(self, idxs):
        idxs = np.ascontiguousarray(idxs, dtype=np.int64).ravel()
        check(lib().pfd_add_pits(self._h, ptr(idxs), idxs.size))

    # -- sweeps -----------------------------------------------------------------------------
    def upstream_area_cell(self, out=None, memspace=PFD_HOST, engine="auto"):
        f = lib().pfd_upstream_area_cell if engine == "auto" else lib().pfd_upstream_area_cell_levels
        if memspace == PFD_HOST:
            out = np.empty(self.n, np.int32)
        check(f(self._h, ptr(out), memspace))
        return out

    def accuflux(self, data, dtype_code, nodata_i=0, nodata_f=0.0, has_nodata=1, direction=PFD_UP,
                 mask_invalid=0, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty_like(data)
        check(lib().pfd_accuflux(self._h, dtype_code, ptr(data), int(nodata_i), float(nodata_f), int(has_nodata),
                                 direction, int(mask_invalid), ptr(out), memspace))
        return out

    def accuflux_rows(self, row_values, dtype_code, nodata_i=0, nodata_f=0.0, has_nodata=1, direction=PFD_UP,
                      mask_invalid=0, out=None, memspace=PFD_HOST):
        """accuflux of a payload that is constant along raster rows (``row_values``: nrow host values)."""
        row_values = np.ascontiguousarray(row_values)
        assert row_values.size == self.nrow
        if memspace == PFD_HOST:
            out = np.empty(self.n, row_values.dtype)
        check(lib().pfd_accuflux_rows(self._h, dtype_code, ptr(row_values), int(nodata_i), float(nodata_f),
                                      int(has_nodata), direction, int(mask_invalid), ptr(out), memspace))
        return out

    def upstream_area_rows_fixed(self, row_values, out=None, memspace=PFD_HOST):
        """Order-free upstream sums of one float64 value per row in 64-bit fixed point (pfd_upstream_area_rows_fixed).
        Returns (out, quantum), or (None, 0.0) when the library did not take the call (the caller runs accuflux_rows)."""
        row_values = np.ascontiguousarray(row_values, dtype=np.float64)
        assert row_values.size == self.nrow
        if memspace == PFD_HOST:
            out = np.empty(self.n, np.float64)
        used, quantum = C.c_int(0), C.c_double(0.0)
        check(lib().pfd_upstream_area_rows_fixed(self._h, ptr(row_values), ptr(out), memspace, C.byref(used),
                                                 C.byref(quantum)))
        return (out, quantum.value) if used.value else (None, 0.0)

    def strahler(self, mask=None, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty(self.n, np.uint8)
        check(lib().pfd_strahler(self._h, ptr(mask), ptr(out), memspace))
        return out

    def basins(self, outlets, ids, out=None, memspace=PFD_HOST):
        outlets = np.ascontiguousarray(outlets, dtype=np.int64).ravel()
        ids = np.ascontiguousarray(ids).ravel()
        if memspace == PFD_HOST:
            out = np.empty(self.n, ids.dtype)
        check(lib().pfd_basins(self._h, ptr(outlets), ptr(ids), outlets.size, ids.dtype.itemsize, ptr(out), memspace))
        return out

    def hand(self, drain, elevtn, elev_code, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty(self.n, np.float64)
        check(lib().pfd_hand(self._h, ptr(drain), elev_code, ptr(elevtn), ptr(out), memspace))
        return out

    def hand_block(self, drain, elevtn, elev_code, halo_seed, out=None, memspace=PFD_HOST, update=False):
        """HAND of a row block whose halo cells take their height from ``halo_seed`` (2 * ncol float64, host); drain,
        elevtn and the result cover the block's device raster (own + halo rows).  ``update``: ``out`` holds an earlier
        result, only its unknown (-inf) cells are recomputed.  Returns (out, boundary rows [2, ncol], unknown own cells)."""
        brows = None
        if not isinstance(halo_seed, DeviceBuffer):  # (a DeviceBuffer: pfd_set_block_io(PFD_DEVICE), rows stay on the device)
            halo_seed = np.ascontiguousarray(halo_seed, dtype=np.float64)
            assert halo_seed.size == 2 * self.ncol
            brows = np.empty((2, self.ncol), np.float64)
        if memspace == PFD_HOST and out is None:
            assert not update
            out = np.empty((self.nrow + sum(self.halo)) * self.ncol, np.float64)
        unk = C.c_int64(0)
        check(lib().pfd_hand_block(self._h, ptr(drain), elev_code, ptr(elevtn), ptr(halo_seed), 1 if update else 0, ptr(out),
                                   memspace, ptr(brows), C.byref(unk)))
        return out, brows, int(unk.value)

    def accuflux_block(self, data, dtype_code, halo_seed, out, nodata_i=0, nodata_f=0.0, has_nodata=0, by_row=False,
                       verify=False, memspace=PFD_HOST, direction=PFD_UP):
        """accuflux (either direction) of a row block whose halo cells hold ``halo_seed`` (2 * ncol values of the result
        type, host); data and ``out`` cover the block's device raster (own + halo rows), ``by_row``: one host value per
        device row.  Returns (boundary rows [2, ncol], own cells failing their local equation — verify only)."""
        brows = None
        if not isinstance(halo_seed, DeviceBuffer):
            halo_seed = np.ascontiguousarray(halo_seed)
            assert halo_seed.size == 2 * self.ncol and _PAYLOAD_CODE[halo_seed.dtype] == dtype_code
            brows = np.empty((2, self.ncol), halo_seed.dtype)
        bad = C.c_int64(0)
        check(lib().pfd_accuflux_block(self._h, dtype_code, ptr(data), 1 if by_row else 0, int(nodata_i), float(nodata_f),
                                       int(has_nodata), int(direction), ptr(halo_seed), 1 if verify else 0, ptr(out),
                                       memspace, ptr(brows), C.byref(bad)))
        return brows, int(bad.value)

    def stream_distance_block(self, mask, step_lengths, halo_seed, out, verify=False, memspace=PFD_HOST):
        """stream_distance of a row block whose halo cells hold ``halo_seed`` (2 * ncol int32, or float32 with
        ``step_lengths``: the table rows of the block's device raster).  Returns (boundary rows, failing own cells)."""
        real = step_lengths is not None
        dt = np.float32 if real else np.int32
        dev_seed = isinstance(halo_seed, DeviceBuffer)
        if not dev_seed:
            halo_seed = np.ascontiguousarray(halo_seed, dtype=dt)
            assert halo_seed.size == 2 * self.ncol
        if real:
            step_lengths = np.ascontiguousarray(step_lengths, dtype=np.float32)
            assert step_lengths.size == 3 * (2 * (self.nrow + sum(self.halo)) - 1)
        brows = None if dev_seed else np.empty((2, self.ncol), dt)
        bad = C.c_int64(0)
        check(lib().pfd_stream_distance_block(self._h, ptr(mask), int(real), ptr(step_lengths), ptr(halo_seed),
                                              1 if verify else 0, ptr(out), memspace, ptr(brows), C.byref(bad)))
        return brows, int(bad.value)

    def trib_info_block(self, uparea, dtype_code, mask=None, upa_min=0.0, out=None, memspace=PFD_HOST):
        """One byte per cell of the block's device raster: slot of the main upstream cell | (more than one upstream cell
        inside the mask) << 4 (include/pfd.h pfd_trib_info_block); the halo rows are the caller's to fill in."""
        if memspace == PFD_HOST:
            out = np.empty((self.nrow + sum(self.halo)) * self.ncol, np.uint8)
        check(lib().pfd_trib_info_block(self._h, int(dtype_code), ptr(uparea), float(upa_min), ptr(mask), ptr(out), memspace))
        return out

    def stream_order_classic_block(self, tinfo, mask, halo_seed, out, verify=False, memspace=PFD_HOST):
        """Classic stream order of a row block whose halo cells hold ``halo_seed`` (2 * ncol uint8).  Returns (boundary
        rows [2, ncol], own cells failing their local equation — verify only)."""
        brows = None
        if not isinstance(halo_seed, DeviceBuffer):
            halo_seed = np.ascontiguousarray(halo_seed, dtype=np.uint8)
            assert halo_seed.size == 2 * self.ncol
            brows = np.empty((2, self.ncol), np.uint8)
        bad = C.c_int64(0)
        check(lib().pfd_stream_order_classic_block(self._h, ptr(tinfo), ptr(mask), ptr(halo_seed), 1 if verify else 0, ptr(out),
                                                   memspace, ptr(brows), C.byref(bad)))
        return brows, int(bad.value)

    def floodplains_block(self, elevtn, elev_code, is_stream, stream_h, halo_seed, state, verify=False, memspace=PFD_HOST):
        """dem.floodplains of a row block whose halo cells hold the floodplain state ``halo_seed`` (2 * ncol records of
        FLOOD_STATE, host, or a DeviceBuffer after set_block_io); ``state`` covers the block's device raster.  Returns
        (boundary rows [2, ncol] of FLOOD_STATE, own cells failing their local equation — verify only)."""
        brows = None
        if not isinstance(halo_seed, DeviceBuffer):
            halo_seed = np.ascontiguousarray(halo_seed, dtype=FLOOD_STATE)
            assert halo_seed.size == 2 * self.ncol
            brows = np.empty((2, self.ncol), FLOOD_STATE)
        bad = C.c_int64(0)
        check(lib().pfd_floodplains_block(self._h, int(elev_code), ptr(elevtn), ptr(is_stream), ptr(stream_h), ptr(halo_seed),
                                          1 if verify else 0, ptr(state), memspace, ptr(brows), C.byref(bad)))
        return brows, int(bad.value)

    def floodplains_block_flags(self, state: "DeviceBuffer", out=None) -> np.ndarray:
        """int8 flags of the block's own rows from its device-resident floodplain state (pfd_floodplains_block_flags);
        ``out``: a C-contiguous int8 array of own_rows * ncol cells to fill (e.g. the block's rows of the whole result)."""
        if out is None:
            out = np.empty((self.nrow, self.ncol), np.int8)
        assert out.dtype == np.int8 and out.flags.c_contiguous and out.size == self.nrow * self.ncol
        check(lib().pfd_floodplains_block_flags(self._h, ptr(state), ptr(out), PFD_HOST))
        return out

    def strahler_block(self, mask, halo_seed, out, verify=False, memspace=PFD_HOST):
        """Strahler order of a row block whose halo cells hold ``halo_seed`` (2 * ncol uint8, host).  Returns
        (boundary rows [2, ncol], own cells failing their local equation — verify only)."""
        brows = None
        if not isinstance(halo_seed, DeviceBuffer):
            halo_seed = np.ascontiguousarray(halo_seed, dtype=np.uint8)
            assert halo_seed.size == 2 * self.ncol
            brows = np.empty((2, self.ncol), np.uint8)
        bad = C.c_int64(0)
        check(lib().pfd_strahler_block(self._h, ptr(mask), ptr(halo_seed), 1 if verify else 0, ptr(out), memspace,
                                       ptr(brows), C.byref(bad)))
        return brows, int(bad.value)

    def verify_basins(self, outlets, ids, labels, memspace=PFD_HOST) -> dict:
        """Local-equation check of a basins() result with uint32 ids (see include/pfd.h)."""
        outlets = np.ascontiguousarray(outlets, dtype=np.int64).ravel()
        ids = np.ascontiguousarray(ids, dtype=np.uint32).ravel()
        a = (C.c_int64 * 4)()
        check(lib().pfd_verify_basins(self._h, ptr(outlets), ptr(ids), outlets.size, ptr(labels), memspace, a))
        return dict(bad_cells=int(a[0]), bad_nodata=int(a[1]), checksum=int(a[2]), n_labelled=int(a[3]))

    def verify_hand(self, drain, elevtn, elev_code, hand, memspace=PFD_HOST) -> dict:
        """Local-equation check of a HAND result, bit for bit (see include/pfd.h)."""
        a = (C.c_int64 * 4)()
        check(lib().pfd_verify_hand(self._h, ptr(drain), elev_code, ptr(elevtn), ptr(hand), memspace, a))
        return dict(bad_cells=int(a[0]), bad_nodata=int(a[1]), checksum=int(a[2]), n_drain=int(a[3]))

    def ucat_area(self, idxs_out, map_dtype, area_rows=None):
        """(map[n] of map_dtype, area[k]); area_rows None: int32 cell counts, else nrow float32/float64 row areas."""
        idxs_out = np.ascontiguousarray(idxs_out, dtype=np.int64).ravel()
        ucmap = np.empty(self.n, map_dtype)
        if area_rows is None:
            code, are = PFD_I32, np.empty(idxs_out.size, np.int32)
        else:
            area_rows = np.ascontiguousarray(area_rows)
            assert area_rows.size == self.nrow and area_rows.dtype in (np.float32, np.float64)
            code, are = (PFD_F32 if area_rows.dtype == np.float32 else PFD_F64), np.empty(idxs_out.size, area_rows.dtype)
        check(lib().pfd_ucat_area(self._h, ptr(idxs_out), idxs_out.size, IDX_CODE[np.dtype(map_dtype)], ptr(ucmap), PFD_HOST,
                                  code, ptr(area_rows), ptr(are)))
        return ucmap, are

    def floodplains(self, elevtn, elev_code, is_stream, stream_h):
        out = np.empty(self.n, np.int8)
        check(lib().pfd_floodplains(self._h, elev_code, ptr(elevtn), ptr(is_stream), ptr(stream_h), ptr(out), PFD_HOST))
        return out

    def snap_downstream(self, idxs, mask, max_hops=-1):
        idxs = np.ascontiguousarray(idxs, dtype=np.int64).ravel()
        out, dist = np.empty(idxs.size, np.int64), np.empty(idxs.size, np.float32)
        check(lib().pfd_snap_downstream(self._h, ptr(idxs), idxs.size, ptr(mask), PFD_HOST, int(max_hops), ptr(out), ptr(dist)))
        return out, dist

    def snap(self, idxs, mask=None, idxs_us_main=None, step_lengths=None, max_length=None):
        """core.snap: downstream, or upstream along ``idxs_us_main`` (int64, negative = none); cells, or metres from
        the float64 ``step_lengths`` table; returns (idxs int64, dists float32)."""
        idxs = np.ascontiguousarray(idxs, dtype=np.int64).ravel()
        out, dist = np.empty(idxs.size, np.int64), np.empty(idxs.size, np.float32)
        up = None if idxs_us_main is None else np.ascontiguousarray(idxs_us_main, dtype=np.int64).ravel()
        tab = None if step_lengths is None else np.ascontiguousarray(step_lengths, dtype=np.float64)
        check(lib().pfd_snap(self._h, ptr(idxs), idxs.size, ptr(mask), ptr(up), ptr(tab), 0 if max_length is None else 1,
                             C.c_double(0.0 if max_length is None else float(max_length)), ptr(out), ptr(dist)))
        return out, dist

    def main_upstream(self, uparea, dtype_code, idx_dtype, upa_min=0.0, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty(self.n, idx_dtype)
        check(lib().pfd_main_upstream(self._h, dtype_code, ptr(uparea), float(upa_min), IDX_CODE[np.dtype(idx_dtype)],
                                      ptr(out), memspace))
        return out

    def upstream_sum(self, data, dtype_code, nodata_i=0, nodata_f=0.0, has_nodata=1, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty_like(data)
        check(lib().pfd_upstream_sum(self._h, dtype_code, ptr(data), int(nodata_i), float(nodata_f), int(has_nodata),
                                     ptr(out), memspace))
        return out

    def stream_order_classic(self, idxs_us_main, mask=None, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty(self.n, np.uint8)
        check(lib().pfd_stream_order_classic(self._h, IDX_CODE[np.dtype(idxs_us_main.dtype)], ptr(idxs_us_main),
                                             ptr(mask), ptr(out), memspace))
        return out

    def stream_distance(self, mask=None, step_lengths=None, out=None, memspace=PFD_HOST, per_cell=False):
        """``step_lengths`` None: int32 cell counts; else float32 with the host table [2*nrow-1, 3] (D8 rasters)
        or one value per cell (``per_cell``: general idxs_ds graphs)."""
        real = step_lengths is not None
        if memspace == PFD_HOST:
            out = np.empty(self.n, np.float32 if real else np.int32)
        if real:
            step_lengths = np.ascontiguousarray(step_lengths, dtype=np.float32)
            assert step_lengths.size == (self.n if per_cell else 3 * (2 * self.nrow - 1))
        check(lib().pfd_stream_distance(self._h, ptr(mask), int(real), ptr(step_lengths), ptr(out), memspace))
        return out


def upstream_area_cell_blocks(handles, outs=None, memspace=PFD_HOST):
    """upstream_area("cell") of a raster held as row blocks by the handles of THIS process."""
    k = len(handles)
    if memspace == PFD_HOST:
        outs = [np.empty(h.n, np.int32) for h in handles]
    hs = (C.c_void_p * k)(*[h._h for h in handles])
    ps = (C.c_void_p * k)(*[ptr(o) for o in outs])
    check(lib().pfd_upstream_area_cell_blocks(hs, k, ps, memspace))
    return outs


def upstream_area_cell_begin(handle, out=None, memspace=PFD_HOST):
    """Local phase of a multi-block pass; returns (out, record) with record = 4*ncol uint32."""
    if memspace == PFD_HOST:
        out = np.empty(handle.n, np.int32)
    rec = np.empty(4 * handle.ncol, np.uint32)
    check(lib().pfd_upstream_area_cell_begin(handle._h, ptr(out), memspace, ptr(rec)))
    return out, rec


def upstream_area_cell_finish(handle, all_records: np.ndarray, nblocks: int, block: int) -> bool:
    """Completes the pass with the records of all blocks (shape [nblocks, 4*ncol]); True if acyclic."""
    all_records = np.ascontiguousarray(all_records, dtype=np.uint32)
    assert all_records.size == nblocks * 4 * handle.ncol
    ok = C.c_int(0)
    check(lib().pfd_upstream_area_cell_finish(handle._h, ptr(all_records), nblocks, block, C.byref(ok)))
    return bool(ok.value)


def basins_begin(handle, outlets, ids, out=None, memspace=PFD_HOST):
    """Local phase of a multi-block basins query; ``outlets`` index the block's own rows.  Returns
    (out, record) with record = 6*ncol uint32."""
    outlets = np.ascontiguousarray(outlets, dtype=np.int64).ravel()
    ids = np.ascontiguousarray(ids).ravel()
    if memspace == PFD_HOST:
        out = np.empty(handle.n, ids.dtype)
    rec = np.empty(6 * handle.ncol, np.uint32)
    check(lib().pfd_basins_begin(handle._h, ptr(outlets), ptr(ids), outlets.size, ids.dtype.itemsize, ptr(out), memspace,
                                 ptr(rec)))
    handle._basins_out = out  # (the library writes it in finish: keep it alive)
    return out, rec


def basins_finish(handle, all_records: np.ndarray, nblocks: int, block: int) -> bool:
    all_records = np.ascontiguousarray(all_records, dtype=np.uint32)
    assert all_records.size == nblocks * 6 * handle.ncol
    ok = C.c_int(0)
    check(lib().pfd_basins_finish(handle._h, ptr(all_records), nblocks, block, C.byref(ok)))
    return bool(ok.value)


class Communicator:
    """RCCL communicator (one rank per GPU/process).  ``uid`` = 128 bytes from ``unique_id()`` of rank 0."""

    UID_BYTES = 128

    def __init__(self, uid: bytes, rank: int, world: int, device: int):
        self._c = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), self.UID_BYTES)
        check(lib().pfd_comm_create(buf, self.UID_BYTES, rank, world, device, C.byref(self._c)))
        self.rank, self.world, self.device = rank, world, device

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Communicator.UID_BYTES)
        check(lib().pfd_comm_unique_id(buf, Communicator.UID_BYTES))
        return buf.raw

    def info(self) -> dict:
        """World size / rank / device as RCCL reports them for this communicator."""
        n, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().pfd_comm_info(self._c, C.byref(n), C.byref(r), C.byref(d)))
        return dict(nranks=n.value, rank=r.value, device=d.value)

    def upstream_area_cell(self, handle, out=None, memspace=PFD_HOST):
        if memspace == PFD_HOST:
            out = np.empty(handle.n, np.int32)
        check(lib().pfd_upstream_area_cell_dist(handle._h, self._c, ptr(out), memspace))
        return out

    def exchange_rows(self, handle, result_buf, itemsize: int, seed_buf, c0: int = 0, c1: int = 0):
        """Neighbour exchange of the block's boundary rows, device to device (pfd_comm_exchange_rows): returns
        (sum of c0 over the ranks, sum of c1, my halo values changed, ranks whose halo values changed)."""
        a = (C.c_int64 * 4)(int(c0), int(c1), 0, 0)
        check(lib().pfd_comm_exchange_rows(self._c, handle._h, ptr(result_buf), int(itemsize), ptr(seed_buf), a))
        return int(a[0]), int(a[1]), bool(a[2]), int(a[3])

    def allgather_host(self, handle, data: bytes) -> list:
        """All-gather of equally sized host byte strings through the device and RCCL."""
        n = len(data)
        src = np.frombuffer(data, np.uint8)
        out = np.empty(n * self.world, np.uint8)
        check(lib().pfd_comm_allgather_host(self._c, handle._h, ptr(np.ascontiguousarray(src)), n, ptr(out)))
        return [out[i * n:(i + 1) * n].tobytes() for i in range(self.world)]

    def close(self):
        if self._c:
            lib().pfd_comm_destroy(self._c)
            self._c = C.c_void_p()


def count_nonfinite(buf, n: int, dtype_code: int, device: int = 0) -> int:
    """NaN / +-inf values among the first n float32 / float64 values of a device buffer."""
    out = C.c_int64(0)
    check(lib().pfd_count_nonfinite(device, int(dtype_code), ptr(buf), int(n), C.byref(out)))
    return int(out.value)


def checksum_i32(buf, n: int, device: int = 0) -> int:
    """Sum (64-bit two's complement) of n int32 values in a device buffer."""
    out = C.c_int64(0)
    check(lib().pfd_checksum_i32(device, ptr(buf), int(n), C.byref(out)))
    return int(out.value)


# -- synthetic rasters generated in HBM ---------------------------------------------------------
def synth_d8_device(nrow, ncol, seed=0, tilt=1 << 26, white=2, nodata_pct=0, row0=0, nrows=None, device=0):
    nrows = nrow - row0 if nrows is None else nrows
    buf = DeviceBuffer(nrows * ncol, device)
    check(lib().pfd_synth_d8(device, seed, nrow, ncol, tilt, white, nodata_pct, row0, nrows, C.c_void_p(buf.addr)))
    return buf


def synth_mosaic_device(base: np.ndarray, nrow, ncol, device=0):
    """A host base raster tiled over nrow x ncol cells in HBM (nodata frame per copy): pfd_synth_mosaic."""
    base = np.ascontiguousarray(base, dtype=np.uint8)
    buf = DeviceBuffer(int(nrow) * int(ncol), device)
    check(lib().pfd_synth_mosaic(device, ptr(base), C.c_int64(base.shape[0]), C.c_int64(base.shape[1]), C.c_int64(nrow),
                                 C.c_int64(ncol), C.c_void_p(buf.addr)))
    return buf


def synth_elev_device(nrow, ncol, seed=0, tilt=1 << 26, white=2, nodata_pct=0, row0=0, nrows=None, device=0):
    nrows = nrow - row0 if nrows is None else nrows
    buf = DeviceBuffer(nrows * ncol * 4, device)
    check(lib().pfd_synth_elev_f32(device, seed, nrow, ncol, tilt, white, nodata_pct, row0, nrows, C.c_void_p(buf.addr)))
    return buf


def synth_weights_device(n, seed=1, i0=0, device=0):
    buf = DeviceBuffer(n * 4, device)
    check(lib().pfd_synth_weights_f32(device, seed, i0, n, C.c_void_p(buf.addr)))
    return buf
