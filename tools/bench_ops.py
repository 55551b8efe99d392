"""Timing of the other hot-path operations on a synthetic raster, everything device-resident.

    python tools/bench_ops.py NROW NCOL [nodata_pct] [tilt]

Prints one line per operation: wall time of a complete call (HIP work + host launch loop) and
Mcells/s over all raster cells.  Used for the C3/C5-shaped measurements quoted in DESIGN.md."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip
if os.environ.get("PFD_TOOL_RESERVE_GIB"):  # one arena for the working buffers (pfd_reserve): no hipMalloc while timing
    from pyflwdir_amd import _hip as _h0
    _h0.reserve(int(float(os.environ["PFD_TOOL_RESERVE_GIB"]) * 2**30))
L = _hip.lib()
nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else nrow
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tilt = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 26
n = nrow * ncol
kw = dict(seed=0, tilt=tilt, white=2, nodata_pct=nd)
d8 = _hip.synth_d8_device(nrow, ncol, **kw)
def sync(): _hip.check(L.pfd_device_synchronize(0))
def timed(name, fn, reps=2):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(reps - 1): fn()
    sync(); t1 = time.perf_counter()
    dt = (t1 - t0) / max(1, reps - 1)
    print(f"{name:22s} {1e3*dt:10.2f} ms  {n/dt/1e6:10.1f} Mcells/s", flush=True)
sync(); t0 = time.perf_counter()
h = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE)
sync(); t1 = time.perf_counter()
print(f"raster {nrow}x{ncol} nodata_pct={nd} tilt={tilt}: create {1e3*(t1-t0):.2f} ms, {h.info()}")
out4 = _hip.DeviceBuffer(n * 4)
timed("upstream_area(cell)", lambda: h.upstream_area_cell(out=out4, memspace=_hip.PFD_DEVICE))
# (first use pays hipMalloc of the sort buffers — ~0.4 s at 30000^2; the caching allocator keeps
#  them, so time the ordering of a second handle)
h0 = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE); h0.order_cells(); h0.close(); del h0
sync(); t0 = time.perf_counter(); h.order_cells(); sync(); t1 = time.perf_counter()
info = h.info()
print(f"{'order_cells':22s} {1e3*(t1-t0):10.2f} ms  {n/(t1-t0)/1e6:10.1f} Mcells/s  levels={info['n_levels']} n_seq={info['n_seq']} "
      f"n_pits={info['n_pits']} bytes_held={info['bytes_held']/1e9:.2f} GB", flush=True)
w = _hip.synth_weights_device(n, seed=1)
# (the plan's temporaries are GB-sized too: a throw-away handle builds one first, so that the line below times the
#  plan build + sweep and not cold hipMallocs — 0.1 vs 0.45 s at 30000^2, depending on what the pool happens to hold)
h0 = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE)
h0.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out4, memspace=_hip.PFD_DEVICE); h0.close(); del h0
h.set_profiling(True)
sync(); t0 = time.perf_counter()
h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out4, memspace=_hip.PFD_DEVICE)
sync(); t1 = time.perf_counter()
print(f"{'first accuflux f32':22s} {1e3*(t1-t0):10.2f} ms  (builds the exact plan) segments", [(s['name'], round(s['ms'], 2), s['launches']) for s in h.last_timing()], flush=True)
h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out4, memspace=_hip.PFD_DEVICE)
print("   warm segments", [(s['name'], round(s['ms'], 2), s['launches']) for s in h.last_timing()], flush=True)
h.set_profiling(False)
timed("accuflux f32 (up)", lambda: h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out4, memspace=_hip.PFD_DEVICE))
timed("accuflux f32 (down)", lambda: h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, direction=_hip.PFD_DOWN, out=out4, memspace=_hip.PFD_DEVICE))
wi = _hip.DeviceBuffer(n * 4)
wi.upload(np.random.default_rng(0).integers(0, 20, n, dtype=np.int32)) if n <= 200_000_000 else None
if n <= 200_000_000:
    timed("accuflux i32 (up, tiled)", lambda: h.accuflux(wi, _hip.PFD_I32, nodata_i=-9999, out=out4, memspace=_hip.PFD_DEVICE))
del wi
out1 = _hip.DeviceBuffer(n)
timed("strahler", lambda: h.strahler(None, out=out1, memspace=_hip.PFD_DEVICE))
# basins from 1000 outlets: the pits + evenly spread cells (ids 1..k)
k = 1000
idxs = (np.arange(k, dtype=np.int64) * (n // k) + ncol // 2) % n
ids = np.arange(1, k + 1, dtype=np.uint32)
timed("basins (1000 outlets)", lambda: h.basins(idxs, ids, out=out4, memspace=_hip.PFD_DEVICE))
del w
elev = _hip.synth_elev_device(nrow, ncol, **kw)
out8 = _hip.DeviceBuffer(n * 8)
drain = out1  # strahler order as a stand-in stream mask (value 1 = drain: the headwater cells... any uint8 works)
timed("hand f32->f64", lambda: h.hand(drain, elev, _hip.PFD_F32, out=out8, memspace=_hip.PFD_DEVICE))
