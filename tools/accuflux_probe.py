"""Timing of float32 accuflux (up) on a synthetic raster, first call and warm calls, with the phase
segments; PROBE_OP=up|down|strahler|hand picks the operation.

    python tools/accuflux_probe.py NROW [NCOL [nodata_pct [tilt]]]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip
if os.environ.get("PFD_TOOL_RESERVE_GIB"):  # one arena for the working buffers (pfd_reserve): no hipMalloc while timing
    from pyflwdir_amd import _hip as _h0
    _h0.reserve(int(float(os.environ["PFD_TOOL_RESERVE_GIB"]) * 2**30))
L = _hip.lib()
nrow = int(sys.argv[1]); ncol = int(sys.argv[2]) if len(sys.argv) > 2 else nrow
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tilt = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 26
n = nrow * ncol
d8 = _hip.synth_d8_device(nrow, ncol, seed=0, tilt=tilt, white=2, nodata_pct=nd)
def sync(): _hip.check(L.pfd_device_synchronize(0))
h = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE)
h.set_profiling(True)
w = _hip.synth_weights_device(n, seed=1)
out = _hip.DeviceBuffer(n * 8)
op = os.environ.get("PROBE_OP", "up")
if op == "hand":
    so = _hip.DeviceBuffer(n)
    h.strahler(None, out=so, memspace=_hip.PFD_DEVICE)
    elev = _hip.synth_elev_device(nrow, ncol, seed=0, tilt=tilt, white=2, nodata_pct=nd)
for it in range(3):
    sync(); t0 = time.perf_counter()
    if op == "up":
        h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out, memspace=_hip.PFD_DEVICE)
    elif op == "down":
        h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, direction=_hip.PFD_DOWN, out=out, memspace=_hip.PFD_DEVICE)
    elif op == "strahler":
        h.strahler(None, out=out, memspace=_hip.PFD_DEVICE)
    elif op == "hand":
        h.hand(so, elev, _hip.PFD_F32, out=out, memspace=_hip.PFD_DEVICE)
    sync(); t1 = time.perf_counter()
    print(it, round(1e3 * (t1 - t0), 2), "ms", [(s['name'], round(s['ms'], 2)) for s in h.last_timing()], flush=True)
