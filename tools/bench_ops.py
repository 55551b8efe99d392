"""Timing of the other hot-path operations (level engine) on a synthetic raster, device-resident."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pyflwdir_amd import _hip
L = _hip.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d8 = _hip.synth_d8_device(n, n, seed=0)
w = _hip.synth_weights_device(n * n, seed=1)
out4 = _hip.DeviceBuffer(n * n * 4)
out1 = _hip.DeviceBuffer(n * n)
def sync(): _hip.check(L.pfd_device_synchronize(0))
h = _hip.RasterHandle(d8, n, n, device=0, memspace=_hip.PFD_DEVICE)
h.set_profiling(True)
sync(); t0 = time.perf_counter(); h.order_cells(); sync(); t1 = time.perf_counter()
print(f"order_cells   {1e3*(t1-t0):9.2f} ms  levels={h.info()['n_levels']}", h.last_timing())
for name, fn in [("accuflux_f32", lambda: h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out4, memspace=_hip.PFD_DEVICE)),
                 ("strahler", lambda: h.strahler(None, out=out1, memspace=_hip.PFD_DEVICE)),
                 ("count_levels", lambda: h.upstream_area_cell(out=out4, memspace=_hip.PFD_DEVICE, engine="levels"))]:
    fn(); sync(); t0 = time.perf_counter(); fn(); sync(); t1 = time.perf_counter()
    print(f"{name:13s} {1e3*(t1-t0):9.2f} ms  {n*n/(t1-t0)/1e6:9.1f} Mcells/s")
