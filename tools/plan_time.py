"""Build time of the exact-order plan on fresh handles of a synthetic raster, allocator warm.

    python tools/plan_time.py NROW NCOL [reps] [nodata_pct] [tilt]

Prints the `exact_plan` segment (HIP events) and the wall time of the first float32 accuflux of each handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyflwdir_amd import _hip
if os.environ.get("PFD_TOOL_RESERVE_GIB"):  # one arena for the working buffers (pfd_reserve): no hipMalloc while timing
    from pyflwdir_amd import _hip as _h0
    _h0.reserve(int(float(os.environ["PFD_TOOL_RESERVE_GIB"]) * 2**30))
L = _hip.lib()
nrow, ncol = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tilt = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 26
n = nrow * ncol
d8 = _hip.synth_d8_device(nrow, ncol, seed=0, tilt=tilt, white=2, nodata_pct=nd)
w = _hip.synth_weights_device(n, seed=1)
out = _hip.DeviceBuffer(n * 4)
def sync(): _hip.check(L.pfd_device_synchronize(0))
plan, wall = [], []
for r in range(reps + 1):  # (the first handle warms the allocator and is not reported)
    h = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE)
    h.upstream_area_cell(out=out, memspace=_hip.PFD_DEVICE)  # normalises the codes: not part of the plan
    h.set_profiling(True)
    sync(); t0 = time.perf_counter()
    h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, out=out, memspace=_hip.PFD_DEVICE)
    sync(); t1 = time.perf_counter()
    seg = {s["name"]: s["ms"] for s in h.last_timing()}
    if r == 0:
        first = round(1e3 * (t1 - t0), 2)  # this process's very first call: the allocator's (or the arena's) cold path
    if r:
        plan.append(round(seg.get("exact_plan", float("nan")), 2)); wall.append(round(1e3 * (t1 - t0), 2))
    h.close()
print(f"{nrow}x{ncol} nodata_pct={nd} tilt={tilt}: exact_plan {plan} ms; first accuflux (plan + sweep) {wall} ms; the process's "
      f"first handle {first} ms = {first / min(wall):.2f} x the warm-allocator first call; allocator {_hip.alloc_stats()}")
