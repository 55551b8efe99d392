"""In-process multi-block timing on ONE GPU (the phases of the multi-GPU protocol, serially).

    python tools/bench_blocks.py ROWS_PER_BLOCK NBLOCKS [NCOL]      (NCOL defaults to ROWS_PER_BLOCK)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyflwdir_amd import _hip, dist
L = _hip.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else n
bufs = []
for b in range(nb):
    top, bot = dist.halo_of(b, nb)
    bufs.append(_hip.synth_d8_device(n * nb, ncol, seed=0, row0=b * n - top, nrows=n + top + bot))
outs = [_hip.DeviceBuffer(n * ncol * 4) for _ in range(nb)]
for it in range(3):
    _hip.check(L.pfd_device_synchronize(0)); t0 = time.perf_counter()
    hs = [_hip.RasterHandle(bufs[b], n, ncol, device=0, memspace=_hip.PFD_DEVICE, halo=dist.halo_of(b, nb), deferred=True) for b in range(nb)]
    for h in hs: h.set_profiling(True)
    _hip.upstream_area_cell_blocks(hs, outs=outs, memspace=_hip.PFD_DEVICE)
    _hip.check(L.pfd_device_synchronize(0)); t1 = time.perf_counter()
    print(f"{nb} blocks of {n}x{ncol} on one GPU (serial): {1e3*(t1-t0):.2f} ms total;",
          [(s["name"], round(s["ms"], 3)) for s in hs[1].last_timing()])
    for h in hs: h.close()
