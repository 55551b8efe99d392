import sys, time
sys.path.insert(0, '.')
import numpy as np
from pyflwdir_amd import _hip
L=_hip.lib()
n=10000
d8=_hip.synth_d8_device(n,n,seed=0)
out=_hip.DeviceBuffer(n*n*4)
def sync(): _hip.check(L.pfd_device_synchronize(0))
for it in range(4):
    sync(); t0=time.perf_counter()
    h=_hip.RasterHandle(d8,n,n,device=0,memspace=_hip.PFD_DEVICE)
    sync(); t1=time.perf_counter()
    h.upstream_area_cell(out=out, memspace=_hip.PFD_DEVICE)
    sync(); t2=time.perf_counter()
    h.close()
    sync(); t3=time.perf_counter()
    print(f"create {1e3*(t1-t0):.3f} ms  op {1e3*(t2-t1):.3f} ms  close {1e3*(t3-t2):.3f} ms")
h=_hip.RasterHandle(d8,n,n,device=0,memspace=_hip.PFD_DEVICE); h.set_profiling(True)
h.upstream_area_cell(out=out, memspace=_hip.PFD_DEVICE); print(h.last_timing())
