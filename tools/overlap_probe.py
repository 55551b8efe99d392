"""Do two passes on two streams overlap?  N passes one after the other vs two host threads with their own handles
(the library's calls synchronise their own stream only; ctypes releases the GIL)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyflwdir_amd import _hip
L = _hip.lib()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
reps = 6
bufs = [_hip.synth_d8_device(size, size, seed=s) for s in (0, 1)]
outs = [_hip.DeviceBuffer(size * size * 4) for _ in range(2)]
def one(k):
    h = _hip.RasterHandle(bufs[k], size, size, memspace=_hip.PFD_DEVICE, deferred=True)
    h.upstream_area_cell(out=outs[k], memspace=_hip.PFD_DEVICE)
    h.close()
for k in (0, 1): one(k)
_hip.check(L.pfd_device_synchronize(0))
t0 = time.perf_counter()
for i in range(reps): one(i & 1)
_hip.check(L.pfd_device_synchronize(0))
t_seq = time.perf_counter() - t0
def worker(k):
    for _ in range(reps // 2): one(k)
ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
_hip.check(L.pfd_device_synchronize(0))
t_par = time.perf_counter() - t0
print(f"{size}x{size}: {reps} passes sequential {t_seq*1e3/reps:.2f} ms/pass, two threads {t_par*1e3/reps:.2f} ms/pass ({t_seq/t_par:.2f}x)")
