"""Where do the one-off stalls come from?  hipMalloc of a size class the cache has not seen (through pfd_malloc: the
library's caching allocator), and pageable host <-> device copies out of / into a FRESH host array.

    python tools/alloc_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip
L = _hip.lib()
sync = lambda: _hip.check(L.pfd_device_synchronize(0))
_hip.DeviceBuffer(1024).free(); sync()
def t_alloc(nbytes):
    sync(); t0 = time.perf_counter(); b = _hip.DeviceBuffer(nbytes); sync(); t1 = time.perf_counter()
    return b, 1e3 * (t1 - t0)
print("raw hipMalloc / hipFree (pfd_malloc does not go through the cache): first call of a size, free, same size again:")
for mb in (2, 14, 64, 258, 1026, 2050, 4098, 8194, 16386, 32770):
    b, cold = t_alloc(mb << 20)
    t0 = time.perf_counter(); b.free(); sync(); tf = 1e3 * (time.perf_counter() - t0)
    b, warm = t_alloc(mb << 20)
    b.free()
    print(f"  {mb:6d} MiB: first {cold:9.3f} ms   free {tf:7.3f} ms   again {warm:7.3f} ms")
# many live blocks at once, like a fresh process that builds a handle + plan: ten new classes in a row
sync(); t0 = time.perf_counter()
bs = [_hip.DeviceBuffer((3000 + 2 * k) << 20) for k in range(10)]
sync(); print(f"ten new ~3 GiB classes in a row: {1e3 * (time.perf_counter() - t0):.1f} ms")
for b in bs: b.free()
dev = _hip.DeviceBuffer(64 << 20)
print("pageable host copies of 11.5 MB (the gathered boundary records of 8 blocks at 90000 columns):")
for kind in ("fresh", "fresh", "reused", "reused"):
    if kind == "fresh" or "arr" not in dir():
        arr = np.ones(4 * 90000 * 8, np.uint32)
    sync(); t0 = time.perf_counter(); dev.upload(arr); sync(); th = 1e3 * (time.perf_counter() - t0)
    out_t0 = time.perf_counter(); got = dev.download(np.uint32, (4 * 90000 * 8,)); sync(); td = 1e3 * (time.perf_counter() - out_t0)
    print(f"  {kind:7s} host array: H2D {th:7.3f} ms   D2H into a fresh array {td:7.3f} ms")
