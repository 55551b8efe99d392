"""Pageable host -> device copies (DeviceBuffer.upload = one hipMemcpy) split over host threads: GB/s for 1, 2, 4, 8 threads.

    python tools/h2d_threads_probe.py [GiB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from pyflwdir_amd import _hip

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(gib * 2**30)
src = np.ones(n, np.uint8)  # (touched pages)
buf = _hip.DeviceBuffer(n, 0)
lib = _hip.lib()
for T in (1, 2, 4, 8, 1):
    step = -(-n // T)
    def part(t):
        o = t * step
        m = min(step, n - o)
        _hip.check(lib.pfd_memcpy_h2d(0, C.c_void_p(buf.addr + o), C.c_void_p(src.ctypes.data + o), C.c_size_t(m)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(part, range(T)))
    dt = time.perf_counter() - t0
    print(f"  {T} thread(s): {n / dt / 1e9:.1f} GB/s", flush=True)
buf.free()
