"""hipMalloc of blocks above 16 GiB: one-off or per call, and what does it scale with?  Each line is a fresh process.
    python tools/alloc_probe_big.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from pyflwdir_amd import _hip
L = _hip.lib()
sync = lambda: _hip.check(L.pfd_device_synchronize(0))
_hip.DeviceBuffer(1024).free(); sync()
out = []
for g in [float(x) for x in sys.argv[1].split(",")]:
    sync(); t0 = time.perf_counter(); b = _hip.DeviceBuffer(int(g * (1 << 30))); sync(); t1 = time.perf_counter()
    keep = g < 0
    t2 = time.perf_counter(); b.free(); sync(); t3 = time.perf_counter()
    out.append(f"{g:g} GiB: malloc {1e3 * (t1 - t0):.1f} ms, free {1e3 * (t3 - t2):.1f} ms")
print("; ".join(out))
''' % ROOT
for seq in ("17", "20", "24", "32", "48", "64", "100", "32,32,33,40,24", "20,20,21", "64,32,48", "17,17.5,18"):
    r = subprocess.run([sys.executable, "-c", CHILD, seq], capture_output=True, text=True)
    print(f"[{seq}] " + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
