"""tools/api_probe.py SIZE [REPS] — the numpy-in / numpy-out path of the front end at SIZE x SIZE: from_array and
upstream_area() wall times with the library's own upload / download clocks (pfd_transfer_stats), in a process that does
NOT reserve explicitly (the front end's default arena is what runs).  VERDICT r05 item 3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import pyflwdir_amd as pyflwdir  # noqa: E402
from pyflwdir_amd import _hip  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t = time.perf_counter()
buf = _hip.synth_d8_device(size, size, seed=0, tilt=1 << 26, white=2, nodata_pct=0)
d8 = buf.download(np.uint8, (size, size))
buf.free()
_hip.check(_hip.lib().pfd_trim(0))
print(f"{size}x{size} = {d8.size / 1e9:.2f} Gcells on the host ({time.perf_counter() - t:.1f} s to make); "
      f"PFD_PREFAULT_THREADS={os.environ.get('PFD_PREFAULT_THREADS', 'default')}", flush=True)
for rep in range(reps):
    _hip.transfer_stats(reset=True)
    t0 = time.perf_counter()
    flw = pyflwdir.from_array(d8, ftype="d8")
    t1 = time.perf_counter()
    a = _hip.transfer_stats(reset=True)
    upa = flw.upstream_area()
    t2 = time.perf_counter()
    b = _hip.transfer_stats(reset=True)
    print(f"  rep {rep}: from_array {t1 - t0:.3f} s (h2d {a['h2d_bytes'] / 1e9:.2f} GB in {a['h2d_ms']:.0f} ms = "
          f"{a['h2d_bytes'] / max(a['h2d_ms'], 1e-9) / 1e6:.1f} GB/s; d2h {a['d2h_ms']:.0f} ms); upstream_area {t2 - t1:.3f} s "
          f"(d2h {b['d2h_bytes'] / 1e9:.2f} GB in {b['d2h_ms']:.0f} ms = {b['d2h_bytes'] / max(b['d2h_ms'], 1e-9) / 1e6:.1f} GB/s, "
          f"prefault {b['prefault_ms']:.0f} ms, device + rest {(t2 - t1) * 1e3 - b['d2h_ms']:.0f} ms); max {int(upa.max())}; "
          f"allocator {_hip.alloc_stats()}", flush=True)
    del upa, flw
