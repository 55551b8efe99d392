"""One `rank` + one `idxs_seq` with 64-bit cell indices (csrc/order64.hip) on a SIZE x SIZE raster generated on the device — the command
tools/prof_cmd.sh profiles for profiles/r05_order64_kernel_stats.txt.

    python tools/order64_run.py [SIZE]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyflwdir_amd import _hip

size = int(sys.argv[1]) if len(sys.argv) > 1 else 66000
_hip.reserve(120 << 30)
d8 = _hip.synth_d8_device(size, size, seed=0)
h = _hip.RasterHandle(d8, size, size, device=0, memspace=_hip.PFD_DEVICE)
assert h.wide_cells()
t0 = time.perf_counter()
rank = h.rank()
t1 = time.perf_counter()
seq = h.idxs_seq(np.int64)
t2 = time.perf_counter()
print(f"{size} x {size}: rank {t1 - t0:.2f} s (max {int(rank.max())}), idxs_seq {t2 - t1:.2f} s ({seq.size} entries) incl. the copies to the host", flush=True)
