"""Per-block cost of the multi-GPU protocol with every block ALONE on the GPU (what one GPU of an N-GPU job sees):
begin() of each block timed separately (phase A + record to the host), then finish() of each block (records from the
host, interface solve, second exit solve, final pass).  tools/bench_blocks.py issues all blocks concurrently instead,
which hides latencies that a real rank pays.

    python tools/bench_blocks_isolated.py ROWS_PER_BLOCK NBLOCKS [NCOL]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip, dist
if os.environ.get("PFD_TOOL_RESERVE_GIB"):  # one arena for the working buffers (pfd_reserve): no hipMalloc while timing
    from pyflwdir_amd import _hip as _h0
    _h0.reserve(int(float(os.environ["PFD_TOOL_RESERVE_GIB"]) * 2**30))
L = _hip.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11250
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else n
PHASES = os.environ.get("PFD_BLOCK_PHASES") == "1"  # HIP-event segments of the middle block (costs a little)
bufs = []
for b in range(nb):
    top, bot = dist.halo_of(b, nb)
    bufs.append(_hip.synth_d8_device(n * nb, ncol, seed=0, row0=b * n - top, nrows=n + top + bot))
outs = [_hip.DeviceBuffer(n * ncol * 4) for _ in range(nb)]
sync = lambda: _hip.check(L.pfd_device_synchronize(0))
for it in range(3):
    ta, tb, hs, recs = [], [], [], []
    for b in range(nb):
        sync(); t0 = time.perf_counter()
        h = _hip.RasterHandle(bufs[b], n, ncol, device=0, memspace=_hip.PFD_DEVICE, halo=dist.halo_of(b, nb), deferred=True)
        if PHASES and b == nb // 2: h.set_profiling(True)
        _, rec = _hip.upstream_area_cell_begin(h, out=outs[b], memspace=_hip.PFD_DEVICE)
        sync(); ta.append(1e3 * (time.perf_counter() - t0))
        if PHASES and b == nb // 2: print("  begin :", [(s["name"], round(s["ms"], 3), s["launches"]) for s in h.last_timing()])
        hs.append(h); recs.append(rec)
    allrec = np.ascontiguousarray(np.stack(recs))
    for b in range(nb):
        sync(); t0 = time.perf_counter()
        ok = _hip.upstream_area_cell_finish(hs[b], allrec, nb, b)
        sync(); tb.append(1e3 * (time.perf_counter() - t0))
        if PHASES and b == nb // 2: print("  finish:", [(s["name"], round(s["ms"], 3), s["launches"]) for s in hs[b].last_timing()])
        assert ok
    for h in hs: h.close()
    tot = [x + y for x, y in zip(ta, tb)]
    print(f"{nb} blocks of {n}x{ncol}, one at a time: begin {np.round(ta, 2).tolist()} finish {np.round(tb, 2).tolist()} ms; "
          f"per block max {max(tot):.2f} mean {np.mean(tot):.2f} ms")
