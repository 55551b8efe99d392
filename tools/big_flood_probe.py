"""floodplains beyond 2**32 - 2 cells (streamed row blocks, dist.floodplains_blocks) at SIZE x SIZE with a breakdown of where
the time goes: the front end's numpy preparation, block construction (uploads), sweeps, results.

    python tools/big_flood_probe.py [SIZE]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip, dist

size = int(sys.argv[1]) if len(sys.argv) > 1 else 90000
_hip.reserve(int(float(os.environ.get("PFD_TOOL_RESERVE_GIB", "100")) * 2**30))
buf = _hip.synth_d8_device(size, size, seed=0)
d8 = buf.download(np.uint8, (size, size))
buf.free()
flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
upa = flw.upstream_area().astype(np.float32)
elev = np.empty(flw.shape, np.float32)  # (touched pages, like a raster that was read from somewhere: np.zeros leaves them unmapped)
for r in range(0, size, 4096):
    elev[r:r + 4096] = 0
T = collections.defaultdict(float)
N = collections.Counter()


def timed(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[tag] += time.perf_counter() - t0
            N[tag] += 1
    setattr(obj, name, g)


timed(dist, "floodplains_blocks", "dist.floodplains_blocks (all)")
timed(dist._StreamedBlock, "_call", "  streamed block: build + sweep + result")
timed(dist._FloodBlock, "__init__", "    _FloodBlock.__init__ (uploads)")
timed(dist._FloodBlock, "_call", "    _FloodBlock._call (plan + sweep)")
timed(dist._FloodBlock, "result", "    _FloodBlock.result (download)")
timed(_hip.RasterHandle, "__init__", "    RasterHandle.__init__")
timed(_hip.DeviceBuffer, "__init__", "      DeviceBuffer.__init__ (allocation)")
timed(_hip.DeviceBuffer, "upload", "      DeviceBuffer.upload")
timed(_hip.DeviceBuffer, "free", "      DeviceBuffer.free")
timed(_hip.RasterHandle, "close", "      RasterHandle.close")
t0 = time.perf_counter()
out = flw.floodplains(elev, uparea=upa, upa_min=1e5)
total = time.perf_counter() - t0
print(f"{size}x{size}: floodplains {total:.1f} s  sweeps per block {dist.LAST_SWEEPS}  flags {np.bincount(out.ravel()[::97] + 1).tolist()} (sampled)")
for k, v in T.items():
    print(f"  {k}: {v:.1f} s in {N[k]} calls")
print(f"  allocator: {_hip.alloc_stats()}")
print(f"  front end outside dist.floodplains_blocks: {total - T['dist.floodplains_blocks (all)']:.1f} s")

# HAND through the same streamed row blocks (dist.hand_blocks)
if os.environ.get("PFD_PROBE_HAND", "1") == "1":
    T.clear(); N.clear()
    timed(dist, "hand_blocks", "dist.hand_blocks (all)")
    timed(dist, "_hand_inputs", "  _hand_inputs (dtype + finite checks)")
    timed(dist._StreamedHandBlock, "sweep", "  streamed block: handle + uploads + sweep + download")
    timed(_hip.RasterHandle, "hand_block", "    RasterHandle.hand_block")
    drain = upa > 1000
    t0 = time.perf_counter()
    hand = flw.hand(drain, elev)
    total = time.perf_counter() - t0
    print(f"{size}x{size}: hand {total:.1f} s  max {float(np.nanmax(hand[::97])):.1f} (sampled rows)")
    for k, v in T.items():
        print(f"  {k}: {v:.1f} s in {N[k]} calls")
    print(f"  front end outside dist.hand_blocks: {total - T['dist.hand_blocks (all)']:.1f} s")
