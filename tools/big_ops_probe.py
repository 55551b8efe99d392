"""The order-sensitive operations of the front end beyond 2**32 - 2 cells (row blocks held or streamed by one process,
pyflwdir_amd/dist.py) at SIZE x SIZE, with the time spent in device allocation / uploads / downloads of the blocks.

    python tools/big_ops_probe.py [SIZE] [op ...]      ops: strahler classic distance accuflux km2"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip, dist
from pyflwdir_amd._affine import Affine

size = int(sys.argv[1]) if len(sys.argv) > 1 else 90000
ops = sys.argv[2:] or ["strahler", "classic", "distance", "accuflux", "km2"]
_hip.reserve(int(float(os.environ.get("PFD_TOOL_RESERVE_GIB", "100")) * 2**30))
buf = _hip.synth_d8_device(size, size, seed=0)
d8 = buf.download(np.uint8, (size, size))
buf.free()
res = 1.0 / 1200.0
flw = pyflwdir.from_array(d8, ftype="d8", transform=Affine(res, 0.0, 5.0, 0.0, -res, 80.0), latlon=True, cache=False)
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


timed(_hip.DeviceBuffer, "__init__", "DeviceBuffer allocation")
timed(_hip.DeviceBuffer, "upload", "DeviceBuffer.upload")
timed(_hip.DeviceBuffer, "download", "DeviceBuffer.download")
timed(_hip.DeviceBuffer, "free", "DeviceBuffer.free")
timed(_hip.RasterHandle, "__init__", "RasterHandle.__init__")
timed(dist, "_concat_rows", "_concat_rows")
timed(dist._StreamedBlock, "_call", "streamed block: build + sweep + result")


def run(tag, fn):
    T.clear(); N.clear()
    t0 = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t0
    print(f"{size}x{size} {tag}: {dt:.1f} s  -> {out}  sweeps per block {list(dist.LAST_SWEEPS)}", flush=True)
    print("    " + "; ".join(f"{k} {v:.1f} s / {N[k]}" for k, v in T.items()), flush=True)


if "strahler" in ops:
    run("stream_order strahler", lambda: int(flw.stream_order().max()))
if "classic" in ops:
    run("stream_order classic", lambda: int(flw.stream_order(type="classic").max()))
if "distance" in ops:
    run("stream_distance (cells)", lambda: int(flw.stream_distance(unit="cell").max()))
if "accuflux" in ops:
    w = np.ones(flw.shape, np.float32)
    run("accuflux float32", lambda: float(flw.accuflux(w).max()))
    del w
if "km2" in ops:
    run("upstream_area km2 exact", lambda: float(flw.upstream_area("km2").max()))
