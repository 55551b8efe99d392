"""Does the Python front end work beyond 2**32 - 2 cells?  A 66000 x 66000 raster (4.36e9 cells) through
pyflwdir_amd.from_array: construction, upstream_area, and the operations that run in row blocks there.

    python tools/big_frontend_probe.py [SIZE]"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip
size = int(sys.argv[1]) if len(sys.argv) > 1 else 66000
_hip.reserve(100 << 30)
buf = _hip.synth_d8_device(size, size, seed=0)
d8 = buf.download(np.uint8, (size, size))
buf.free()
print(f"{size}x{size} = {d8.size / 1e9:.2f} Gcells on the host", flush=True)
def step(name, fn):
    t0 = time.perf_counter()
    try:
        r = fn()
        print(f"  {name}: ok in {time.perf_counter() - t0:.1f} s -> {r}", flush=True)
        return True
    except Exception as exc:  # noqa: BLE001
        print(f"  {name}: {type(exc).__name__}: {str(exc)[:200]}", flush=True)
        return False
flw = None
def make():
    global flw
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    return (flw.shape, flw.idxs_pit.dtype, flw.idxs_pit.size)
if step("from_array", make):
    upa = None
    def f_upa():
        global upa
        upa = flw.upstream_area()
        return (upa.dtype, int(upa.max()))
    step("upstream_area", f_upa)
    step("stream_order strahler", lambda: int(flw.stream_order().max()))
    step("stream_order classic", lambda: int(flw.stream_order(type="classic").max()))
    step("n_upstream", lambda: np.bincount(flw.n_upstream.ravel()[:10_000_000] + 9)[:3].tolist())
    step("idxs_ds dtype", lambda: (flw.idxs_ds.dtype, int(flw.idxs_ds[-1])))
    step("main_upstream", lambda: (lambda mu: (mu.dtype, int(mu.max())))(flw.main_upstream(upa)))
    step("upstream_sum", lambda: float(flw.upstream_sum(np.ones(flw.shape, np.float32)).max()))
    step("add_pits + idxs_pit", lambda: (flw.add_pits(idxs=np.array([int(np.argmax(upa)) - 5 * size])), flw.idxs_pit.size)[1])
    step("upstream_area after add_pits", lambda: int(flw.upstream_area().max()))
    step("floodplains", lambda: np.bincount(flw.floodplains(np.zeros(flw.shape, np.float32), uparea=upa.astype(np.float32), upa_min=1e5).ravel() + 1).tolist())
    # the operations that needed 32-bit cell indices until the end of round 5: csrc/order64.hip, 64-bit snap walks, ucat_area
    step("rank", lambda: (flw.rank.dtype, int(flw.rank.max())))
    step("idxs_seq (walk)", lambda: (flw.idxs_seq.dtype, int(flw.idxs_seq.size), int(flw.idxs_seq[0])))
    top = np.argsort(upa.ravel()[:: 4097])[-500:].astype(np.int64) * 4097
    step("ucat_area cell, 500 outlets", lambda: [int(v) for v in np.sort(flw.ucat_area(top, unit="cell")[1])[-2:]])
    step("ucat_area km2, 500 outlets", lambda: [float(v) for v in np.sort(flw.ucat_area(top, unit="km2")[1])[-2:]])
    step("snap down, 1000 points", lambda: float(flw.snap(idxs=np.arange(1000, dtype=np.int64) * 4000003 + 7, mask=upa > 1000)[1].max()))
    step("stream_distance (cells)", lambda: int(flw.stream_distance(unit="cell").max()))
    step("hand", lambda: float(np.nanmax(flw.hand(upa > 1000, np.zeros(flw.shape, np.float32)))))
