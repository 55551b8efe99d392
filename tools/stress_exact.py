"""Stress of the exact-order engine against the level engine on random larger rasters (device generator),
all sweep operations; prints one line per raster.   python tools/stress_exact.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("PFD_ENABLE_KNOBS", "1")  # PFD_EXACT_LEVELS below is a test-only switch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    nrow, ncol = int(rng.integers(200, 9000)), int(rng.integers(200, 9000))
    if rng.random() < 0.15:
        nrow, ncol = int(rng.integers(20000, 30001)), int(rng.integers(20000, 30001))
    kw = dict(seed=int(rng.integers(0, 1000)), tilt=int(rng.choice([1 << 26, 100000, 3000, 300])), white=2,
              nodata_pct=int(rng.choice([0, 0, 10, 40])))
    n = nrow * ncol
    print(f"    next: {nrow}x{ncol} {kw}", flush=True)
    d8 = _hip.synth_d8_device(nrow, ncol, **kw)
    w = _hip.synth_weights_device(n, seed=int(rng.integers(0, 99)))
    elev = _hip.synth_elev_device(nrow, ncol, **kw)
    res = {}
    keep = {}
    for engine in ("exact", "levels", "exact2", "levels2"):
        if engine.startswith("levels"):
            os.environ["PFD_EXACT_LEVELS"] = "1"
        else:
            os.environ.pop("PFD_EXACT_LEVELS", None)
        sys.stderr.write(f"[handle] {it} {engine}\n"); sys.stderr.flush()
        h = _hip.RasterHandle(d8, nrow, ncol, memspace=_hip.PFD_DEVICE)
        o4, o1, o8 = _hip.DeviceBuffer(n * 4), _hip.DeviceBuffer(n), _hip.DeviceBuffer(n * 8)
        print('     ', engine, 'up', flush=True)
        h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, has_nodata=1, out=o4, memspace=_hip.PFD_DEVICE)
        a = _hip.checksum_i32(o4, n)
        h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, has_nodata=1, direction=_hip.PFD_DOWN, out=o4, memspace=_hip.PFD_DEVICE)
        b = _hip.checksum_i32(o4, n)
        print('      down ok', flush=True)
        h.strahler(None, out=o1, memspace=_hip.PFD_DEVICE)
        s = _hip.checksum_i32(o1, n // 4)
        keep[engine] = o1
        print('      strahler ok', flush=True)
        h.hand(o1, elev, _hip.PFD_F32, out=o8, memspace=_hip.PFD_DEVICE)
        hd = _hip.checksum_i32(o8, 2 * n)
        print('      hand ok', flush=True)
        h.stream_distance(None, None, out=o4, memspace=_hip.PFD_DEVICE)
        sd = _hip.checksum_i32(o4, n)
        res[engine] = (a, b, s, hd, sd)
        h.close()
        for bb in (o4, o8):
            bb.free()
    ok = res["exact"] == res["levels"] == res["exact2"] == res["levels2"]
    print(f"{it:3d} {nrow}x{ncol} {kw} {'ok' if ok else 'MISMATCH ' + str(res)}", flush=True)
    sys.stderr.write(f"[result] {it} {'ok' if ok else 'MISMATCH'}\n"); sys.stderr.flush()
    for bb in (d8, w, elev):
        bb.free()
    if not ok:
        ref = keep["levels"].download(np.uint8, (nrow, ncol))
        for e in ("exact", "exact2", "levels2"):
            got = keep[e].download(np.uint8, (nrow, ncol))
            bad = np.argwhere(got != ref)
            print("   strahler", e, "differs at", len(bad), "cells; first", bad[:8].tolist(), "last", bad[-3:].tolist(),
                  "tiles", sorted(set((int(r) // 64, int(c) // 64) for r, c in bad[:2000]))[:12], flush=True)
            if len(bad):
                print("   values got/ref", got[tuple(bad[:8].T)].tolist(), ref[tuple(bad[:8].T)].tolist(), flush=True)
        sys.exit(1)
    for bb in keep.values():
        bb.free()
print("done", it)
