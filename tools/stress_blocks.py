"""Randomised stress of the row-block sweeps (exact-order engine on blocks, seeded halo rows): many rasters of random
shape / terrain / block count; float32 accuflux (both directions), the Strahler order and HAND over row blocks must
equal the whole raster on one handle bit for bit, and pass their local-equation checks.

    python tools/stress_blocks.py [SECONDS] [SEED] [MAX_SIDE]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip, dist

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 2600
t_end, n, cells = time.time() + budget, 0, 0
while time.time() < t_end:
    nrow, ncol = int(rng.integers(40, max_side)), int(rng.integers(40, max_side))
    kw = dict(seed=int(rng.integers(1, 1 << 30)), tilt=int(rng.choice([3000, 100000, 1 << 20, 1 << 26])), white=2,
              nodata_pct=int(rng.choice([0, 0, 5, 20, 40])))
    nb = int(rng.integers(2, min(9, nrow // 4)))
    buf = _hip.synth_d8_device(nrow, ncol, **kw)
    d8 = buf.download(np.uint8, (nrow, ncol)); buf.free()
    ebuf = _hip.synth_elev_device(nrow, ncol, **kw)
    elev = ebuf.download(np.float32, (nrow, ncol)); ebuf.free()
    try:
        flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    except ValueError:  # (no pits: all nodata)
        continue
    data = (rng.random((nrow, ncol)) * 4).astype(np.float32)
    data[rng.random((nrow, ncol)) < 0.002] = -9999
    upa = flw.upstream_area()
    drain = upa > max(2, int(np.percentile(upa[upa > 0], rng.choice([80, 97, 99.9]))))
    tag = f"{nrow}x{ncol} nb={nb} {kw}"
    for direction in ("up", "down"):
        got, _, bad = dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), verify=True, direction=direction)
        assert bad == 0, (tag, direction, bad)
        assert np.array_equal(got.view(np.uint32), flw.accuflux(data, direction=direction).view(np.uint32)), (tag, direction)
    got, _, bad = dist.strahler_blocks(d8, nb, verify=True)
    assert bad == 0 and np.array_equal(got, flw.stream_order()), tag
    got, _ = dist.hand_blocks(d8, nb, drain, elev)
    assert np.array_equal(got.view(np.uint64), flw.hand(drain, elev).view(np.uint64)), tag
    n += 1; cells += nrow * ncol
print(f"stress_blocks: {n} rasters, {cells/1e6:.0f} Mcells, all row-block results == whole raster")
