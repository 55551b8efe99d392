"""One-off stress of the row-block protocol: random rasters, 2-16 blocks (down to one row per block)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from pyflwdir_amd import dist
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for it in range(N):
    nb = int(rng.integers(2, 17))
    nrow = int(rng.integers(nb, 2600))
    ncol = int(rng.integers(3, 900))
    kw = dict(tilt=int(rng.choice([1 << 26, 100000, 3000000])), white=2, nodata_pct=int(rng.choice([0, 10, 40])))
    seed = int(rng.integers(0, 1 << 30))
    d8 = O.synth_d8(nrow, ncol, seed=seed, **kw)
    exp = O.upstream_area_cell(d8)[0]
    got = dist.upstream_area_blocks(d8, nb, deferred=bool(rng.integers(0, 2)))
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        print("FAIL it", it, (nrow, ncol), "nb", nb, "seed", seed, kw, "nbad", len(bad), "rows", bad[:, 0].min(), bad[:, 0].max(), flush=True)
        sys.exit(1)
print("stress blocks:", N, "cases ok")
