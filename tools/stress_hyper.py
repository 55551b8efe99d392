"""One-off stress on rasters that span several hypertiles (level 3 in LDS + level 4): counts, ranks, labels."""
import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from pyflwdir_amd import _hip, dist
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for it in range(N):
    shape = (int(rng.integers(2049, 5200)), int(rng.integers(2049, 5200)))
    kw = dict(tilt=int(rng.choice([1 << 26, 100000, 3000000])), white=2, nodata_pct=int(rng.choice([0, 10, 30])))
    seed = int(rng.integers(0, 1 << 30))
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    n = d8.size
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    exp = O.upstream_area_cell(d8)[0].ravel()
    tag = f"it {it} {shape} seed {seed} {kw}"
    h = _hip.RasterHandle(d8, shape[0], shape[1], deferred=bool(rng.integers(0, 2)))
    ok = np.array_equal(h.upstream_area_cell(), exp)
    ok_rank = np.array_equal(h.rank(), O.rank(idxs_ds)[0])
    oidx = np.unique(rng.integers(0, n, 50)).astype(np.int64)
    oids = (np.arange(oidx.size) + 1).astype(np.uint32)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    ok_bas = np.array_equal(h.basins(oidx, oids), O.basins(idxs_ds, oidx.astype(idxs_ds.dtype), seq, oids))
    h.close()
    nb = int(rng.integers(2, 5))
    ok_blk = np.array_equal(dist.upstream_area_blocks(d8, nb, deferred=True).ravel(), exp)
    print(tag, "uparea", ok, "rank", ok_rank, "basins", ok_bas, f"blocks{nb}", ok_blk, flush=True)
    if not (ok and ok_rank and ok_bas and ok_blk):
        sys.exit(1)
print("stress hyper:", N, "cases ok")
