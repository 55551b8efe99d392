"""One-off stress on larger synthetic rasters (hundreds of tiles, several supertiles): river / rough
regimes with nodata, injected cycles and random outlets, every engine and operation against the oracle."""
import sys, os
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip, dist

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for it in range(N):
    shape = (int(rng.integers(300, 1800)), int(rng.integers(300, 1800)))
    kw = dict(tilt=int(rng.choice([1 << 26, 100000, 3000000])), white=2, nodata_pct=int(rng.choice([0, 5, 30])))
    seed = int(rng.integers(0, 1 << 30))
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    ncyc = int(rng.choice([0, 0, 1, 3]))
    for _ in range(ncyc):  # inject 2-cycles
        r, c = int(rng.integers(0, shape[0])), int(rng.integers(0, shape[1] - 1))
        d8[r, c], d8[r, c + 1] = 1, 16
    n = d8.size
    idxs_ds, idxs_pit, nvalid = O.from_array(d8)
    if idxs_pit.size == 0:
        continue
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    rank = O.rank(idxs_ds)[0]
    has_cycle = bool((rank == -1).any())
    upa_o = O.accuflux(idxs_ds, seq, np.ones(n, np.int32), nodata=-9999); upa_o[idxs_ds == -1] = -9999
    tag = f"it {it} shape {shape} seed {seed} {kw} cycles {ncyc}"
    try:
        flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
        assert np.array_equal(flw.upstream_area().ravel(), upa_o), "uparea"
        h = _hip.RasterHandle(d8, shape[0], shape[1], deferred=True)
        first = int(rng.integers(0, 3))
        if first == 1: h.rank()
        if first == 2: h.idxs_pit(np.int32)
        assert np.array_equal(h.upstream_area_cell(), upa_o), "deferred"
        h.close()
        assert np.array_equal(flw.rank.ravel(), rank), "rank"
        assert np.array_equal(flw.idxs_seq, seq), "idxs_seq"
        k = int(rng.integers(1, 40))
        oidx = np.unique(rng.integers(0, n, k))
        if has_cycle:
            oidx = np.unique(np.concatenate([oidx, np.flatnonzero(rank == -1)[:2]]))
        oids = (np.arange(oidx.size) + 3).astype(np.uint32)
        assert np.array_equal(flw.basins(idxs=oidx, ids=oids).ravel(), O.basins(idxs_ds, oidx.astype(idxs_ds.dtype), seq, oids)), "basins"
        w = O.synth_weights_f32(n, seed=it + 1)
        assert np.array_equal(flw.accuflux(w.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, w)), "accuflux f32"
        assert np.array_equal(flw.accuflux(w.reshape(shape), direction="down").ravel(), O.accuflux(idxs_ds, seq, w, direction="down")), "accuflux down"
        wi = (w * 50).astype(np.int32)
        assert np.array_equal(flw.accuflux(wi.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wi)), "accuflux i32"
        mask = w < 0.7
        assert np.array_equal(flw.stream_order(mask=mask.reshape(shape)).ravel(), O.strahler_order(idxs_ds, seq, mask)), "strahler"
        elev = O.synth_elev_f32(shape[0], shape[1], seed=seed, **kw)
        drain = upa_o > 200
        assert np.array_equal(flw.hand(drain.reshape(shape), elev).ravel(), O.height_above_nearest_drain(idxs_ds, seq, drain, elev.ravel()), equal_nan=True), "hand"
        main = O.main_upstream(idxs_ds, upa_o)
        assert np.array_equal(flw.idxs_us_main, main), "main_upstream"
        assert np.array_equal(flw.stream_order(type="classic").ravel(), O.stream_order_classic(idxs_ds, seq, main)), "classic"
        assert np.array_equal(flw.stream_distance(mask=mask.reshape(shape)).ravel(), O.stream_distance(idxs_ds, seq, shape[1], mask=mask, real_length=False)), "distance"
        nb = int(rng.integers(2, 6))
        try:
            got = dist.upstream_area_blocks(d8, nb, deferred=bool(rng.integers(0, 2)))
            assert not has_cycle, "blocks accepted a cyclic raster"
            assert np.array_equal(got.ravel(), upa_o), f"blocks {nb}"
        except NotImplementedError:
            assert has_cycle, "blocks rejected an acyclic raster"
    except AssertionError as exc:
        print("FAIL", tag, "->", exc, flush=True)
        raise
print("stress large:", N, "cases ok")
