"""Randomised parity: uniformly random D8 rasters (every code incl. both pit codes and nodata, so
full of short paths, cycles, flow off the raster and into nodata) at shapes around the tile /
supertile edges, every operation against the oracle.  Random codes are the adversarial regime for
the tiled engines (cycles that cross tile borders, tiles without any exit, partial tiles) and for
the fallbacks to the level engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODES = np.array([1, 2, 4, 8, 16, 32, 64, 128, 0, 255, 247], np.uint8)
SHAPES = [(1, 2), (2, 3), (7, 5), (63, 65), (64, 64), (65, 63), (64, 129), (130, 67), (200, 511), (513, 520),
          (1, 700), (700, 1), (1030, 90)]


def random_d8(rng, shape, p_nodata, p_pit, coherent):
    n = shape[0] * shape[1]
    p_dir = (1.0 - p_nodata - p_pit) / 8
    d8 = rng.choice(CODES, size=n, p=[p_dir] * 8 + [p_pit / 2, p_pit / 2, p_nodata]).reshape(shape)
    if coherent > 0:  # mostly one direction: long paths, plus noise
        keep = rng.random(shape) < 0.7
        d8 = np.where(keep & (d8 != 247), np.uint8(coherent), d8)
    if coherent == -1:  # acyclic: only E / SE / S / SW (row, then column, strictly increases along a path)
        dirs = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=n).reshape(shape)
        d8 = np.where(np.isin(d8, [16, 32, 64, 128]), dirs, d8)
    if not np.isin(d8, [0, 255]).any():
        d8.flat[rng.integers(0, n)] = 0
    return np.ascontiguousarray(d8, dtype=np.uint8)


@pytest.mark.parametrize("seed", range(len(SHAPES) * 3))
def test_random_rasters(gpu_lib, oracle, seed):
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip

    O = oracle
    rng = np.random.default_rng(1000 + seed)
    shape = SHAPES[seed % len(SHAPES)]
    d8 = random_d8(rng, shape, p_nodata=rng.choice([0.0, 0.1, 0.4]), p_pit=rng.choice([0.002, 0.05]),
                   coherent=rng.choice([0, 4, 2, 1, -1, -1]))
    idxs_ds, idxs_pit, nvalid = O.from_array(d8)
    if idxs_pit.size == 0:
        pytest.skip("no pit survived")
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    n = d8.size
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert np.array_equal(flw.idxs_ds, idxs_ds) and np.array_equal(flw.idxs_pit, idxs_pit)
    assert np.array_equal(flw.idxs_seq, seq)
    assert np.array_equal(flw.rank.ravel(), O.rank(idxs_ds)[0])
    upa_o = O.accuflux(idxs_ds, seq, np.ones(n, np.int32), nodata=-9999)
    upa_o[idxs_ds == -1] = -9999
    upa = flw.upstream_area()
    assert np.array_equal(upa.ravel(), upa_o)
    # deferred handle: decode fused into the tile pass (cycles make it fall back to the level engine)
    h = _hip.RasterHandle(d8, shape[0], shape[1], deferred=True)
    assert np.array_equal(h.upstream_area_cell(), upa_o)
    h.close()
    w = rng.random(n).astype(np.float32)
    w[rng.random(n) < 0.05] = -9999
    assert np.array_equal(flw.accuflux(w.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, w))
    assert np.array_equal(flw.accuflux(w.reshape(shape), direction="down").ravel(),
                          O.accuflux(idxs_ds, seq, w, direction="down"))
    wi32 = rng.integers(0, 1000, n).astype(np.int32)  # tiled engine if the raster is acyclic
    assert np.array_equal(flw.accuflux(wi32.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wi32))
    wi = rng.integers(-5, 1000, n).astype(np.int64)
    assert np.array_equal(flw.accuflux(wi.reshape(shape), nodata=-3).ravel(), O.accuflux(idxs_ds, seq, wi, nodata=-3))
    mask = rng.random(n) < 0.6
    assert np.array_equal(flw.stream_order().ravel(), O.strahler_order(idxs_ds, seq))
    assert np.array_equal(flw.stream_order(mask=mask.reshape(shape)).ravel(), O.strahler_order(idxs_ds, seq, mask))
    assert np.array_equal(flw.basins().ravel(), O.basins(idxs_ds, idxs_pit, seq))
    k = min(n, 17)
    oidx = np.unique(rng.integers(0, n, k))
    oids = (np.arange(oidx.size) + 5).astype(np.uint16)
    exp = O.basins(idxs_ds, oidx.astype(idxs_ds.dtype), seq, oids)
    assert np.array_equal(flw.basins(idxs=oidx, ids=oids).ravel(), exp)
    elev = (rng.random(n) * 100).astype(np.float32)
    drain = rng.random(n) < 0.1
    assert np.array_equal(flw.hand(drain.reshape(shape), elev.reshape(shape)).ravel(),
                          O.height_above_nearest_drain(idxs_ds, seq, drain, elev), equal_nan=True)
    main = O.main_upstream(idxs_ds, upa_o)
    assert np.array_equal(flw.idxs_us_main, main)
    assert np.array_equal(flw.stream_order(type="classic", mask=mask.reshape(shape)).ravel(),
                          O.stream_order_classic(idxs_ds, seq, main, mask))
    assert np.array_equal(flw.stream_distance(mask=mask.reshape(shape)).ravel(),
                          O.stream_distance(idxs_ds, seq, shape[1], mask=mask, real_length=False))
    assert np.array_equal(flw.stream_distance(unit="m").ravel(),
                          O.stream_distance(idxs_ds, seq, shape[1], latlon=False, transform=tuple(flw.transform)[:6]))
    if shape[0] >= 4:
        from pyflwdir_amd import dist

        for nb in (2, 3):
            try:
                got = dist.upstream_area_blocks(d8, nb)
            except NotImplementedError:  # rasters with cycles are rejected by the block protocol
                assert (O.rank(idxs_ds)[0] == -1).any()
                continue
            assert np.array_equal(got.ravel(), upa_o)
