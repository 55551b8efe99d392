"""Full-size checks the oracle cannot do in seconds: size-independent properties and cross-engine
equality (LDS-tiled engine vs level engine vs row blocks) on a raster that spans several hypertiles
(20000 x 12000 = 240 Mcells: 313 x 188 tiles, 40 x 24 supertiles, 10 x 6 hypertiles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(tilt=1 << 26, white=2, nodata_pct=0), dict(tilt=100000, white=2, nodata_pct=30)])
def test_engines_agree_at_scale(gpu_lib, kw):
    from pyflwdir_amd import _hip

    nrow, ncol = 20000, 12000
    n = nrow * ncol
    d8 = _hip.synth_d8_device(nrow, ncol, seed=4, **kw)
    h = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE)
    info = h.info()
    tiled = h.upstream_area_cell()                     # LDS-tiled engine (3 hierarchy levels + global)
    levels = h.upstream_area_cell(engine="levels")     # rank sort + level sweeps (validated against the oracle)
    assert np.array_equal(tiled, levels)
    # reference invariant (tests/test_streams_basins.py:24-27): pit sums == number of valid cells
    pits = h.idxs_pit(np.int32)
    assert pits.size == info["n_pits"]
    assert int(tiled[pits].astype(np.int64).sum()) == info["n_valid"]
    codes = d8.download(np.uint8, (n,))
    assert np.all(tiled[codes == 247] == -9999) and tiled[codes != 247].min() == 1
    # basins through the tiled label query == basins through the level engine; sizes == upstream area of the pits
    bas = h.basins(pits.astype(np.int64), np.arange(1, pits.size + 1, dtype=np.uint32))
    sizes = np.bincount(bas, minlength=pits.size + 1)[1:]
    assert np.array_equal(sizes, tiled[pits])
    # rank (tiled path query): pits have rank 0, every other valid cell is one step further than its downstream cell
    rank = h.rank()
    idxs_ds = h.idxs_ds(np.int32)
    valid = idxs_ds >= 0
    assert np.all(rank[pits] == 0) and np.all(rank[~valid] == -9999)
    nonpit = valid & (idxs_ds != np.arange(n, dtype=np.int32))
    assert np.all(rank[nonpit] == rank[idxs_ds[nonpit]] + 1)
    # deferred handle: decode / validation / counts fused into the first tile pass
    hd = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE, deferred=True)
    assert np.array_equal(hd.upstream_area_cell(), tiled)
    assert hd.info()["n_valid"] == info["n_valid"] and hd.info()["n_pits"] == info["n_pits"]
    hd.close()
    # int32 payload on the tiled engine == the level engine (PFD_ACCUFLUX_LEVELS forces the latter)
    import os

    w = ((np.arange(n, dtype=np.int64) * 2654435761) % 7).astype(np.int32)
    acc_tiled = h.accuflux(w, _hip.PFD_I32, nodata_i=-9999)
    os.environ["PFD_ACCUFLUX_LEVELS"] = "1"
    try:
        acc_levels = h.accuflux(w, _hip.PFD_I32, nodata_i=-9999)
    finally:
        del os.environ["PFD_ACCUFLUX_LEVELS"]
    assert np.array_equal(acc_tiled, acc_levels)
    h.close()
    # row blocks (multi-GPU protocol, in-process): 3 blocks of the same raster
    from pyflwdir_amd import dist

    got = dist.upstream_area_blocks(codes.reshape(nrow, ncol), 3)
    assert np.array_equal(got.ravel(), tiled)
