"""pfd_graph_stats and pfd_verify_upstream_area_cell (the full-size, oracle-free checker bench.py and the
large-size tests rely on) against the oracle: the verifier must accept the reference's result and reject
any perturbed one; the statistics must equal numpy's on the oracle's arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,kw", [((300, 417), dict(seed=3, tilt=100000, white=2, nodata_pct=25)),
                                      ((1100, 900), dict(seed=5, tilt=1 << 26, white=2, nodata_pct=0)),
                                      ((64, 2500), dict(seed=6, tilt=3000, white=2, nodata_pct=10))])
def test_stats_and_verifier(gpu_lib, oracle, shape, kw):
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(shape[0], shape[1], **kw)
    idxs_ds, idxs_pit, n_valid = oracle.from_array(d8)
    seq = oracle.idxs_seq(idxs_ds, idxs_pit)
    upa = oracle.upstream_area_cell(d8)[0].ravel()
    h = _hip.RasterHandle(d8, shape[0], shape[1])
    st = h.graph_stats()
    assert st["n_valid"] == n_valid and st["n_pits"] == idxs_pit.size
    assert st["max_rank"] == int(oracle.rank(idxs_ds)[0].max())
    nup = oracle.upstream_count(idxs_ds)
    assert st["indegree_hist"] == [int(np.sum(nup == k)) for k in range(9)]
    v = h.verify_upstream_area_cell(upa)
    assert v["bad_cells"] == 0 and v["bad_nodata"] == 0
    assert v["pit_sum"] == n_valid == v["n_valid"] and v["n_pits"] == idxs_pit.size
    assert v["checksum"] == int(upa.astype(np.int64).sum())
    # the device result passes, a single wrong cell does not
    got = h.upstream_area_cell()
    assert h.verify_upstream_area_cell(got)["bad_cells"] == 0
    bad = upa.copy()
    i = int(seq[len(seq) // 2])
    bad[i] += 1
    vb = h.verify_upstream_area_cell(bad)
    assert vb["bad_cells"] >= 1
    if n_valid < d8.size:
        bad = upa.copy()
        bad[np.flatnonzero(d8.ravel() == 247)[0]] = 0
        assert h.verify_upstream_area_cell(bad)["bad_nodata"] == 1
    h.close()


def test_tile_round_statistics(gpu_lib):
    """pfd_set_profiling(h, 2) counts the pointer-doubling rounds of the tile passes (pfd_graph_stats[12..15]): a
    raster whose every 64 x 64 tile is one 4096-cell path needs log2(4096) = 12 rounds of the value-carrying final pass
    in every tile — and of the local pass in the frame tiles (general kernel: one pointer jump per round), fewer in the
    interior tiles (two jumps per round, and a jump may use a pointer its owner has already advanced); a raster of
    isolated pits needs none beyond the first."""
    from pyflwdir_amd import _hip

    t = np.empty((64, 64), np.uint8)
    t[0::2, :] = 1
    t[1::2, :] = 16
    t[0::2, 63] = 4
    t[1::2, 0] = 4
    t[63, 0] = 0
    d8 = np.ascontiguousarray(np.tile(t, (5, 7)))
    h = _hip.RasterHandle(d8, d8.shape[0], d8.shape[1], deferred=True)
    h.set_profiling(2)
    upa = h.upstream_area_cell().reshape(d8.shape)
    st = h.graph_stats()["tile_rounds"]
    h.close()
    assert upa.max() == 4096 and st["local_max"] == st["final_max"] == 12
    assert st["final_mean"] == 12.0 and 6.0 <= st["local_mean"] < 12.0 and h is not None
    pits = np.zeros((130, 70), np.uint8)
    h = _hip.RasterHandle(pits, 130, 70, deferred=True)
    h.set_profiling(2)
    assert (h.upstream_area_cell() == 1).all()
    st = h.graph_stats()["tile_rounds"]
    h.close()
    assert st["local_max"] == st["final_max"] == 1
