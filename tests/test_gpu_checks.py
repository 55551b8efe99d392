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
