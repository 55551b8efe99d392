"""Deferred handles (pfd_raster_create_deferred): decoding, the pit rule, validation and the
counts are fused into the first tile pass of upstream_area(cell).  Results, counts and errors must
be those of an eagerly created handle (= the reference's core_d8.from_array, core_d8.py:42-67)."""
import numpy as np
import pytest

from conftest import case_names
from golden_util import Case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", case_names())
def test_golden_cases(gpu_lib, manifest, name):
    from pyflwdir_amd import _hip

    case = Case(name, manifest)
    nrow, ncol = case.shape
    h = _hip.RasterHandle(case.d8, nrow, ncol, deferred=True)
    if case.d8.size >= 4:
        assert h.info()["n_valid"] == -1  # nothing has looked at the codes yet
    upa = h.upstream_area_cell().reshape(case.shape)
    case.check("uparea_cell", upa)
    st = case.entry["stats"]
    info = h.info()
    assert info["n_valid"] == st["n_valid"] and info["n_pits"] == st["n_pits"]
    # the normalised codes written by the fused pass serve every later operation
    case.check("idxs_ds_int32", h.idxs_ds(np.int32))
    case.check("idxs_seq_int32", h.idxs_seq(np.int32))
    case.check("uparea_cell", h.upstream_area_cell().reshape(case.shape))
    h.close()
    # another operation first: normalised by the standalone pass
    h = _hip.RasterHandle(case.d8, nrow, ncol, deferred=True)
    case.check("idxs_pit_int32", h.idxs_pit(np.int32))
    case.check("uparea_cell", h.upstream_area_cell().reshape(case.shape))
    h.close()


@pytest.mark.parametrize("shape,seed,kw", [
    ((1500, 2101), 3, dict(tilt=1 << 26, white=2, nodata_pct=0)),     # ncol % 4 != 0: unaligned rows
    ((2048, 2048), 4, dict(tilt=100000, white=2, nodata_pct=30)),     # n % 4096 == 0: no slack behind the buffer
    ((3000, 1003), 5, dict(tilt=1 << 26, white=2, nodata_pct=20)),
    ((64, 64), 6, dict(tilt=100000, white=2, nodata_pct=10)),
    ((1, 7), 7, dict(tilt=100000, white=2, nodata_pct=0)),
    ((5, 1), 8, dict(tilt=100000, white=2, nodata_pct=0)),
])
def test_vs_oracle_host_and_device_input(gpu_lib, oracle, shape, seed, kw):
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(shape[0], shape[1], seed=seed, **kw)
    d8[d8 == 0] = np.where(np.arange((d8 == 0).sum()) % 2 == 0, 0, 255).astype(np.uint8)  # both pit codes
    exp, _, _ = oracle.upstream_area_cell(d8)
    idxs_ds, idxs_pit, nvalid = oracle.from_array(d8)
    for memspace in (_hip.PFD_HOST, _hip.PFD_DEVICE):
        if memspace == _hip.PFD_DEVICE:
            buf = _hip.DeviceBuffer(d8.size, 0)  # exactly n bytes: the guarded loads must stay inside
            buf.upload(d8)
            h = _hip.RasterHandle(buf.addr, shape[0], shape[1], memspace=memspace, deferred=True)
        else:
            h = _hip.RasterHandle(d8, shape[0], shape[1], deferred=True)
        got = h.upstream_area_cell().reshape(shape)
        assert np.array_equal(got, exp)
        info = h.info()
        assert info["n_valid"] == nvalid and info["n_pits"] == idxs_pit.size
        assert np.array_equal(h.idxs_ds(np.int32), idxs_ds)
        h.close()


def test_errors_surface_at_first_operation(gpu_lib):
    from pyflwdir_amd import _hip

    rng = np.random.default_rng(0)
    d8 = np.full((200, 300), 4, np.uint8)  # everything flows south, off the raster: pits in the last row
    good = _hip.RasterHandle(d8, 200, 300, deferred=True).upstream_area_cell().reshape(d8.shape)
    assert good[-1, 0] == 200
    for badval in (3, 254, 246, 129):
        bad = d8.copy()
        bad[rng.integers(0, 200), rng.integers(0, 300)] = badval
        h = _hip.RasterHandle(bad, 200, 300, deferred=True)  # accepted: nothing has looked yet
        with pytest.raises(ValueError, match="not D8 codes"):
            h.upstream_area_cell()
        with pytest.raises(ValueError, match="not D8 codes"):  # and it stays rejected
            h.idxs_pit(np.int32)
        h.close()
    loop = np.tile(np.array([[1, 16]], np.uint8), (100, 50))  # 2-cycles only
    h = _hip.RasterHandle(loop, 100, 100, deferred=True)
    with pytest.raises(ValueError, match="no pits found"):
        h.upstream_area_cell()
    h.close()


@pytest.mark.parametrize("nblocks", [2, 3, 5])
def test_deferred_row_blocks(gpu_lib, oracle, nblocks):
    from pyflwdir_amd import dist

    for shape, seed, kw in [((700, 901), 11, dict(tilt=1 << 26, white=2, nodata_pct=0)),
                            ((1030, 517), 12, dict(tilt=100000, white=2, nodata_pct=30))]:
        d8 = oracle.synth_d8(shape[0], shape[1], seed=seed, **kw)
        exp, _, _ = oracle.upstream_area_cell(d8)
        assert np.array_equal(dist.upstream_area_blocks(d8, nblocks, deferred=True), exp)
