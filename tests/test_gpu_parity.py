"""GPU parity: the HIP path, called through the FlwdirRaster front end and the C-ABI, must
reproduce the reference bit for bit on every golden case (ints, labels, Strahler, the exact
idxs_seq order AND the float32/float64 accumulations and HAND — the pull sweeps add children
in the reference's order, so no tolerance is needed; the north-star tolerance for float32
accuflux is 1e-6 relative, asserted here as exact equality).

Mirrors reference tests/test_pyflwdir.py:218-289,390-407 and tests/test_streams_basins.py.
"""
import numpy as np
import pytest

from conftest import case_names
from golden_util import Case, derived_inputs
from oracle import golden_inputs as GI

pytestmark = pytest.mark.gpu

CASES = case_names()


@pytest.fixture(scope="module", params=CASES)
def case(request, manifest):
    return Case(request.param, manifest)


@pytest.fixture(scope="module", params=["exact", "levels"])
def flw(request, case, gpu_lib):
    """Every golden case runs twice: order-sensitive sweeps through the exact-order engine (exact.hip; the
    default on rasters without cycles) and, with PFD_EXACT_LEVELS=1, through the level engine."""
    import os

    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    if request.param == "levels":
        os.environ["PFD_EXACT_LEVELS"] = "1"
    try:
        yield pyflwdir.from_array(case.d8, ftype="d8", transform=Affine(*case.transform), latlon=case.latlon,
                                  cache=False)
    finally:
        os.environ.pop("PFD_EXACT_LEVELS", None)


def test_graph_exports(case, flw):
    st = case.entry["stats"]
    info = flw._h.info()
    assert info["n_valid"] == st["n_valid"] and info["n_pits"] == st["n_pits"]
    case.check("idxs_ds_int32", flw.idxs_ds)
    case.check("idxs_pit_int32", flw.idxs_pit)
    case.check("idxs_outlet", flw.idxs_outlet)
    case.check("n_upstream", flw.n_upstream.ravel())
    case.check("idxs_seq_int32", flw.idxs_seq)
    case.check("rank", flw.rank.ravel())
    assert flw.nnodes == st["n_seq"] == flw.ncells
    assert flw._h.info()["n_levels"] == st["max_rank"] + 1
    assert flw.isvalid == (st["n_loop_cells"] == 0)
    # other index dtypes of the reference's ladder (pyflwdir.py:105-127)
    for dt in (np.uint32, np.int64):
        sfx = np.dtype(dt).name
        if f"idxs_ds_{sfx}" in case.digests:
            case.check(f"idxs_ds_{sfx}", flw._h.idxs_ds(dt))
            case.check(f"idxs_pit_{sfx}", flw._h.idxs_pit(dt))
            case.check(f"idxs_seq_{sfx}", flw._h.idxs_seq(dt))


def test_upstream_area(case, flw):
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    upa = flw.upstream_area()
    case.check("uparea_cell", upa)
    # the generic level engine must agree with the fused path
    assert np.array_equal(flw._h.upstream_area_cell(engine="levels").reshape(case.shape), upa)
    case.check("uparea_km2_latlon", flw.upstream_area("km2"))
    flw_proj = pyflwdir.from_array(case.d8, ftype="d8", transform=Affine(*GI.PROJ_TRANSFORM), latlon=False, cache=False)
    case.check("uparea_ha_proj", flw_proj.upstream_area("ha"))
    # reference test_uparea (tests/test_pyflwdir.py:259-270)
    assert upa.dtype == np.int32 and upa.shape == case.shape
    if (~flw.mask).any():
        assert upa.min() == -9999
    assert upa[upa != -9999].min() == 1
    acc = flw.accuflux(np.ones(flw.shape))
    assert np.all(acc.flat[flw.mask] == upa.flat[flw.mask])


def test_accuflux(case, flw):
    P = GI.payloads(case.shape)
    case.check("accuflux_f32", flw.accuflux(P["w32"]))
    case.check("accuflux_f64", flw.accuflux(P["w64"]))
    case.check("accuflux_ds_f32", flw.accuflux(P["w32"], direction="down"))
    case.check("accuflux_i32_nodata", flw.accuflux(P["wi32_nodata"], nodata=-9999))
    case.check("accuflux_ds_i32_nodata", flw.accuflux(P["wi32_nodata"], nodata=-9999, direction="down"))
    case.check("accuflux_f32_nodata_m1", flw.accuflux(P["wf32_nodata_m1"], nodata=-1))
    case.check("accuflux_i64", flw.accuflux(P["wi64"]))


def test_strahler_basins_hand(case, flw):
    upa = flw.upstream_area()
    D = derived_inputs(case, upa, flw.idxs_pit)
    sto = flw.stream_order()
    case.check("strahler", sto)
    case.check("strahler_mask_upa", flw.stream_order(mask=D["mask_upa"]))
    case.check("strahler_mask_rand", flw.stream_order(mask=D["mask_rand"]))
    assert sto.dtype == np.uint8 and sto.flat[flw.mask].min() >= 0
    bas = flw.basins()
    case.check("basins", bas)
    assert bas.dtype == np.uint32 and bas.max() <= flw.idxs_pit.size
    case.check("basins_sub_i16", flw.basins(idxs=D["basins_idxs"], ids=D["basins_ids"]))
    # the same outlets given as coordinates (reference pyflwdir.py:564-599: idxs = self.index(*xy))
    xs, ys = flw.xy(D["basins_idxs"])
    case.check("basins_sub_i16", flw.basins(xy=(xs, ys), ids=D["basins_ids"]))
    case.check("hand_f32", flw.hand(D["drain"], D["elevtn"]))
    case.check("hand_f64", flw.hand(D["drain"], D["elevtn"].astype(np.float64) * 1.000001))


def test_streams_next(case, flw):
    """main upstream cell, classic stream order, stream distance (reference
    tests/test_streams_basins.py:154-168, tests/test_pyflwdir.py:310-327)."""
    upa = flw.upstream_area()
    D = derived_inputs(case, upa, flw.idxs_pit)
    case.check("idxs_us_main", flw.idxs_us_main)
    case.check("idxs_us_main_km2", flw.main_upstream(uparea=flw.upstream_area("km2")))
    sto = flw.stream_order(type="classic")
    case.check("strord_classic", sto)
    case.check("strord_classic_mask", flw.stream_order(type="classic", mask=D["mask_upa"]))
    assert sto.dtype == np.uint8 and np.all(sto.flat[flw.idxs_pit] == 1)
    dist = flw.stream_distance(unit="cell")
    case.check("strdist_cell", dist)
    assert dist.dtype == np.int32 and dist.max() == flw.rank.max()  # reference tests/test_pyflwdir.py:315-317
    case.check("strdist_cell_mask", flw.stream_distance(mask=D["mask_upa"], unit="cell"))
    case.check("strdist_m_latlon", flw.stream_distance(unit="m"))
    case.check("strdist_m_mask", flw.stream_distance(mask=D["mask_rand"], unit="m"))
    allmask = flw.stream_distance(mask=np.ones(case.shape, dtype=bool))
    assert np.all(allmask[allmask != -9999] == 0)
    with pytest.raises(ValueError, match="Unknown unit"):
        flw.stream_distance(unit="km")
    with pytest.raises(ValueError, match="size does not match"):
        flw.stream_distance(mask=np.ones((2, 1)) if case.n != 2 else np.ones((3, 1)))
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    flw_proj = pyflwdir.from_array(case.d8, ftype="d8", transform=Affine(*GI.PROJ_TRANSFORM), latlon=False, cache=False)
    case.check("strdist_m_proj", flw_proj.stream_distance(unit="m"))


def test_codecs(case, flw):
    """re-encoding to D8 / LDD and LDD rasters as input (reference
    tests/test_pyflwdir.py:19-34, tests/test_core_xx.py:12-62)."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    case.check("to_array_d8", flw.to_array("d8"))
    ldd = flw.to_array("ldd")
    case.check("to_array_ldd", ldd)
    assert np.all(pyflwdir.from_array(flw.to_array()).idxs_ds == flw.idxs_ds)  # tests/test_pyflwdir.py:24
    flw_ldd = pyflwdir.from_array(ldd, ftype="infer", transform=Affine(*case.transform), latlon=case.latlon, cache=False)
    if "ldd_idxs_ds" in case.digests:  # (the interpreted reference parses LDD input on narrow rasters only)
        if "ldd_inferred_is_ldd" in case.full:
            assert (flw_ldd.ftype == "ldd") == bool(case.full["ldd_inferred_is_ldd"])
        case.check("ldd_idxs_ds", flw_ldd.idxs_ds)
        case.check("ldd_idxs_outlet", flw_ldd.idxs_outlet)
        case.check("ldd_uparea_cell", flw_ldd.upstream_area())
    # an LDD raster is the same graph as its D8 twin
    assert np.array_equal(flw_ldd.idxs_ds, flw.idxs_ds) and np.array_equal(flw_ldd.upstream_area(), flw.upstream_area())
    if flw_ldd.ftype == "ldd":
        case.check("to_array_ldd", flw_ldd.to_array())  # default: the input's own type
    with pytest.raises(ValueError, match="unknown"):
        flw.to_array("unknown")  # tests/test_pyflwdir.py:32-33


def test_constructor_from_idxs_ds(case, flw):
    """FlwdirRaster(idxs_ds, shape, "d8") like reference tests/conftest.py:49-54."""
    import pyflwdir_amd as pyflwdir

    if case.n <= 1:
        pytest.skip("single cell")
    flw2 = pyflwdir.FlwdirRaster(flw.idxs_ds.copy(), case.shape, "d8", idxs_pit=flw.idxs_pit.copy(), cache=False)
    case.check("uparea_cell", flw2.upstream_area())
    case.check("idxs_seq_int32", flw2.idxs_seq)


def test_errors(case, flw):
    """Error behaviour of the wrappers (reference tests/test_pyflwdir.py:230-235,271-278,403-407)."""
    with pytest.raises(ValueError, match="Unknown unit"):
        flw.upstream_area(unit="km")
    with pytest.raises(ValueError, match="size does not match"):
        flw.accuflux(np.ones((2, 1)) if case.n != 2 else np.ones((3, 1)))
    with pytest.raises(ValueError, match="Unknown flow direction"):
        flw.accuflux(np.ones((1, 1)), direction="???")
    k = flw.idxs_pit.size
    if k > 1:
        with pytest.raises(ValueError, match="size does not match"):
            flw.basins(ids=np.arange(k - 1))
    with pytest.raises(ValueError, match="IDs cannot contain a value zero"):
        flw.basins(ids=np.zeros(k, dtype=np.int16))
    with pytest.raises(ValueError, match="size does not match"):
        flw.hand(np.ones(case.shape, bool), np.ones((case.shape[0] + 1, 1)))
