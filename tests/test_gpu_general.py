"""NEXTXY rasters and FlwdirRaster(idxs_ds=...) with links outside the 8 neighbours run on the
general idxs_ds engine (csrc/general.hip); every operation against the reference's recorded outputs
(tests/golden/wide_general.npz, oracle/gen_golden_wide.py), bit-exact incl. floats; dump / load round trip with
the reference's dictionary layout (pyflwdir.py:275-286)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["flwdir0", "flwdir_large", "synth_loops_96x80"]


def _check_ops(W, tag, flw, elv):
    from oracle import golden_inputs as GI

    def eq(key, got):
        exp = W[f"out_{tag}_{key}"]
        got = np.asarray(got)
        assert got.dtype == exp.dtype, (tag, key, got.dtype, exp.dtype)
        assert got.shape == exp.shape, (tag, key)
        assert np.array_equal(got, exp, equal_nan=True), (tag, key, np.flatnonzero(got.ravel() != exp.ravel())[:5])

    upa = flw.upstream_area()
    eq("idxs_ds", flw.idxs_ds)
    eq("idxs_pit", flw.idxs_pit)
    if f"out_{tag}_idxs_outlet" in W.files:
        eq("idxs_outlet", flw.idxs_outlet)
    else:
        assert flw.idxs_outlet is None  # FlwdirRaster(idxs_ds=...) built without outlets, like the reference
    eq("idxs_seq", flw.idxs_seq)
    eq("rank", flw.rank)
    eq("n_upstream", flw.n_upstream)
    eq("upa", upa)
    eq("upa_km2", flw.upstream_area("km2"))
    P = GI.payloads(flw.shape)
    eq("accu_f32", flw.accuflux(P["w32"]))
    eq("accu_ds_f64", flw.accuflux(P["w64"], direction="down"))
    eq("accu_i32_nd", flw.accuflux(P["wi32_nodata"], nodata=-9999))
    eq("strahler", flw.stream_order())
    eq("strahler_mask", flw.stream_order(mask=GI.random_mask(flw.shape)))
    eq("classic", flw.stream_order(type="classic"))
    eq("idxs_us_main", flw.idxs_us_main)
    eq("basins", flw.basins())
    eq("basins_sub", flw.basins(idxs=W[f"in_{tag}_basins_idxs"], ids=W[f"in_{tag}_basins_ids"]))
    thr = GI.threshold(upa)
    eq("hand", flw.hand(upa > thr, elv))
    eq("dist_cell", flw.stream_distance(unit="cell"))
    eq("dist_m", flw.stream_distance(unit="m"))
    eq("dist_m_mask", flw.stream_distance(mask=upa > thr, unit="m"))
    flw.order_cells(method="sort")
    eq("idxs_seq_sort", flw.idxs_seq)


@pytest.mark.parametrize("name", CASES)
def test_nextxy_and_arbitrary_links(gpu_lib, name, tmp_path):
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import nextxy
    from pyflwdir_amd._affine import Affine

    W = np.load(os.path.join(GOLD, "wide_general.npz"))
    ent = json.load(open(os.path.join(GOLD, "manifest.json")))[name]
    A = Affine(*ent["transform"])
    elv = W[f"in_{name}_elevtn"]
    nxy = W[f"in_{name}_nextxy"]
    assert nextxy.isvalid(nxy) == bool(W[f"out_{name}_nextxy_isvalid"])
    for ftype in ("nextxy", "infer"):
        flwn = pyflwdir.from_array(nxy, ftype=ftype, transform=A, latlon=ent["latlon"], cache=False)
        assert flwn.ftype == "nextxy"
    _check_ops(W, name + "_nextxy", flwn, elv)
    assert np.array_equal(flwn.to_array(), W[f"out_{name}_nextxy_to_array"])
    assert np.array_equal(flwn.to_array("d8"), W[f"out_{name}_nextxy_to_d8"])  # (these links ARE neighbours)
    # the tuple form of a NEXTXY raster
    flwt = pyflwdir.from_array((nxy[0], nxy[1]), ftype="nextxy", cache=False)
    assert np.array_equal(flwt.idxs_ds, flwn.idxs_ds)
    # arbitrary links
    ds2 = W[f"in_{name}_ds2"]
    flw2 = pyflwdir.FlwdirRaster(idxs_ds=ds2, shape=nxy.shape[1:], ftype="d8", transform=A, latlon=ent["latlon"], cache=False)
    _check_ops(W, name + "_ds2", flw2, elv)
    assert np.array_equal(flw2.to_array("nextxy"), W[f"out_{name}_ds2_to_nextxy"])
    with pytest.raises(ValueError, match="outside 8 neighbors"):
        flw2.to_array("d8")
    # read_nextxy: the CaMa-Flood binary layout (raw int32 [2, nrow, ncol]), core_nextxy.py:122-144
    fn = tmp_path / "nextxy.bin"
    nxy.astype("i4").tofile(fn)
    data, tr = pyflwdir.read_nextxy(fn, nxy.shape[1], nxy.shape[2], bbox=[0, -nxy.shape[1], nxy.shape[2], 0])
    assert np.array_equal(data, nxy) and tuple(tr)[:6] == (1.0, 0.0, 0.0, 0.0, -1.0, 0.0)
    # dump / load: same dictionary keys as the reference, identical object after the round trip
    pkl = tmp_path / "flw.pkl"
    for flw in (flwn, flw2):
        assert ",".join(sorted(flw._dict.keys())) == str(W["dump_keys"])
        flw.dump(pkl)
        back = pyflwdir.FlwdirRaster.load(pkl)
        assert back.ftype == flw.ftype and back.shape == flw.shape and back.latlon == flw.latlon
        assert np.array_equal(back.idxs_ds, flw.idxs_ds) and np.array_equal(back.idxs_pit, flw.idxs_pit)
        assert np.array_equal(back.upstream_area(), flw.upstream_area())


def test_dump_load_d8(gpu_lib, tmp_path):
    import pyflwdir_amd as pyflwdir

    z = np.load(os.path.join(GOLD, "flwdir_large.npz"))
    flw = pyflwdir.from_array(z["d8"], ftype="d8")
    flw.order_cells("walk")
    flw.dump(tmp_path / "a.pkl")
    back = pyflwdir.FlwdirRaster.load(tmp_path / "a.pkl")
    assert back.ftype == "d8" and back.shape == flw.shape
    assert np.array_equal(back.idxs_ds, flw.idxs_ds) and np.array_equal(back.to_array(), flw.to_array())
    assert np.array_equal(back.upstream_area(), flw.upstream_area()) and back.nnodes == flw.nnodes


@pytest.mark.parametrize("nm", ["odd_nbr", "odd_far"])
def test_unchecked_d8_values(gpu_lib, nm):
    """from_array(ftype="d8", check_ftype=False) decodes values outside the alphabet like core_d8.drdc
    (reference pyflwdir/core_d8.py:20-37): neighbour offsets on the D8 engines, the -2 column offsets of the
    values 9..15 as a general graph."""
    import pyflwdir_amd as pyflwdir

    W = np.load(os.path.join(GOLD, "wide_general.npz"))
    d = W[f"in_{nm}"]
    with pytest.raises(ValueError, match="is invalid"):
        pyflwdir.from_array(d, ftype="d8")
    f = pyflwdir.from_array(d, ftype="d8", check_ftype=False, cache=False)
    for key in ("idxs_ds", "idxs_pit", "idxs_outlet", "rank"):
        exp = W[f"out_{nm}_{key}"]
        got = getattr(f, key)
        assert got.dtype == exp.dtype and np.array_equal(got, exp), (nm, key)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("kind", ["nextxy", "ds2"])
def test_add_pits_on_general_graphs(gpu_lib, name, kind):
    """add_pits on a NEXTXY raster / a raster of non-neighbour links while a sorted cell order is installed (reference
    pyflwdir/flwdir.py:261-279 resets the order; tests/golden/wide_general_pits.npz, oracle/gen_golden_wide.py): the
    device must forget the installed order with the edit (it kept it and indexed with a stale length)."""
    import pyflwdir_amd as pyflwdir
    from oracle import golden_inputs as GI
    from pyflwdir_amd._affine import Affine

    W = np.load(os.path.join(GOLD, "wide_general.npz"))
    Pz = np.load(os.path.join(GOLD, "wide_general_pits.npz"))
    ent = json.load(open(os.path.join(GOLD, "manifest.json")))[name]
    A = Affine(*ent["transform"])
    nxy = W[f"in_{name}_nextxy"]
    if kind == "nextxy":
        flw = pyflwdir.from_array(nxy, ftype="nextxy", transform=A, latlon=ent["latlon"], cache=False)
    else:
        flw = pyflwdir.FlwdirRaster(idxs_ds=W[f"in_{name}_ds2"], shape=nxy.shape[1:], ftype="d8", transform=A,
                                    latlon=ent["latlon"], cache=False)
    tag = f"{name}_{kind}"
    flw.upstream_area()
    flw.order_cells(method="sort")
    flw.add_pits(idxs=Pz[f"in_{tag}_pits"])

    def eq(key, got):
        exp = Pz[f"out_{tag}_{key}"]
        got = np.asarray(got)
        assert got.dtype == exp.dtype and got.shape == exp.shape, (tag, key, got.dtype, exp.dtype)
        assert np.array_equal(got, exp, equal_nan=True), (tag, key, np.flatnonzero(got.ravel() != exp.ravel())[:5])

    eq("idxs_pit", flw.idxs_pit)
    eq("idxs_ds", flw.idxs_ds)
    eq("idxs_seq", flw.idxs_seq)
    eq("rank", flw.rank)
    upa = flw.upstream_area()
    eq("upa", upa)
    P = GI.payloads(flw.shape)
    eq("accu_f32", flw.accuflux(P["w32"]))
    eq("accu_ds_f64", flw.accuflux(P["w64"], direction="down"))
    eq("strahler", flw.stream_order())
    eq("basins", flw.basins())
    eq("hand", flw.hand(upa > GI.threshold(upa), W[f"in_{name}_elevtn"]))


def test_walk_after_sort_is_breadth_first(gpu_lib):
    """order_cells("walk") after order_cells("sort") on a general graph returns the reference's breadth-first
    core.idxs_seq order, not the installed argsort order; a sequence that repeats a cell is refused."""
    import pyflwdir_amd as pyflwdir

    W = np.load(os.path.join(GOLD, "wide_general.npz"))
    name = "flwdir_large"
    flw = pyflwdir.FlwdirRaster(idxs_ds=W[f"in_{name}_ds2"], shape=W[f"in_{name}_nextxy"].shape[1:], ftype="d8", cache=False)
    flw.order_cells(method="sort")
    assert np.array_equal(flw.idxs_seq, W[f"out_{name}_ds2_idxs_seq_sort"])
    flw.order_cells(method="walk")
    assert np.array_equal(flw.idxs_seq, W[f"out_{name}_ds2_idxs_seq"])
    assert np.array_equal(flw.upstream_area(), W[f"out_{name}_ds2_upa"])
    seq = W[f"out_{name}_ds2_idxs_seq_sort"].copy()
    rnk = flw.rank.ravel()
    same = np.flatnonzero(rnk[seq[1:]] == rnk[seq[:-1]])
    seq[same[0] + 1] = seq[same[0]]  # still rank-ordered, but one cell twice and one missing
    with pytest.raises(ValueError, match="repeats a cell"):
        flw._h.set_idxs_seq(seq)
    assert np.array_equal(flw.upstream_area(), W[f"out_{name}_ds2_upa"])  # the handle is still usable
