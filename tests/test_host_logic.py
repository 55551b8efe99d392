"""Host-side logic of the front end that needs no GPU: index dtype ladder, D8 validation,
idxs_ds -> D8 re-encoding, cell-area grid, coordinate lookups, payload/nodata mapping,
argument errors raised before any device call."""
import numpy as np
import pytest

from pyflwdir_amd import gis, raster
from pyflwdir_amd._affine import Affine


def test_idxs_dtype_ladder():
    # reference tests/test_pyflwdir.py:42-51 / pyflwdir.py:105-127
    assert raster._get_idxs_dtype(10) == np.int32
    assert raster._get_idxs_dtype(2147483646) == np.int32
    assert raster._get_idxs_dtype(2147483647) == np.uint32
    assert raster._get_idxs_dtype(4294967293) == np.uint32
    assert raster._get_idxs_dtype(4294967294) == np.int64
    assert raster._get_idxs_dtype(90000 * 90000) == np.int64
    assert raster._get_idxs_dtype(72000 * 36000) == np.uint32


def test_d8_isvalid_and_infer():
    good = np.array([[1, 2, 4], [8, 16, 32], [64, 128, 0], [247, 255, 1]], dtype=np.uint8)
    assert raster.d8_isvalid(good)
    assert not raster.d8_isvalid(good.astype(np.int32))
    assert not raster.d8_isvalid(good.ravel())
    bad = good.copy()
    bad[0, 0] = 3
    assert not raster.d8_isvalid(bad)
    assert raster._infer_ftype(good) == "d8"
    with pytest.raises(ValueError, match="could not be inferred"):
        raster._infer_ftype(bad)


def test_from_array_argument_errors():
    good = np.array([[1, 2], [0, 0]], dtype=np.uint8)
    with pytest.raises(ValueError, match="should be 2 dimensional"):
        raster.from_array(good.ravel(), ftype="d8")
    with pytest.raises(ValueError, match="is invalid"):
        raster.from_array(np.array([[3, 0], [0, 0]], dtype=np.uint8), ftype="d8")
    with pytest.raises(ValueError, match='"mask" shape does not match'):
        raster.from_array(good, ftype="d8", mask=np.ones((3, 3)))
    with pytest.raises(ValueError, match="Unknown flow direction type"):
        raster.from_array(good, ftype="xyz")
    with pytest.raises(ValueError, match='type "ldd" is invalid'):
        raster.from_array(good, ftype="ldd")  # D8 values are not LDD values
    with pytest.raises(ValueError, match='type "nextxy" is invalid'):  # (negative values that are no pit / nodata code)
        raster.from_array(np.full((2, 2, 2), -3, np.int32), ftype="nextxy")
    with pytest.raises(TypeError, match="NEXTXY flwdir data not understood"):
        raster.from_array(np.zeros((2, 2), np.int32), ftype="nextxy", check_ftype=False)


def test_d8_from_idxs_ds_roundtrip(oracle):
    d8 = oracle.synth_d8(40, 50, seed=2, tilt=100000, nodata_pct=30)
    idxs_ds, idxs_pit, _ = oracle.from_array(d8)
    back = raster._d8_from_idxs_ds(idxs_ds, d8.shape, np.intp(-1))
    exp = d8.copy()
    exp.flat[idxs_pit] = 0
    assert np.array_equal(back, exp)
    idxs_u, _, _ = oracle.from_array(d8, dtype=np.uint32)
    assert np.array_equal(raster._d8_from_idxs_ds(idxs_u, d8.shape, np.uint32(4294967295)), exp)
    far = idxs_ds.copy()
    far[0] = d8.size - 1
    with pytest.raises(ValueError, match="outside 8 neighbors"):
        raster._d8_from_idxs_ds(far, d8.shape, np.intp(-1))


def test_area_grid_dtypes_and_values():
    tr = Affine(1 / 120.0, 0.0, 5.0, 0.0, -1 / 120.0, 50.0)
    a = gis.area_grid(tr, (7, 5), latlon=True, unit="km2")
    assert a.dtype == np.float64 and a.shape == (7, 5)
    lat = 50.0 - (np.arange(7) + 0.5) / 120.0
    exp = 6371e3**2 * np.radians(1 / 120.0) * (np.sin(np.radians(lat + 1 / 240.0)) - np.sin(np.radians(lat - 1 / 240.0)))
    assert np.allclose(a[:, 0] * 1e6, exp, rtol=1e-8)
    p = gis.area_grid(Affine(30.0, 0.0, 0.0, 0.0, -30.0, 0.0), (3, 4), latlon=False, unit="ha")
    assert p.dtype == np.float32 and np.all(p == np.float32(900.0 / 1e4))
    assert gis.area_grid(tr, (2, 2), unit="cell").dtype == np.int32
    with pytest.raises(ValueError, match="Unknown unit"):
        gis.area_grid(tr, (2, 2), unit="km")


def test_coords_roundtrip():
    tr = Affine(0.5, 0.0, 10.0, 0.0, -0.5, 60.0)
    shape = (20, 30)
    idxs = np.array([0, 29, 30, 599])
    xs, ys = gis.idxs_to_coords(idxs, tr, shape)
    assert np.array_equal(gis.coords_to_idxs(xs, ys, tr, shape), idxs)
    with pytest.raises(IndexError):
        gis.coords_to_idxs(np.array([9.0]), np.array([59.0]), tr, shape)
    with pytest.raises(IndexError):
        gis.idxs_to_coords(np.array([600]), tr, shape)


def test_affine_algebra():
    a = Affine(2.0, 0.0, 1.0, 0.0, -3.0, 5.0)
    x, y = a * (np.array([1.0, 2.0]), np.array([0.0, 1.0]))
    assert np.allclose(x, [3.0, 5.0]) and np.allclose(y, [5.0, 2.0])
    inv = ~a
    bx, by = inv * (x, y)
    assert np.allclose(bx, [1.0, 2.0]) and np.allclose(by, [0.0, 1.0])
    t = a * Affine.translation(0.5, 0.5)
    assert np.allclose(t * (0.0, 0.0), a * (0.5, 0.5))
    assert a[0] == 2.0 and a[4] == -3.0 and a.xoff == 1.0 and a.yoff == 5.0


def test_payload_args():
    f = raster._payload_args
    v, code, ndi, ndf, has = f(np.ones(3, np.float32), -9999)
    assert (code, ndf, has) == (4, -9999.0, 1)
    assert f(np.ones(3, np.float64), float("nan"))[4] == 0
    assert f(np.ones(3, np.int32), -9999)[2:] == (-9999, 0.0, 1)
    assert f(np.ones(3, np.int32), -9999.5)[4] == 0
    assert f(np.ones(3, np.int32), 2**40)[4] == 0
    v, code, ndi, ndf, has = f(np.ones(3, np.uint32), -9999)
    assert v.dtype == np.int32 and has == 0
    v, code, ndi, ndf, has = f(np.ones(3, np.uint32), 4294967295)
    assert (ndi, has) == (-1, 1)
    with pytest.raises(NotImplementedError):
        f(np.ones(3, np.int16), -9999)


def test_area_rows_equal_area_grid_rows():
    """upstream_area(unit != "cell") hands the device one area per row (pfd_accuflux_rows): the row
    values must be, element for element, the reference's area grid (gis_utils.area_grid,
    reference pyflwdir/gis_utils.py:388-402)."""
    import numpy as np

    from pyflwdir_amd import gis
    from pyflwdir_amd._affine import Affine

    cases = [((1 / 120.0, 0, 5.0, 0, -1 / 120.0, 50.0), True, (682, 997)),
             ((30.0, 0, 1000.0, 0, -30.0, 5000.0), False, (20, 25)),
             ((1 / 120.0, 0, 5.0, 0, -1 / 120.0, 50.0), True, (300, 1)),
             ((0.1, 0, 0, 0, -0.1, 80.0), True, (1, 300))]
    for tr, latlon, shape in cases:
        for unit in ("m2", "km2", "ha"):
            with np.errstate(all="ignore"):
                grid = gis.area_grid(Affine(*tr), shape, latlon, unit)
                rows = gis.area_rows(Affine(*tr), shape, latlon, unit)
            assert grid.dtype == rows.dtype
            assert np.array_equal(grid, np.broadcast_to(rows[:, None], shape), equal_nan=True)


def test_ldd_codec_tables():
    """LDD is the D8 graph with other labels (reference pyflwdir/core_ldd.py:11-17, core_d8.py:14-19):
    the two 256-entry tables are inverse on the alphabets and keep pits / nodata apart."""
    import numpy as np

    from pyflwdir_amd import raster as R

    assert np.array_equal(R._LDD_TO_D8[R.LDD_DS], R.D8_DS) and R._LDD_TO_D8[255] == 247
    assert np.array_equal(R._D8_TO_LDD[R.D8_DS], R.LDD_DS) and R._D8_TO_LDD[247] == 255
    assert R._LDD_TO_D8[5] == 0 and R._D8_TO_LDD[0] == 5
    for bad in (0, 10, 200, 254):  # not LDD values: must not become valid D8 codes
        assert R._LDD_TO_D8[bad] not in R.D8_ALL
    ldd = np.array([[7, 8, 9], [4, 5, 6], [1, 2, 3]], np.uint8)
    assert R.ldd_isvalid(ldd) and not R.d8_isvalid(ldd) and R._infer_ftype(ldd) == "ldd"
    d8 = R.D8_DS.copy()
    assert R._infer_ftype(d8) == "d8"  # D8 first, like the reference's FTYPES order
    with pytest.raises(ValueError, match="could not be inferred"):
        R._infer_ftype(np.array([[11, 12]], np.uint8))


def test_nextxy_codec_host():
    """core_nextxy from_array / to_array / isvalid restated in numpy (pyflwdir_amd/nextxy.py) on a hand-made
    raster: 1-based (x, y) targets, -9 river mouth, -10 inland pit, -9999 nodata, a target off the raster and a
    target on a nodata cell both become pits (reference pyflwdir/core_nextxy.py:41-68)."""
    from pyflwdir_amd import nextxy

    mv = -9999
    nextx = np.array([[2, 3, -9], [1, mv, 3], [9, 2, -10]], np.int32)
    nexty = np.array([[1, 1, -9], [1, mv, 1], [9, 2, -10]], np.int32)
    assert nextxy.isvalid((nextx, nexty)) and nextxy.isvalid(np.stack([nextx, nexty]))
    ds, pits, n = nextxy.from_array((nextx, nexty), dtype=np.int32)
    # cell 6 points at (9, 9): outside -> pit; cell 7 points at (2, 2) = cell 4 = nodata -> pit
    assert ds.tolist() == [1, 2, 2, 0, -1, 2, 6, 7, 8] and pits.tolist() == [2, 6, 7, 8] and n == 8
    back = nextxy.to_array(ds, (3, 3), mv=-1)
    assert back.shape == (2, 3, 3) and back.dtype == np.int32
    assert back[0].ravel().tolist() == [2, 3, -9, 1, mv, 3, -9, -9, -9]
    assert back[1].ravel().tolist() == [1, 1, -9, 1, mv, 1, -9, -9, -9]
    dsu = nextxy.from_array((nextx, nexty), dtype=np.uint32)[0]
    assert dsu[4] == np.uint32(4294967295)
    assert not nextxy.isvalid(np.zeros((3, 3), np.int32))


def test_bench_spread_rasters_are_valid_and_acyclic(oracle):
    """The host-built rasters of bench.py's workload-spread lines (Rhine mosaic, serpentine tiles) hold only D8
    codes and no cycle: every valid cell is ordered by the oracle's idxs_seq, so the timed pass never leaves the
    tiled engine for them."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for d8 in (bench.rhine_mosaic(1500), bench.serpentine(200, 330)):
        assert d8.dtype == np.uint8 and d8.flags["C_CONTIGUOUS"]
        assert np.isin(d8, (0, 1, 2, 4, 8, 16, 32, 64, 128, 247, 255)).all()
        idxs_ds, idxs_pit, n_valid = oracle.from_array(d8)
        seq = oracle.idxs_seq(idxs_ds, idxs_pit)
        assert seq.size == n_valid == int((d8 != 247).sum())
    s = bench.serpentine(64, 64)
    idxs_ds, idxs_pit, _ = oracle.from_array(s)
    assert idxs_pit.size == 1 and oracle.rank(idxs_ds)[0].max() == 4095  # one 4096-cell path per tile


def test_reggrid_helpers_match_the_reference_expressions():
    """reggrid_area / reggrid_dx / reggrid_dy (public in the reference's gis_utils.__all__): a per-row column times a
    matrix of ones; area_grid on a lat/lon transform is reggrid_area / factor; dtypes follow numpy's promotion."""
    from pyflwdir_amd._affine import get_affine

    Affine = get_affine()
    tr = Affine(0.25, 0.0, 5.0, 0.0, -0.25, 52.0)
    shape = (9, 6)
    lon, lat = gis.affine_to_coords(tr, shape)
    a = gis.reggrid_area(lat, lon)
    assert a.shape == shape and a.dtype == np.float64
    assert np.array_equal(a / 1e6, gis.area_grid(tr, shape, latlon=True, unit="km2"))
    assert np.array_equal(a[:, 0], gis.cellarea(lat, 0.25, 0.25)) and np.all(a == a[:, :1])
    dx, dy = gis.reggrid_dx(lat, lon), gis.reggrid_dy(lat, lon)
    assert dx.shape == shape and dy.shape == shape and dx.dtype == lat.dtype
    assert np.array_equal(dx[:, 3], gis.degree_metres_x(lat) * 0.25) and np.array_equal(dy[:, 0], gis.degree_metres_y(lat) * 0.25)
    assert {"reggrid_area", "reggrid_dx", "reggrid_dy"} <= set(gis.__all__)


def test_relevant_halo_cells_of_a_row_block():
    """dist.relevant_halo: a down-sweep depends on the halo cells its boundary cells drain into, an up-sweep on the halo
    cells that drain into a boundary cell; flow into nodata ends where it is."""
    from pyflwdir_amd.dist import relevant_halo

    MV = 247
    rows = np.array([[4, 2, MV, 8, 1, 64],      # top halo row:   S  SE  nodata  SW  E  N
                     [64, MV, 128, 32, 16, 4],  # own first row:  N  nodata  NE  NW  W  S
                     [4, 8, 2, 1, 64, 4],       # own last row:   S  SW  SE  E  N  S
                     [64, MV, 32, 128, 1, MV]],  # bottom halo row: N nodata NW NE E nodata
                    dtype=np.uint8)
    ncol = rows.shape[1]
    down = relevant_halo(rows, (1, 1), True)
    # own first row: col 0 N -> halo col 0; col 2 NE -> halo col 3; col 3 NW -> halo col 2 is nodata: ends there
    assert down[:ncol].tolist() == [True, False, False, True, False, False]
    # own last row: col 0 S -> 0; col 1 SW -> 0; col 2 SE -> 3; col 5 S -> nodata
    assert down[ncol:].tolist() == [True, False, False, True, False, False]
    up = relevant_halo(rows, (1, 1), False)
    # top halo: col 0 S -> own 0 valid; col 1 SE -> own 2 valid; col 3 SW -> own 2 valid
    assert up[:ncol].tolist() == [True, True, False, True, False, False]
    # bottom halo: col 0 N -> own 0; col 2 NW -> own 1; col 3 NE -> own 4
    assert up[ncol:].tolist() == [True, False, True, True, False, False]
    assert not relevant_halo(rows[1:], (0, 1), True)[:ncol].any()  # (no top halo row: nothing to depend on)


def test_host_thread_helpers_of_the_wide_front_end(monkeypatch):
    """The host-side pieces that assemble results of billions of cells (round 6): the same bytes as the one-piece numpy
    expressions they replace — concatenation of row blocks, the finite check, the nodata fill — also across piece edges."""
    from pyflwdir_amd import dist, raster

    rng = np.random.default_rng(5)
    parts = [rng.random((r, 3001)) for r in (700, 1, 1299, 350)]  # (> 2**28 bytes would take the threaded form: forced below)
    exp = np.concatenate(parts, axis=0)
    assert np.array_equal(dist._concat_rows(parts), exp)
    big = [np.broadcast_to(rng.random((1, 1 << 16)), (1100, 1 << 16)) for _ in range(4)]  # 4 x 577 MB views, no memory behind them
    got = dist._concat_rows(big)
    assert got.shape == (4400, 1 << 16) and all(np.array_equal(got[1100 * i], big[i][0]) and np.array_equal(got[1100 * i + 1099], big[i][0])
                                                for i in range(4))
    del got
    a = rng.random(70_000_001).astype(np.float32)
    assert dist._all_finite(a) and dist._all_finite(a.reshape(1, -1))
    for bad in (np.inf, -np.inf, np.nan):
        a[-1] = bad
        assert not dist._all_finite(a)
    a[-1] = 0
    a[(1 << 25) - 1] = np.nan  # (the last element of the first piece)
    assert not dist._all_finite(a)
    codes = rng.choice(np.array([1, 2, 4, 247, 0], np.uint8), size=(1000, 777))
    out = rng.random((1000, 777))
    exp = out.copy()
    exp[codes == 247] = -9999
    raster._fill_where(out, codes, 247, -9999, rows_per=37)  # (28 pieces on the host threads)
    assert np.array_equal(out, exp)
    out2 = exp.copy()
    raster._fill_where(out2, codes, 255, -1.0)  # (nothing to fill, one piece)
    assert np.array_equal(out2, exp)
