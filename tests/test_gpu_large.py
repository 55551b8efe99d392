"""GPU-vs-oracle parity on rasters too large for the interpreted reference, plus
size-independent invariants (reference tests/test_streams_basins.py:14-50: pit sums == number
of cells, basin sizes == upstream area at the pits) and device-generator == host-generator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,seed,kw", [
    ((1500, 2100), 3, dict(tilt=1 << 26, white=2, nodata_pct=0)),
    ((2048, 2048), 4, dict(tilt=100000, white=2, nodata_pct=30)),
    ((3000, 1000), 5, dict(tilt=1 << 26, white=2, nodata_pct=20)),
])
def test_vs_oracle(gpu_lib, oracle, shape, seed, kw):
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    # device generator is the bit-exact twin of the host generator
    buf = _hip.synth_d8_device(shape[0], shape[1], seed=seed, **kw)
    assert np.array_equal(buf.download(np.uint8, shape), d8)
    ebuf = _hip.synth_elev_device(shape[0], shape[1], seed=seed, **kw)
    elev = O.synth_elev_f32(shape[0], shape[1], seed=seed, **kw)
    assert np.array_equal(ebuf.download(np.float32, shape), elev)
    w = O.synth_weights_f32(d8.size, seed=1)
    assert np.array_equal(_hip.synth_weights_device(d8.size, seed=1).download(np.float32, (d8.size,)), w)

    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    idxs_ds, idxs_pit, nvalid = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    assert np.array_equal(flw.idxs_ds, idxs_ds) and np.array_equal(flw.idxs_pit, idxs_pit)
    assert np.array_equal(flw.idxs_seq, seq)
    upa_o, _, _ = O.upstream_area_cell(d8)
    upa = flw.upstream_area()
    assert np.array_equal(upa, upa_o)
    # invariants
    assert upa.flat[idxs_pit].astype(np.int64).sum() == seq.size
    bas = flw.basins()
    assert np.array_equal(bas.ravel(), O.basins(idxs_ds, idxs_pit, seq))
    sizes = np.bincount(bas.ravel(), minlength=idxs_pit.size + 1)[1:]
    assert np.array_equal(sizes, upa.flat[idxs_pit])
    # float32 accumulation: bit-exact (tolerance of the north star: 1e-6 relative)
    acc = flw.accuflux(w.reshape(shape))
    acc_o = O.accuflux(idxs_ds, seq, w).reshape(shape)
    assert np.array_equal(acc, acc_o)
    sto = flw.stream_order()
    assert np.array_equal(sto.ravel(), O.strahler_order(idxs_ds, seq))
    drain = upa > 500
    hand = flw.hand(drain, elev)
    assert np.array_equal(hand.ravel(), O.height_above_nearest_drain(idxs_ds, seq, drain.ravel(), elev.ravel()))
    assert np.array_equal(flw.rank.ravel(), O.rank(idxs_ds)[0])
    # int32 payloads: the tiled engine where the nodata rule provably cannot interfere, else the level engine
    flw._h.set_profiling(True)
    wi = (w * 1000).astype(np.int32)
    assert np.array_equal(flw.accuflux(wi.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wi))
    assert any(s["name"] == "tile_local" for s in flw._h.last_timing())       # non-negative, no nodata hit: tiled
    wneg = wi - 300
    assert np.array_equal(flw.accuflux(wneg.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wneg))
    assert not any(s["name"] == "tile_local" for s in flw._h.last_timing())   # negative values: level engine
    wbig = np.full(d8.size, 2000, np.int32)                                   # total >= 2^31: int32 wrap, level engine
    assert np.array_equal(flw.accuflux(wbig.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wbig))
    wnd = wi.copy()
    wnd[::13] = -9999                                                          # nodata inside the domain: level engine
    assert np.array_equal(flw.accuflux(wnd.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wnd))
    wu = wi.astype(np.uint32)                                                  # unsigned view, nodata can never match
    assert np.array_equal(flw.accuflux(wu.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wu))
    flw._h.set_profiling(False)
    # SURVEY 8(f)-1 functions
    main = O.main_upstream(idxs_ds, upa_o.ravel())
    assert np.array_equal(flw.idxs_us_main, main)
    assert np.array_equal(flw.main_upstream(uparea=acc), O.main_upstream(idxs_ds, acc_o.ravel()))  # float32 areas
    assert np.array_equal(flw.stream_order(type="classic").ravel(), O.stream_order_classic(idxs_ds, seq, main))
    assert np.array_equal(flw.stream_order(type="classic", mask=drain).ravel(),
                          O.stream_order_classic(idxs_ds, seq, main, drain.ravel()))
    assert np.array_equal(flw.stream_distance(unit="cell").ravel(),
                          O.stream_distance(idxs_ds, seq, shape[1], real_length=False))
    assert np.array_equal(flw.stream_distance(mask=drain, unit="m").ravel(),
                          O.stream_distance(idxs_ds, seq, shape[1], mask=drain.ravel(), latlon=False,
                                            transform=tuple(flw.transform)[:6]))


def test_add_pits(gpu_lib, oracle):
    """add_pits turns cells into pits and invalidates the order (reference flwdir.py:261-279)."""
    import pyflwdir_amd as pyflwdir

    O = oracle
    d8 = O.synth_d8(300, 400, seed=9)
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    upa = flw.upstream_area()
    new = np.argsort(upa.ravel())[-5:-2]
    flw.add_pits(idxs=new)
    d8b = d8.copy()
    d8b.flat[new] = 0
    idxs_ds, idxs_pit, _ = O.from_array(d8b)
    assert np.array_equal(flw.idxs_pit, idxs_pit) and np.array_equal(flw.idxs_ds, idxs_ds)
    assert np.array_equal(flw.upstream_area(), O.upstream_area_cell(d8b)[0])
    assert np.array_equal(flw.idxs_seq, O.idxs_seq(idxs_ds, idxs_pit))


def test_invalid_rasters(gpu_lib):
    import pyflwdir_amd as pyflwdir

    with pytest.raises(ValueError, match="no pits found"):
        pyflwdir.from_array(np.array([[1, 16], [1, 16]], dtype=np.uint8), ftype="d8")  # two 2-cycles
    with pytest.raises(ValueError, match="is invalid"):
        pyflwdir.from_array(np.array([[3, 0], [0, 0]], dtype=np.uint8), ftype="d8")
    # check_ftype=False lets values outside the alphabet through, decoded like core_d8.drdc (3 -> south)
    flw = pyflwdir.from_array(np.array([[3, 0], [0, 0]], dtype=np.uint8), ftype="d8", check_ftype=False)
    assert flw.idxs_ds.tolist() == [2, 1, 2, 3]
    # the C-ABI itself takes D8 codes only
    from pyflwdir_amd import _hip

    with pytest.raises(ValueError, match="not D8 codes"):
        _hip.RasterHandle(np.array([[3, 0], [0, 0]], dtype=np.uint8), 2, 2)


def test_hypertile_overflow_fallback(gpu_lib, oracle, monkeypatch):
    """A hypertile with more super-exits than fit in LDS makes the pass fall back to the flat
    level-3 id range (forced here by lowering the capacity); single handle and row blocks.  (A WHOLE raster of at most 64
    hypertiles takes the flat forest without host round trips since round 6: PFD_HYPER_SMALL keeps the hypertile solves
    for this one, so that their overflow path stays under test at a size the oracle answers.)"""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(2300, 2600, seed=8, tilt=100000, white=2, nodata_pct=10)  # 2 x 2 hypertiles
    exp, _, _ = oracle.upstream_area_cell(d8)
    monkeypatch.setenv("PFD_HYPER_SMALL", "1")
    monkeypatch.setenv("PFD_TEST_HCAP", "100")
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert np.array_equal(flw.upstream_area(), exp)
    assert np.array_equal(dist.upstream_area_blocks(d8, 2), exp)


@pytest.mark.parametrize("scap", ["0", "5000"])
def test_supertile_dense_capacity_fallback(gpu_lib, oracle, monkeypatch, scap):
    """The level-2 solve keeps only the EXITS of a supertile in LDS (dense ids, SCAP of them); a supertile holding more
    is taken by the positional kernel instead, decided on the device per supertile.  Forced here by lowering the
    capacity: every supertile (0) / the fuller ones (5000) take the fallback; single handle, deferred, row blocks."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip, dist

    d8 = oracle.synth_d8(2300, 2600, seed=8, tilt=100000, white=2, nodata_pct=10)
    exp, _, _ = oracle.upstream_area_cell(d8)
    monkeypatch.setenv("PFD_TEST_SCAP", scap)
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert np.array_equal(flw.upstream_area(), exp)
    h = _hip.RasterHandle(d8, 2300, 2600, deferred=True)
    assert np.array_equal(h.upstream_area_cell().reshape(2300, 2600), exp)
    h.close()
    assert np.array_equal(dist.upstream_area_blocks(d8, 3), exp)


@pytest.mark.parametrize("hyper_small", [None, "1"])
def test_level4_round_budget_miss(gpu_lib, oracle, monkeypatch, hyper_small):
    """Level 4 of the exit graph — and the flat level-3 forest a whole raster of few hypertiles takes instead (round 6) —
    issues a fixed number of doubling rounds without asking the host; too few (forced here) must be noticed at the end of
    the pass and repaired by a longer re-run.  Both forms of the single handle, and row blocks."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(4200, 4300, seed=9, tilt=1 << 26, white=2, nodata_pct=0)  # 3 x 3 hypertiles, long rivers
    exp, _, _ = oracle.upstream_area_cell(d8)
    if hyper_small:
        monkeypatch.setenv("PFD_HYPER_SMALL", hyper_small)
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert np.array_equal(flw.upstream_area(), exp)  # (no knob: the budget suffices)
    monkeypatch.setenv("PFD_TEST_ROUNDS4", "1")
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert np.array_equal(flw.upstream_area(), exp)
    assert np.array_equal(dist.upstream_area_blocks(d8, 2), exp)


@pytest.mark.parametrize("hyper_small", [None, "1"])
def test_both_level3_forms_of_a_small_raster(gpu_lib, oracle, monkeypatch, hyper_small):
    """10 x 11 hypertiles' worth of nothing special: the flat forest (default up to 64 hypertiles) and the hypertile solves
    (PFD_HYPER_SMALL) give the oracle's counts on a river and on a rough raster with nodata, eager and deferred."""
    from pyflwdir_amd import _hip

    if hyper_small:
        monkeypatch.setenv("PFD_HYPER_SMALL", hyper_small)
    for shape, kw in (((6200, 7100), dict(seed=3, tilt=1 << 26, white=2, nodata_pct=0)),
                      ((5000, 9000), dict(seed=4, tilt=100000, white=6, nodata_pct=30))):
        d8 = oracle.synth_d8(shape[0], shape[1], **kw)
        exp, _, _ = oracle.upstream_area_cell(d8)
        for deferred in (False, True):
            h = _hip.RasterHandle(d8, shape[0], shape[1], deferred=deferred)
            assert np.array_equal(h.upstream_area_cell().reshape(shape), exp)
            h.close()


@pytest.mark.parametrize("engine", ["exact", "levels", "exact:PFD_SCAN_UNFUSED", "exact:PFD_TAILS_RASTER",
                                    "exact:PFD_ROUNDS_EARLY", "exact:PFD_DSCAN_LDS", "exact:PFD_DSCAN_GLOBAL", "exact:PFD_TEST_FUSE_MIN=1048576"])
def test_exact_engine_accuflux(gpu_lib, oracle, monkeypatch, engine):
    """float / int accuflux through the exact-order engine (tile leaves + heavy-chain trunk, exact.hip) and,
    forced by PFD_EXACT_LEVELS=1, through the level engine: both bit-identical to the reference's serial
    loop, with and without in-domain nodata, rasters spanning many tiles.  The exact engine in its default form
    (chain ends listed tile by tile, gather + fold of the short chains fused in k_xtrunk_prescan) and with each of the
    round-6 choices switched the other way: the two-kernel gather / fold, the raster-ordered chain list, the "earliest
    round" labels of the short chains, the down-fold of the short chains through LDS for every operation / for none."""
    import pyflwdir_amd as pyflwdir

    O = oracle
    if engine == "levels":
        monkeypatch.setenv("PFD_EXACT_LEVELS", "1")
    if ":" in engine:
        engine, knob = engine.split(":")
        knob, _, val = knob.partition("=")
        monkeypatch.setenv(knob, val or "1")
    for shape, seed, kw in [((1500, 2100), 3, dict(tilt=1 << 26, white=2, nodata_pct=0)),
                            ((1024, 1024), 4, dict(tilt=100000, white=2, nodata_pct=30)),
                            ((700, 900), 5, dict(tilt=3000, white=2, nodata_pct=3))]:
        d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
        idxs_ds, idxs_pit, _ = O.from_array(d8)
        seq = O.idxs_seq(idxs_ds, idxs_pit)
        flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
        flw._h.set_profiling(True)
        w = O.synth_weights_f32(d8.size, seed=1)
        assert np.array_equal(flw.accuflux(w.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, w))
        ran = [s["name"] for s in flw._h.last_timing()]
        assert ("exact_accuflux_up" in ran) == (engine == "exact"), ran
        w64 = w.astype(np.float64) * 3.25
        w64[::17] = -9999.0
        assert np.array_equal(flw.accuflux(w64.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, w64))
        assert np.array_equal(flw.accuflux(w64.reshape(shape), direction="down").ravel(),
                              O.accuflux(idxs_ds, seq, w64, direction="down"))
        wi = (w * 1000).astype(np.int32)
        wi[::13] = -7
        assert np.array_equal(flw.accuflux(wi.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wi))
        wl = wi.astype(np.int64) * 100000
        assert np.array_equal(flw.accuflux(wl.reshape(shape)).ravel(), O.accuflux(idxs_ds, seq, wl))
        assert np.array_equal(flw.accuflux(w.reshape(shape), direction="down").ravel(),
                              O.accuflux(idxs_ds, seq, w, direction="down"))
        assert np.array_equal(flw.stream_order().ravel(), O.strahler_order(idxs_ds, seq))
        elev = O.synth_elev_f32(shape[0], shape[1], seed=seed, **kw)
        drain = (np.arange(d8.size) % 97 == 0).reshape(shape)
        assert np.array_equal(flw.hand(drain, elev).ravel(),
                              O.height_above_nearest_drain(idxs_ds, seq, drain.ravel(), elev.ravel()))


def test_basins_outlet_on_a_cycle(gpu_lib, oracle):
    """An outlet seeded on (or downstream of) a cycle: the cells draining into it never reach a pit, the
    reference never visits them (they are not in idxs_seq) and they keep label 0 — the tiled label query
    alone would stop at the outlet and label them (found by tools/stress_fuzz.py)."""
    import pyflwdir_amd as pyflwdir

    d8 = np.array([[4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 4],
                   [4, 2, 1, 4, 4, 4, 1, 255, 4, 32, 4, 128],
                   [4, 4, 1, 4, 4, 4, 4, 4, 4, 4, 64, 4],
                   [4, 4, 1, 4, 4, 4, 4, 32, 4, 8, 4, 4],
                   [2, 4, 4, 4, 4, 4, 4, 32, 4, 4, 4, 4]], np.uint8)
    idxs_ds, idxs_pit, _ = oracle.from_array(d8)
    seq = oracle.idxs_seq(idxs_ds, idxs_pit)
    assert (oracle.rank(idxs_ds)[0] == -1).any()  # the raster holds a cycle (cells 22, 34)
    oidx = np.array([5, 6, 7, 19, 20, 21, 32, 34, 35, 36, 37, 39, 40, 41, 50, 52, 54])
    oids = (np.arange(oidx.size) + 5).astype(np.uint16)
    exp = oracle.basins(idxs_ds, oidx.astype(idxs_ds.dtype), seq, oids)
    for first in ("basins", "uparea"):  # with and without an earlier operation that already knows about the cycle
        flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
        if first == "uparea":
            flw.upstream_area()
        assert np.array_equal(flw.basins(idxs=oidx, ids=oids).ravel(), exp)
    # a larger raster: synthetic rivers with injected cycles and outlets on them
    d8 = oracle.synth_d8(300, 400, seed=31, tilt=1 << 26, white=2, nodata_pct=5)
    d8[100, 100], d8[100, 101] = 1, 16
    d8[200, 50], d8[201, 51], d8[201, 50] = 2, 16, 64
    idxs_ds, idxs_pit, _ = oracle.from_array(d8)
    seq = oracle.idxs_seq(idxs_ds, idxs_pit)
    oidx = np.array([100 * 400 + 101, 201 * 400 + 50, 50 * 400 + 7, 299 * 400 + 3])
    oids = np.array([3, 4, 5, 6], np.uint32)
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert np.array_equal(flw.basins(idxs=oidx, ids=oids).ravel(), oracle.basins(idxs_ds, oidx.astype(idxs_ds.dtype), seq, oids))


def test_exact_plan_is_deterministic(gpu_lib):
    """The plan of the exact-order engine (leaf steps per tile) must not depend on thread timing: repeated builds
    on one raster give identical leaf steps.  (Regression: a missing barrier in k_plan_tile let the waves of a tile
    disagree on the first step's length about once in 10^5 tiles — tools/stress_exact.py found it as a sporadic
    wrong result / memory fault on 30000-cell-wide rasters.)"""
    import ctypes as C

    from pyflwdir_amd import _hip

    nrow, ncol = 14000, 15000
    d8 = _hip.synth_d8_device(nrow, ncol, seed=573, tilt=3000, white=2, nodata_pct=10)
    L = _hip.lib()
    L.pfd_debug_xplan.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    ref = None
    for _ in range(10):
        h = _hip.RasterHandle(d8, nrow, ncol, memspace=_hip.PFD_DEVICE)
        info = (C.c_int64 * 8)()
        lh = np.empty((nrow, ncol), np.uint8)
        _hip.check(L.pfd_debug_xplan(h._h, info, lh.ctypes.data_as(C.c_void_p)))
        h.close()
        assert info[0] == 1
        if ref is None:
            ref = lh
        else:
            assert np.array_equal(lh, ref)
    d8.free()
