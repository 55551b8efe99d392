"""Multi-GPU protocol, exercised on ONE GPU: the raster is cut into row blocks (one handle per
block, all on device 0) and solved with pfd_upstream_area_cell_blocks — the same kernels, records
and interface solve as the RCCL path, with device copies in place of ncclAllGather.  The result
must be identical to the single-handle result and to the oracle for every block count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,seed,kw", [
    ((700, 900), 11, dict(tilt=1 << 26, white=2, nodata_pct=0)),      # long rivers crossing every block
    ((1030, 517), 12, dict(tilt=100000, white=2, nodata_pct=30)),     # all 8 directions: flow crosses both ways
    ((333, 2100), 13, dict(tilt=1 << 26, white=2, nodata_pct=20)),
])
@pytest.mark.parametrize("nblocks", [1, 2, 3, 5, 8])
def test_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks):
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(shape[0], shape[1], seed=seed, **kw)
    exp, _, _ = oracle.upstream_area_cell(d8)
    got = dist.upstream_area_blocks(d8, nblocks)
    assert got.shape == exp.shape and got.dtype == np.int32
    assert np.array_equal(got, exp)


def test_blocks_on_reference_rasters(gpu_lib, manifest):
    """Real rasters from the reference (Rhine, 160x200 fixture) cut into blocks."""
    from golden_util import Case
    from pyflwdir_amd import dist

    for name in ("rhine", "flwdir_large", "flwdir0"):
        case = Case(name, manifest)
        for nb in (2, 4, 7):
            if case.shape[0] < nb:
                continue
            case.check("uparea_cell", dist.upstream_area_blocks(case.d8, nb))


def test_blocks_thin_rows(gpu_lib, oracle):
    """Blocks of a single row each (every row is first AND last row of its block)."""
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(6, 300, seed=3, tilt=100000, white=2, nodata_pct=10)
    exp, _, _ = oracle.upstream_area_cell(d8)
    assert np.array_equal(dist.upstream_area_blocks(d8, 6), exp)


def test_blocks_reject_cycles(gpu_lib, oracle):
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(200, 200, seed=5)
    d8[100, 50], d8[100, 51] = 1, 16
    with pytest.raises(NotImplementedError, match="cycles"):
        dist.upstream_area_blocks(d8, 2)


def test_split_phase_api(gpu_lib, oracle):
    """begin()/finish(): the caller moves the boundary records itself (host transport of the multi-GPU path)."""
    from pyflwdir_amd import _hip, dist

    d8 = oracle.synth_d8(500, 640, seed=17, tilt=100000, white=2, nodata_pct=15)
    exp, _, _ = oracle.upstream_area_cell(d8)
    nb = 3
    handles, outs, recs = [], [], []
    for b, (r0, r1) in enumerate(dist.block_rows(d8.shape[0], nb)):
        a, e = dist.block_slice(d8.shape[0], nb, b)
        handles.append(_hip.RasterHandle(d8[a:e], r1 - r0, d8.shape[1], halo=dist.halo_of(b, nb)))
    for h in handles:
        o, r = _hip.upstream_area_cell_begin(h)
        outs.append(o)
        recs.append(r)
    allrec = np.stack(recs)
    for b, h in enumerate(handles):
        assert _hip.upstream_area_cell_finish(h, allrec, nb, b)
    got = np.concatenate([o.reshape(-1, d8.shape[1]) for o in outs])
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("rows_per_block,nblocks", [(2100, 2), (1050, 4), (513, 3), (1024, 2), (577, 5), (4100, 2)])
def test_blocks_supertile_edge_geometries(gpu_lib, oracle, rows_per_block, nblocks):
    """The first exit-graph solve of a block delivers only where the halo sinks need it (supertile rows
    of tile rows 0-1 and ntr-2..ntr-1).  Geometries where the last supertile row holds one or two tile
    rows, where a block spans 2..9 supertile rows, and where flow weaves across the block border."""
    from pyflwdir_amd import dist

    nrow, ncol = rows_per_block * nblocks, 700
    for kw in (dict(tilt=1 << 26, white=2, nodata_pct=0), dict(tilt=100000, white=2, nodata_pct=15)):
        d8 = oracle.synth_d8(nrow, ncol, seed=21, **kw)
        exp, _, _ = oracle.upstream_area_cell(d8)
        assert np.array_equal(dist.upstream_area_blocks(d8, nblocks), exp)
        assert np.array_equal(dist.upstream_area_blocks(d8, nblocks, deferred=True), exp)


@pytest.mark.parametrize("shape,seed,kw,nblocks", [
    ((900, 700), 41, dict(tilt=1 << 26, white=2, nodata_pct=0), 2),
    ((1300, 1100), 42, dict(tilt=100000, white=2, nodata_pct=25), 3),
    ((2100, 1500), 43, dict(tilt=3000, white=2, nodata_pct=5), 5),
    ((1600, 900), 44, dict(tilt=1 << 26, white=2, nodata_pct=10), 8),
    ((64, 300), 45, dict(tilt=100000, white=2, nodata_pct=0), 8),   # 8 rows per block
])
def test_basins_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks):
    """BASELINE config 5 shape of work: basins from 1000 outlets (largest upstream areas, so many of them are
    nested in one another's basins) on a raster split into row blocks == the oracle on the whole raster."""
    from pyflwdir_amd import dist

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    upa = O.upstream_area_cell(d8)[0].ravel()
    rng = np.random.default_rng(seed)
    k = min(1000, int((upa > 0).sum()) // 4)
    big = np.argsort(upa)[-k // 2:]                                  # nested along the main stems
    rnd = rng.choice(np.flatnonzero(upa > 0), size=k - big.size, replace=False)
    outl = np.unique(np.concatenate([big, rnd]))
    rng.shuffle(outl)
    for ids in (np.arange(1, outl.size + 1, dtype=np.uint32), (rng.permutation(outl.size) + 7).astype(np.int16),
                (rng.integers(1, 2**62, outl.size)).astype(np.int64)):
        exp = O.basins(idxs_ds, outl.astype(idxs_ds.dtype), seq, ids)
        got = dist.basins_blocks(d8, nblocks, outl, ids)
        assert got.dtype == ids.dtype
        assert np.array_equal(got.ravel(), exp)
    # default: one basin per pit
    assert np.array_equal(dist.basins_blocks(d8, nblocks, idxs_pit).ravel(), O.basins(idxs_ds, idxs_pit, seq))


@pytest.mark.parametrize("shape,seed,kw,nblocks,edtype", [
    ((900, 700), 51, dict(tilt=1 << 26, white=2, nodata_pct=0), 2, np.float32),
    ((1300, 1100), 52, dict(tilt=100000, white=2, nodata_pct=25), 3, np.float32),
    ((2100, 1500), 53, dict(tilt=3000, white=2, nodata_pct=5), 5, np.float64),
    ((1600, 900), 54, dict(tilt=1 << 26, white=2, nodata_pct=10), 8, np.float32),
    ((64, 300), 55, dict(tilt=100000, white=2, nodata_pct=0), 8, np.float64),   # 8 rows per block
])
def test_hand_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks, edtype):
    """BASELINE config 5's second operation over row blocks: HAND (reference pyflwdir/dem.py:299-330) of a raster
    split into 2-8 row blocks == the oracle on the whole raster, BIT FOR BIT — a path that crosses a block edge
    continues the neighbour's float64 sum (sharded HAND).  Sparse drains make paths cross several edges."""
    from pyflwdir_amd import dist

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    upa = O.upstream_area_cell(d8)[0].ravel()
    elev = O.synth_elev_f32(shape[0], shape[1], seed=seed, **kw).astype(edtype).ravel()
    for thr in (np.percentile(upa[upa > 0], 80), np.percentile(upa[upa > 0], 99.7), upa.max() + 1):
        drain = upa > thr  # the last one: no drain cell at all, every path runs to its pit
        exp = O.height_above_nearest_drain(idxs_ds, seq, drain, elev)
        got, iters = dist.hand_blocks(d8, nblocks, drain, elev)
        assert got.dtype == np.float64 and iters >= 1
        assert np.array_equal(got.ravel().view(np.uint64), exp.view(np.uint64)), (thr, iters)
    # all blocks at once == the whole-raster engines
    import pyflwdir_amd as pyflwdir

    flw = pyflwdir.from_array(d8, ftype="d8")
    assert np.array_equal(flw.hand(drain.reshape(shape), elev.reshape(shape)).ravel().view(np.uint64), got.ravel().view(np.uint64))


def test_front_end_cuts_huge_rasters_into_row_blocks(gpu_lib, oracle, monkeypatch):
    """FlwdirRaster.basins / .hand on a raster beyond 32-bit cell indices run the row-block protocols inside the one
    process (the threshold is lowered here): same results as the oracle on the whole raster, bit for bit."""
    import pyflwdir_amd as pyflwdir

    O = oracle
    shape = (1500, 1100)
    d8 = O.synth_d8(shape[0], shape[1], seed=61, tilt=100000, white=2, nodata_pct=15)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    upa = O.upstream_area_cell(d8)[0]
    elev = O.synth_elev_f32(shape[0], shape[1], seed=61, tilt=100000, white=2, nodata_pct=15)
    drain = upa > np.percentile(upa[upa > 0], 95)
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "400000")  # 1.65 Mcells -> 5 row blocks
    assert flw._row_blocks_needed() == 5
    got = flw.hand(drain, elev)
    exp = O.height_above_nearest_drain(idxs_ds, seq, drain.ravel(), elev.ravel())
    assert np.array_equal(got.ravel().view(np.uint64), exp.view(np.uint64))
    outl = np.argsort(upa.ravel())[-300:]
    ids = (np.arange(outl.size) + 11).astype(np.uint32)
    assert np.array_equal(flw.basins(idxs=outl, ids=ids).ravel(), O.basins(idxs_ds, outl.astype(idxs_ds.dtype), seq, ids))
    assert np.array_equal(flw.basins().ravel(), O.basins(idxs_ds, idxs_pit, seq))
    # the up-sweeps: float32 accuflux, upstream_area in km2 (float64 row areas of a lat/lon grid), Strahler order
    data = (np.random.default_rng(5).random(shape) * 2.5).astype(np.float32)
    exp = O.accuflux(idxs_ds, seq, data.ravel(), nodata=-9999)
    assert np.array_equal(flw.accuflux(data).ravel().view(np.uint32), exp.view(np.uint32))
    assert np.array_equal(flw.stream_order().ravel(), O.strahler_order(idxs_ds, seq))
    exp = O.accuflux(idxs_ds, seq, data.ravel(), nodata=-9999, direction="down")
    assert np.array_equal(flw.accuflux(data, direction="down").ravel().view(np.uint32), exp.view(np.uint32))
    assert np.array_equal(flw.stream_distance().ravel(), O.stream_distance(idxs_ds, seq, shape[1], real_length=False))
    monkeypatch.delenv("PFD_TEST_BIG_CELLS")
    from pyflwdir_amd import gis

    tr = gis.Affine(0.01, 0.0, 3.0, 0.0, -0.01, 52.0)  # 15 x 11 degrees around 45 N
    whole = pyflwdir.from_array(d8, ftype="d8", cache=False, latlon=True, transform=tr)
    blocked = pyflwdir.from_array(d8, ftype="d8", cache=False, latlon=True, transform=tr)
    exp_km2 = whole.upstream_area("km2")
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "400000")
    got_km2 = blocked.upstream_area("km2")
    assert got_km2.dtype == exp_km2.dtype and np.array_equal(got_km2.view(np.uint64), exp_km2.view(np.uint64))
    got_m = blocked.stream_distance(unit="m")
    monkeypatch.delenv("PFD_TEST_BIG_CELLS")
    exp_m = whole.stream_distance(unit="m")
    assert got_m.dtype == np.float32 and np.array_equal(got_m.view(np.uint32), exp_m.view(np.uint32))


@pytest.mark.parametrize("shape,seed,kw,nblocks,dtype", [
    ((900, 700), 71, dict(tilt=100000, white=2, nodata_pct=20), 2, np.float32),   # all 8 directions: flow crosses both ways
    ((1200, 800), 72, dict(tilt=1 << 26, white=2, nodata_pct=0), 3, np.float32),
    ((2100, 1500), 73, dict(tilt=3000, white=2, nodata_pct=5), 5, np.float64),
    ((1600, 900), 74, dict(tilt=1 << 26, white=2, nodata_pct=10), 8, np.float32),
    ((64, 300), 75, dict(tilt=100000, white=2, nodata_pct=0), 8, np.float32),    # 8 rows per block
    ((700, 900), 76, dict(tilt=100000, white=2, nodata_pct=10), 4, np.int32),
    ((700, 900), 77, dict(tilt=100000, white=2, nodata_pct=10), 3, np.int64),
])
def test_accuflux_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks, dtype):
    """accuflux (direction "up", reference pyflwdir/streams.py:15-41) of a raster split into 2-8 row blocks == the
    oracle on the whole raster BIT FOR BIT, floats included: a boundary cell adds the halo cells draining into it in
    the serial loop's position.  With a nodata value in the payload, and the all-cell local-equation check."""
    from pyflwdir_amd import dist

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        data = (rng.random(d8.size) * 3.7 + 1e-3).astype(dtype)
    else:
        data = rng.integers(0, 1000, d8.size).astype(dtype)
    data[rng.random(d8.size) < 0.001] = -9999  # nodata cells poison nothing: the reference skips the addition
    exp = O.accuflux(idxs_ds, seq, data, nodata=-9999)
    got, rounds, bad = dist.accuflux_blocks(d8, nblocks, data, (-9999, -9999.0, 1), verify=True)
    assert got.dtype == dtype and rounds >= 1 and bad == 0
    assert np.array_equal(got.ravel().view(np.uint8), exp.view(np.uint8)), rounds
    # one value per row (upstream_area in area units): float64 areas of a lat/lon grid
    if dtype == np.float64:
        rows = np.cos(np.linspace(-1.2, 1.2, shape[0])) * 1234.5
        exp = O.accuflux(idxs_ds, seq, np.repeat(rows, shape[1]), nodata=-9999)
        got, _, bad = dist.accuflux_blocks(d8, nblocks, rows, (-9999, -9999.0, 1), by_row=True, verify=True)
        assert bad == 0 and np.array_equal(got.ravel().view(np.uint64), exp.view(np.uint64))


@pytest.mark.parametrize("shape,seed,kw,nblocks", [
    ((900, 700), 81, dict(tilt=100000, white=2, nodata_pct=20), 2),
    ((1200, 800), 82, dict(tilt=1 << 26, white=2, nodata_pct=0), 3),
    ((1600, 900), 83, dict(tilt=3000, white=2, nodata_pct=10), 8),
    ((40, 300), 84, dict(tilt=100000, white=2, nodata_pct=0), 8),
])
def test_strahler_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks):
    """Strahler order (reference pyflwdir/streams.py:228-269) over row blocks == the oracle, with and without a mask."""
    from pyflwdir_amd import dist

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    upa = O.upstream_area_cell(d8)[0].ravel()
    for mask in (None, (upa > 5).astype(np.uint8)):
        exp = O.strahler_order(idxs_ds, seq, mask)
        got, rounds, bad = dist.strahler_blocks(d8, nblocks, mask, verify=True)
        assert got.dtype == np.uint8 and bad == 0
        assert np.array_equal(got.ravel(), exp), rounds


def test_up_block_verifier_sees_a_wrong_cell(gpu_lib, oracle):
    """The local-equation check of a blocked result is not vacuous: one changed own cell is reported (itself and the
    cell it drains into)."""
    from pyflwdir_amd import _hip, dist

    O = oracle
    d8 = O.synth_d8(500, 400, seed=91, tilt=100000, white=2, nodata_pct=5)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    data = np.random.default_rng(1).random(d8.size).astype(np.float32)
    exp = O.accuflux(idxs_ds, O.idxs_seq(idxs_ds, idxs_pit), data, nodata=-9999).reshape(d8.shape)
    a, e = dist.block_slice(500, 2, 1)
    h = _hip.RasterHandle(d8[a:e], 500 - (a + 1), 400, halo=dist.halo_of(1, 2))
    try:
        out = np.ascontiguousarray(exp[a:e]).copy()
        seed = np.concatenate([exp[a], np.zeros(400, np.float32)])
        payload = np.ascontiguousarray(data.reshape(d8.shape)[a:e])
        code = _hip._PAYLOAD_CODE[np.dtype(np.float32)]
        assert h.accuflux_block(payload, code, seed, out, -9999, -9999.0, 1, verify=True)[1] == 0
        r, c = np.argwhere((d8[a + 1:e] != 247) & (d8[a + 1:e] != 0))[1234]
        out[r + 1, c] += np.float32(1.0)
        assert h.accuflux_block(payload, code, seed, out, -9999, -9999.0, 1, verify=True)[1] == 2
    finally:
        h.close()


def test_up_blocks_report_a_cycle_through_the_block_edges(gpu_lib, oracle):
    """A cycle that crosses a block edge never settles (its sums grow with every exchange): the iteration is bounded
    and says so; a cycle inside one block is harmless (its cells are in no ordering and keep their input, like in the
    reference)."""
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(200, 120, seed=5, tilt=100000, white=2, nodata_pct=0)
    data = np.ones(d8.shape, np.float32)
    inside = d8.copy()
    inside[40, 50], inside[40, 51] = 1, 16        # E <-> W inside block 0
    got, _, bad = dist.accuflux_blocks(inside, 2, data, (-9999, -9999.0, 1), verify=True)
    idxs_ds, idxs_pit, _ = oracle.from_array(inside)
    exp = oracle.accuflux(idxs_ds, oracle.idxs_seq(idxs_ds, idxs_pit), data.ravel(), nodata=-9999)
    assert np.array_equal(got.ravel().view(np.uint32), exp.view(np.uint32))
    assert bad > 0  # (the cells on the cycle keep their input: the local equations do not hold there, and the check says so)
    across = d8.copy()
    across[99, 50], across[100, 50] = 4, 64       # S <-> N across the edge between the blocks (rows 99 | 100)
    with pytest.raises(NotImplementedError, match="did not settle"):
        dist.accuflux_blocks(across, 2, data, (-9999, -9999.0, 1), max_iter=12)
    # (a Strahler order can settle on such a cycle, with values where the reference has none: the front end therefore
    #  refuses rasters with cycles before it cuts them into blocks)
    import pyflwdir_amd as pyflwdir

    flw = pyflwdir.from_array(across, ftype="d8", cache=False)
    import os
    os.environ["PFD_TEST_BIG_CELLS"] = "10000"
    try:
        assert flw._row_blocks_needed() > 1
        for call in (lambda: flw.stream_order(), lambda: flw.accuflux(data), lambda: flw.upstream_area("km2")):
            with pytest.raises(NotImplementedError, match="cycle"):
                call()
    finally:
        del os.environ["PFD_TEST_BIG_CELLS"]


def test_row_blocks_on_the_level_engine(gpu_lib, oracle, monkeypatch):
    """Row blocks sweep with the exact-order engine by default; the level engine stays the fallback (a block whose plan
    cannot be built: a cycle inside the block, PFD_BLOCK_LEVELS) and must give the same bits."""
    from pyflwdir_amd import dist

    O = oracle
    shape = (900, 700)
    d8 = O.synth_d8(shape[0], shape[1], seed=97, tilt=100000, white=2, nodata_pct=10)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    upa = O.upstream_area_cell(d8)[0].ravel()
    elev = O.synth_elev_f32(shape[0], shape[1], seed=97, tilt=100000, white=2, nodata_pct=10).ravel()
    drain = upa > np.percentile(upa[upa > 0], 97)
    data = (np.random.default_rng(3).random(d8.size) * 2).astype(np.float32)
    exp_h = O.height_above_nearest_drain(idxs_ds, seq, drain, elev)
    exp_a = O.accuflux(idxs_ds, seq, data, nodata=-9999)
    exp_s = O.strahler_order(idxs_ds, seq)
    for knob in (None, "1"):
        if knob:
            monkeypatch.setenv("PFD_BLOCK_LEVELS", knob)
        got_h, _ = dist.hand_blocks(d8, 3, drain, elev)
        got_a, _, bad_a = dist.accuflux_blocks(d8, 3, data, (-9999, -9999.0, 1), verify=True)
        got_s, _, bad_s = dist.strahler_blocks(d8, 3, verify=True)
        got_d, _, bad_d = dist.accuflux_blocks(d8, 3, data, (-9999, -9999.0, 1), verify=True, direction="down")
        assert bad_d == 0 and np.array_equal(got_d.ravel().view(np.uint32),
                                             O.accuflux(idxs_ds, seq, data, nodata=-9999, direction="down").view(np.uint32)), knob
        got_l, _, _ = dist.stream_distance_blocks(d8, 3)
        assert np.array_equal(got_l.ravel(), O.stream_distance(idxs_ds, seq, shape[1], real_length=False)), knob
        assert np.array_equal(got_h.ravel().view(np.uint64), exp_h.view(np.uint64)), knob
        assert bad_a == 0 and np.array_equal(got_a.ravel().view(np.uint32), exp_a.view(np.uint32)), knob
        assert bad_s == 0 and np.array_equal(got_s.ravel(), exp_s), knob
    monkeypatch.delenv("PFD_BLOCK_LEVELS")
    # a cycle inside one block: no plan for that block (level engine there), the other blocks keep theirs
    cyc = d8.copy()
    r, c = np.argwhere((d8[100:200, 100:600] != 247))[50] + (100, 100)
    cyc[r, c], cyc[r, c + 1] = 1, 16
    idxs_ds, idxs_pit, _ = O.from_array(cyc)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    exp_h = O.height_above_nearest_drain(idxs_ds, seq, drain, elev)
    got_h, _ = dist.hand_blocks(cyc, 3, drain, elev)
    assert np.array_equal(got_h.ravel().view(np.uint64), exp_h.view(np.uint64))


@pytest.mark.parametrize("shape,seed,kw,nblocks,dtype", [
    ((900, 700), 101, dict(tilt=100000, white=2, nodata_pct=20), 2, np.float32),
    ((1600, 900), 102, dict(tilt=1 << 26, white=2, nodata_pct=10), 8, np.float32),
    ((1100, 800), 103, dict(tilt=3000, white=2, nodata_pct=5), 5, np.float64),
    ((64, 300), 104, dict(tilt=100000, white=2, nodata_pct=0), 8, np.int32),
])
def test_accuflux_down_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks, dtype):
    """accuflux(direction="down") (reference pyflwdir/streams.py:44-70: every cell adds the value of its downstream
    cell) over row blocks == the oracle on the whole raster, bit for bit; every cell's local equation checked."""
    from pyflwdir_amd import dist

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    rng = np.random.default_rng(seed)
    data = (rng.random(d8.size) * 3.1).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(0, 50, d8.size).astype(dtype)
    data[rng.random(d8.size) < 0.001] = -9999
    exp = O.accuflux(idxs_ds, seq, data, nodata=-9999, direction="down")
    got, rounds, bad = dist.accuflux_blocks(d8, nblocks, data, (-9999, -9999.0, 1), verify=True, direction="down")
    assert got.dtype == dtype and rounds >= 1 and bad == 0
    assert np.array_equal(got.ravel().view(np.uint8), exp.view(np.uint8)), rounds


@pytest.mark.parametrize("shape,seed,kw,nblocks", [
    ((900, 700), 111, dict(tilt=100000, white=2, nodata_pct=20), 2),
    ((1600, 900), 112, dict(tilt=1 << 26, white=2, nodata_pct=10), 8),
    ((64, 300), 113, dict(tilt=100000, white=2, nodata_pct=0), 8),
])
def test_stream_distance_blocks_vs_oracle(gpu_lib, oracle, shape, seed, kw, nblocks):
    """stream_distance (reference pyflwdir/streams.py:272-315) over row blocks == the oracle on the whole raster: cell
    counts and float32 metres on a lat/lon grid (row-dependent step lengths: every block gets its slice of the table),
    with and without a mask; every cell's local equation checked."""
    from pyflwdir_amd import dist, gis

    O = oracle
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    upa = O.upstream_area_cell(d8)[0].ravel()
    tr = gis.Affine(0.01, 0.0, 3.0, 0.0, -0.01, 52.0)
    tab = gis.step_length_table(shape[0], True, tr)
    for mask in (None, (upa > 30).astype(np.uint8)):
        exp = O.stream_distance(idxs_ds, seq, shape[1], mask=mask, real_length=False)
        got, rounds, bad = dist.stream_distance_blocks(d8, nblocks, mask)
        assert got.dtype == np.int32 and bad is None and rounds >= 1
        assert np.array_equal(got.ravel(), exp)
        exp = O.stream_distance(idxs_ds, seq, shape[1], mask=mask, real_length=True, latlon=True, transform=tuple(tr)[:6])
        got, _, bad = dist.stream_distance_blocks(d8, nblocks, mask, tab, verify=True)
        assert got.dtype == np.float32 and bad == 0
        assert np.array_equal(got.ravel().view(np.uint32), exp.view(np.uint32))


def test_block_sweeps_edge_cases(gpu_lib, oracle):
    """Degenerate shapes of the row-block sweeps: one and three columns, one row per block, a single block, blocks that
    start on a tile edge, a block of nothing but nodata, nodata rows along a block edge — accuflux (both directions),
    Strahler, stream distance and HAND == the oracle on the whole raster."""
    from pyflwdir_amd import dist

    O = oracle
    rng = np.random.default_rng(5)

    def check(d8, nb, tag):
        idxs_ds, idxs_pit, _ = O.from_array(d8)
        seq = O.idxs_seq(idxs_ds, idxs_pit)
        data = (rng.random(d8.size) * 3).astype(np.float32)
        for direction in ("up", "down"):
            exp = O.accuflux(idxs_ds, seq, data, nodata=-9999, direction=direction)
            got, _, bad = dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), verify=True, direction=direction)
            assert bad == 0 and np.array_equal(got.ravel().view(np.uint32), exp.view(np.uint32)), (tag, d8.shape, direction)
        got, _, bad = dist.strahler_blocks(d8, nb, verify=True)
        assert bad == 0 and np.array_equal(got.ravel(), O.strahler_order(idxs_ds, seq)), (tag, d8.shape)
        got, _, _ = dist.stream_distance_blocks(d8, nb)
        assert np.array_equal(got.ravel(), O.stream_distance(idxs_ds, seq, d8.shape[1], real_length=False)), (tag, d8.shape)
        elev = rng.random(d8.size).astype(np.float32) * 100
        drain = O.upstream_area_cell(d8)[0].ravel() > 3
        got, _ = dist.hand_blocks(d8, nb, drain, elev)
        exp = O.height_above_nearest_drain(idxs_ds, seq, drain, elev)
        assert np.array_equal(got.ravel().view(np.uint64), exp.view(np.uint64)), (tag, d8.shape)

    for shape, nb in (((40, 1), 4), ((40, 3), 5), ((9, 200), 9), ((130, 65), 2), ((300, 5), 7), ((64, 64), 1), ((129, 129), 3)):
        check(O.synth_d8(shape[0], shape[1], seed=int(rng.integers(1, 999)), tilt=100000, white=2, nodata_pct=10), nb, "random")
    d8 = O.synth_d8(300, 200, seed=7, tilt=100000, white=2, nodata_pct=0)
    d8[100:200] = 247
    check(d8, 3, "middle block all nodata")
    d8 = O.synth_d8(300, 200, seed=8, tilt=100000, white=2, nodata_pct=0)
    d8[99:101] = 247
    check(d8, 3, "nodata rows at the first edge")
    check(O.synth_d8(256, 256, seed=9, tilt=1 << 26, white=2, nodata_pct=0), 4, "tile-aligned blocks")


def test_whole_raster_operations_refuse_row_block_handles(gpu_lib, oracle):
    """A whole-raster entry point on a row-block handle would return block-local values (halo cells as roots, nothing
    entering from the neighbours): only the *_block / _begin / _finish / _blocks / _dist entry points accept one."""
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(300, 200, seed=4, tilt=100000, white=2, nodata_pct=5).reshape(300, 200)
    h = _hip.RasterHandle(d8[99:201], 100, 200, halo=(1, 1))
    n = 102 * 200
    f32 = np.ones(n, np.float32)
    calls = {
        "accuflux": lambda: h.accuflux(f32, _hip.PFD_F32, nodata_f=-9999.0),
        "strahler": lambda: h.strahler(),
        "hand": lambda: h.hand(np.zeros(n, np.uint8), f32, _hip.PFD_F32),
        "stream_distance": lambda: h.stream_distance(),
        "rank": lambda: h.rank(),
        "idxs_seq": lambda: h.idxs_seq(np.int32),
        "order_cells": lambda: h.order_cells(),
        "upstream_area_cell": lambda: h.upstream_area_cell(),
        "upstream_area_cell_levels": lambda: h.upstream_area_cell(engine="levels"),
        "basins": lambda: h.basins(np.array([5], np.int64), np.array([1], np.uint32)),
    }
    for name, call in calls.items():
        with pytest.raises(NotImplementedError, match="row-block handle"):
            call()
    # the block entry points still work on the same handle
    out = np.zeros(n, np.float32)
    h.accuflux_block(f32, _hip.PFD_F32, np.zeros(2 * 200, np.float32), out)
    h.close()


def test_hand_blocks_refuse_non_finite_elevations(gpu_lib, oracle):
    from pyflwdir_amd import dist

    d8 = oracle.synth_d8(120, 90, seed=1, tilt=100000, white=2).reshape(120, 90)
    elev = np.ones((120, 90), np.float32)
    elev[50, 40] = np.inf
    with pytest.raises(NotImplementedError, match="finite elevations"):
        dist.hand_blocks(d8, 2, np.zeros((120, 90), np.uint8), elev)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,dtype", [("accuflux", np.float32), ("accuflux", np.float64), ("strahler", np.uint8)])
def test_block_updates_fold_only_below_changed_seeds(gpu_lib, oracle, kind, dtype):
    """pfd_set_block_update: the second and later sweeps of a block (other halo seeds, result in place) take the update
    path — no tile pass, only the chains below a changed seed — and give the full sweep's bits; an update request
    without a kept sweep of the operation, or into another buffer, is a full sweep."""
    from pyflwdir_amd import _hip, dist

    O = oracle
    nrow, ncol, nb = 900, 700, 3
    d8 = O.synth_d8(nrow, ncol, seed=171, tilt=100000, white=2, nodata_pct=10)
    rng = np.random.default_rng(3)
    data = (rng.random(d8.shape) * 2.5 + 1e-3).astype(dtype) if kind == "accuflux" else None

    def run(incremental):
        blocks, names = [], []
        for b, (r0, r1) in enumerate(dist.block_rows(nrow, nb)):
            a, e = dist.block_slice(nrow, nb, b)
            h = _hip.RasterHandle(d8[a:e], r1 - r0, ncol, halo=dist.halo_of(b, nb))
            h.set_profiling(True)
            blk = dist._UpBlock(h, kind, dtype, payload=None if data is None else data[a:e], nodata=(-9999, -9999.0, 1))
            blk.incremental = incremental
            blocks.append(blk)
        try:
            seeds = [np.zeros(2 * ncol, dtype) for _ in range(nb)]
            for it in range(16):
                swept = []
                for b, blk in enumerate(blocks):
                    swept.append(blk.sweep(seeds[b]))
                    if swept[-1]:
                        names.append((it, [s["name"] for s in blk.h.last_timing()]))
                if not any(swept):
                    break
                for b in range(nb):
                    if b > 0:
                        seeds[b][:ncol] = blocks[b - 1].brows[1]
                    if b + 1 < nb:
                        seeds[b][ncol:] = blocks[b + 1].brows[0]
            bad = sum(blk.verify(seeds[b]) for b, blk in enumerate(blocks))
            return np.concatenate([blk.result() for blk in blocks], axis=0), names, bad
        finally:
            for blk in blocks:
                blk.close()

    full, names_full, bad_full = run(False)
    inc, names_inc, bad_inc = run(True)
    assert bad_full == 0 and bad_inc == 0
    assert np.array_equal(full.view(np.uint8), inc.view(np.uint8))
    assert all("exact_up_block_update" not in n for _, n in names_full)
    assert all(("exact_up_block_update" in n) == (it > 0) for it, n in names_inc), names_inc
    assert any(it > 0 for it, _ in names_inc)
    # an update request with nothing kept, and one into another buffer: full sweeps with the same result
    a, e = dist.block_slice(nrow, nb, 1)
    h = _hip.RasterHandle(d8[a:e], dist.block_rows(nrow, nb)[1][1] - dist.block_rows(nrow, nb)[1][0], ncol, halo=dist.halo_of(1, nb))
    try:
        h.set_profiling(True)
        blk = dist._UpBlock(h, kind, dtype, payload=None if data is None else data[a:e], nodata=(-9999, -9999.0, 1))
        blk.sweeps = 1  # (asks for an update straight away)
        seed = np.zeros(2 * ncol, dtype)
        seed[:ncol] = full[a]  # the final row above the block; the row below stays 0
        blk.sweep(seed)
        assert "exact_up_block_update" not in [s["name"] for s in h.last_timing()]
        first = blk.result().copy()
        other = _hip.DeviceBuffer(blk.out.nbytes, h.device)
        keep, blk.out = blk.out, other
        blk.swept_with = None
        blk.sweep(seed)
        assert "exact_up_block_update" not in [s["name"] for s in h.last_timing()]
        assert np.array_equal(blk.result().view(np.uint8), first.view(np.uint8))
        blk.swept_with = None
        seed[:ncol] = 0
        blk.sweep(seed)  # now an update of the sweep kept for `other`
        assert "exact_up_block_update" in [s["name"] for s in h.last_timing()]
        blk.out = keep
        other.free()
        blk.close(close_handle=False)
    finally:
        h.close()


@pytest.mark.gpu
def test_block_update_modes_refuse_a_host_result(gpu_lib, oracle):
    """pfd_set_block_update modes 1 / 2 keep pointers into `out` between calls: a HOST `out` is a temporary device buffer
    that is gone on return — refused with a message, the default mode takes it, and a refused call leaves
    nothing behind that a later device-resident update could mistake for a kept sweep."""
    from pyflwdir_amd import _hip, dist

    O = oracle
    nrow, ncol, nb = 600, 500, 2
    d8 = O.synth_d8(nrow, ncol, seed=5, tilt=100000, white=2, nodata_pct=5)
    data = (np.random.default_rng(1).random(d8.shape) + 0.01).astype(np.float32)
    a, e = dist.block_slice(nrow, nb, 1)
    r0, r1 = dist.block_rows(nrow, nb)[1]
    h = _hip.RasterHandle(d8[a:e], r1 - r0, ncol, halo=dist.halo_of(1, nb))
    try:
        seed = np.zeros(2 * ncol, np.float32)
        out0 = np.empty((e - a, ncol), np.float32)
        h.accuflux_block(data[a:e], _hip.PFD_F32, seed, out0, nodata_f=-9999.0, has_nodata=1)
        for mode in (1, 2):
            h.set_block_update(mode)
            out = np.empty((e - a, ncol), np.float32)
            with pytest.raises(Exception, match="DEVICE memory"):
                h.accuflux_block(data[a:e], _hip.PFD_F32, seed, out, nodata_f=-9999.0, has_nodata=1)
            with pytest.raises(Exception, match="DEVICE memory"):
                h.strahler_block(None, np.zeros(2 * ncol, np.uint8), np.empty((e - a, ncol), np.uint8))
        h.set_block_update(0)
        out = np.empty((e - a, ncol), np.float32)
        h.accuflux_block(data[a:e], _hip.PFD_F32, seed, out, nodata_f=-9999.0, has_nodata=1)
        assert np.array_equal(out.view(np.uint32), out0.view(np.uint32))
    finally:
        h.close()


def _zigzag(nrow, ncol, row):
    """A river that crosses the edge between `row` and `row + 1` at EVERY step (SE, NE, SE, ...): the interface forest of
    the row blocks holds one path of ncol - 1 hops; tributaries feed it from above and below."""
    d8 = np.full((nrow, ncol), 247, np.uint8)
    for c in range(ncol - 1):
        if c % 2 == 0:
            d8[row, c] = 2  # SE -> (row + 1, c + 1)
        else:
            d8[row + 1, c] = 128  # NE -> (row, c + 1)
    d8[row + ((ncol - 1) % 2), ncol - 1] = 0  # the river's pit
    d8[:row, ::2] = 4  # columns of S-flowing cells above the even crossings ...
    d8[row + 2:, 1::2] = 64  # ... and of N-flowing cells below the odd ones
    return d8


@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True])
def test_interface_paths_longer_than_the_chase(gpu_lib, oracle, monkeypatch, deferred):
    """The interface forest of a block pass is chased (one launch, paths of a handful of hops); a path with more hops than
    the chase allows — here a river zigzagging across a block edge 299 times — raises the miss bit and the pass is redone
    with the doubling rounds; PFD_TEST_IFACE_HOPS=1 forces that on an ordinary raster.  Results as the oracle's."""
    from pyflwdir_amd import dist

    d8 = _zigzag(100, 300, 49)  # block_rows(100, 2) = [0, 50), [50, 100): the river runs on rows 49 / 50
    exp = oracle.upstream_area_cell(d8)[0]
    assert exp.max() > 5000
    for nb in (2, 4):
        got = dist.upstream_area_blocks(d8, nb, deferred=deferred)
        assert np.array_equal(got, exp), nb
    d8 = oracle.synth_d8(1300, 900, seed=4, tilt=1 << 26, white=2, nodata_pct=3)
    exp = oracle.upstream_area_cell(d8)[0]
    monkeypatch.setenv("PFD_TEST_IFACE_HOPS", "1")
    assert np.array_equal(dist.upstream_area_blocks(d8, 5, deferred=deferred), exp)


@pytest.mark.gpu
@pytest.mark.parametrize("masked", [False, True])
def test_classic_stream_order_over_row_blocks(gpu_lib, oracle, monkeypatch, masked):
    """FlwdirRaster.stream_order(type="classic") beyond 32-bit cell indices (threshold lowered): one byte per cell about
    the downstream cell's main upstream cell instead of the index array, swapped once between neighbouring blocks, then
    seeded down-sweeps — the same orders as on one handle (whose classic order is pinned on the reference's goldens),
    every own cell's local equation checked."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import dist

    O = oracle
    shape = (1400, 900)
    d8 = O.synth_d8(shape[0], shape[1], seed=23, tilt=100000, white=2, nodata_pct=12)
    mask = None
    if masked:
        mask = (np.random.default_rng(2).random(shape) < 0.8)
    whole = pyflwdir.from_array(d8, ftype="d8", cache=False)
    exp = whole.stream_order(type="classic", mask=mask)
    assert exp.max() >= 5
    blocked = pyflwdir.from_array(d8, ftype="d8", cache=False)
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "300000")  # 1.26 Mcells -> 5 row blocks
    assert blocked._row_blocks_needed() == 5
    got = blocked.stream_order(type="classic", mask=mask)
    assert got.dtype == exp.dtype and np.array_equal(got, exp)
    monkeypatch.delenv("PFD_TEST_BIG_CELLS")
    upa = whole.upstream_area()
    m8 = None if mask is None else mask.astype(np.uint8)
    for nb in (2, 7):
        res, rounds, bad = dist.classic_blocks(d8, nb, upa, m8, verify=True)
        assert bad == 0 and rounds >= 1 and np.array_equal(res, exp), nb


@pytest.mark.gpu
def test_front_end_exports_in_row_slices(gpu_lib, oracle, monkeypatch):
    """Beyond 32-bit cell indices the front end assembles its cell-local exports — idxs_ds, idxs_pit (needed to CONSTRUCT
    the object), n_upstream, main_upstream, upstream_sum — from plain handles on row slices with one context row per inner
    side (threshold lowered here); add_pits edits the whole raster with 64-bit indices.  Everything equal to the
    one-handle object's, which the goldens pin."""
    import pyflwdir_amd as pyflwdir

    O = oracle
    shape = (1300, 700)
    d8 = O.synth_d8(shape[0], shape[1], seed=31, tilt=100000, white=2, nodata_pct=10)
    whole = pyflwdir.from_array(d8, ftype="d8", cache=False)
    data = (np.random.default_rng(4).random(shape) * 3).astype(np.float32)
    exp = dict(ds=whole.idxs_ds.copy(), pit=whole.idxs_pit.copy(), nup=whole.n_upstream.copy(), mu=whole.main_upstream().copy(),
               us=whole.upstream_sum(data).copy(), outlet=whole.idxs_outlet.copy())
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "200000")  # 0.91 Mcells: 5 row blocks, slices of 285 rows
    sliced = pyflwdir.from_array(d8, ftype="d8", cache=False)  # (the constructor reads idxs_pit)
    assert sliced._row_blocks_needed() == 5
    assert np.array_equal(sliced.idxs_pit, exp["pit"]) and np.array_equal(sliced.idxs_outlet, exp["outlet"])
    assert np.array_equal(sliced.idxs_ds, exp["ds"])
    assert np.array_equal(sliced.n_upstream, exp["nup"])
    assert np.array_equal(sliced.main_upstream(), exp["mu"])
    assert np.array_equal(sliced.upstream_sum(data).view(np.uint32), exp["us"].view(np.uint32))
    # add_pits, then the pit list and an accumulation through the row blocks
    new = np.argsort(whole.upstream_area().ravel())[-3:]
    sliced.add_pits(idxs=new)
    monkeypatch.delenv("PFD_TEST_BIG_CELLS")
    whole.add_pits(idxs=new)
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "200000")
    assert np.array_equal(sliced.idxs_pit, whole.idxs_pit)
    got = sliced.stream_order(type="classic")
    monkeypatch.delenv("PFD_TEST_BIG_CELLS")
    assert np.array_equal(got, whole.stream_order(type="classic"))
    assert np.array_equal(sliced.upstream_area(), whole.upstream_area())


@pytest.mark.gpu
@pytest.mark.parametrize("latlon", [False, True])
def test_front_end_order_ucat_snap_beyond_32_bit_indices(gpu_lib, oracle, monkeypatch, latlon):
    """The last operations that needed 32-bit cell indices: rank / idxs_seq / order_cells through csrc/order64.hip,
    ucat_area composed from the label query and the 64-bit sequence (float64 areas added in sequence order: bit for bit),
    snap with 64-bit walks — thresholds lowered, everything equal to the one-handle object's (which the goldens pin)."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    shape = (900, 1100)
    d8 = oracle.synth_d8(shape[0], shape[1], seed=37, tilt=100000, white=2, nodata_pct=9)
    tf = Affine(0.01, 0, 4.0, 0, -0.01, 52.0) if latlon else Affine(30.0, 0, 0, 0, -30.0, 0)
    whole = pyflwdir.from_array(d8, ftype="d8", transform=tf, latlon=latlon, cache=False)
    upa = whole.upstream_area()
    rng = np.random.default_rng(6)
    valid = np.flatnonzero(d8.ravel() != 247)
    outs = np.concatenate([np.argsort(upa.ravel())[-40:], rng.choice(valid, 60), np.flatnonzero(d8.ravel() == 247)[:2]])
    outs = np.concatenate([outs, outs[:5]]).astype(whole.idxs_ds.dtype)  # repeated outlets: the last label wins
    outs[7] = -1  # the reference's missing value
    outs = outs[: outs.size // 4 * 4].reshape(4, -1)
    exp = dict(rank=whole.rank.copy(), seq=whole.idxs_seq.copy(), n=whole.nnodes)
    for unit in ("cell", "km2"):
        exp[unit] = whole.ucat_area(outs, unit=unit)
    pts = rng.choice(valid, 200)
    streams = upa > 50
    exp["snap_down"] = whole.snap(idxs=pts, mask=streams, unit="m")
    exp["snap_up"] = whole.snap(idxs=pts[:50], mask=upa < 3, direction="up", max_length=40)
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "250000")
    monkeypatch.setenv("PFD_TEST_ORDER64", "1")
    big = pyflwdir.from_array(d8, ftype="d8", transform=tf, latlon=latlon, cache=False)
    assert big._row_blocks_needed() == 4 and big._wide()
    assert big.isvalid and big.nnodes == exp["n"]
    assert np.array_equal(big.rank, exp["rank"]) and np.array_equal(big.idxs_seq, exp["seq"])
    # ucat_area runs in the library on the whole raster (round 6: the tiled label query at any size + the float sums over
    # the 64-bit sequence in pieces, csrc/subgrid.hip ucat_float_wide) — pieces of 7001 and of 64 entries, and one piece
    big._h.set_profiling(True)
    for piece in ("7001", "64", None):
        if piece:
            monkeypatch.setenv("PFD_UCAT_PIECE", piece)
        else:
            monkeypatch.delenv("PFD_UCAT_PIECE")
        for unit in ("cell", "km2"):
            m, a = big.ucat_area(outs, unit=unit)
            names = [s["name"] for s in big._h.last_timing()]
            assert "tile_labels" in names and ("ucat_sums" in names) == (unit == "km2"), names
            assert m.dtype == exp[unit][0].dtype and np.array_equal(m, exp[unit][0]), unit
            assert a.dtype == exp[unit][1].dtype and a.shape == outs.shape, unit
            assert a.tobytes() == exp[unit][1].tobytes(), unit
    # the composed form (what a raster with cycles falls back to) stays pinned too
    idx64 = np.where(outs.ravel() == big._mv, -1, outs.ravel().astype(np.int64))
    from pyflwdir_amd import gis
    rows = np.ascontiguousarray(gis.area_rows(big.transform, big.shape, big.latlon, unit="m2") / gis.AREA_FACTORS["km2"])
    m, a = big._ucat_area_wide(idx64, rows)
    assert np.array_equal(m.reshape(shape), exp["km2"][0]) and a.tobytes() == exp["km2"][1].tobytes()
    # basins: one label query on the whole raster (any id width)
    for dt in (np.uint32, np.uint8, np.int64):
        ids = (np.arange(40) + 3).astype(dt)
        got = big.basins(idxs=np.argsort(upa.ravel())[-40:], ids=ids)
        assert "tile_labels" in [s["name"] for s in big._h.last_timing()]
        assert got.dtype == dt and np.array_equal(got, whole.basins(idxs=np.argsort(upa.ravel())[-40:], ids=ids))
    for key, got in (("snap_down", big.snap(idxs=pts, mask=streams, unit="m")),
                     ("snap_up", big.snap(idxs=pts[:50], mask=upa < 3, direction="up", max_length=40))):
        assert np.array_equal(got[0], exp[key][0]) and np.array_equal(got[1], exp[key][1]), key


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["accuflux_down", "distance", "classic", "strahler"])
def test_blocks_sweep_only_when_a_relevant_halo_value_changed(gpu_lib, oracle, monkeypatch, op):
    """The fixpoint iteration over row blocks compares only the halo values a block's result depends on
    (dist.relevant_halo): same results as with every halo value counted, in fewer sweeps — flow that crosses the blocks
    one way lets the downstream blocks of a down-sweep settle at once."""
    from pyflwdir_amd import dist

    shape, nb = (1400, 600), 5
    d8 = oracle.synth_d8(shape[0], shape[1], seed=5, tilt=400, white=2, nodata_pct=6)  # (a strong tilt: flow mostly one way)
    data = (np.random.default_rng(3).random(shape) * 4).astype(np.float32)
    upa = oracle.upstream_area_cell(d8)[0].reshape(shape)

    def run():
        if op == "accuflux_down":
            return dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), verify=True, direction="down")
        if op == "distance":
            return dist.stream_distance_blocks(d8, nb, verify=True)
        if op == "classic":
            return dist.classic_blocks(d8, nb, upa, verify=True)
        return dist.strahler_blocks(d8, nb, verify=True)

    got, rounds, bad = run()
    gated = list(dist.LAST_SWEEPS)
    monkeypatch.setattr(dist, "relevant_halo", lambda rows, halo, down: None)
    exp, rounds_all, bad_all = run()
    every = list(dist.LAST_SWEEPS)
    assert bad == 0 and bad_all == 0 and got.tobytes() == exp.tobytes()
    assert all(g <= e for g, e in zip(gated, every)) and sum(gated) <= sum(every), (gated, every)
    if op != "strahler":  # (the down-sweeps of this tilted raster: at least one block is spared a sweep)
        assert sum(gated) < sum(every), (gated, every)


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["accuflux_up", "accuflux_down", "distance", "strahler", "classic", "floodplains", "hand"])
def test_streamed_row_blocks_equal_resident_ones(gpu_lib, oracle, monkeypatch, op):
    """Row blocks whose state would not fit one GPU go through the device one at a time (dist._stream_blocks: a block is
    built, swept and released per sweep, its result waits on the host).  Forced here; bit for bit the resident result —
    which the other tests of this file pin against the oracle."""
    from pyflwdir_amd import dist

    shape, nb = (900, 500), 4
    d8 = oracle.synth_d8(shape[0], shape[1], seed=8, tilt=100000, white=2, nodata_pct=8)
    upa = oracle.upstream_area_cell(d8)[0].reshape(shape)
    data = (np.random.default_rng(5).random(shape) * 2).astype(np.float32)
    elev = oracle.synth_elev_f32(shape[0], shape[1], seed=8, tilt=100000, white=2, nodata_pct=8)

    def run():
        if op.startswith("accuflux"):
            return dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), direction=op.split("_")[1])[0]
        if op == "distance":
            return dist.stream_distance_blocks(d8, nb)[0]
        if op == "strahler":
            return dist.strahler_blocks(d8, nb)[0]
        if op == "classic":
            return dist.classic_blocks(d8, nb, upa)[0]
        if op == "floodplains":
            stream = (upa > 200).astype(np.uint8)
            hs = np.where(stream, upa.astype(np.float32) ** np.float32(0.3), 0).astype(np.float32)
            return dist.floodplains_blocks(d8, nb, elev, stream, hs)[0]
        return dist.hand_blocks(d8, nb, upa > 100, elev)[0]

    monkeypatch.setenv("PFD_TEST_STREAM_BLOCKS", "0")
    held = run()
    monkeypatch.setenv("PFD_TEST_STREAM_BLOCKS", "1")
    streamed = run()
    assert streamed.dtype == held.dtype and streamed.shape == held.shape and streamed.tobytes() == held.tobytes()
    if op != "hand":
        with pytest.raises(NotImplementedError, match="resident"):
            {"accuflux_up": lambda: dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), verify=True),
             "accuflux_down": lambda: dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), verify=True, direction="down"),
             "distance": lambda: dist.stream_distance_blocks(d8, nb, verify=True),
             "strahler": lambda: dist.strahler_blocks(d8, nb, verify=True),
             "classic": lambda: dist.classic_blocks(d8, nb, upa, verify=True),
             "floodplains": lambda: dist.floodplains_blocks(d8, nb, elev, (upa > 200).astype(np.uint8),
                                                            np.zeros(shape, np.float32), verify=True)}[op]()
