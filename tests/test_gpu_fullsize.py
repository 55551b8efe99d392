"""Parity at BASELINE.json's full sizes: what the oracle can still check in a minute or two
is compared cell by cell; beyond that the size-independent properties of the domain take over.

 * C3  30000 x 30000: float32 accuflux and Strahler order against the oracle (bit-exact), every cell.
 * C4  90000 x 90000 (8.1 Gcells, beyond 2^32 tile addressing): every cell's local equation
       upa == 1 + sum over the cells draining into it, -9999 on nodata, pit sum == n_valid (the reference's
       own invariant, tests/test_streams_basins.py:24-27) — on an acyclic raster these equations have one
       solution; the first rows against the oracle; and the 8-row-block run (the multi-GPU protocol, blocks
       held by this process) must produce the same checksum and pass the same checks block by block.
 * C4 again, the order-sensitive sweeps (level engine, 32-bit cell addressing): three row blocks of 2.7 Gcells with
       seeded halo rows — int32 accuflux of ones == the tiled engine's upstream area block by block, float32
       accuflux of row areas and the Strahler order pass every own cell's local equation, first rows vs the oracle.
 * the uint32 rung of the index ladder (2^31 .. 2^32 cells, reference pyflwdir.py:105-127): idxs_ds /
   idxs_pit exports against the oracle on sampled row bands.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RIVER = dict(seed=0, tilt=1 << 26, white=2, nodata_pct=0)


def test_c2_10000_upstream_area_vs_oracle(gpu_lib, oracle):
    """BASELINE configs[1] exactly as stated: 10000 x 10000 synthetic D8 (the bench's river raster), upstream cell
    counts on one GPU, bit-exact against the oracle's serial pipeline — through the FlwdirRaster surface and through
    the deferred handle bench.py times."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip

    n = 10000
    buf = _hip.synth_d8_device(n, n, seed=0, tilt=1 << 26, white=2, nodata_pct=0)
    d8 = buf.download(np.uint8, (n, n))
    exp, _, st = oracle.upstream_area_cell(d8)
    flw = pyflwdir.from_array(d8, ftype="d8")
    got = flw.upstream_area()
    assert got.dtype == np.int32 and np.array_equal(got, exp)
    h = _hip.RasterHandle(buf, n, n, memspace=_hip.PFD_DEVICE, deferred=True)
    assert np.array_equal(h.upstream_area_cell().reshape(n, n), exp)
    assert h.info()["n_valid"] == n * n == int(exp[d8 == 0].astype(np.int64).sum())  # pit sums == valid cells
    h.close()
    buf.free()


def test_c3_30000_accuflux_strahler_vs_oracle(gpu_lib, oracle, monkeypatch):
    # (the production threshold of the fused chain kernels — conftest.py lowers it for the small rasters: at this size
    #  there are rounds on both sides of it)
    monkeypatch.setenv("PFD_TEST_FUSE_MIN", str(1 << 20))
    from pyflwdir_amd import _hip

    O = oracle
    size = 30000
    n = size * size
    d8_buf = _hip.synth_d8_device(size, size, **RIVER)
    d8 = d8_buf.download(np.uint8, (size, size))
    assert np.array_equal(d8[:50], O.synth_d8(size, size, nrows=50, **RIVER))  # device generator == host generator
    h = _hip.RasterHandle(d8_buf, size, size, memspace=_hip.PFD_DEVICE)
    w_buf = _hip.synth_weights_device(n, seed=1)
    out = _hip.DeviceBuffer(n * 4)
    h.accuflux(w_buf, _hip.PFD_F32, nodata_f=-9999.0, has_nodata=1, out=out, memspace=_hip.PFD_DEVICE)
    acc = out.download(np.float32, (n,))
    sto = _hip.DeviceBuffer(n)
    h.strahler(None, out=sto, memspace=_hip.PFD_DEVICE)
    strord = sto.download(np.uint8, (n,))
    idxs_ds, idxs_pit, _ = O.from_array(d8)
    seq = O.idxs_seq(idxs_ds, idxs_pit)
    assert seq.size == n
    w = O.synth_weights_f32(n, seed=1)
    assert np.array_equal(acc, O.accuflux(idxs_ds, seq, w))
    assert np.array_equal(strord, O.strahler_order(idxs_ds, seq))
    # int32 accuflux of small non-negative weights rides the tiled engine (weights instead of ones in both tile
    # passes): on the eager handle and on a deferred one, whose first tile pass also decodes the raster
    wi = (np.arange(n, dtype=np.int64) * 2654435761 >> 7 & 3).astype(np.int32)
    exp = O.accuflux(idxs_ds, seq, wi)
    h.set_profiling(True)
    got = h.accuflux(wi, _hip.PFD_I32, nodata_i=-9999, has_nodata=1)
    assert "tile_local" in [s_["name"] for s_ in h.last_timing()]
    assert np.array_equal(got, exp)
    h.close()
    h = _hip.RasterHandle(d8_buf, size, size, memspace=_hip.PFD_DEVICE, deferred=True)
    assert np.array_equal(h.accuflux(wi, _hip.PFD_I32, nodata_i=-9999, has_nodata=1), exp)
    h.close()


def test_c4_90000_properties_and_blocks(gpu_lib, oracle):
    from pyflwdir_amd import _hip
    from pyflwdir_amd import dist as pdist

    size = 90000
    d8_buf = _hip.synth_d8_device(size, size, **RIVER)
    out = _hip.DeviceBuffer(size * size * 4)
    h = _hip.RasterHandle(d8_buf, size, size, memspace=_hip.PFD_DEVICE, deferred=True)
    h.upstream_area_cell(out=out, memspace=_hip.PFD_DEVICE)
    info = h.info()
    v = h.verify_upstream_area_cell(out, memspace=_hip.PFD_DEVICE)
    assert v["bad_cells"] == 0 and v["bad_nodata"] == 0
    assert v["n_valid"] == info["n_valid"] == size * size and v["n_pits"] == info["n_pits"]
    assert v["pit_sum"] == info["n_valid"]
    st = h.graph_stats()
    assert st["max_rank"] >= size - 1 and sum(st["indegree_hist"]) == info["n_valid"]
    h.close()
    # the first rows against the oracle (no cell below them drains upwards: their upstream area is complete)
    rows = 1200
    d8_top = d8_buf.download(np.uint8, (rows + 1, size))
    assert not np.isin(d8_top[rows], (32, 64, 128)).any()
    exp = oracle.upstream_area_cell(d8_top[:rows])[0]
    assert np.array_equal(out.download(np.int32, (rows, size)), exp)
    whole_sum = v["checksum"]
    out.free()
    # 8 row blocks (the C4 split: 11250 rows each), all held by this process on the one GPU
    nblocks = 8
    handles, outs, bufs = [], [], []
    for b, (r0, r1) in enumerate(pdist.block_rows(size, nblocks)):
        top, bot = pdist.halo_of(b, nblocks)
        blk = _hip.synth_d8_device(size, size, row0=r0 - top, nrows=(r1 - r0) + top + bot, **RIVER)
        bufs.append(blk)
        handles.append(_hip.RasterHandle(blk, r1 - r0, size, memspace=_hip.PFD_DEVICE, halo=(top, bot), deferred=True))
        outs.append(_hip.DeviceBuffer((r1 - r0) * size * 4))
    d8_buf.free()
    _hip.upstream_area_cell_blocks(handles, outs, memspace=_hip.PFD_DEVICE)
    total = 0
    for b, (r0, r1) in enumerate(pdist.block_rows(size, nblocks)):
        total += _hip.checksum_i32(outs[b], (r1 - r0) * size)
    assert total == whole_sum
    # spot check: the first rows of block 3 against the local equation with the row above (last row of block 2)
    r0 = pdist.block_rows(size, nblocks)[3][0]
    up3 = outs[3].download(np.int32, (2, size))
    up2 = outs[2].download(np.int32, (1, size), offset_bytes=(pdist.block_rows(size, nblocks)[2][1] - pdist.block_rows(size, nblocks)[2][0] - 1) * size * 4)
    codes = bufs[3].download(np.uint8, (3, size))  # halo row (= last row of block 2), first two own rows
    exp = np.ones(size, np.int64)
    for code, dc in ((2, 1), (4, 0), (8, -1)):  # cells of the row above that drain SE / S / SW into row r0
        src = np.arange(size) - dc
        ok = (src >= 0) & (src < size)
        hit = ok & (codes[0, np.clip(src, 0, size - 1)] == code)
        exp += np.where(hit, up2[0, np.clip(src, 0, size - 1)].astype(np.int64), 0)
    for code, dc in ((1, 1), (16, -1)):  # E / W neighbours in the same row
        src = np.arange(size) - dc
        ok = (src >= 0) & (src < size)
        hit = ok & (codes[1, np.clip(src, 0, size - 1)] == code)
        exp += np.where(hit, up3[0, np.clip(src, 0, size - 1)].astype(np.int64), 0)
    for code, dc in ((128, 1), (64, 0), (32, -1)):  # cells of the row below that drain NE / N / NW upwards
        src = np.arange(size) - dc
        ok = (src >= 0) & (src < size)
        hit = ok & (codes[2, np.clip(src, 0, size - 1)] == code)
        exp += np.where(hit, up3[1, np.clip(src, 0, size - 1)].astype(np.int64), 0)
    assert np.array_equal(up3[0].astype(np.int64), exp)
    for hh in handles:
        hh.close()


def test_c4_90000_order_sensitive_sweeps_in_row_blocks(gpu_lib, oracle):
    """Float accuflux, area-unit upstream_area and the Strahler order at 8.1 Gcells: beyond 2^32 - 2 cells the level
    engine cannot address the raster, so it is cut into row blocks whose halo cells carry the neighbours' values
    (pfd_accuflux_block / pfd_strahler_block, pyflwdir_amd/dist.py).  Checked here with everything device-resident:
      * int32 accuflux of ones over the blocks == upstream_area("cell") of the tiled engine on the whole raster
        (independent engines; checksum of every block's rows);
      * float32 accuflux of one value per row (what upstream_area("ha") sums on a projected grid) and the Strahler
        order: the iteration reaches its fixpoint and EVERY own cell satisfies its local equation bit for bit against
        the values in place (halo rows = the neighbours' final rows) — on an acyclic raster these equations have
        one solution; the first 1200 rows against the oracle."""
    from pyflwdir_amd import _hip
    from pyflwdir_amd import dist as pdist

    size, nblocks = 90000, 3
    rows_of = pdist.block_rows(size, nblocks)
    d8_buf = _hip.synth_d8_device(size, size, **RIVER)
    out = _hip.DeviceBuffer(size * size * 4)
    h = _hip.RasterHandle(d8_buf, size, size, memspace=_hip.PFD_DEVICE, deferred=True)
    h.upstream_area_cell(out=out, memspace=_hip.PFD_DEVICE)
    whole = [_hip.checksum_i32(out.addr + r0 * size * 4, (r1 - r0) * size) for r0, r1 in rows_of]
    h.close()
    out.free()
    top_rows = 1200
    d8_top = d8_buf.download(np.uint8, (top_rows + 1, size))
    assert not np.isin(d8_top[top_rows], (32, 64, 128)).any()  # nothing below drains up into the first rows
    d8_buf.free()
    idxs_ds, idxs_pit, _ = oracle.from_array(d8_top[:top_rows])
    seq = oracle.idxs_seq(idxs_ds, idxs_pit)
    handles, bufs = [], []
    for b, (r0, r1) in enumerate(rows_of):
        top, bot = pdist.halo_of(b, nblocks)
        assert ((r1 - r0) + top + bot) * size <= 4294967294
        bufs.append(_hip.synth_d8_device(size, size, row0=r0 - top, nrows=(r1 - r0) + top + bot, **RIVER))
        handles.append(_hip.RasterHandle(bufs[-1], r1 - r0, size, memspace=_hip.PFD_DEVICE, halo=(top, bot)))

    def run(kind, dtype, row_values=None):
        blocks = []
        try:
            for b, hh in enumerate(handles):
                a, e = pdist.block_slice(size, nblocks, b)
                blocks.append(pdist._UpBlock(hh, kind, dtype, payload=None if row_values is None else row_values[a:e],
                                             by_row=True, nodata=(-9999, -9999.0, 1)))
            rounds, bad = pdist._up_blocks_run(blocks, size, dtype, verify=True)
            sums = None
            if dtype == np.int32:
                sums = [_hip.checksum_i32(blk.out.addr + hh.halo[0] * size * 4, hh.nrow * size)
                        for blk, hh in zip(blocks, handles)]
            first = blocks[0].out.download(dtype, (top_rows, size))
            return rounds, bad, sums, first
        finally:
            for blk in blocks:
                blk.close(close_handle=False)

    try:
        rounds, bad, sums, first = run("accuflux", np.int32, np.ones(size, np.int32))
        assert bad == 0 and rounds >= 2 and sums == whole
        areas = (np.cos(np.linspace(-0.9, 0.9, size)) * 0.81).astype(np.float32)  # "hectares" per cell of a row
        rounds, bad, _, first = run("accuflux", np.float32, areas)
        assert bad == 0 and rounds >= 2
        exp = oracle.accuflux(idxs_ds, seq, np.repeat(areas[:top_rows], size), nodata=-9999)
        assert np.array_equal(first.ravel().view(np.uint32), exp.view(np.uint32))
        rounds, bad, _, first = run("strahler", np.uint8)
        assert bad == 0 and rounds >= 2
        assert np.array_equal(first.ravel(), oracle.strahler_order(idxs_ds, seq))
    finally:
        for hh in handles:
            hh.close()
        for bb in bufs:
            bb.free()


def test_uint32_index_rung(gpu_lib, oracle):
    """50000 x 50000 = 2.5e9 cells: idxs_ds / idxs_pit in uint32 (mv = 4294967295), checked on row bands."""
    from pyflwdir_amd import _hip

    size = 50000
    n = size * size
    assert 2**31 - 1 <= n < 2**32 - 2
    kw = dict(seed=3, tilt=100000, white=2, nodata_pct=20)
    d8_buf = _hip.synth_d8_device(size, size, **kw)
    h = _hip.RasterHandle(d8_buf, size, size, memspace=_hip.PFD_DEVICE)
    ds = _hip.DeviceBuffer(n * 4)
    _hip.check(_hip.lib().pfd_idxs_ds(h._h, _hip.PFD_U32, C.c_void_p(ds.addr), _hip.PFD_DEVICE))
    npits = h.info()["n_pits"]
    pits = np.empty(npits, np.uint32)
    _hip.check(_hip.lib().pfd_idxs_pit(h._h, _hip.PFD_U32, _hip.ptr(pits), _hip.PFD_HOST))
    assert np.all(np.diff(pits.astype(np.int64)) > 0)
    for r0 in (0, 21474, 42949, size - 40):  # incl. the band where the linear index crosses 2^31
        rows = min(40, size - r0)
        a0, a1 = max(0, r0 - 1), min(size, r0 + rows + 1)
        band = d8_buf.download(np.uint8, (a1 - a0, size), offset_bytes=a0 * size)
        loc_ds, loc_pit, _ = oracle.from_array(band, dtype=np.int64)
        loc = loc_ds.reshape(a1 - a0, size)[r0 - a0:r0 - a0 + rows]
        exp = np.where(loc < 0, np.int64(0xFFFFFFFF), loc + a0 * size)
        got = ds.download(np.uint32, (rows, size), offset_bytes=r0 * size * 4).astype(np.int64)
        # (the band carries one context row on every inner side, so the compared rows see the same
        #  neighbours as in the whole raster)
        assert np.array_equal(got, exp)
        in_band = pits[(pits >= r0 * size) & (pits < (r0 + rows) * size)].astype(np.int64)
        exp_p = loc_pit[(loc_pit // size >= r0 - a0) & (loc_pit // size < r0 - a0 + rows)] + a0 * size
        assert np.array_equal(in_band, exp_p)
    h.close()


def test_c5_basins_hand_at_size(gpu_lib, oracle):
    """BASELINE config 5 at full size (36000 x 72000 = 2.592e9 cells, 30 % nodata, rough regime): basins from
    1000 outlets and HAND on ONE GPU (it fits — 2.6 GB codes + 10.4 GB elevation + 2.6 GB drain + 20.7 GB float64
    result + the sweep plan; the sharded forms are the 4-block basins check at the end of this test and tests/test_gpu_blocks.py), checked at every cell on the device and by their local equations recomputed by numpy on sampled rows
    (label(x) == id(x) at an outlet, else label(downstream cell), 0 at pits; hand(x) == 0 on drains, else
    hand(downstream) + (double)(float32)(elev(x) - elev(downstream))) — on an acyclic raster the equations have
    one solution.  Basins sharded over 4 row blocks (the multi-GPU protocol, blocks held by this process) must
    give the identical labels."""
    from pyflwdir_amd import _hip
    from pyflwdir_amd import dist as pdist

    nrow, ncol = 36000, 72000
    n = nrow * ncol
    kw = dict(seed=2, tilt=100000, white=2, nodata_pct=30)
    d8_buf = _hip.synth_d8_device(nrow, ncol, **kw)
    h = _hip.RasterHandle(d8_buf, nrow, ncol, memspace=_hip.PFD_DEVICE)
    upa = _hip.DeviceBuffer(n * 4)
    h.upstream_area_cell(out=upa, memspace=_hip.PFD_DEVICE)
    # outlets: the largest upstream areas of 1000 sampled rows (distinct cells, many of them nested)
    rng = np.random.default_rng(5)
    outl = []
    for r in np.unique(rng.integers(0, nrow, 1000)):
        row = upa.download(np.int32, (ncol,), offset_bytes=int(r) * ncol * 4)
        outl.append(int(r) * ncol + int(np.argmax(row)))
    outl = np.array(outl[:1000], np.int64)
    ids = np.arange(1, outl.size + 1, dtype=np.uint32)
    lab = _hip.DeviceBuffer(n * 4)
    h.basins(outl, ids, out=lab, memspace=_hip.PFD_DEVICE)
    elev = _hip.synth_elev_device(nrow, ncol, **kw)
    # drain = cells with more than 100 upstream cells; built band by band on the host (uint8, 2.6 GB on the device)
    drain = _hip.DeviceBuffer(n)
    band = 2000
    for r0 in range(0, nrow, band):
        u = upa.download(np.int32, (band, ncol), offset_bytes=r0 * ncol * 4)
        _hip.check(_hip.lib().pfd_memcpy_h2d(0, C.c_void_p(drain.addr + r0 * ncol), _hip.ptr(np.ascontiguousarray(u > 100).view(np.uint8)),
                                             C.c_size_t(band * ncol)))
    upa.free()
    hand = _hip.DeviceBuffer(n * 8)
    h.hand(drain, elev, _hip.PFD_F32, out=hand, memspace=_hip.PFD_DEVICE)
    # every cell's local equation on the device (streaming kernels that share nothing with the engines; on an
    # acyclic raster each system has one solution, the reference's): labels and HAND at all 2.59 Gcells
    vb = h.verify_basins(outl, ids, lab, memspace=_hip.PFD_DEVICE)
    assert vb["bad_cells"] == 0 and vb["bad_nodata"] == 0 and vb["n_labelled"] > 1000, vb
    vh = h.verify_hand(drain, elev, _hip.PFD_F32, hand, memspace=_hip.PFD_DEVICE)
    assert vh["bad_cells"] == 0 and vh["bad_nodata"] == 0 and vh["n_drain"] > 0, vh
    # ... and the checkers themselves notice a single wrong cell
    probe = int(outl[0]) + 1
    keep = lab.download(np.uint32, (1,), offset_bytes=probe * 4)
    _hip.check(_hip.lib().pfd_memcpy_h2d(0, C.c_void_p(lab.addr + probe * 4), _hip.ptr(keep + np.uint32(1)), C.c_size_t(4)))
    assert h.verify_basins(outl, ids, lab, memspace=_hip.PFD_DEVICE)["bad_cells"] >= 1
    _hip.check(_hip.lib().pfd_memcpy_h2d(0, C.c_void_p(lab.addr + probe * 4), _hip.ptr(keep), C.c_size_t(4)))
    # (the HAND twin: one height off by one ulp at a cell that is no drain cell)
    for r in (12345, 23456):
        D = drain.download(np.uint8, (ncol,), offset_bytes=r * ncol)
        Hrow = hand.download(np.float64, (ncol,), offset_bytes=r * ncol * 8)
        cand = np.nonzero((D == 0) & (Hrow != -9999.0) & np.isfinite(Hrow))[0]
        if cand.size:
            break
    probe = r * ncol + int(cand[cand.size // 2])
    keep = hand.download(np.float64, (1,), offset_bytes=probe * 8)
    _hip.check(_hip.lib().pfd_memcpy_h2d(0, C.c_void_p(hand.addr + probe * 8), _hip.ptr(np.nextafter(keep, np.inf)), C.c_size_t(8)))
    assert h.verify_hand(drain, elev, _hip.PFD_F32, hand, memspace=_hip.PFD_DEVICE)["bad_cells"] >= 1
    _hip.check(_hip.lib().pfd_memcpy_h2d(0, C.c_void_p(hand.addr + probe * 8), _hip.ptr(keep), C.c_size_t(8)))
    # the same equations recomputed by numpy (independent of the device decode) on sampled rows
    DR = {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0), 128: (-1, 1)}
    for r in (1, 7777, 18000, 25113, nrow - 2):
        d = d8_buf.download(np.uint8, (3, ncol), offset_bytes=(r - 1) * ncol)
        L = lab.download(np.uint32, (3, ncol), offset_bytes=(r - 1) * ncol * 4)
        H = hand.download(np.float64, (3, ncol), offset_bytes=(r - 1) * ncol * 8)
        E = elev.download(np.float32, (3, ncol), offset_bytes=(r - 1) * ncol * 4)
        D = drain.download(np.uint8, (ncol,), offset_bytes=r * ncol)
        cols = np.arange(ncol)
        code = d[1]
        valid = code != 247
        # downstream cell of every cell of row r (pit rule: target outside the raster or nodata -> pit)
        tr = np.ones(ncol, np.int64)
        tc = cols.copy()
        pit = ~np.isin(code, list(DR))
        for cde, (dr, dc) in DR.items():
            m = code == cde
            tr[m] = 1 + dr
            tc[m] = cols[m] + dc
        off = (tc < 0) | (tc >= ncol)
        tcc = np.clip(tc, 0, ncol - 1)
        pit |= off | (d[tr, tcc] == 247)
        # labels
        out_id = np.zeros(ncol, np.uint32)
        sel = (outl // ncol) == r
        out_id[outl[sel] % ncol] = ids[sel]
        expL = np.where(out_id != 0, out_id, np.where(pit, 0, L[tr, tcc]))
        assert np.array_equal(L[1][valid], expL[valid]) and np.all(L[1][~valid] == 0)
        # HAND
        dz = (E[1] - np.where(pit, E[1], E[tr, tcc])).astype(np.float32)
        expH = np.where(D == 1, 0.0, np.where(pit, 0.0, H[tr, tcc]) + dz.astype(np.float64))
        assert np.array_equal(H[1][valid], expH[valid]) and np.all(H[1][~valid] == -9999.0)
    h.close()
    # HAND sharded over 4 row blocks (config 5 names 4 GPUs) == the single-GPU result, bit for bit: the blocks'
    # inputs are slices of the device-resident rasters, the boundary rows travel through the host
    nb = 4
    rows = pdist.block_rows(nrow, nb)
    seeds = [np.full(2 * ncol, -np.inf) for _ in range(nb)]
    hbs, bouts, brows, nunk, prev_seeds = [], [], [None] * nb, [None] * nb, [None] * nb
    for b, (r0, r1) in enumerate(rows):
        a, e = pdist.block_slice(nrow, nb, b)
        hbs.append(_hip.RasterHandle(d8_buf.addr + a * ncol, r1 - r0, ncol, memspace=_hip.PFD_DEVICE, halo=pdist.halo_of(b, nb)))
        bouts.append(_hip.DeviceBuffer((e - a) * ncol * 8))
    # (one exchange per block edge a path crosses before it meets a drain cell: a path that weaves along an edge needs
    #  several; the first pass sweeps the block, later ones only relax the cells that are still unknown)
    sweeps = 0
    for it in range(1, 65):
        for b, (r0, r1) in enumerate(rows):
            if prev_seeds[b] is not None and np.array_equal(prev_seeds[b].view(np.uint64), seeds[b].view(np.uint64)):
                continue
            a, e = pdist.block_slice(nrow, nb, b)
            _, brows[b], nunk[b] = hbs[b].hand_block(drain.addr + a * ncol, elev.addr + a * ncol * 4, _hip.PFD_F32, seeds[b],
                                                     out=bouts[b], memspace=_hip.PFD_DEVICE, update=prev_seeds[b] is not None)
            prev_seeds[b] = seeds[b].copy()
            sweeps += 1
        if sum(nunk) == 0:
            break
        for b in range(nb):
            if b > 0:
                seeds[b][:ncol] = brows[b - 1][1]
            if b + 1 < nb:
                seeds[b][ncol:] = brows[b + 1][0]
    assert sum(nunk) == 0, (nunk, it)
    print(f"sharded HAND at {nrow}x{ncol}: {it} exchanges, {sweeps} block passes")
    for b, (r0, r1) in enumerate(rows):  # every block's rows == the whole raster's, bit for bit
        a, e = pdist.block_slice(nrow, nb, b)
        own = bouts[b].download(np.float64, (r1 - r0, ncol), offset_bytes=(r0 - a) * ncol * 8)
        whole = hand.download(np.float64, (r1 - r0, ncol), offset_bytes=r0 * ncol * 8)
        assert np.array_equal(own.view(np.uint64), whole.view(np.uint64)), b
        del own, whole
        hbs[b].close()
        bouts[b].free()
    for b in (hand, elev, drain):
        b.free()
    # sharded over 4 row blocks == the single-GPU labels (compared through checksums of the uint32 labels)
    d8 = d8_buf.download(np.uint8, (nrow, ncol))
    d8_buf.free()
    got = pdist.basins_blocks(d8, 4, outl, ids)
    whole = lab.download(np.uint32, (nrow, ncol))
    assert np.array_equal(got, whole)
