"""Body of tests/test_gpu_order64.py::test_true_size_beyond_2_32_cells, run in a CHILD process: the run holds a 120 GiB arena
and tens of GB of host arrays, which must not outlive the test inside the pytest process (a later test that starts eight
bench.py ranks on the same GPU found the HBM nearly full once).  Exit code 0 = every assertion held."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _monotone_windows(seq, rank, nwin=48, wlen=1 << 21):
    """rank[seq] non-decreasing inside windows spread over the sequence, and from window to window (tests/test_core.py:82;
    the full gather of 4.4e9 random reads takes a minute on the host: sampled)."""
    ok, prev = True, -1
    for w in range(nwin):
        i = (seq.size - wlen) * w // max(1, nwin - 1) if seq.size > wlen else 0
        r = rank[seq[i:i + wlen]]
        ok = ok and int(r[0]) >= prev and bool(np.all(np.diff(r) >= 0))
        prev = int(r[-1])
    return ok


def main():
    """66000 x 66000 = 4.356e9 cells (> 2**32 - 2: the int64 rung of pyflwdir.py:105-127) through the front end at TRUE size —
    no lowered threshold: rank and the exact idxs_seq order by the properties that define them (core.py:17-47, :87-117,
    tests/test_core.py:66-82), the classic stream order over row blocks by its local properties (streams.py:191-225), and the
    same raster with a cycle injected: the cells that never reach a pit read -1 and are left out of the sequence."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip

    size = 66000
    free = _hip.mem_info()["free"]
    assert free > (150 << 30), f"only {free >> 30} GiB of HBM free: this run needs ~150"
    _hip.reserve(min(120 << 30, int(0.55 * free)))
    buf = _hip.synth_d8_device(size, size, seed=0)
    d8 = buf.download(np.uint8, (size, size))
    buf.free()
    n = d8.size
    assert n > 2**32 - 2
    from pyflwdir_amd._affine import Affine

    res = 1.0 / 1200.0  # a 3-arc-second lat/lon grid from 60 N down to 5 N
    flw = pyflwdir.from_array(d8, ftype="d8", transform=Affine(res, 0.0, 5.0, 0.0, -res, 60.0), latlon=True, cache=False)
    assert flw._wide() and flw.idxs_pit.dtype == np.int64
    # upstream_area in km2, order-free in fixed point (csrc/wide.h) on ONE handle of 4.36e9 cells, against the exact form in
    # seeded row blocks on sampled rows: within the stated n_cells / 2**63 x (mean / min area) relative, and the local
    # equation area(x) = own + sum of the upstream cells' areas to the same tolerance
    t0 = time.perf_counter()
    fx = flw.upstream_area("km2", exact=False)
    t_fixed = time.perf_counter() - t0
    assert fx.dtype == np.float64 and fx.shape == (size, size) and float(fx.min()) > 0.0
    t0 = time.perf_counter()
    ex = flw.upstream_area("km2")
    t_exact = time.perf_counter() - t0
    print(f"66000^2 upstream_area('km2'): fixed point {t_fixed:.2f} s, exact (row blocks) {t_exact:.2f} s")
    for r0 in (0, 33000, size - 2000):
        a, b = fx[r0:r0 + 2000], ex[r0:r0 + 2000]
        assert float(np.max(np.abs(a - b) / b)) <= 2e-9
    del ex, fx
    rank = flw.rank.ravel()
    assert rank.dtype == np.int32 and int(rank.min()) == 0  # (no nodata, no cycle)
    pits = flw.idxs_pit
    assert np.all(rank[pits] == 0) and int(np.count_nonzero(rank[: 1 << 28] == 0)) == int(np.count_nonzero(pits < (1 << 28)))
    # rank is the distance to the pit: one more than the downstream cell's, on sampled row windows
    dr = np.array([0, 1, 1, 1, 0, -1, -1, -1])
    dc = np.array([1, 1, 0, -1, -1, -1, 0, 1])
    r2 = rank.reshape(size, size)
    for r0 in (0, 31000, size - 1001):
        blk, rk = d8[r0:r0 + 1001], r2[r0:r0 + 1001]
        for k in range(8):
            rr, cc = np.nonzero(blk[1:-1, 1:-1] == (1 << k))
            assert np.all(rk[rr + 1 + dr[k], cc + 1 + dc[k]] + 1 == rk[rr + 1, cc + 1])
    seq = flw.idxs_seq
    assert seq.dtype == np.int64 and seq.size == n
    assert np.array_equal(seq[: pits.size], pits)  # pits first, ascending
    assert _monotone_windows(seq, rank)
    # every cell exactly once: the sums of the indices and of their squares (mod 2^64) are those of 0 .. n - 1
    s1 = s2 = 0
    for i in range(0, n, 1 << 28):
        part = seq[i:i + (1 << 28)].view(np.uint64)
        s1 = (s1 + int(part.sum(dtype=np.uint64))) & (2**64 - 1)
        s2 = (s2 + int((part * part).sum(dtype=np.uint64))) & (2**64 - 1)
    assert s1 == (n * (n - 1) // 2) & (2**64 - 1) and s2 == ((n - 1) * n * (2 * n - 1) // 6) & (2**64 - 1)
    # the upstream cells of a dequeued cell are contiguous and ascending at the queue's own positions (a prefix)
    j = pits.size
    for i in range(20000):
        r, c = divmod(int(seq[i]), size)
        ch = sorted((r + dr[k]) * size + c + dc[k] for k in range(8)
                    if 0 <= r + dr[k] < size and 0 <= c + dc[k] < size and d8[r + dr[k], c + dc[k]] == (1 << ((k + 4) & 7)))
        assert seq[j:j + len(ch)].tolist() == ch
        j += len(ch)
    del seq
    # classic ("Hack") stream order over row blocks (the raster is beyond one handle's 32-bit cell indices; reference
    # streams.py:191-225): a pit has order 1; a cell has its downstream cell's order, or one more — only where that cell has
    # more than one upstream cell
    so = flw.stream_order(type="classic")
    assert so.dtype == np.uint8 and so.shape == (size, size)
    assert np.all(so.ravel()[pits] == 1)
    nup = flw.n_upstream
    for r0 in (0, 40000, size - 1001):
        blk, o, nu = d8[r0:r0 + 1001], so[r0:r0 + 1001], nup[r0:r0 + 1001]
        for k in range(8):
            rr, cc = np.nonzero(blk[1:-1, 1:-1] == (1 << k))
            od, ou = o[rr + 1 + dr[k], cc + 1 + dc[k]].astype(int), o[rr + 1, cc + 1].astype(int)
            nd = nu[rr + 1 + dr[k], cc + 1 + dc[k]]
            assert np.all((ou == od) | (ou == od + 1)) and np.all(nd[ou > od] > 1) and np.all(ou[nd == 1] == od[nd == 1])
    del so, nup
    # ---- basins and ucat_area on ONE handle (round 6: the tiled label query at any size; the float64 sums of ucat_area over
    # the 64-bit sequence on the device) by what defines them (basins.py:12-18, subgrid.py:51-93): an outlet carries its own
    # label, every other cell its downstream cell's; a catchment's cell count is the number of cells with its label, its
    # area the sum of their row areas (the device adds them in sequence order, a row-count dot product in another:
    # equal to 1e-12 relative)
    from pyflwdir_amd import gis

    upa_s = flw.upstream_area()[::, ::4097]  # (a strided sample to pick outlets from)
    cols = np.arange(0, size, 4097)
    flat = np.argsort(upa_s.ravel())[-300:]
    outs = (flat // cols.size).astype(np.int64) * size + cols[flat % cols.size]
    del upa_s
    t0 = time.perf_counter()
    bas = flw.basins(idxs=outs)
    t_b = time.perf_counter() - t0
    assert bas.dtype == np.uint32 and np.array_equal(bas.ravel()[outs], np.arange(1, outs.size + 1, dtype=np.uint32))
    is_out = np.zeros(n, bool)
    is_out[outs] = True
    is_out = is_out.reshape(size, size)
    for r0 in (0, 20000, size - 1001):
        blk, lb, io = d8[r0:r0 + 1001], bas[r0:r0 + 1001], is_out[r0:r0 + 1001]
        for k in range(8):
            rr, cc = np.nonzero((blk[1:-1, 1:-1] == (1 << k)) & ~io[1:-1, 1:-1])
            assert np.array_equal(lb[rr + 1, cc + 1], lb[rr + 1 + dr[k], cc + 1 + dc[k]])
        assert np.all(lb[(blk == 0) & ~io] == 0)  # a pit that is no outlet: label 0
    del is_out
    t0 = time.perf_counter()
    m_c, a_c = flw.ucat_area(outs, unit="cell")
    t_c = time.perf_counter() - t0
    assert m_c.dtype == np.int64 and a_c.dtype == np.int32
    t0 = time.perf_counter()
    m_k, a_k = flw.ucat_area(outs, unit="km2")
    t_k = time.perf_counter() - t0
    assert a_k.dtype == np.float64
    rows = gis.area_rows(flw.transform, flw.shape, flw.latlon, unit="m2") / gis.AREA_FACTORS["km2"]
    big3 = np.argsort(a_c)[-3:]
    per_row = np.zeros((3, size), np.int64)
    for r0 in range(0, size, 4096):
        assert np.array_equal(m_c[r0:r0 + 4096], bas[r0:r0 + 4096]) and np.array_equal(m_k[r0:r0 + 4096], bas[r0:r0 + 4096])
        for j, i in enumerate(big3):
            per_row[j, r0:r0 + 4096] = np.count_nonzero(bas[r0:r0 + 4096] == i + 1, axis=1)
    for j, i in enumerate(big3):
        assert int(per_row[j].sum()) == int(a_c[i])
        ref = float(np.dot(per_row[j].astype(np.float64), rows))
        assert abs(float(a_k[i]) - ref) <= 1e-9 * ref
    print(f"66000^2, 300 outlets on one handle: basins {t_b:.2f} s, ucat_area cell {t_c:.2f} s, km2 {t_k:.2f} s")
    del bas, m_c, m_k
    # ---- floodplains over streamed row blocks (dem.py:333-379) by its local equations: a stream cell (upstream area >= upa_min)
    # is floodplain; any other cell is floodplain only if its downstream cell is, and — where the downstream cell is a stream
    # cell, whose (z0, h0) are its own elevation and uparea ** b — exactly when elevtn - z0 <= h0 in float32
    # (any elevation serves the local equations; the synthetic surface's float32 values are all tilt at this size — a periodic
    #  pattern with steps of the order of the height thresholds instead: some cells beside every stream are floodplain, most not)
    base = ((np.arange(4094, dtype=np.int64)[:, None] * 7 + np.arange(size, dtype=np.int64)[None, :] * 13) % 23).astype(np.float32)
    elev = np.empty((size, size), np.float32)
    for r0 in range(0, size, 4094):  # (4094 = 23 * 178: the pattern continues across the pieces)
        elev[r0:r0 + 4094] = base[: min(4094, size - r0)]
    del base
    upa = flw.upstream_area().astype(np.float32)
    upa_min, bexp = 3000.0, 0.3
    t0 = time.perf_counter()
    fld = flw.floodplains(elev, uparea=upa, upa_min=upa_min, b=bexp)
    t_f = time.perf_counter() - t0
    assert fld.dtype == np.int8 and fld.shape == (size, size) and int(fld.min()) >= 0  # (no nodata, no cycle: nothing off the sequence)
    nflood = 0
    for r0 in (0, 25000, size - 1001):
        blk, f, u, z = d8[r0:r0 + 1001], fld[r0:r0 + 1001], upa[r0:r0 + 1001], elev[r0:r0 + 1001]
        assert np.all(f[u >= upa_min] == 1)
        for k in range(8):
            rr, cc = np.nonzero((blk[1:-1, 1:-1] == (1 << k)) & (u[1:-1, 1:-1] < upa_min))
            rd, cd = rr + 1 + dr[k], cc + 1 + dc[k]
            fd, fu = f[rd, cd], f[rr + 1, cc + 1]
            assert np.all(fu <= fd)  # floodplain only below a floodplain cell
            st = u[rd, cd] >= upa_min  # the downstream cell starts a floodplain itself
            h0 = (u[rd, cd][st] ** bexp).astype(np.float32)
            dh = z[rr + 1, cc + 1][st] - z[rd, cd][st]
            assert np.array_equal(fu[st] == 1, dh <= h0)
            nflood += int(fu.sum())
    assert nflood > 1000
    print(f"66000^2 floodplains over streamed row blocks: {t_f:.2f} s ({nflood} floodplain cells off the streams in the sampled windows)")
    del elev, upa, fld
    # ---- the same raster with a cycle: two neighbouring headwater-side cells made to drain into each other ----------
    r, c = 5, 40000
    d8[r, c], d8[r, c + 1] = 1, 16  # E and W
    loop = {r * size + c, r * size + c + 1}
    todo = list(loop)
    while todo:  # everything that drains to the cycle (a handful of cells this close to the raster's upper edge)
        x = todo.pop()
        rr, cc = divmod(x, size)
        for k in range(8):
            a, b = rr + dr[k], cc + dc[k]
            if 0 <= a < size and 0 <= b < size and d8[a, b] == (1 << ((k + 4) & 7)) and a * size + b not in loop:
                loop.add(a * size + b)
                todo.append(a * size + b)
    assert len(loop) < 10**6
    flw2 = pyflwdir.from_array(d8, ftype="d8", cache=False)
    assert not flw2.isvalid
    rank2 = flw2.rank.ravel()
    li = np.fromiter(loop, np.int64)
    assert np.all(rank2[li] == -1) and int(np.count_nonzero(rank2 == -1)) == li.size
    rank[li] = -1
    assert np.array_equal(rank2, rank)  # everybody else keeps its distance to its pit
    seq2 = flw2.idxs_seq
    assert seq2.size == n - li.size and _monotone_windows(seq2, rank2)
    assert int(rank2[seq2[:: 4099]].min()) >= 0 and flw2.nnodes == seq2.size


if __name__ == "__main__":
    main()
    print("true size: ok")
