"""Host side of the multi-GPU path: a raster row-tiled over the GPUs of one node, one process
per GPU, one small RCCL all-gather per pass (kernels and protocol: csrc/dist.hip).

The reference has no distributed code at all; this module only adds what a
driver script needs: the row partition, halo handling, and the rendezvous of the RCCL unique id
through a host-side process group — by default ``hostgroup.HostGroup``: plain TCP, no PyTorch.
"""
from __future__ import annotations

import numpy as np

from . import _hip


def block_rows(nrow: int, nblocks: int):
    """Row ranges [r0, r1) of ``nblocks`` contiguous row blocks, sizes differing by at most 1."""
    if nblocks < 1 or nrow < nblocks:
        raise ValueError(f"cannot split {nrow} rows into {nblocks} blocks")
    base, extra = divmod(nrow, nblocks)
    out, r0 = [], 0
    for b in range(nblocks):
        r1 = r0 + base + (1 if b < extra else 0)
        out.append((r0, r1))
        r0 = r1
    return out


def halo_of(block: int, nblocks: int):
    """(top, bottom) halo rows of a block: one row towards every existing neighbour."""
    return (1 if block > 0 else 0, 1 if block + 1 < nblocks else 0)


def block_slice(nrow: int, nblocks: int, block: int):
    """Rows [a, b) of the full raster that block ``block`` must hold, halo rows included."""
    r0, r1 = block_rows(nrow, nblocks)[block]
    top, bot = halo_of(block, nblocks)
    return r0 - top, r1 + bot


def _concat_rows(parts):
    """``np.concatenate(parts, axis=0)`` of the blocks' results.  Tens of GB at 8.1 Gcells (HAND: 65 GB): the fresh pages
    of the result are touched, and the rows copied, by host threads in pieces — one thread takes 8 s for what 16 do in 2."""
    parts = [np.asarray(p) for p in parts]
    if sum(p.nbytes for p in parts) < (1 << 28):
        return np.concatenate(parts, axis=0)
    from concurrent.futures import ThreadPoolExecutor
    import os

    out = np.empty((sum(p.shape[0] for p in parts),) + parts[0].shape[1:], parts[0].dtype)
    rows_per = max(1, (1 << 26) // max(1, parts[0][:1].nbytes))
    jobs, r0 = [], 0
    for p in parts:
        for a in range(0, p.shape[0], rows_per):
            jobs.append((r0 + a, p, a, min(p.shape[0], a + rows_per)))
        r0 += p.shape[0]

    def copy(j):
        o, p, a, e = j
        out[o:o + (e - a)] = p[a:e]

    with ThreadPoolExecutor(min(16, len(jobs), os.cpu_count() or 1)) as ex:
        list(ex.map(copy, jobs))
    return out


def upstream_area_blocks(d8: np.ndarray, nblocks: int, devices=None, deferred: bool = False) -> np.ndarray:
    """``upstream_area("cell")`` of a host raster computed as ``nblocks`` row blocks held by this one
    process (on one or several GPUs).  Same kernels and protocol as the RCCL path."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    devices = devices or [0] * nblocks
    handles = []
    for b, (r0, r1) in enumerate(block_rows(nrow, nblocks)):
        a, e = block_slice(nrow, nblocks, b)
        handles.append(_hip.RasterHandle(d8[a:e], r1 - r0, ncol, device=devices[b], halo=halo_of(b, nblocks),
                                          deferred=deferred))
    outs = _hip.upstream_area_cell_blocks(handles)
    for h in handles:
        h.close()
    return _concat_rows([o.reshape(-1, ncol) for o in outs])


def _split_outlets(idxs, ids, nrow, ncol, nblocks):
    """Outlets (global linear indices) per row block, as indices into the block's own rows, in input order."""
    idxs = np.asarray(idxs, dtype=np.int64).ravel()
    ids = np.asarray(ids).ravel()
    rows = idxs // ncol
    out = []
    for r0, r1 in block_rows(nrow, nblocks):
        sel = (rows >= r0) & (rows < r1)
        out.append((idxs[sel] - r0 * ncol, ids[sel]))
    return out


def basins_blocks(d8: np.ndarray, nblocks: int, idxs, ids=None, devices=None) -> np.ndarray:
    """``basins(idxs, ids)`` of a host raster computed as ``nblocks`` row blocks held by this one process
    (same kernels and protocol as the one-process-per-GPU path; the records are moved by numpy)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    idxs = np.asarray(idxs, dtype=np.int64).ravel()
    ids = np.arange(1, idxs.size + 1, dtype=np.uint32) if ids is None else np.asarray(ids).ravel()
    devices = devices or [0] * nblocks
    parts = _split_outlets(idxs, ids, nrow, ncol, nblocks)
    handles, outs, recs = [], [], []
    for b, (r0, r1) in enumerate(block_rows(nrow, nblocks)):
        a, e = block_slice(nrow, nblocks, b)
        h = _hip.RasterHandle(d8[a:e], r1 - r0, ncol, device=devices[b], halo=halo_of(b, nblocks))
        o, rec = _hip.basins_begin(h, parts[b][0], parts[b][1].astype(ids.dtype))
        handles.append(h), outs.append(o), recs.append(rec)
    allrec = np.stack(recs)
    ok = True
    for b, h in enumerate(handles):
        ok &= _hip.basins_finish(h, allrec, nblocks, b)
        h.close()
    if not ok:
        raise NotImplementedError("the raster holds a cycle through several row blocks")
    return _concat_rows([o.reshape(-1, ncol) for o in outs])


_ELEV_CODE = {np.dtype(np.float32): _hip.PFD_F32, np.dtype(np.float64): _hip.PFD_F64}


def _all_finite(a) -> bool:
    """``np.isfinite(a).all()`` in host-thread pieces (2 s of a HAND call at 8.1 Gcells in one piece)."""
    flat = a.reshape(-1)
    step = 1 << 25
    if flat.size <= step:
        return bool(np.isfinite(flat).all())
    from concurrent.futures import ThreadPoolExecutor
    import os

    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        return all(ex.map(lambda i: bool(np.isfinite(flat[i:i + step]).all()), range(0, flat.size, step)))


def _hand_inputs(drain, elevtn):
    drain = np.ascontiguousarray(drain).astype(np.uint8, copy=False)
    elevtn = np.ascontiguousarray(elevtn)
    if elevtn.dtype not in _ELEV_CODE:
        raise NotImplementedError(f"elevation dtype {elevtn.dtype} is not supported on the HIP path")
    # the row-block protocol marks "height not known yet" with -inf: an elevation difference of +-inf or NaN could produce
    # that very value (or turn an unknown into NaN) and the blocks would never agree that they are done
    if not _all_finite(elevtn):
        raise NotImplementedError("hand over row blocks needs finite elevations (-inf marks heights that are not known "
                                  "yet); mask or fill inf / NaN cells first")
    return drain, elevtn, _ELEV_CODE[elevtn.dtype]


def hand_blocks(d8: np.ndarray, nblocks: int, drain, elevtn, devices=None, max_iter=None):
    """``hand(drain, elevtn)`` (reference pyflwdir/dem.py:299-330) of a host raster computed as ``nblocks`` row blocks
    held by this one process; same kernels and protocol as ``DistributedRaster.hand``, the boundary rows are moved
    by numpy.  Returns (hand, iterations).

    HAND is a sum along the path to the nearest drain cell, float64, one addition per step — not associative, so a
    path that crosses a block edge must CONTINUE the neighbour's sum.  Every block sweeps down- to upstream with its
    halo cells (the neighbour's boundary cells) holding the neighbour's height; a height that is not known yet is
    -inf, which every sum depending on it inherits.  The blocks exchange their boundary rows and sweep again until
    no owned cell is -inf: once per block edge a path crosses before it meets a drain cell (rivers are drain cells
    themselves, so two sweeps are the rule)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    drain, elevtn, code = _hand_inputs(drain, elevtn)
    drain, elevtn = drain.reshape(nrow, ncol), elevtn.reshape(nrow, ncol)
    devices = devices or [0] * nblocks
    rows = block_rows(nrow, nblocks)
    stream = _stream_blocks(d8.size, 1 + elevtn.dtype.itemsize + 8 + 28, devices)
    blocks = []
    try:
        for b, (r0, r1) in enumerate(rows):
            a, e = block_slice(nrow, nblocks, b)
            if stream:
                blocks.append(_StreamedHandBlock(d8[a:e], r1 - r0, ncol, devices[b], halo_of(b, nblocks), drain[a:e], elevtn[a:e], code))
                continue
            h = _hip.RasterHandle(d8[a:e], r1 - r0, ncol, device=devices[b], halo=halo_of(b, nblocks))
            blocks.append(_HandBlock(h, drain[a:e], elevtn[a:e], code))
        seeds = [np.full(2 * ncol, -np.inf) for _ in range(nblocks)]
        unknown_before, it = None, 0
        while True:
            it += 1
            for b, blk in enumerate(blocks):
                blk.sweep(seeds[b])
            unknown = sum(blk.unknown for blk in blocks)
            if unknown == 0:
                break
            if unknown == unknown_before or (max_iter is not None and it >= max_iter):
                raise NotImplementedError("hand: heights that depend on each other through several row blocks "
                                          "(a cycle through the block edges)")
            unknown_before = unknown
            for b in range(nblocks):  # halo rows = the neighbours' boundary rows
                if b > 0:
                    seeds[b][:ncol] = blocks[b - 1].brows[1]
                if b + 1 < nblocks:
                    seeds[b][ncol:] = blocks[b + 1].brows[0]
        return _concat_rows([blk.result() for blk in blocks]), it
    finally:
        for blk in blocks:
            blk.close()


def relevant_halo(d8_rows: np.ndarray, halo, down: bool) -> np.ndarray:
    """Which of a row block's 2 * ncol halo cells its result depends on.  A down-sweep (a cell takes its value from the
    cell it drains into) reads the halo cells an own boundary cell drains INTO; an up-sweep those that drain into an own
    boundary cell.  The neighbour's other boundary values change from round to round of the fixpoint iteration without
    meaning anything to this block: a block whose RELEVANT halo values are those of its last sweep is not swept again
    (flow crossing the blocks one way: the downstream blocks of a down-sweep sweep once or twice instead of every round).
    ``d8_rows``: the block's device rows (halo rows included), raw D8 codes."""
    ncol = d8_rows.shape[1]
    rel = np.zeros(2 * ncol, bool)

    def flows(src, dst, codes):  # (source columns, target columns) of the cells of row `src` that drain into row `dst`
        cs, cd = [], []
        for j, code in enumerate(codes):  # codes[j] points at column c + j - 1
            c = np.flatnonzero(src == code)
            d = c + (j - 1)
            ok = (d >= 0) & (d < ncol)
            c, d = c[ok], d[ok]
            ok = dst[d] != 247  # (flow into nodata ends where it is: the cell is a pit of its own row)
            cs.append(c[ok])
            cd.append(d[ok])
        return np.concatenate(cs), np.concatenate(cd)

    if halo[0]:
        own, hal = d8_rows[1], d8_rows[0]
        rel[:ncol][flows(own, hal, (32, 64, 128))[1] if down else flows(hal, own, (8, 4, 2))[0]] = True  # NW N NE / SW S SE
    if halo[1]:
        own, hal = d8_rows[-2], d8_rows[-1]
        rel[ncol:][flows(own, hal, (8, 4, 2))[1] if down else flows(hal, own, (32, 64, 128))[0]] = True
    return rel


class _SeedGate:
    """``sweep(seed)``: sweep with these halo values unless the ones that matter are those of the last sweep."""

    relevant = None  # bool [2 * ncol] (relevant_halo), or None: every halo value counts
    calls = 0        # sweeps that ran

    def _seed_bits(self, seed):
        part = seed if self.relevant is None else np.ascontiguousarray(seed[self.relevant])
        return part.view(np.uint8)

    def sweep(self, seed):
        """False if the relevant halo values are the ones of the last sweep (nothing can have changed)."""
        bits = self._seed_bits(seed)
        if self.swept_with is not None and np.array_equal(self.swept_with, bits):
            return False
        self.swept_with = bits.copy()
        self.brows, _ = self._call(seed, False)
        self.calls += 1
        return True


HBM_BUDGET = 200 << 30  # bytes of an MI355X's 288 GB that the row blocks of ONE call may hold at the same time


def hbm_budget(device: int = 0) -> int:
    """What the row blocks of one call may hold on ``device``: 70 % of its HBM (200 GiB of an MI355X's 288 GB — the
    module constant, kept for callers that read it — and the same share of a smaller part), never more than what is
    free right now less a working margin of an eighth of the device."""
    try:
        mi = _hip.mem_info(device)
    except Exception:  # noqa: BLE001 - (no device query: the MI355X figure)
        return HBM_BUDGET
    return max(1 << 30, min(int(0.7 * mi["total"]), mi["free"] + _hip.alloc_stats()["reserved_free"] - mi["total"] // 8))


def _stream_blocks(cells: int, bytes_per_cell: int, devices) -> bool:
    """Do the row blocks of a call go through the device one at a time?  Resident blocks keep payloads, results and
    their sweep plans (~26 bytes per cell) in HBM between the exchanges; when that does not fit one GPU — floodplains of
    8.1 Gcells: ~85 bytes per cell — a block is built, swept and released per sweep instead, its result waiting on the
    host (the flow-order schedule of _up_blocks_run keeps the sweeps few).  Blocks spread over several devices are
    held.  (PFD_TEST_STREAM_BLOCKS with PFD_ENABLE_KNOBS=1 forces either.)"""
    import os

    if os.environ.get("PFD_ENABLE_KNOBS") == "1" and os.environ.get("PFD_TEST_STREAM_BLOCKS"):
        return os.environ["PFD_TEST_STREAM_BLOCKS"] == "1"
    return len(set(devices)) == 1 and cells * bytes_per_cell > hbm_budget(list(devices)[0] if devices else 0)


class _StreamedBlock(_SeedGate):
    """A row block that is on the device only while it sweeps (``make()`` builds it from the host arrays)."""

    def __init__(self, make):
        self.make, self.host = make, None
        self.swept_with, self.brows = None, None
        self.into = None  # (optional: the block's rows of the caller's result array — result(out=) of the block fills them)

    def _call(self, seed, verify):
        if verify:
            raise NotImplementedError("verify=True needs the blocks resident: these are streamed through the device "
                                      "(their state exceeds HBM_BUDGET)")
        blk = self.make()
        try:
            if hasattr(blk, "incremental"):
                blk.incremental = False  # (nothing is kept between the sweeps)
            out = blk._call(seed, False)
            self.host = blk.result() if self.into is None else blk.result(self.into)
            return out
        finally:
            blk.close()

    def verify(self, seed):
        return self._call(seed, True)[1]

    def result(self):
        return self.host

    def close(self, close_handle=True):
        pass


def _blocks_of(nblocks, make, relevant, stream):
    """The blocks of a call: ``make(b)`` now, or a streamed stand-in; ``relevant(b)`` = its relevant_halo mask."""
    import functools

    blocks = []
    try:
        for b in range(nblocks):
            blk = _StreamedBlock(functools.partial(make, b)) if stream else make(b)
            blocks.append(blk)
            blk.relevant = relevant(b)
    except Exception:
        for blk in blocks:
            blk.close()
        raise
    return blocks


def _result_array(blocks, rows, ncol, dtype):
    """The whole result of a call, allocated up front: streamed blocks fill their rows when they sweep (``into``), resident
    ones when ``_collect`` asks — no per-block host arrays, no concatenation (at 8.1 Gcells those were 32-65 GB allocated,
    copied and freed again: seconds of page faults and munmap)."""
    out = np.empty((rows[-1][1], ncol), dtype)
    for b, blk in enumerate(blocks):
        if isinstance(blk, _StreamedBlock):
            blk.into = out[rows[b][0]:rows[b][1]]
    return out


def _collect(blocks, rows, out):
    for b, blk in enumerate(blocks):
        if not isinstance(blk, _StreamedBlock):
            blk.result(out[rows[b][0]:rows[b][1]])
    return out


LAST_SWEEPS = []  # sweeps per block of the last fixpoint iteration of this process (diagnostics, tools/bench_down_blocks.py)


class _UpBlock(_SeedGate):
    """Device-resident state of one row block of an up-sweep (accuflux, Strahler) between the exchanges: payload and
    result stay in HBM; only the two boundary rows travel."""

    def __init__(self, handle, kind, dtype, payload=None, by_row=False, nodata=(0, 0.0, 0), mask=None, direction=_hip.PFD_UP,
                 out=None):
        self.h, self.kind, self.dtype, self.by_row, self.nodata = handle, kind, np.dtype(dtype), by_row, nodata
        self.direction = direction
        ncol, dev = handle.ncol, handle.device
        self.nrows_dev = handle.nrow + sum(handle.halo)
        self.payload = self.mask = None
        if kind == "accuflux":
            payload = np.ascontiguousarray(payload, dtype=self.dtype)
            assert payload.size == (self.nrows_dev if by_row else self.nrows_dev * ncol)
            self.payload = payload if by_row else _hip.DeviceBuffer(payload.nbytes, dev).upload(payload)
        elif mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.size == self.nrows_dev * ncol
            self.mask = _hip.DeviceBuffer(mask.nbytes, dev).upload(mask)
        if kind == "distance":
            self.payload = None if payload is None else np.ascontiguousarray(payload, dtype=np.float32)  # step-length rows
        # (``out``: the caller's DEVICE buffer for the result over the block's device rows — stays resident, is not freed here,
        #  and ``result()`` hands it back instead of a host copy)
        self.out_given = out is not None
        if self.out_given:
            assert out.nbytes >= self.nrows_dev * ncol * self.dtype.itemsize
        self.out = out if self.out_given else _hip.DeviceBuffer(self.nrows_dev * ncol * self.dtype.itemsize, dev)
        self.swept_with, self.brows = None, None
        # accuflux "up" and Strahler: from the second sweep on only the chains below a changed halo seed are folded
        # again (pfd_set_block_update; the down-sweeps of accuflux "down" / stream_distance sweep the block each time)
        self.incremental = kind == "strahler" or (kind == "accuflux" and direction == _hip.PFD_UP)
        self.sweeps = 0

    def _call(self, seed, verify):
        if self.incremental and not verify:
            self.h.set_block_update(2 if self.sweeps else 1)
            self.sweeps += 1
        if self.kind == "accuflux":
            nd_i, nd_f, has_nd = self.nodata
            return self.h.accuflux_block(self.payload, _hip._PAYLOAD_CODE[self.dtype], seed, self.out, nd_i, nd_f, has_nd,
                                         by_row=self.by_row, verify=verify, memspace=_hip.PFD_DEVICE,
                                         direction=self.direction)
        if self.kind == "distance":
            return self.h.stream_distance_block(self.mask, self.payload, seed, self.out, verify=verify, memspace=_hip.PFD_DEVICE)
        return self.h.strahler_block(self.mask, seed, self.out, verify=verify, memspace=_hip.PFD_DEVICE)

    def sweep_dev(self, seed_buf):
        """Sweep with the halo values in the DEVICE buffer ``seed_buf`` (the handle reads device seeds:
        ``set_block_io(PFD_DEVICE)``); the boundary rows stay in ``self.out`` for the RCCL exchange."""
        self._call(seed_buf, False)

    def verify(self, seed):
        """Own cells whose value is not the one their upstream cells (halo values included) give."""
        return self._call(seed, True)[1]

    def result(self, out=None):
        if self.out_given:
            return self.out
        ncol, sz = self.h.ncol, self.dtype.itemsize
        return self.out.download(self.dtype, (self.h.nrow, ncol), offset_bytes=self.h.halo[0] * ncol * sz, out=out)

    def close(self, close_handle=True):
        for b in (self.payload, self.mask, None if self.out_given else self.out):
            if isinstance(b, _hip.DeviceBuffer):
                b.free()
        if close_handle:
            self.h.close()
        elif self.incremental and self.sweeps:
            self.h.set_block_update(0)  # (releases the kept sweep)


MAX_ROUNDS = 256  # default bound of the fixpoint iterations below (sharded HAND needs 11 on the roughest test raster)


def _not_settled(rounds):
    return NotImplementedError(f"row blocks: the boundary rows did not settle in {rounds} rounds — a cycle through "
                               "the block edges (the reference leaves cells on cycles untouched; blocks cannot), or a "
                               "path that crosses block edges more often than that (pass max_iter)")


def _up_blocks_run(blocks, ncol, dtype, max_iter=MAX_ROUNDS, verify=False):
    """Exchange boundary rows and sweep until no halo value changes: the fixpoint is the whole raster's result (the
    graph is acyclic; a value is final after as many exchanges as its longest upstream path crosses block edges).
    Returns the number of rounds in which some block swept.  A cycle through the block edges never settles (its sums
    grow with every round): the iteration is bounded by ``max_iter`` and raises.

    Order of the sweeps: a block whose relevant halo values (relevant_halo) all come from blocks that are FINAL is swept
    once and is final itself — flow that crosses the blocks one way costs every block one sweep, in the order of the
    flow, instead of one per block it has downstream.  Blocks that wait for each other (a path that leaves a block and
    comes back) iterate as before: every one of them whose relevant halo values changed sweeps, until none does."""
    nb = len(blocks)
    seeds = [np.zeros(2 * ncol, dtype) for _ in range(nb)]

    def deps(b):  # the neighbouring blocks whose boundary rows block b's result depends on
        rel = blocks[b].relevant
        up = b > 0 and (rel is None or bool(rel[:ncol].any()))
        dn = b + 1 < nb and (rel is None or bool(rel[ncol:].any()))
        return ([b - 1] if up else []) + ([b + 1] if dn else [])

    dep = [deps(b) for b in range(nb)]
    final = [False] * nb
    it = 0
    while not all(final):
        ready = [b for b in range(nb) if not final[b] and all(final[d] for d in dep[b])]
        if ready:
            swept = [blocks[b].sweep(seeds[b]) for b in ready]
            for b in ready:
                final[b] = True
        else:  # mutual dependence among the blocks that are left
            swept = [blocks[b].sweep(seeds[b]) for b in range(nb) if not final[b]]
            if not any(swept):
                break
        if any(swept):
            it += 1
            if max_iter is not None and it > max_iter:
                raise _not_settled(max_iter)
        for b in range(nb):  # halo rows = the neighbours' boundary rows
            if b > 0 and blocks[b - 1].brows is not None:
                seeds[b][:ncol] = blocks[b - 1].brows[1]
            if b + 1 < nb and blocks[b + 1].brows is not None:
                seeds[b][ncol:] = blocks[b + 1].brows[0]
    LAST_SWEEPS[:] = [blk.calls for blk in blocks]
    bad = sum(blk.verify(seeds[b]) for b, blk in enumerate(blocks)) if verify else None
    return it, bad


def accuflux_blocks(d8: np.ndarray, nblocks: int, data, nodata_args=(0, 0.0, 0), by_row=False, devices=None,
                    verify=False, max_iter=MAX_ROUNDS, direction="up"):
    """``accuflux(data, direction)`` (reference pyflwdir/streams.py:15-41, :44-70) of a host raster computed as ``nblocks``
    row blocks held by this one process — for rasters beyond 2**32 - 2 cells, and the in-process form of the
    multi-GPU protocol.  ``data``: the payload raster (int32 / int64 / float32 / float64), or with ``by_row`` one value
    per raster row (cell areas of a regular grid).  ``nodata_args`` = (nodata_i, nodata_f, has_nodata) as
    raster._payload_args returns them.  Returns (result, rounds, bad cells or None).

    Float sums are not associative: a boundary cell adds the values of the halo cells draining into it in the serial
    loop's position, so every sum has the whole raster's operands in the whole raster's order (bit-identical)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    data = np.asarray(data)
    dtype = data.dtype
    if dtype not in _hip._PAYLOAD_CODE:
        raise NotImplementedError(f"payload dtype {dtype} is not supported by the row-block accuflux")
    data = data.reshape(nrow) if by_row else data.reshape(nrow, ncol)
    devices = devices or [0] * nblocks
    rows = block_rows(nrow, nblocks)
    dirc = _hip.PFD_UP if direction == "up" else _hip.PFD_DOWN

    def make(b):
        a, e = block_slice(nrow, nblocks, b)
        h = _hip.RasterHandle(d8[a:e], rows[b][1] - rows[b][0], ncol, device=devices[b], halo=halo_of(b, nblocks))
        return _UpBlock(h, "accuflux", dtype, payload=data[a:e], by_row=by_row, nodata=nodata_args, direction=dirc)

    blocks = _blocks_of(nblocks, make, lambda b: relevant_halo(d8[slice(*block_slice(nrow, nblocks, b))], halo_of(b, nblocks),
                                                               down=direction != "up"),
                        _stream_blocks(d8.size, (1 if by_row else 2) * dtype.itemsize + 28, devices))
    try:
        out = _result_array(blocks, rows, ncol, dtype)
        it, bad = _up_blocks_run(blocks, ncol, dtype, max_iter=max_iter, verify=verify)
        return _collect(blocks, rows, out), it, bad
    finally:
        for blk in blocks:
            blk.close()


def stream_distance_blocks(d8: np.ndarray, nblocks: int, mask=None, step_lengths=None, devices=None, verify=False,
                           max_iter=MAX_ROUNDS):
    """``stream_distance`` (reference pyflwdir/streams.py:272-315) of a host raster computed as ``nblocks`` row blocks
    held by this one process: int32 cell counts, or float32 metres with ``step_lengths`` (the whole raster's table,
    [2 * nrow - 1, 3]: gis.step_length_table).  Returns (distances, rounds, bad cells or None)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(nrow, ncol)
    tab = None if step_lengths is None else np.ascontiguousarray(step_lengths, dtype=np.float32).reshape(2 * nrow - 1, 3)
    dtype = np.int32 if tab is None else np.float32
    devices = devices or [0] * nblocks
    brows = block_rows(nrow, nblocks)

    def make(b):
        a, e = block_slice(nrow, nblocks, b)
        h = _hip.RasterHandle(d8[a:e], brows[b][1] - brows[b][0], ncol, device=devices[b], halo=halo_of(b, nblocks))
        rows = None if tab is None else tab[2 * a:2 * a + 2 * (e - a) - 1]  # (row sums 2a .. 2(e-1): the block's steps)
        return _UpBlock(h, "distance", dtype, payload=rows, mask=None if mask is None else mask[a:e])

    blocks = _blocks_of(nblocks, make, lambda b: relevant_halo(d8[slice(*block_slice(nrow, nblocks, b))], halo_of(b, nblocks), down=True),
                        _stream_blocks(d8.size, 4 + (mask is not None) + 28, devices))
    try:
        out = _result_array(blocks, brows, ncol, dtype)
        it, bad = _up_blocks_run(blocks, ncol, dtype, max_iter=max_iter, verify=verify)
        return _collect(blocks, brows, out), it, bad
    finally:
        for blk in blocks:
            blk.close()


def strahler_blocks(d8: np.ndarray, nblocks: int, mask=None, devices=None, verify=False, max_iter=MAX_ROUNDS):
    """Strahler stream order (reference pyflwdir/streams.py:228-269) of a host raster computed as ``nblocks`` row
    blocks held by this one process; ``mask`` (uint8, optional) as in ``stream_order``.  Returns (order, rounds,
    bad cells or None).  The raster must be acyclic: on a cycle through a block edge the orders can settle on values
    the reference never assigns (FlwdirRaster checks with one rank query before it cuts a raster into blocks)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(nrow, ncol)
    devices = devices or [0] * nblocks
    brows = block_rows(nrow, nblocks)

    def make(b):
        a, e = block_slice(nrow, nblocks, b)
        h = _hip.RasterHandle(d8[a:e], brows[b][1] - brows[b][0], ncol, device=devices[b], halo=halo_of(b, nblocks))
        return _UpBlock(h, "strahler", np.uint8, mask=None if mask is None else mask[a:e])

    blocks = _blocks_of(nblocks, make, lambda b: relevant_halo(d8[slice(*block_slice(nrow, nblocks, b))], halo_of(b, nblocks), down=False),
                        _stream_blocks(d8.size, 1 + (mask is not None) + 28, devices))
    try:
        out = _result_array(blocks, brows, ncol, np.uint8)
        it, bad = _up_blocks_run(blocks, ncol, np.uint8, max_iter=max_iter, verify=verify)
        return _collect(blocks, brows, out), it, bad
    finally:
        for blk in blocks:
            blk.close()


class _ClassicBlock(_SeedGate):
    """Device-resident state of one row block of the classic stream order between the exchanges: the per-cell byte of
    pfd_trib_info_block (halo rows from the neighbours), the mask and the orders stay in HBM."""

    def __init__(self, handle, tinfo_rows, mask_rows):
        self.h = handle
        ncol, dev = handle.ncol, handle.device
        self.nrows_dev = handle.nrow + sum(handle.halo)
        self.tinfo = _hip.DeviceBuffer(tinfo_rows.nbytes, dev).upload(np.ascontiguousarray(tinfo_rows))
        self.mask = None if mask_rows is None else _hip.DeviceBuffer(mask_rows.nbytes, dev).upload(np.ascontiguousarray(mask_rows))
        self.out = _hip.DeviceBuffer(self.nrows_dev * ncol, dev)
        self.swept_with, self.brows = None, None

    def _call(self, seed, verify):
        return self.h.stream_order_classic_block(self.tinfo, self.mask, seed, self.out, verify=verify, memspace=_hip.PFD_DEVICE)

    def verify(self, seed):
        return self._call(seed, True)[1]

    def result(self, out=None):
        ncol = self.h.ncol
        return self.out.download(np.uint8, (self.h.nrow, ncol), offset_bytes=self.h.halo[0] * ncol, out=out)

    def close(self, close_handle=True):
        for b in (self.tinfo, self.mask, self.out):
            if b is not None:
                b.free()
        if close_handle:
            self.h.close()


def classic_blocks(d8: np.ndarray, nblocks: int, uparea, mask=None, upa_min=0.0, devices=None, verify=False,
                   max_iter=MAX_ROUNDS):
    """Classic stream order (reference pyflwdir/streams.py:191-225: the order grows by one at every confluence where the
    cell is NOT its downstream cell's main upstream cell, core.py:191-219) of a host raster computed as ``nblocks`` row
    blocks held by this one process — what ``FlwdirRaster.stream_order(type="classic")`` runs beyond 2**32 - 2 cells.
    ``uparea``: the raster that ranks the upstream cells (the reference's default: the upstream cell count).  No index
    array is built: a block computes one byte per cell (main upstream SLOT, several-upstream-cells flag), the blocks
    swap the bytes of their boundary rows once, then sweep down- to upstream with their halo cells holding the
    neighbours' orders until no boundary row changes.  Returns (uint8 orders, rounds, bad cells or None)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    uparea = np.ascontiguousarray(uparea).reshape(nrow, ncol)
    if uparea.dtype not in _hip._PAYLOAD_CODE:
        raise NotImplementedError(f"uparea dtype {uparea.dtype} is not supported by the row-block stream order")
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(nrow, ncol)
    devices = devices or [0] * nblocks
    brows = block_rows(nrow, nblocks)
    stream = _stream_blocks(d8.size, 3 + uparea.dtype.itemsize + 28, devices)

    def handle(b):
        a, e = block_slice(nrow, nblocks, b)
        return _hip.RasterHandle(d8[a:e], brows[b][1] - brows[b][0], ncol, device=devices[b], halo=halo_of(b, nblocks))

    handles, blocks = {}, []
    try:
        infos = []
        for b in range(nblocks):
            a, e = block_slice(nrow, nblocks, b)
            h = handle(b)
            try:
                infos.append(h.trib_info_block(uparea[a:e], _hip._PAYLOAD_CODE[uparea.dtype], None if mask is None else mask[a:e],
                                               upa_min).reshape(e - a, ncol))
            finally:
                if stream:
                    h.close()
                else:
                    handles[b] = h  # (kept: the block below owns it)
        for b in range(nblocks):  # a halo cell's byte is its owner's: the neighbouring block's boundary row
            top, bot = halo_of(b, nblocks)
            if top:
                infos[b][0] = infos[b - 1][-1 - halo_of(b - 1, nblocks)[1]]
            if bot:
                infos[b][-1] = infos[b + 1][halo_of(b + 1, nblocks)[0]]

        def make(b):
            a, e = block_slice(nrow, nblocks, b)
            return _ClassicBlock(handles.pop(b) if b in handles else handle(b), infos[b], None if mask is None else mask[a:e])

        blocks = _blocks_of(nblocks, make, lambda b: relevant_halo(d8[slice(*block_slice(nrow, nblocks, b))], halo_of(b, nblocks),
                                                                   down=True), stream)
        out = _result_array(blocks, brows, ncol, np.uint8)
        it, bad = _up_blocks_run(blocks, ncol, np.uint8, max_iter=max_iter, verify=verify)
        return _collect(blocks, brows, out), it, bad
    finally:
        for h in handles.values():
            h.close()
        for blk in blocks:
            blk.close()


class _FloodBlock(_SeedGate):
    """Device-resident state of one row block of dem.floodplains between the exchanges: elevation, stream flags, height
    thresholds and the floodplain state (16 bytes per cell) stay in HBM; only the two boundary rows of the state travel."""

    def __init__(self, handle, elevtn_rows, elev_code, is_stream_rows, stream_h_rows, pool=None):
        self.h, self.code = handle, elev_code
        ncol, dev = handle.ncol, handle.device
        self.nrows_dev = handle.nrow + sum(handle.halo)
        # ``pool`` (streamed blocks: {"rows": the most device rows any block has}): the four buffers are allocated once
        # for the largest block and serve every block of the call — a hipMalloc of tens of GB takes longer than filling it
        self.pool = pool

        def buf(name, bytes_per_cell):
            if pool is None:
                return _hip.DeviceBuffer(self.nrows_dev * ncol * bytes_per_cell, dev)
            if name not in pool:
                pool[name] = _hip.DeviceBuffer(max(pool["rows"], self.nrows_dev) * ncol * bytes_per_cell, dev)
            return pool[name]

        up = lambda name, a: buf(name, a.dtype.itemsize).upload(np.ascontiguousarray(a))  # noqa: E731
        self.elev, self.stream, self.hs = up("elev", elevtn_rows), up("stream", is_stream_rows), up("hs", stream_h_rows)
        self.state = buf("state", _hip.FLOOD_STATE.itemsize)
        self.swept_with, self.brows = None, None

    def _call(self, seed, verify):
        return self.h.floodplains_block(self.elev, self.code, self.stream, self.hs, seed, self.state, verify=verify,
                                        memspace=_hip.PFD_DEVICE)

    def verify(self, seed):
        return self._call(seed, True)[1]

    def result(self, out=None):
        """int8 flags of the own rows (``out``: the block's rows of the whole result, filled in place) — extracted from the
        16-byte state records on the device (pfd_floodplains_block_flags): 1 byte per cell crosses PCIe."""
        return self.h.floodplains_block_flags(self.state, out)

    def close(self, close_handle=True):
        if self.pool is None:
            for b in (self.elev, self.stream, self.hs, self.state):
                b.free()
        if close_handle:
            self.h.close()


def floodplains_blocks(d8: np.ndarray, nblocks: int, elevtn, is_stream, stream_h, devices=None, verify=False,
                       max_iter=MAX_ROUNDS):
    """``dem.floodplains`` (reference pyflwdir/dem.py:333-379) of a host raster computed as ``nblocks`` row blocks held by
    this one process — what ``FlwdirRaster.floodplains`` runs beyond 2**32 - 2 cells.  ``is_stream`` (uint8: upstream area
    >= upa_min) and ``stream_h`` (float32: uparea ** b on the stream cells, evaluated by the caller in the reference's
    dtype) as ``pfd_floodplains`` takes them.  A cell joins the floodplain of its DOWNSTREAM cell, so a block needs the
    state (z, h, flag) of the cells it drains into across its edges: the halo rows hold the neighbours' boundary rows
    of the state, exchanged until no row changes.  Returns (int8 flags, rounds, bad cells or None)."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    elevtn = np.ascontiguousarray(elevtn).reshape(nrow, ncol)
    if elevtn.dtype not in _ELEV_CODE:
        raise NotImplementedError(f"elevation dtype {elevtn.dtype} is not supported on the HIP path")
    is_stream = np.ascontiguousarray(is_stream, dtype=np.uint8).reshape(nrow, ncol)
    stream_h = np.ascontiguousarray(stream_h, dtype=np.float32).reshape(nrow, ncol)
    devices = devices or [0] * nblocks
    brows = block_rows(nrow, nblocks)

    stream = _stream_blocks(d8.size, elevtn.dtype.itemsize + 5 + 2 * _hip.FLOOD_STATE.itemsize + 28, devices)
    pool = {"rows": max(e - a for a, e in (block_slice(nrow, nblocks, b) for b in range(nblocks)))} if stream else None

    def make(b):
        a, e = block_slice(nrow, nblocks, b)
        h = _hip.RasterHandle(d8[a:e], brows[b][1] - brows[b][0], ncol, device=devices[b], halo=halo_of(b, nblocks))
        return _FloodBlock(h, elevtn[a:e], _ELEV_CODE[elevtn.dtype], is_stream[a:e], stream_h[a:e], pool=pool)

    blocks = _blocks_of(nblocks, make, lambda b: relevant_halo(d8[slice(*block_slice(nrow, nblocks, b))], halo_of(b, nblocks), down=True),
                        stream)
    try:
        out = np.empty((nrow, ncol), np.int8)  # (every block writes its own rows: no concatenation of 8 GB at 8.1 Gcells)
        for b, blk in enumerate(blocks):
            if isinstance(blk, _StreamedBlock):
                blk.into = out[brows[b][0]:brows[b][1]]
        it, bad = _up_blocks_run(blocks, ncol, _hip.FLOOD_STATE, max_iter=max_iter, verify=verify)
        for b, blk in enumerate(blocks):
            if not isinstance(blk, _StreamedBlock):
                blk.result(out[brows[b][0]:brows[b][1]])
        return out, it, bad
    finally:
        for blk in blocks:
            blk.close()
        for b in (pool or {}).values():
            if isinstance(b, _hip.DeviceBuffer):
                b.free()


class _HandBlock:
    """Device-resident state of one row block's HAND between the exchanges: drain, elevation and the heights stay in
    HBM; only the two boundary rows and the number of unknown cells travel."""

    def __init__(self, handle, drain_rows, elevtn_rows, code, out=None):
        self.h, self.code = handle, code
        ncol = handle.ncol
        self.nrows_dev = handle.nrow + sum(handle.halo)
        dev = handle.device
        self.owned = not isinstance(drain_rows, _hip.DeviceBuffer)  # (device-resident inputs stay the caller's)
        if self.owned:
            drain_rows = np.ascontiguousarray(drain_rows)
            elevtn_rows = np.ascontiguousarray(elevtn_rows)
            assert drain_rows.size == elevtn_rows.size == self.nrows_dev * ncol
            self.drain = _hip.DeviceBuffer(drain_rows.nbytes, dev).upload(drain_rows)
            self.elev = _hip.DeviceBuffer(elevtn_rows.nbytes, dev).upload(elevtn_rows)
        else:
            self.drain, self.elev = drain_rows, elevtn_rows
        self.out = out if out is not None else _hip.DeviceBuffer(self.nrows_dev * ncol * 8, dev)
        self.keep_out = out is not None  # (the caller's buffer, or result_device(): the caller takes the result buffer)
        self.swept_with, self.brows, self.unknown = None, None, None

    def sweep(self, seed):
        if self.swept_with is not None and np.array_equal(self.swept_with.view(np.uint64), seed.view(np.uint64)):
            return  # the heights of the halo cells did not change: neither did the block
        update = self.swept_with is not None
        self.swept_with = seed.copy()
        _, self.brows, self.unknown = self.h.hand_block(self.drain, self.elev, self.code, seed, out=self.out,
                                                        memspace=_hip.PFD_DEVICE, update=update)

    def sweep_dev(self, seed_buf, update):
        """Sweep (or, ``update``, relax the unknown cells) with the halo heights in the DEVICE buffer ``seed_buf``."""
        _, _, self.unknown = self.h.hand_block(self.drain, self.elev, self.code, seed_buf, out=self.out,
                                               memspace=_hip.PFD_DEVICE, update=update)

    def result(self):
        ncol = self.h.ncol
        return self.out.download(np.float64, (self.h.nrow, ncol), offset_bytes=self.h.halo[0] * ncol * 8)

    def result_device(self):
        """The device buffer of the block's result (own + halo rows); the caller frees it."""
        self.keep_out = True
        return self.out

    def close(self, close_handle=True):
        for b in ((self.drain, self.elev) if self.owned else ()) + (() if self.keep_out else (self.out,)):
            b.free()
        if close_handle:
            self.h.close()


class _StreamedHandBlock:
    """A row block of HAND that is on the device only while it sweeps (_stream_blocks): the heights wait on the host,
    own and halo rows; a later sweep sends them back and relaxes the cells that are still unknown (pfd_hand_block with
    host memory rebuilds its list of unknown cells by one scan)."""

    def __init__(self, d8_rows, own_rows, ncol, device, halo, drain_rows, elevtn_rows, code):
        self.args = (d8_rows, own_rows, ncol, device, halo)
        self.drain, self.elev, self.code = np.ascontiguousarray(drain_rows), np.ascontiguousarray(elevtn_rows), code
        self.own_rows, self.ncol, self.top = own_rows, ncol, halo[0]
        self.out, self.swept_with, self.brows, self.unknown = None, None, None, None

    def sweep(self, seed):
        if self.swept_with is not None and np.array_equal(self.swept_with.view(np.uint64), seed.view(np.uint64)):
            return
        update = self.swept_with is not None
        self.swept_with = seed.copy()
        d8_rows, own_rows, ncol, device, halo = self.args
        h = _hip.RasterHandle(d8_rows, own_rows, ncol, device=device, halo=halo)
        try:
            self.out, self.brows, self.unknown = h.hand_block(self.drain, self.elev, self.code, seed, out=self.out,
                                                              memspace=_hip.PFD_HOST, update=update)
        finally:
            h.close()

    def result(self):
        return self.out.reshape(-1, self.ncol)[self.top:self.top + self.own_rows]

    def close(self, close_handle=True):
        pass


def exchange_unique_id(rank: int, world: int, group=None) -> bytes:
    """Rank 0 creates the RCCL unique id, everybody receives it through the host group."""
    group = group or _default_group(rank, world)
    uid = _hip.Communicator.unique_id() if rank == 0 else b""
    return group.bcast(uid, 0)


_GROUP = None


def _default_group(rank, world):
    """The process-wide HostGroup (plain TCP rendezvous, MASTER_ADDR / MASTER_PORT + 23): created on first use."""
    global _GROUP
    if _GROUP is None or _GROUP.world != world:
        from .hostgroup import HostGroup

        _GROUP = HostGroup(rank, world)
    return _GROUP


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: a caller's stdout (bench.py: ONE JSON line) stays clean
    whatever a native library prints while it initialises."""

    def __enter__(self):
        import os
        import sys

        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import os
        import sys

        sys.stdout.flush()
        try:  # (the library printed through C stdio, which buffers when stdout is a pipe: flush it while fd 1 is stderr)
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _try_rccl(uid, rank, world, device, timeout):
    """Create the RCCL communicator in a watchdog thread: a bootstrap that cannot reach its peers
    blocks forever instead of failing.  A creation that finishes after the watchdog gave up destroys its
    communicator again."""
    import threading

    box, lock = {}, threading.Lock()

    def work():
        try:
            comm = _hip.Communicator(uid, rank, world, device)
        except Exception as exc:  # noqa: BLE001 - any failure means "use the host transport"
            comm, box["err"] = None, exc
        with lock:
            if box.get("abandoned") and comm is not None:
                comm.close()
            else:
                box["comm"] = comm

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout)
    with lock:
        if "comm" not in box:
            box["abandoned"] = True
        return box.get("comm")


class DistributedRaster:
    """The row block of this rank; ``upstream_area()`` is collective over the process group.

    ``group``: anything with ``bcast(bytes, src) / allgather(bytes) -> [bytes] / allreduce(x, op)`` —
    by default a :class:`pyflwdir_amd.hostgroup.HostGroup` (plain TCP, no PyTorch, no MPI).

    transport="rccl": one ``ncclAllGather`` of the 4*ncol-word boundary record over xGMI (default).
    transport="host": the same record travels through ``group.allgather``; identical kernels on either
    side of the exchange (split-phase C-ABI).  With transport="auto" RCCL is tried first and every rank
    falls back to "host" if any rank could not create its communicator.
    """

    def __init__(self, d8_block, own_rows: int, ncol: int, rank: int, world: int, device: int,
                 memspace=_hip.PFD_HOST, transport="auto", group=None, rccl_timeout=120.0, deferred=False):
        self.rank, self.world, self.device = rank, world, device
        self.group = group or _default_group(rank, world)
        # a rank whose block cannot be created (bad code, out of memory) still takes part in the
        # collectives below and raises afterwards: nobody is left waiting
        err = None
        self.handle = None
        self._edge = None  # host callers: the first / last two device rows of the raw codes (relevant_halo)
        if isinstance(d8_block, np.ndarray) and d8_block.size >= 2 * ncol:
            self._edge = np.asarray(d8_block, np.uint8).reshape(-1, ncol)[[0, 1, -2, -1]].copy()
        try:
            self.handle = _hip.RasterHandle(d8_block, own_rows, ncol, device=device, memspace=memspace,
                                            halo=halo_of(rank, world), deferred=deferred)
        except Exception as exc:  # noqa: BLE001
            err = exc
        self.comm = None
        if transport in ("auto", "rccl"):
            with _stdout_to_stderr():  # (RCCL prints a banner — host name, library path — on stdout when it comes up)
                uid = exchange_unique_id(rank, world, self.group)
                if err is None:
                    self.comm = _try_rccl(uid, rank, world, device, rccl_timeout)
            ok = self.group.allreduce(1 if self.comm is not None else 0, "min")
            if ok == 0:
                if self.comm is not None:
                    self.comm.close()  # created here, unusable elsewhere
                    self.comm = None
                if transport == "rccl" and err is None:
                    err = RuntimeError("RCCL communicator could not be created on every rank")
        all_ok = self.group.allreduce(0 if err is not None else 1, "min")
        if err is not None:
            raise err
        if all_ok == 0:
            raise RuntimeError("another rank could not create its row block")
        self.transport = "rccl" if self.comm is not None else "host"
        self.exchanges = []  # (transport, bytes) of every boundary exchange of the collectives below: what a caller reports

    def upstream_area(self, out=None, memspace=_hip.PFD_HOST):
        if self.comm is not None:
            res = self.comm.upstream_area_cell(self.handle, out=out, memspace=memspace)
        else:
            # every rank reaches both collectives whatever happens locally (a rank that raised early
            # would leave the others waiting): a local failure travels with the final agreement
            err, res, ok = None, out, False
            rec = np.zeros(4 * self.handle.ncol, np.uint32)
            try:
                res, rec = _hip.upstream_area_cell_begin(self.handle, out=out, memspace=memspace)
            except Exception as exc:  # noqa: BLE001
                err = exc
            parts = self.group.allgather(rec.tobytes())
            if err is None:
                try:
                    allrec = np.stack([np.frombuffer(p, np.uint32) for p in parts])
                    ok = _hip.upstream_area_cell_finish(self.handle, allrec, self.world, self.rank)
                except Exception as exc:  # noqa: BLE001
                    err = exc
            flag = self.group.allreduce(1 if (ok and err is None) else 0, "min")
            if err is not None:
                raise err
            if flag == 0:
                raise NotImplementedError("a row block failed or the raster holds cells that never reach a pit "
                                          "(cycles); the multi-GPU path requires a valid flow direction raster "
                                          "(FlwdirRaster.isvalid)")
        return res.reshape(self.handle.nrow, self.handle.ncol) if memspace == _hip.PFD_HOST else res

    def basins(self, idxs_global, ids, nrow_total: int, out=None, memspace=_hip.PFD_HOST):
        """Collective ``basins``: every rank passes the same outlets (global linear indices) and ids; returns
        the labels of this rank's rows.  The 6*ncol-word boundary records travel in one all-gather — RCCL when the
        communicator exists (``pfd_comm_allgather_host``), else the host group."""
        ncol = self.handle.ncol
        mine = _split_outlets(idxs_global, ids, nrow_total, ncol, self.world)[self.rank]
        err, res, ok = None, out, False
        rec = np.zeros(6 * ncol, np.uint32)
        try:
            res, rec = _hip.basins_begin(self.handle, mine[0], mine[1], out=out, memspace=memspace)
        except Exception as exc:  # noqa: BLE001 - the failure travels with the final agreement
            err = exc
        parts = self._allgather(rec.tobytes())
        if err is None:
            try:
                ok = _hip.basins_finish(self.handle, np.stack([np.frombuffer(p, np.uint32) for p in parts]), self.world,
                                        self.rank)
            except Exception as exc:  # noqa: BLE001
                err = exc
        flag = self.group.allreduce(1 if (ok and err is None) else 0, "min")
        if err is not None:
            raise err
        if flag == 0:
            raise NotImplementedError("a row block failed or the raster holds a cycle through several row blocks")
        return res.reshape(self.handle.nrow, ncol) if memspace == _hip.PFD_HOST else res

    def hand(self, drain_block, elevtn_block, max_iter=None, elev_code=None, out=None, check_finite=True):
        """Collective ``hand(drain, elevtn)`` (reference pyflwdir/dem.py:299-330): every rank passes the rows of its
        block INCLUDING its halo rows (like the D8 codes); returns (float64 heights of the rank's own rows,
        iterations).  Bit-identical to the whole raster: see :func:`hand_blocks`.  Per iteration the two boundary
        rows travel — device to device between neighbours with the RCCL transport, through one all-gather of the host
        group otherwise — plus one agreement on the number of unknown cells.  Device-resident inputs
        (``_hip.DeviceBuffer`` + ``elev_code``) are used in place and the result stays on the device (a DeviceBuffer
        covering own + halo rows, owned by the caller; ``out``: use this device buffer for it).  Device-resident
        elevations are checked for NaN / inf on the device (one read of the array, like numpy's check of host inputs)
        unless ``check_finite=False`` — for a caller that repeats the call on a buffer it has checked."""
        h = self.handle
        ncol = h.ncol
        self._device_io = isinstance(drain_block, _hip.DeviceBuffer)
        if self._device_io:
            drain, elevtn, code = drain_block, elevtn_block, elev_code
            nonfinite = None
            if check_finite and _hip.count_nonfinite(elevtn, (h.nrow + sum(h.halo)) * ncol, code, h.device):
                # (raised below, inside the block set-up: the failure travels with the agreement)
                nonfinite = NotImplementedError("hand over row blocks needs finite elevations (-inf marks heights that "
                                                "are not known yet); mask or fill inf / NaN cells first")
        else:
            drain, elevtn, code = _hand_inputs(drain_block, elevtn_block)
            nonfinite = None
        seed = np.full(2 * ncol, -np.inf)
        unknown_before, it = None, 0
        blk, err = None, None
        try:
            if nonfinite is not None:
                raise nonfinite
            blk = _HandBlock(h, drain, elevtn, code, out=out)
        except Exception as exc:  # noqa: BLE001 - the failure travels with the agreement: nobody is left waiting
            err = exc
        if self.comm is not None:
            return self._hand_rccl(blk, err, seed, max_iter)
        try:
            while True:
                it += 1
                rec, mine = np.zeros(2 * ncol), -1
                if err is None:
                    try:
                        blk.sweep(seed)
                        rec, mine = blk.brows.ravel(), blk.unknown
                    except Exception as exc:  # noqa: BLE001
                        err = exc
                parts = self.group.allgather(rec.tobytes())
                self.exchanges.append(("host_allgather", rec.nbytes))
                counts = [int(x) for x in np.frombuffer(b"".join(self.group.allgather(np.int64(mine).tobytes())), np.int64)]
                if err is not None:
                    raise err
                if min(counts) < 0:
                    raise RuntimeError("another rank failed in hand()")
                unknown = sum(counts)
                if unknown == 0:
                    return (blk.result_device() if self._device_io else blk.result()), it
                if unknown == unknown_before or (max_iter is not None and it >= max_iter):
                    raise NotImplementedError("hand: heights that depend on each other through several row blocks "
                                              "(a cycle through the block edges)")
                unknown_before = unknown
                if self.rank > 0:
                    seed[:ncol] = np.frombuffer(parts[self.rank - 1], np.float64)[ncol:]
                if self.rank + 1 < self.world:
                    seed[ncol:] = np.frombuffer(parts[self.rank + 1], np.float64)[:ncol]
        finally:
            if blk is not None:
                blk.close(close_handle=False)

    def _allgather(self, data: bytes):
        """One-shot all-gather of a boundary record: RCCL (through the device) when the communicator exists."""
        if self.world == 1:  # (nobody to exchange with: the record is its own gather)
            self.exchanges.append(("rccl_allgather" if self.comm is not None else "host_allgather", 0))
            return [bytes(data)]
        if self.comm is not None:
            self.exchanges.append(("rccl_allgather", len(data)))
            return self.comm.allgather_host(self.handle, data)
        self.exchanges.append(("host_allgather", len(data)))
        return self.group.allgather(data)

    def _agree_ready(self, err):
        """Before an RCCL loop: a rank whose block could not be set up has no device rows to exchange — one host
        agreement per CALL (not per exchange), then every rank raises together."""
        bad = self.group.allreduce(1 if err is not None else 0, "sum")
        if err is not None:
            raise err
        if bad:
            raise RuntimeError("another rank could not set up its row block")

    def _hand_rccl(self, blk, err, seed0, max_iter):
        """hand(): the exchange loop with the boundary rows travelling device to device (ncclSend / ncclRecv between
        neighbours, one ncclAllReduce of the unknown / failed / changed counts per exchange; nothing but 32 bytes of
        counts crosses PCIe per exchange)."""
        h, ncol = self.handle, self.handle.ncol
        try:
            self._agree_ready(err)
            seed = _hip.DeviceBuffer(seed0.nbytes, h.device).upload(seed0)
            h.set_block_io(_hip.PFD_DEVICE)
            try:
                it, unknown_before, changed = 0, None, True
                while True:
                    it += 1
                    mine, fail = 0, None
                    try:
                        if it == 1 or changed:
                            blk.sweep_dev(seed, update=it > 1)
                        mine = blk.unknown
                    except Exception as exc:  # noqa: BLE001 - travels with the counts: nobody is left waiting
                        fail = exc
                    unknown, nfail, changed, _ = self.comm.exchange_rows(h, blk.out, 8, seed, 0 if fail else mine, 1 if fail else 0)
                    self.exchanges.append(("rccl_sendrecv", 2 * ncol * 8))
                    if fail is not None:
                        raise fail
                    if nfail:
                        raise RuntimeError("another rank failed in hand()")
                    if unknown == 0:
                        return (blk.result_device() if self._device_io else blk.result()), it
                    if unknown == unknown_before or (max_iter is not None and it >= max_iter):
                        raise NotImplementedError("hand: heights that depend on each other through several row blocks "
                                                  "(a cycle through the block edges)")
                    unknown_before = unknown
            finally:
                h.set_block_io(_hip.PFD_HOST)
                seed.free()
        finally:
            if blk is not None:
                blk.close(close_handle=False)

    def _up_rccl(self, blk, err, dtype, max_iter):
        """The fixpoint loop of an up-sweep with the boundary rows travelling device to device (see _hand_rccl)."""
        h, ncol = self.handle, self.handle.ncol
        try:
            self._agree_ready(err)
            seed = _hip.DeviceBuffer(2 * ncol * dtype.itemsize, h.device).upload(np.zeros(2 * ncol, dtype))
            h.set_block_io(_hip.PFD_DEVICE)
            try:
                it, changed = 0, True
                while True:
                    swept, fail = 0, None
                    try:
                        if changed:  # (the halo values differ from the ones of the last sweep; first round: sweep)
                            blk.sweep_dev(seed)
                            swept = 1
                    except Exception as exc:  # noqa: BLE001
                        fail = exc
                    nswept, nfail, changed, _ = self.comm.exchange_rows(h, blk.out, dtype.itemsize, seed, swept, 1 if fail else 0)
                    self.exchanges.append(("rccl_sendrecv", 2 * ncol * dtype.itemsize))
                    if fail is not None:
                        raise fail
                    if nfail:
                        raise RuntimeError("another rank failed in the row-block sweep")
                    if nswept == 0:
                        return blk.result(), it
                    it += 1
                    if max_iter is not None and it > max_iter:
                        raise _not_settled(max_iter)
            finally:
                h.set_block_io(_hip.PFD_HOST)
                seed.free()
        finally:
            if blk is not None:
                blk.close(close_handle=False)

    def _up_collective(self, make_block, dtype, max_iter=MAX_ROUNDS):
        """Collective fixpoint of an up-sweep over the ranks' blocks (see :func:`_up_blocks_run`): per round one
        all-gather of the two boundary rows and one agreement on whether any rank swept."""
        ncol = self.handle.ncol
        dtype = np.dtype(dtype)
        seed = np.zeros(2 * ncol, dtype)
        blk, err, it = None, None, 0
        try:
            blk = make_block()
            if getattr(self, "_edge", None) is not None and isinstance(blk, _UpBlock):  # (host transport: a rank sweeps again only when a halo value it depends on changed;
                #  the RCCL loop counts every changed halo value on the device)
                down = blk.kind == "distance" or (blk.kind == "accuflux" and blk.direction == _hip.PFD_DOWN)
                blk.relevant = relevant_halo(self._edge, self.handle.halo, down)
        except Exception as exc:  # noqa: BLE001 - the failure travels with the agreement: nobody is left waiting
            err = exc
        if self.comm is not None:
            return self._up_rccl(blk, err, dtype, max_iter)
        try:
            while True:
                rec, mine = np.zeros(2 * ncol, dtype), -1
                if err is None:
                    try:
                        mine = 1 if blk.sweep(seed) else 0
                        rec = blk.brows.ravel()
                    except Exception as exc:  # noqa: BLE001
                        err = exc
                parts = self.group.allgather(rec.tobytes())
                self.exchanges.append(("host_allgather", rec.nbytes))
                flags = [int(x) for x in np.frombuffer(b"".join(self.group.allgather(np.int64(mine).tobytes())), np.int64)]
                if err is not None:
                    raise err
                if min(flags) < 0:
                    raise RuntimeError("another rank failed in the row-block sweep")
                if max(flags) == 0:
                    return blk.result(), it
                it += 1
                if max_iter is not None and it > max_iter:
                    raise _not_settled(max_iter)
                if self.rank > 0:
                    seed[:ncol] = np.frombuffer(parts[self.rank - 1], dtype)[ncol:]
                if self.rank + 1 < self.world:
                    seed[ncol:] = np.frombuffer(parts[self.rank + 1], dtype)[:ncol]
        finally:
            if blk is not None:
                blk.close(close_handle=False)

    def accuflux(self, data_block, nodata_args=(0, 0.0, 0), by_row=False, max_iter=MAX_ROUNDS, direction="up", out=None):
        """Collective ``accuflux(data, direction="up")`` (reference pyflwdir/streams.py:15-41): every rank passes the
        payload of its block INCLUDING its halo rows (``by_row``: one value per device row); returns (the rank's own
        rows, rounds).  Bit-identical to the whole raster, floats included: see :func:`accuflux_blocks`.  ``out``: a
        DEVICE buffer over the block's device rows (own + halo) that receives the result and is returned instead."""
        data = np.asarray(data_block)
        if data.dtype not in _hip._PAYLOAD_CODE:
            raise NotImplementedError(f"payload dtype {data.dtype} is not supported by the row-block accuflux")
        dirc = _hip.PFD_UP if direction == "up" else _hip.PFD_DOWN
        return self._up_collective(lambda: _UpBlock(self.handle, "accuflux", data.dtype, payload=data, by_row=by_row,
                                                    nodata=nodata_args, direction=dirc, out=out), data.dtype, max_iter)

    def stream_distance(self, mask_block=None, step_lengths_block=None, max_iter=MAX_ROUNDS):
        """Collective ``stream_distance`` (reference pyflwdir/streams.py:272-315): int32 cell counts, or float32 metres
        with ``step_lengths_block`` = the rows of the whole raster's step-length table that belong to the block's device
        rows ([2 * rows - 1, 3], see :func:`stream_distance_blocks`).  Returns (distances of the own rows, rounds)."""
        dtype = np.int32 if step_lengths_block is None else np.float32
        return self._up_collective(lambda: _UpBlock(self.handle, "distance", dtype, payload=step_lengths_block,
                                                    mask=mask_block), dtype, max_iter)

    def stream_order(self, mask_block=None, max_iter=MAX_ROUNDS, out=None):
        """Collective Strahler order (reference pyflwdir/streams.py:228-269); ``mask_block`` covers the block's device
        rows.  Returns (uint8 orders of the rank's own rows, rounds); ``out`` as in :meth:`accuflux`."""
        return self._up_collective(lambda: _UpBlock(self.handle, "strahler", np.uint8, mask=mask_block, out=out), np.uint8,
                                   max_iter)

    def close(self):
        if self.handle is not None:
            self.handle.close()
        if self.comm is not None:
            self.comm.close()
