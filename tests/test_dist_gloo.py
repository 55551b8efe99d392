"""N > 1 path on CPU: two real processes over the gloo backend.

What can run without a GPU is the host side of the multi-GPU path and the PROTOCOL itself:
 * row partition / halo bookkeeping (pyflwdir_amd.dist),
 * the rendezvous of the 128-byte RCCL unique id through the host group interface (here the
   torch.distributed adapter; the torch-free TCP group is tested in tests/test_hostgroup.py),
 * the block protocol of csrc/dist.hip restated in numpy — local solve with the halo rows as
   weightless sinks, all-gather of the boundary records {L_top, L_bottom, sink_first, sink_last},
   redundant interface-forest solve, final local solve with the boundary inflow — with the
   oracle as the per-block solver and dist.all_gather as the transport.  Its result must equal
   the oracle on the whole raster.  (The same protocol with the HIP kernels is tested on the GPU
   box in tests/test_gpu_blocks.py.)
 * the seeded up-sweep protocol (float accuflux over row blocks): the real collective driver
   DistributedRaster._up_collective — agreement, seed routing, termination at the fixpoint — around a numpy block
   sweep that adds the halo cells draining into a boundary cell in the serial loop's position; bit-identical
   float32 sums against the oracle on the whole raster.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition():
    from pyflwdir_amd import dist

    assert dist.block_rows(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert dist.block_rows(8, 8) == [(i, i + 1) for i in range(8)]
    with pytest.raises(ValueError):
        dist.block_rows(3, 4)
    assert [dist.halo_of(b, 4) for b in range(4)] == [(0, 1), (1, 1), (1, 1), (1, 0)]
    assert dist.halo_of(0, 1) == (0, 0)
    assert [dist.block_slice(10, 3, b) for b in range(3)] == [(0, 5), (3, 8), (6, 10)]
    rows = dist.block_rows(90000, 8)
    assert rows[0] == (0, 11250) and rows[-1] == (78750, 90000)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from pyflwdir_amd import _hip
    from pyflwdir_amd import dist as pdist

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # 1) unique-id rendezvous (the id itself comes from RCCL on a GPU box; here a stand-in)
        _hip.Communicator.unique_id = staticmethod(lambda: bytes(range(128)))
        from pyflwdir_amd.hostgroup import TorchGroup

        grp = TorchGroup()
        uid = pdist.exchange_unique_id(rank, world, grp)
        assert uid == bytes(range(128))
        assert grp.allreduce(rank + 1, "min") == 1 and grp.allreduce(float(rank), "max") == world - 1

        # 2) the block protocol, numpy + oracle + gloo all_gather
        for shape, seed, kw in [((97, 120), 21, dict(tilt=1 << 26, white=2, nodata_pct=0)),
                                ((64, 75), 22, dict(tilt=100000, white=2, nodata_pct=30))]:
            d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
            nrow, ncol = d8.shape
            r0, r1 = pdist.block_rows(nrow, world)[rank]
            top, bot = pdist.halo_of(rank, world)
            a, e = pdist.block_slice(nrow, world, rank)
            blk = d8[a:e].copy()
            own = np.zeros(blk.shape, bool)
            own[top:top + (r1 - r0)] = True
            # halo rows: weightless sinks (valid cells become pits), like k_normalise does
            halo_valid = (~own) & (blk != 247)
            blk_local = blk.copy()
            blk_local[halo_valid] = 0
            idxs_ds, idxs_pit, _ = O.from_array(blk_local)
            seq = O.idxs_seq(idxs_ds, idxs_pit)
            w0 = (own & (blk != 247)).astype(np.int64).ravel()
            acc0 = O.accuflux(idxs_ds, seq, w0, nodata=-1).reshape(blk.shape)
            # terminal cell of every cell's path (pointer jumping)
            end = np.where(idxs_ds >= 0, idxs_ds, np.arange(idxs_ds.size)).astype(np.int64)
            for _ in range(20):
                end = end[end]
            NONE = 0xFFFFFFFF
            rec = np.zeros(4 * ncol, np.int64)
            rows_h = ([0] if top else [None]) + ([blk.shape[0] - 1] if bot else [None])
            for side, hr in enumerate(rows_h):
                if hr is not None:
                    rec[side * ncol:(side + 1) * ncol] = np.where(halo_valid[hr], acc0[hr], 0)
            first, last = top, top + (r1 - r0) - 1
            for side, br in enumerate((first, last)):
                for c in range(ncol):
                    v = NONE
                    if blk[br, c] != 247:
                        t = end[br * ncol + c]
                        tr, tc = divmod(int(t), ncol)
                        if halo_valid[tr, tc]:
                            v = (1 << 31) | ((1 << 30) if tr > last else 0) | tc
                    rec[(2 + side) * ncol + c] = v
            gathered = [torch.zeros(4 * ncol, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(rec))
            allrec = np.stack([g.numpy() for g in gathered])
            # interface forest, solved redundantly: F(h) = L(h) + sum F(h'') with next(h'') == h
            nn = world * 2 * ncol
            L = np.zeros(nn, np.int64)
            nxt = np.full(nn, -1, np.int64)
            for b in range(world):
                for side in range(2):
                    for c in range(ncol):
                        nid = (b * 2 + side) * ncol + c
                        L[nid] = allrec[b, side * ncol + c]
                        owner = b + (1 if side else -1)
                        if 0 <= owner < world:
                            s = int(allrec[owner, (2 + (1 - side)) * ncol + c])
                            if s != NONE:
                                nxt[nid] = (owner * 2 + (1 if s & (1 << 30) else 0)) * ncol + (s & 0x3FFFFFFF)
            F = L.copy()
            order = []  # topological order by repeated relaxation (tiny graph)
            indeg = np.zeros(nn, np.int64)
            for i in range(nn):
                if nxt[i] >= 0:
                    indeg[nxt[i]] += 1
            stack = [i for i in range(nn) if indeg[i] == 0]
            while stack:
                i = stack.pop()
                j = nxt[i]
                if j >= 0:
                    F[j] += F[i]
                    indeg[j] -= 1
                    if indeg[j] == 0:
                        stack.append(j)
            w1 = w0.reshape(blk.shape).copy()
            if rank > 0:
                w1[first] += np.where(blk[first] != 247, F[((rank - 1) * 2 + 1) * ncol:((rank - 1) * 2 + 2) * ncol], 0)
            if rank + 1 < world:
                w1[last] += np.where(blk[last] != 247, F[((rank + 1) * 2 + 0) * ncol:((rank + 1) * 2 + 1) * ncol], 0)
            acc1 = O.accuflux(idxs_ds, seq, w1.ravel(), nodata=-1).reshape(blk.shape)[top:top + (r1 - r0)]
            res = np.where(blk[top:top + (r1 - r0)] == 247, -9999, acc1).astype(np.int32)
            exp = O.upstream_area_cell(d8)[0][r0:r1]
            assert np.array_equal(res, exp), f"rank {rank}: block protocol differs from the oracle"
        # 2b) seeded up-sweeps: the collective driver of pyflwdir_amd/dist.py around a numpy block sweep
        import types

        DR = {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0), 128: (-1, 1)}

        class NumpyUpBlock:
            """float32 accuflux of one row block; its halo cells hold the seeds (same interface as dist._UpBlock)."""

            def __init__(self, blk, top, nown, data):
                self.blk, self.top, self.nown, self.data = blk, top, nown, data
                self.nr, self.nc = blk.shape
                self.out = data.copy()
                self.swept_with = self.brows = None
                own = lambda r: top <= r < top + nown
                self.kids = {}   # own cell -> upstream cells (own cells and halo cells), descending linear index
                self.ds = {}
                for r in range(self.nr):
                    for c in range(self.nc):
                        code = int(blk[r, c])
                        if code not in DR:
                            continue
                        rr, cc = r + DR[code][0], c + DR[code][1]
                        if 0 <= rr < self.nr and 0 <= cc < self.nc and own(rr) and blk[rr, cc] != 247:
                            self.kids.setdefault((rr, cc), []).append((r, c))
                            if own(r):
                                self.ds[(r, c)] = (rr, cc)
                for k in self.kids.values():
                    k.sort(reverse=True)
                # own cells, upstream cells first (Kahn over the links between own cells)
                indeg = {}
                for x, d in self.ds.items():
                    indeg[d] = indeg.get(d, 0) + 1
                cells = [(r, c) for r in range(top, top + nown) for c in range(self.nc) if blk[r, c] != 247]
                stack = [x for x in cells if indeg.get(x, 0) == 0]
                self.order = []
                while stack:
                    x = stack.pop()
                    self.order.append(x)
                    d = self.ds.get(x)
                    if d is not None:
                        indeg[d] -= 1
                        if indeg[d] == 0:
                            stack.append(d)
                assert len(self.order) == len(cells)

            def sweep(self, seed):
                bits = seed.view(np.uint8)
                if self.swept_with is not None and np.array_equal(self.swept_with, bits):
                    return False
                self.swept_with = bits.copy()
                out = self.data.copy()
                if self.top:
                    out[self.top - 1] = seed[:self.nc]
                if self.top + self.nown < self.nr:
                    out[self.top + self.nown] = seed[self.nc:]
                for x in self.order:
                    acc = self.data[x]
                    for k in self.kids.get(x, ()):
                        if acc != np.float32(-9999) and out[k] != np.float32(-9999):
                            acc = np.float32(acc + out[k])
                    out[x] = acc
                self.out = out
                self.brows = np.stack([out[self.top], out[self.top + self.nown - 1]])
                return True

            def result(self):
                return self.out[self.top:self.top + self.nown]

            def close(self, close_handle=True):
                pass

        for shape, seed, kw in [((64, 75), 31, dict(tilt=100000, white=2, nodata_pct=10)),
                                ((40, 90), 32, dict(tilt=3000, white=2, nodata_pct=0))]:
            d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
            idxs_ds, idxs_pit, _ = O.from_array(d8)
            data = (np.random.default_rng(seed).random(shape) * 3).astype(np.float32)
            data[5, 7] = -9999
            exp = O.accuflux(idxs_ds, O.idxs_seq(idxs_ds, idxs_pit), data.ravel(), nodata=-9999).reshape(shape)
            r0, r1 = pdist.block_rows(shape[0], world)[rank]
            a, e = pdist.block_slice(shape[0], world, rank)
            dr = object.__new__(pdist.DistributedRaster)
            dr.group, dr.rank, dr.world, dr.handle = grp, rank, world, types.SimpleNamespace(ncol=shape[1])
            dr.comm, dr.exchanges = None, []  # (host transport: the boundary rows travel through the group)
            got, rounds = dr._up_collective(lambda: NumpyUpBlock(d8[a:e], pdist.halo_of(rank, world)[0], r1 - r0, data[a:e]),
                                            np.float32)
            assert rounds >= 1 and np.array_equal(got.view(np.uint32), exp[r0:r1].view(np.uint32)), f"rank {rank}: seeded up-sweep"
        # 3) bench-style timing reduction: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == float(world)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_process_gloo(tmp_path):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    for r, p in enumerate(procs):
        assert p.exitcode == 0, f"rank {r} failed (exit code {p.exitcode})"
        assert (tmp_path / f"ok{r}").exists()


def test_up_block_asks_for_updates_from_the_second_sweep_on():
    """Host logic of the incremental block up-sweeps (no GPU): the first sweep of a block keeps its sweep
    (pfd_set_block_update mode 1), every later one asks for an update (mode 2), verification does not touch the mode,
    a block whose handle lives on releases the kept sweep when it is closed, and the down-sweeps never ask."""
    from pyflwdir_amd import _hip, dist

    calls = []

    class Handle:
        nrow, ncol, device, halo = 6, 5, 0, (1, 1)

        def set_block_update(self, mode):
            calls.append(("mode", mode))

        def accuflux_block(self, *a, **k):
            calls.append(("sweep", k.get("verify", False)))
            return np.zeros((2, 5), np.float32), 0

        def close(self):
            calls.append(("close",))

    def block(direction):
        b = dist._UpBlock.__new__(dist._UpBlock)
        b.h, b.kind, b.dtype, b.by_row, b.nodata, b.direction = Handle(), "accuflux", np.dtype(np.float32), True, (0, 0.0, 0), direction
        b.payload, b.mask, b.out, b.out_given = np.zeros(8, np.float32), None, None, True
        b.swept_with, b.brows, b.sweeps = None, None, 0
        b.incremental = direction == _hip.PFD_UP
        return b

    b = block(_hip.PFD_UP)
    seed = np.zeros(10, np.float32)
    assert b.sweep(seed) and not b.sweep(seed)  # (same seeds: nothing can have changed, no call at all)
    seed[3] = 1.0
    assert b.sweep(seed)
    b.verify(seed)
    b.close(close_handle=False)
    assert calls == [("mode", 1), ("sweep", False), ("mode", 2), ("sweep", False), ("sweep", True), ("mode", 0)]
    del calls[:]
    d = block(_hip.PFD_DOWN)
    assert d.sweep(seed)
    seed[4] = 2.0
    assert d.sweep(seed)
    d.close(close_handle=True)
    assert calls == [("sweep", False), ("sweep", False), ("close",)]
