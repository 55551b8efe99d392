"""Host side of the multi-GPU path: a raster row-tiled over the GPUs of one node, one process
per GPU, one small RCCL all-gather per pass (kernels and protocol: csrc/dist.hip).

The reference has no distributed code at all; this module only adds what a
driver script needs: the row partition, halo handling, and the rendezvous of the RCCL unique id
through an existing ``torch.distributed`` process group (any backend — ``gloo`` on CPU works,
nothing but 128 bytes travels through it).
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


def upstream_area_blocks(d8: np.ndarray, nblocks: int, devices=None) -> np.ndarray:
    """``upstream_area("cell")`` of a host raster computed as ``nblocks`` row blocks held by this one
    process (on one or several GPUs).  Same kernels and protocol as the RCCL path."""
    d8 = np.ascontiguousarray(d8, dtype=np.uint8)
    nrow, ncol = d8.shape
    devices = devices or [0] * nblocks
    handles = []
    for b, (r0, r1) in enumerate(block_rows(nrow, nblocks)):
        a, e = block_slice(nrow, nblocks, b)
        handles.append(_hip.RasterHandle(d8[a:e], r1 - r0, ncol, device=devices[b], halo=halo_of(b, nblocks)))
    outs = _hip.upstream_area_cell_blocks(handles)
    for h in handles:
        h.close()
    return np.concatenate([o.reshape(-1, ncol) for o in outs], axis=0)


def exchange_unique_id(rank: int, world: int, group=None) -> bytes:
    """Rank 0 creates the RCCL unique id, everybody receives it through torch.distributed."""
    import torch
    import torch.distributed as dist

    t = torch.zeros(_hip.Communicator.UID_BYTES, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(_hip.Communicator.unique_id()), dtype=torch.uint8).clone()
    if world > 1:
        dist.broadcast(t, src=0, group=group)
    return bytes(t.numpy().tobytes())


class DistributedRaster:
    """The row block of this rank plus the communicator; ``upstream_area()`` is collective."""

    def __init__(self, d8_block, own_rows: int, ncol: int, rank: int, world: int, device: int, uid: bytes,
                 memspace=_hip.PFD_HOST):
        self.rank, self.world, self.device = rank, world, device
        self.handle = _hip.RasterHandle(d8_block, own_rows, ncol, device=device, memspace=memspace,
                                        halo=halo_of(rank, world))
        self.comm = _hip.Communicator(uid, rank, world, device)

    def upstream_area(self, out=None, memspace=_hip.PFD_HOST):
        res = self.comm.upstream_area_cell(self.handle, out=out, memspace=memspace)
        return res.reshape(self.handle.nrow, self.handle.ncol) if memspace == _hip.PFD_HOST else res

    def close(self):
        self.handle.close()
        self.comm.close()
