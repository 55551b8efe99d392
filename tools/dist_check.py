"""One rank of a 2+-process check of the collective entry points over the torch-free TCP group:
DistributedRaster.upstream_area, .basins, .hand, .accuflux, .stream_distance and .stream_order of a row block against the oracle on the
whole raster.
Launched by tests/test_gpu_dist.py (all ranks on the one GPU of the test box, records through the host; with
WORLD_SIZE=1 and PFD_DIST_TRANSPORT=rccl the RCCL code path of every collective — ncclSend/ncclRecv group,
ncclAllReduce of the counts, ncclAllGather of the basins records — runs with a communicator of one rank).

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tools/dist_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402  (checker only)
from pyflwdir_amd import _hip  # noqa: E402
from pyflwdir_amd import dist as pdist  # noqa: E402
from pyflwdir_amd.hostgroup import HostGroup  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
grp = HostGroup(rank, world)
shape = tuple(int(x) for x in os.environ.get("DIST_CHECK_SHAPE", "1700x1300").split("x"))
d8 = O.synth_d8(shape[0], shape[1], seed=77, tilt=100000, white=2, nodata_pct=20)
upa = O.upstream_area_cell(d8)[0]
r0, r1 = pdist.block_rows(shape[0], world)[rank]
a, e = pdist.block_slice(shape[0], world, rank)
dr = pdist.DistributedRaster(d8[a:e], r1 - r0, shape[1], rank, world, 0, transport=os.environ.get("PFD_DIST_TRANSPORT", "host"),
                             group=grp)
got = dr.upstream_area()
assert np.array_equal(got, upa[r0:r1]), f"rank {rank}: upstream_area differs"
if os.environ.get("DIST_CHECK_ONLY") == "upstream_area":  # (the forced-miss runs of tests/test_gpu_rccl_loopback.py)
    got = dr.upstream_area()  # once more on the same handle: the remedies of the first pass must not linger wrongly
    assert np.array_equal(got, upa[r0:r1]), f"rank {rank}: second upstream_area differs"
    dr.close()
    grp.barrier()
    grp.close()
    print(f"rank {rank} of {world}: ok ({dr.transport}); upstream_area only")
    sys.exit(0)
idxs_ds, idxs_pit, _ = O.from_array(d8)
seq = O.idxs_seq(idxs_ds, idxs_pit)
outl = np.argsort(upa.ravel())[-400:]
ids = (np.arange(outl.size) + 3).astype(np.uint32)
lab = dr.basins(outl, ids, shape[0])
exp = O.basins(idxs_ds, outl.astype(idxs_ds.dtype), seq, ids).reshape(shape)
assert np.array_equal(lab, exp[r0:r1]), f"rank {rank}: basins differ"
# collective HAND: bit-identical to the oracle on the whole raster (the block passes its rows incl. halo rows)
elev = O.synth_elev_f32(shape[0], shape[1], seed=77, tilt=100000, white=2, nodata_pct=20)
drain = upa > np.percentile(upa[upa > 0], 97)
exp_h = O.height_above_nearest_drain(idxs_ds, seq, drain.ravel(), elev.ravel()).reshape(shape)
got_h, iters = dr.hand(drain[a:e], elev[a:e])
assert np.array_equal(got_h.view(np.uint64), exp_h[r0:r1].view(np.uint64)), f"rank {rank}: hand differs"
# collective float32 accuflux and Strahler order (seeded up-sweeps)
data = (np.random.default_rng(7).random(shape) * 1.7).astype(np.float32)
exp_a = O.accuflux(idxs_ds, seq, data.ravel(), nodata=-9999).reshape(shape)
got_a, rounds = dr.accuflux(data[a:e], (-9999, -9999.0, 1))
assert np.array_equal(got_a.view(np.uint32), exp_a[r0:r1].view(np.uint32)), f"rank {rank}: accuflux differs"
got_l, _ = dr.stream_distance()
assert np.array_equal(got_l, O.stream_distance(idxs_ds, seq, shape[1], real_length=False).reshape(shape)[r0:r1]), f"rank {rank}: stream distance differs"
got_s, _ = dr.stream_order()
assert np.array_equal(got_s, O.strahler_order(idxs_ds, seq).reshape(shape)[r0:r1]), f"rank {rank}: stream order differs"
kinds = sorted({k for k, _ in dr.exchanges})
if dr.transport == "rccl":  # no boundary row may have travelled through the host group
    assert kinds and all(k.startswith("rccl") for k in kinds), kinds
dr.close()
grp.barrier()
grp.close()
print(f"rank {rank} of {world}: ok ({dr.transport}); {len(dr.exchanges)} boundary exchanges: {kinds}; hand {iters} iterations, accuflux {rounds} rounds")
