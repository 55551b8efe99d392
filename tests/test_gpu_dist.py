"""The one-process-per-GPU entry points on the single GPU of the test box:
 * pfd_comm_create + pfd_upstream_area_cell_dist with world = 1 (RCCL communicator of one rank);
 * bench.py launched as 2 ranks by torch.distributed.run and by plain processes over the torch-free TCP
   group, both ranks on the one GPU: RCCL refuses two ranks on one device, so the records travel through
   the host group — everything else (row blocks with halos, deferred handles, split-phase C-ABI, interface
   solve, final pass, agreement) is the N > 1 path.  The result checksum must equal the single-GPU run's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_world_1(gpu_lib, oracle):
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(700, 900, seed=11, tilt=1 << 26, white=2, nodata_pct=5)
    exp = oracle.upstream_area_cell(d8)[0]
    comm = _hip.Communicator(_hip.Communicator.unique_id(), 0, 1, 0)
    for deferred in (False, True):
        h = _hip.RasterHandle(d8, 700, 900, deferred=deferred)
        got = comm.upstream_area_cell(h)
        assert np.array_equal(got.reshape(700, 900), exp)
        h.close()
    comm.close()


def _bench(args, env_extra, launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29651",
                 PFD_BENCH_GROUP="tcp")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    return json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])


def test_bench_spawns_its_own_ranks(gpu_lib):
    """`python bench.py --gpus 2` with NO launcher — the way the driver calls it: bench.py starts its two ranks itself
    (here both on the one GPU, so the records travel through the host group), prints one JSON line, exits 0."""
    args = ["--size", "6000", "--steps", "2", "--warmup", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + args,
                         capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["steps"] == 2 and two["warmup"] == 1
    cfg = two["config"]
    assert cfg["launcher"] == "self-spawned" and cfg["host_group"] == "HostGroup"
    from pyflwdir_amd import _hip

    if _hip.device_count() >= 2:
        assert cfg["transport"] == "rccl" and cfg["rccl_world_size"] == 2
    else:
        assert cfg["transport"] == "host" and cfg["rccl_world_size"] is None
    assert two["invariants"]["result_checksum"] == ref["invariants"]["result_checksum"]
    assert two["invariants"]["result_checksum_equals_n1"] is True and two["speedup_vs_n1"] > 0
    assert two["invariants"]["last_row_pit_sum_equals_n_valid"] is True


@pytest.mark.parametrize("launcher", ["torchrun", "tcp"])
def test_two_ranks_on_one_gpu(gpu_lib, launcher):
    args = ["--size", "6000", "--steps", "2", "--warmup", "1"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + args,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    two = _bench(args, dict(PFD_DIST_TRANSPORT="host", **({"PFD_BENCH_GROUP": "torch"} if launcher == "torchrun" else {})),
                 launcher)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["transport"] == "host"
    assert two["config"]["n_valid"] == ref["config"]["n_valid"] == 36_000_000
    assert two["config"]["n_pits"] == ref["config"]["n_pits"]
    assert two["invariants"]["last_row_pit_sum_equals_n_valid"] is True
    assert two["invariants"]["result_checksum"] == ref["invariants"]["result_checksum"]
    assert ref["invariants"]["all_cells_upa_equals_1_plus_children"] and ref["invariants"]["pit_sum_equals_n_valid"]


@pytest.mark.parametrize("world", [2, 3])
def test_collective_upstream_area_and_basins_over_tcp(gpu_lib, world):
    """DistributedRaster.upstream_area / .basins of every rank's row block against the oracle on the whole raster;
    plain processes + the torch-free TCP group (tools/dist_check.py)."""
    procs = []
    for r in range(world):
        e = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + world),
                 HSA_ENABLE_IPC_MODE_LEGACY="0", PFD_DIST_TRANSPORT="host")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert all("ok (host)" in o[0] for o in outs)


def test_collectives_over_rccl_with_one_rank(gpu_lib):
    """The RCCL transport of every sharded collective (neighbour ncclSend/ncclRecv in one group, ncclAllReduce of the
    unknown / changed counts, ncclAllGather of the basins records) with a communicator of ONE rank — all the test box
    can host: device-resident halo seeds, no boundary row through the host group, results against the oracle."""
    e = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29733",
             HSA_ENABLE_IPC_MODE_LEGACY="0", PFD_DIST_TRANSPORT="rccl")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")], env=e, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ok (rccl)" in out.stdout and "rccl_sendrecv" in out.stdout and "rccl_allgather" in out.stdout


@pytest.mark.parametrize("op", ["hand", "basins", "accuflux", "strahler"])
def test_bench_op_lines_one_and_two_ranks(gpu_lib, op):
    """`bench.py --op hand|basins` (BASELINE configs[4], here on a small tile): one rank over RCCL (world 1: device-resident
    halo seeds, ncclAllReduce of the counts), then two self-spawned ranks (on the one GPU of the test box: host transport);
    the result checksum must not depend on the number of row blocks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    args = ["--op", op, "--rows", "2600", "--cols", "3100", "--steps", "2", "--warmup", "1"]
    lines = []
    for gpus in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + args, capture_output=True,
                             text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-2500:]
        js = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1
        lines.append(json.loads(js[0]))
    one, two = lines
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and one["metric"].startswith("Mcells/s " + op)
    assert one["config"]["transport"] == "rccl" and all(k.startswith("rccl") for k in one["config"]["exchange_kinds"])
    assert one["invariants"]["result_checksum"] == two["invariants"]["result_checksum"]
    assert two["invariants"]["result_checksum_equals_n1"] is True and two["speedup_vs_n1"] > 0
    if op == "hand":
        assert two["config"]["iterations"] >= 2 and two["config"]["exchanges_per_step"] == two["config"]["iterations"]
    if op in ("accuflux", "strahler"):  # (the seeded up-sweeps: at least one exchange that changed a halo value, then one that did not)
        assert two["config"]["iterations"] >= 1 and two["config"]["exchanges_per_step"] >= 2


def test_device_resident_hand_inputs_get_the_finite_check(gpu_lib, oracle):
    """DistributedRaster.hand with DEVICE buffers: a NaN / inf elevation is refused with the message of the host path
    (it would imitate the "-inf = not known yet" marker of the row-block protocol) — pfd_count_nonfinite; ADVICE r04."""
    from pyflwdir_amd import _hip
    from pyflwdir_amd import dist as pdist
    from pyflwdir_amd.hostgroup import HostGroup

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29777"
    shape = (300, 260)
    d8 = oracle.synth_d8(shape[0], shape[1], seed=3, tilt=100000, white=2, nodata_pct=5)
    elev = oracle.synth_elev_f32(shape[0], shape[1], seed=3, tilt=100000, white=2, nodata_pct=5)
    grp = HostGroup(0, 1)
    dr = pdist.DistributedRaster(d8, shape[0], shape[1], 0, 1, 0, transport="host", group=grp)
    try:
        drain = _hip.DeviceBuffer(d8.size).upload(np.zeros(shape, np.uint8))
        good = _hip.DeviceBuffer(elev.nbytes).upload(elev)
        res, it = dr.hand(drain, good, elev_code=_hip.PFD_F32)
        assert it >= 1
        res.free()
        elev[100, 100] = np.inf
        bad = _hip.DeviceBuffer(elev.nbytes).upload(elev)
        with pytest.raises(NotImplementedError, match="finite elevations"):
            dr.hand(drain, bad, elev_code=_hip.PFD_F32)
        for b in (drain, good, bad):
            b.free()
    finally:
        dr.close()
        grp.close()
