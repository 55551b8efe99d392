"""The world > 1 RCCL branches of pyflwdir_amd/csrc/dist.hip on the ONE GPU of the test box.

Real RCCL refuses two ranks on one device, so `pfd_upstream_area_cell_dist` (ncclAllGather with world > 1 feeding the
interface solve, the agreement ncclAllReduce), `pfd_comm_exchange_rows` (ncclGroupStart / ncclSend / ncclRecv /
ncclGroupEnd with a neighbour, k_seed_update, ncclAllReduce of the counts) and `pfd_comm_allgather_host` had never run
anywhere.  Here 2, 4 and 8 ranks preload tests/rccl_loopback/librccl_loopback.so — a test-only stand-in for the thirteen
RCCL entry points the library binds (bytes through a shared mapping; see its header) — and run the SAME library code
with PFD_DIST_TRANSPORT=rccl: every collective against the oracle on the whole raster, and the 90000 x 90000 headline
raster in 8 row blocks with the result checksum of the single-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "rccl_loopback")
SHIM = os.path.join(SHIM_DIR, "librccl_loopback.so")


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-C", SHIM_DIR], stdout=subprocess.DEVNULL)
    return SHIM


def _free_port():
    """A port nobody listens on right now (bench.py's own helper: two suites on one box cannot collide on a fixed number)."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench

    return bench.free_port()


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PFD_DIST_TRANSPORT")}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_every_collective_through_the_rccl_branches(gpu_lib, shim, world):
    """tools/dist_check.py: upstream_area (pfd_upstream_area_cell_dist), basins (pfd_comm_allgather_host), hand / float
    accuflux / stream_distance / Strahler (pfd_comm_exchange_rows) of `world` row blocks, all against the oracle on the
    whole raster; no boundary row may travel through the host group."""
    procs = []
    port = _free_port()
    for r in range(world):
        e = dict(_clean_env(), RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 HSA_ENABLE_IPC_MODE_LEGACY="0", PFD_DIST_TRANSPORT="rccl", LD_PRELOAD=shim, PFD_LOOPBACK_TIMEOUT_S="180")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    for o in outs:
        assert "ok (rccl)" in o[0] and "rccl_sendrecv" in o[0] and "rccl_allgather" in o[0], o[0]


@pytest.mark.parametrize("knob", ["PFD_TEST_IFACE_HOPS", "PFD_TEST_ROUNDS4", "PFD_TEST_HCAP"])
def test_a_stage_that_falls_short_is_redone_by_every_rank(gpu_lib, shim, knob):
    """pfd_upstream_area_cell_dist runs a pass without host round trips; a stage that falls short (the interface chase out of
    hops, level 4 out of rounds, a hypertile's id range overflowed — forced by the test knobs) raises a sticky bit on the
    device, the verdict travels through the agreement all-reduce and EVERY rank repeats the pass with the remedy.  Three
    ranks, results against the oracle (tools/dist_check.py)."""
    world = 3
    procs = []
    port = _free_port()
    for r in range(world):
        e = dict(_clean_env(), RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 HSA_ENABLE_IPC_MODE_LEGACY="0", PFD_DIST_TRANSPORT="rccl", LD_PRELOAD=shim, PFD_LOOPBACK_TIMEOUT_S="180",
                 PFD_ENABLE_KNOBS="1", DIST_CHECK_ONLY="upstream_area", DIST_CHECK_SHAPE="6400x4300")  # (3 x 3 hypertiles per block)
        e[knob] = "100" if knob == "PFD_TEST_HCAP" else "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert all("ok (rccl)" in o[0] for o in outs)


def _line(args, env, timeout=1500):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2500:]
    js = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(js) == 1
    return json.loads(js[0])


@pytest.mark.parametrize("size,world", [(6000, 2), (90000, 8)])
def test_bench_ranks_share_the_gpu_over_the_rccl_branches(gpu_lib, shim, size, world):
    """`bench.py --gpus N --rccl-loopback` with no launcher: N self-spawned ranks on the one GPU, boundary records in ONE
    ncclAllGather per pass (stand-in bound, and the line says so); the checksum of the N blocks equals the N = 1 run's.
    (90000, 8) is BASELINE configs[3] split exactly as the driver's 8-GPU run splits it."""
    env = _clean_env()
    common = ["--size", str(size), "--steps", "2", "--warmup", "1"]
    one = _line(common + ["--no-cpu-baseline", "--no-secondary"], env)
    many = _line(common + ["--gpus", str(world), "--rccl-loopback"], env)
    cfg = many["config"]
    assert many["n_gpus"] == world and cfg["transport"] == "rccl" and cfg["rccl_world_size"] == world
    assert cfg["rccl_binding"].startswith("loopback stand-in")
    assert cfg["n_valid"] == one["config"]["n_valid"] and cfg["n_pits"] == one["config"]["n_pits"]
    assert many["invariants"]["result_checksum"] == one["invariants"]["result_checksum"]
    assert many["invariants"]["result_checksum_equals_n1"] is True and many["speedup_vs_n1"] > 0
    assert many["invariants"]["last_row_pit_sum_equals_n_valid"] is True


@pytest.mark.parametrize("op", ["hand", "accuflux"])
def test_bench_op_over_the_neighbour_exchange(gpu_lib, shim, op):
    """configs[4]'s collectives with 4 ranks on the one GPU: the boundary rows travel in ncclSend / ncclRecv groups
    between neighbours (stand-in bound), the checksum equals the one-rank run's."""
    env = _clean_env()
    args = ["--op", op, "--rows", "2600", "--cols", "3100", "--steps", "2", "--warmup", "1"]
    one = _line(["--gpus", "1"] + args, env)
    four = _line(["--gpus", "4", "--rccl-loopback"] + args, env)
    assert four["n_gpus"] == 4 and four["config"]["transport"] == "rccl"
    assert four["config"]["rccl_binding"].startswith("loopback stand-in")
    assert all(k.startswith("rccl") for k in four["config"]["exchange_kinds"])
    assert four["invariants"]["result_checksum"] == one["invariants"]["result_checksum"]
    assert four["config"]["iterations"] >= 2
