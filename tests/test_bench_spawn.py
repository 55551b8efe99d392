"""bench.py --gpus N without a launcher starts its own ranks (the way the driver calls it): the spawn logic on CPU,
with a stand-in rank script."""
import os
import subprocess
import sys
import textwrap
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_rank_environments():
    b = _bench()
    envs = b.rank_environments(4, 8, 29555, base_env={"PATH": "/usr/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29555" for e in envs)
    assert all(e["PFD_BENCH_GROUP"] == "tcp" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert all("PFD_DIST_TRANSPORT" not in e for e in envs)  # one GPU per rank: RCCL
    shared = b.rank_environments(2, 1, 29555, base_env={})
    assert all(e["PFD_DIST_TRANSPORT"] == "host" for e in shared)  # ranks sharing a GPU: host transport
    kept = b.rank_environments(2, 1, 29555, base_env={"PFD_DIST_TRANSPORT": "rccl", "PFD_BENCH_GROUP": "torch"})
    assert all(e["PFD_DIST_TRANSPORT"] == "rccl" and e["PFD_BENCH_GROUP"] == "torch" for e in kept)
    lb = b.rank_environments(2, 1, 29555, base_env={"LD_PRELOAD": "/x.so"}, loopback=True)  # --rccl-loopback rehearsal
    assert all(e["PFD_DIST_TRANSPORT"] == "rccl" and e["LD_PRELOAD"] == b.LOOPBACK_SO + ":/x.so" for e in lb)


def test_free_port_pair():
    b = _bench()
    p = b.free_port()
    assert 1024 < p < 65535 - 64


def _script(tmp_path, body):
    f = tmp_path / "rank.py"
    f.write_text(textwrap.dedent(body))
    return str(f)


def test_spawn_ranks_ok(tmp_path, capfd):
    """All ranks run with the launcher's variables and meet in the torch-free TCP group; only rank 0 owns stdout."""
    b = _bench()
    script = _script(tmp_path, f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        from pyflwdir_amd.hostgroup import HostGroup
        g = HostGroup()
        s = g.allreduce(int(os.environ["LOCAL_RANK"]) + 1, "sum")
        g.barrier()
        print("line from rank", g.rank, "sum", s, "world", g.world, sys.argv[1:])
        g.close()
        """)
    rc = b.spawn_ranks(types.SimpleNamespace(gpus=3), argv=["--gpus", "3"], n_devices=1, timeout=120, script=script)
    out, err = capfd.readouterr()
    assert rc == 0
    assert out.strip() == "line from rank 0 sum 6 world 3 ['--gpus', '3']"
    assert "line from rank 1" in err and "line from rank 2" in err


def test_spawn_ranks_failure_takes_the_others_down(tmp_path):
    b = _bench()
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)
        """)
    import time

    t0 = time.time()
    rc = b.spawn_ranks(types.SimpleNamespace(gpus=2), argv=[], n_devices=2, timeout=300, script=script)
    assert rc == 7 and time.time() - t0 < 60


def test_plain_call_does_not_ask_for_a_launcher():
    """`python bench.py --gpus 2` used to exit 1 with "launch multi-GPU runs with torch.distributed.run"; without a
    GPU it must now fail for the only honest reason (no device), from the spawn path."""
    sys.path.insert(0, ROOT)
    from pyflwdir_amd import _hip

    if _hip.device_count() > 0:
        import pytest

        pytest.skip("a device is visible: covered by tests/test_gpu_dist.py::test_bench_spawns_its_own_ranks")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode != 0 and "no HIP device" in out.stderr and "torch.distributed.run" not in out.stderr


def test_spawn_falls_back_to_the_host_transport(tmp_path, capfd, monkeypatch):
    """First contact with real multi-GPU hardware: if the run with the automatic transport (RCCL) fails, the same job
    runs once more with PFD_DIST_TRANSPORT=host, and only the successful attempt's line reaches stdout."""
    import functools

    b = _bench()
    monkeypatch.delenv("PFD_DIST_TRANSPORT", raising=False)
    script = _script(tmp_path, """
        import os, sys
        if os.environ["RANK"] == "0":
            print("line of rank 0 with transport", os.environ.get("PFD_DIST_TRANSPORT", "auto"), flush=True)
        if os.environ.get("PFD_DIST_TRANSPORT") != "host":
            sys.exit(3)  # (after rank 0 has printed: that line must not get out)
        """)
    spawn = functools.partial(b.spawn_ranks, argv=[], n_devices=2, script=script)
    rc = b.spawn_with_fallback(types.SimpleNamespace(gpus=2), spawn=spawn, first_timeout=60)
    out, err = capfd.readouterr()
    assert rc == 0 and out.strip() == "line of rank 0 with transport host"
    assert "retrying with the host transport" in err
    # an explicitly chosen transport is not second-guessed
    monkeypatch.setenv("PFD_DIST_TRANSPORT", "rccl")
    rc = b.spawn_with_fallback(types.SimpleNamespace(gpus=2), spawn=spawn, first_timeout=60)
    out, _ = capfd.readouterr()
    assert rc == 3 and out.strip() == ""
