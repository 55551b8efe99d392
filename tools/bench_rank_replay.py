"""What ONE rank of an N-GPU `upstream_area` job spends per pass — through the REAL collective entry point
(`pfd_upstream_area_cell_dist`: phase A, ncclAllGather, interface solve, phase B, agreement ncclAllReduce, one
synchronisation) — measured on a one-GPU box with the rank ALONE on the GPU and its peers answering at once.

Two stages, both with the test-only RCCL stand-in preloaded (tests/rccl_loopback, see its header):

  1. record:  N ranks share the GPU and run one pass; rank 0 writes the result of every collective to a directory
              (PFD_LOOPBACK_RECORD).  The result checksum is compared with `--checksum` when given.
  2. replay:  one process per rank, one after the other, each alone on the GPU, creates "rank k of N" without peers
              (PFD_LOOPBACK_REPLAY): a collective is ONE device-to-device copy of the recorded result on the caller's
              stream = an ideal transport.  Each times `--steps` passes on fresh deferred handles (the step of bench.py).

The serial estimate of the N-GPU speed-up is (N = 1 step) / (slowest rank's pass); what real hardware adds is the all-gather
of N x 4 x ncol words over xGMI and the rank skew.

    python tools/bench_rank_replay.py --size 90000 --gpus 8 [--steps 20] [--n1-ms 38.0]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHIM = os.path.join(ROOT, "tests", "rccl_loopback", "librccl_loopback.so")


def rank_main(a):
    import numpy as np

    from pyflwdir_amd import _hip
    from pyflwdir_amd import dist as pdist

    rank, world = a.rank, a.gpus
    nrow = ncol = a.size
    if a.stage == "replay" and a.reserve_gib > 0:
        # one arena for the working buffers, like bench.py's ranks (pfd_reserve): the 13.7 / 6.0 ms steps of
        # profiles/r05f_rank_replay.txt were hipMalloc calls of a process that had none
        _hip.reserve(int(a.reserve_gib * 2**30))
    r0, r1 = pdist.block_rows(nrow, world)[rank]
    own = r1 - r0
    top, bot = pdist.halo_of(rank, world)
    d8 = _hip.synth_d8_device(nrow, ncol, seed=0, row0=r0 - top, nrows=own + top + bot)
    out = _hip.DeviceBuffer(own * ncol * 4)
    if a.stage == "record":
        from pyflwdir_amd.hostgroup import HostGroup

        grp = HostGroup(rank, world)
        uid = grp.bcast(_hip.Communicator.unique_id() if rank == 0 else b"", 0)
    else:
        grp, uid = None, _hip.Communicator.unique_id()
    comm = _hip.Communicator(uid, rank, world, 0)
    sync = lambda: _hip.check(_hip.lib().pfd_device_synchronize(0))  # noqa: E731

    def step(profile=False):
        h = _hip.RasterHandle(d8, own, ncol, device=0, memspace=_hip.PFD_DEVICE, halo=(top, bot), deferred=True)
        if profile:
            h.set_profiling(True)
        comm.upstream_area_cell(h, out=out, memspace=_hip.PFD_DEVICE)
        segs = h.last_timing() if profile else None
        h.close()
        return segs

    if a.stage == "record":
        step()
        csum = grp.allreduce(int(_hip.checksum_i32(out, own * ncol)), "sum")
        if rank == 0:
            print(json.dumps(dict(stage="record", checksum=csum)))
        grp.barrier()
        comm.close()
        grp.close()
        return
    for _ in range(a.warmup):
        step()
    ts = []
    m0 = _hip.alloc_stats()["hipmalloc_calls"]
    for _ in range(a.steps):
        sync()
        t0 = time.perf_counter()
        step()
        sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    mallocs = _hip.alloc_stats()["hipmalloc_calls"] - m0
    segs = step(profile=True)
    comm.close()
    ts = np.array(ts)
    slow = [(int(i), round(float(ts[i]), 3)) for i in np.argsort(ts)[-3:][::-1]]
    print(json.dumps(dict(stage="replay", rank=rank, ms_median=round(float(np.median(ts)), 3), ms_min=round(float(ts.min()), 3),
                          ms_max=round(float(ts.max()), 3), ms_mean=round(float(ts.mean()), 3), ms_p99=round(float(np.percentile(ts, 99)), 3),
                          slowest_steps=slow, hipmalloc_calls_timed=int(mallocs), reserved_GiB=a.reserve_gib,
                          segments={s["name"]: round(s["ms"], 3) for s in segs}, checksum=int(_hip.checksum_i32(out, own * ncol)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=90000)
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n1-ms", type=float, default=0.0, help="ms per step of the N = 1 run on this box (bench.py), for the estimate")
    ap.add_argument("--checksum", type=int, default=None, help="result checksum of the N = 1 run")
    ap.add_argument("--reserve-gib", type=float, default=24.0, help="arena of a replayed rank (0: none, the round-5 state)")
    ap.add_argument("--stage", default=None)
    ap.add_argument("--rank", type=int, default=0)
    a = ap.parse_args()
    if a.stage:
        return rank_main(a)
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-C", os.path.dirname(SHIM)], stdout=subprocess.DEVNULL)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PFD_DIST_TRANSPORT")}
    base.update(LD_PRELOAD=SHIM, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29911")
    common = [sys.executable, os.path.abspath(__file__), "--size", str(a.size), "--gpus", str(a.gpus), "--steps", str(a.steps),
              "--warmup", str(a.warmup), "--reserve-gib", str(a.reserve_gib)]
    with tempfile.TemporaryDirectory(prefix="pfd_replay_") as d:
        procs = [subprocess.Popen(common + ["--stage", "record", "--rank", str(r)],
                                  env=dict(base, RANK=str(r), WORLD_SIZE=str(a.gpus), PFD_LOOPBACK_RECORD=d),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(a.gpus)]
        outs = [p.communicate(timeout=1200) for p in procs]
        if any(p.returncode for p in procs):
            raise SystemExit("record stage failed:\n" + "\n".join(o[1][-800:] for o in outs))
        rec = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
        nfiles = len(os.listdir(d))
        print(f"recorded {nfiles} collectives of one pass of {a.gpus} ranks; checksum {rec['checksum']}"
              + ("" if a.checksum is None else f" (N = 1: {a.checksum}, equal: {rec['checksum'] == a.checksum})"))
        rows = []
        for r in range(a.gpus):
            # (the set-up agreement is the first recorded call: the passes loop over the calls behind it)
            out = subprocess.run(common + ["--stage", "replay", "--rank", str(r)],
                                 env=dict(base, PFD_LOOPBACK_REPLAY=d, PFD_LOOPBACK_REPLAY_LOOP=str(nfiles - 2)),
                                 capture_output=True, text=True, timeout=1200)
            if out.returncode:
                raise SystemExit(f"replay of rank {r} failed:\n{out.stderr[-1500:]}")
            rows.append(json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]))
            print(f"rank {r} of {a.gpus} alone on the GPU, ideal transport: median {rows[-1]['ms_median']:.3f} ms "
                  f"(min {rows[-1]['ms_min']:.3f}, p99 {rows[-1]['ms_p99']:.3f}, max {rows[-1]['ms_max']:.3f}; slowest steps "
                  f"{rows[-1]['slowest_steps']}; hipMalloc calls while timing {rows[-1]['hipmalloc_calls_timed']}) per pass; "
                  f"segments {rows[-1]['segments']}")
    worst = max(r["ms_median"] for r in rows)
    summary = dict(size=a.size, ranks=a.gpus, steps=a.steps, reserved_GiB_per_rank=a.reserve_gib, slowest_rank_ms=worst, mean_rank_ms=round(sum(r["ms_median"] for r in rows) / len(rows), 3),
                   max_over_median=round(max(r["ms_max"] / r["ms_median"] for r in rows), 2),
                   checksum=sum(r["checksum"] for r in rows), checksum_equals_record=sum(r["checksum"] for r in rows) == rec["checksum"])
    if a.n1_ms:
        summary.update(n1_ms=a.n1_ms, serial_estimate_speedup=round(a.n1_ms / worst, 2))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
