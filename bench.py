#!/usr/bin/env python3
"""bench.py — headline benchmark: Mcells/s of FlwdirRaster.upstream_area("cell") on a synthetic
D8 raster (BASELINE.json configs[1]: 10000 x 10000, 1 MI355X), with the HBM-roofline fraction of
the dominant kernel and the single-thread CPU baseline (the oracle restatement of the
reference's serial algorithm) timed on the same host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--no-cpu-baseline]

A "step" is one complete pass of the hot path on one raster: device-resident uint8 D8 codes in
-> device-resident int32 upstream cell counts out, INCLUDING the decode/normalisation of the
raster and the construction of whatever ordering structure the kernels need (a fresh raster
handle is created every step; nothing is cached between steps).  Inputs are generated in HBM
by the device twin of the oracle's synthetic generator and are resident before the timed
region starts; the result stays in HBM.

N > 1 (launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`):
the raster is row-tiled over the N GPUs (weak scaling: every rank owns `size` rows); see
DESIGN.md §Multi-GPU.  torch.distributed is used only for rendezvous/barrier/max-reduce.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pyflwdir_amd import _hip  # noqa: E402

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per cell of upstream_area("cell"): 29 B/cell in total,
# split over the phases of the level engine as documented in DESIGN.md §Roofline
B_ALG_TOTAL = 29.0
B_ALG_PHASE = {"order_cells": 8.0, "init": 4.0, "sweep_count_up": 17.0,
               "tile_local": 8.0, "exit_graph": 4.0, "tile_final": 17.0}
# segment -> the kernel it times (names as rocprofv3 prints them); single-launch segments only
KERNEL_OF = {"tile_local": "void k_tile<false, true>(TileArgs)", "tile_final": "void k_tile<true, false>(TileArgs)",
             "order_cells": "k_bfs_level(...)", "sweep_count_up": "void k_sweep<CountUp>(...)"}
# HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3
# --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of this same command, 10000 x 10000):
# bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024, raw counter values (calibration in DESIGN.md)
PMC_TRAFFIC = {("void k_tile<true, false>(TileArgs)", 10000): (147178.969 + 390625.000) * 1024,  # profiles/r01g_*
               ("void k_tile<false, true>(TileArgs)", 10000): (176827.453 + 221867.906) * 1024}


def roofline_of(segs, n, size, ms_per_step):
    """Roofline object for the dominant KERNEL (the tile pass that takes longest; multi-launch
    segments such as the exit graph are reported in phases_ms but are not one kernel)."""
    cand = [s for s in segs if s["name"] in KERNEL_OF] or segs
    dom = max(cand, key=lambda s: s["ms"])
    b_alg = B_ALG_PHASE.get(dom["name"], B_ALG_TOTAL)
    launches = max(1, dom["launches"])
    avg_ms = dom["ms"] / launches
    achieved = (b_alg * n / launches) / (avg_ms * 1e-3) / 1e9
    kname = KERNEL_OF.get(dom["name"], dom["name"])
    traffic = PMC_TRAFFIC.get((kname, size))
    whole = B_ALG_TOTAL * n / (ms_per_step * 1e-3) / 1e9
    return dict(bound="hbm", achieved=round(achieved, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                frac=round(achieved / PEAK_HBM_GBS, 5), traffic=traffic, kernel=kname, launches=dom["launches"],
                avg_launch_ms=round(avg_ms, 5), alg_bytes_per_cell=b_alg,
                whole_pass=dict(alg_bytes_per_cell=B_ALG_TOTAL, achieved=round(whole, 2),
                                frac=round(whole / PEAK_HBM_GBS, 5)),
                phases_ms={s["name"]: round(s["ms"], 3) for s in segs})
SYNTH = dict(seed=0, tilt=1 << 26, white=2, nodata_pct=0)  # "river" regime: max_rank = nrow-1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=10000, help="raster is size x size per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the raster given to the CPU baseline (0=auto)")
    return ap.parse_args()


def one_step(d8_buf, out_buf, nrow, ncol, device, profile=False):
    """One pass of the hot path.  With profile=True the library brackets every phase with HIP
    events on its own stream (6 events per pass) and the phase times are returned."""
    h = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE,
                          deferred=not os.environ.get("PFD_BENCH_EAGER"))  # (eager create: A/B knob)
    if profile:
        h.set_profiling(True)
    h.upstream_area_cell(out=out_buf, memspace=_hip.PFD_DEVICE)
    res = (h.last_timing(), h.info()) if profile else None
    h.close()
    return res


def mean_segments(all_segs):
    """Average the per-phase HIP-event times over the timed steps."""
    acc = {}
    for segs in all_segs:
        for s in segs:
            a = acc.setdefault(s["name"], dict(name=s["name"], ms=0.0, launches=s["launches"]))
            a["ms"] += s["ms"] / len(all_segs)
    return list(acc.values())


def cpu_baseline(d8_host, rows):
    """Single-thread oracle (restatement of the reference's serial pipeline) on a bounded sample:
    the first `rows` rows of the same raster (a self-contained raster: flow is southwards, the
    cut edge simply becomes an outlet row)."""
    from oracle import oracle as O

    sample = np.ascontiguousarray(d8_host[:rows])
    t0 = time.perf_counter()
    upa, tim, st = O.upstream_area_cell(sample)
    dt = time.perf_counter() - t0
    return dict(value=round(sample.size / dt / 1e6, 3), unit="Mcells/s", cores=1, kind="port",
                sample=f"first {rows} of {d8_host.shape[0]} rows x {d8_host.shape[1]} cols of the same raster "
                       f"({sample.size / 1e6:.0f} Mcells, {dt:.1f} s; decode {tim['decode_s']:.2f} s, idxs_seq "
                       f"{tim['idxs_seq_s']:.2f} s, accuflux {tim['accuflux_s']:.2f} s)",
                host_cpus=os.cpu_count()), upa


def run_distributed(a, rank, world, local):
    """N > 1: one rank per GPU, weak scaling — every rank owns a size x size row block of the
    (N*size) x size raster (+ one halo row per inner edge), generated directly in its HBM."""
    import torch
    import torch.distributed as dist

    from pyflwdir_amd import dist as pdist

    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)  # (only missing for the single-process PFD_BENCH_FORCE_DIST run)
    dist.init_process_group(backend="gloo")  # rendezvous / barrier / max-reduce only (CPU, 128-byte id)
    device = local % max(1, _hip.device_count())  # (one rank per GPU; the modulo only matters on test boxes)
    ncol = a.size
    nrow_total = a.size * world
    r0, r1 = rank * a.size, (rank + 1) * a.size
    top, bot = pdist.halo_of(rank, world)
    d8_buf = _hip.synth_d8_device(nrow_total, ncol, row0=r0 - top, nrows=(r1 - r0) + top + bot, device=device, **SYNTH)
    out_buf = _hip.DeviceBuffer(a.size * ncol * 4, device)
    # RCCL communicator (all-gather over xGMI); if it cannot be brought up on every rank the same
    # protocol runs with the records travelling through torch.distributed (transport named in the output)
    probe = pdist.DistributedRaster(d8_buf, a.size, ncol, rank, world, device, memspace=_hip.PFD_DEVICE,
                                    transport=os.environ.get("PFD_DIST_TRANSPORT", "auto"))
    comm, transport = probe.comm, probe.transport
    probe.handle.close()

    def step(profile=False):
        # a fresh handle per step, like the single-GPU bench: decode + local solve + exchange + final pass
        h = _hip.RasterHandle(d8_buf, a.size, ncol, device=device, memspace=_hip.PFD_DEVICE, halo=(top, bot),
                              deferred=True)
        if profile:
            h.set_profiling(True)
        if comm is not None:
            comm.upstream_area_cell(h, out=out_buf, memspace=_hip.PFD_DEVICE)
        else:
            probe.handle = h
            probe.upstream_area(out=out_buf, memspace=_hip.PFD_DEVICE)
        res = (h.last_timing(), h.info()) if profile else None
        h.close()
        return res

    for _ in range(a.warmup):
        step()
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    dist.barrier()
    t0 = time.perf_counter()
    timed = [step(profile=True) for _ in range(a.steps)]
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt[0])
    segs, info = mean_segments([t[0] for t in timed]), timed[-1][1]
    # cross-rank invariant: the cells draining off the last row of the raster carry every cell
    stats = torch.tensor([info["n_valid"]], dtype=torch.int64)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    if rank == 0:
        n = nrow_total * ncol
        ms_per_step = dt / a.steps * 1e3
        roof = roofline_of(segs, n // world, a.size, ms_per_step)
        roof["per_gpu"] = True
        out = dict(metric="Mcells/s upstream_area on D8 raster", value=round(n * a.steps / dt / 1e6, 2), unit="Mcells/s",
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_per_step, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="int32", data="synthetic",
                   config=dict(workload=f"{nrow_total}x{ncol} synthetic D8 (river regime, seed 0) row-tiled over {world} "
                                        f"GPUs ({a.size} rows each + halo), upstream_area(unit='cell') int32, "
                                        "decode+local solve+RCCL all-gather+final pass per step",
                               n_valid=int(stats[0]), parallelism=f"{world} row blocks, 1 all-gather/pass", transport=transport),
                   roofline=roof)
        print(json.dumps(out))
    dist.barrier()
    if comm is not None:
        comm.close()
    dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if world > 1 or os.environ.get("PFD_BENCH_FORCE_DIST"):  # the env knob runs the RCCL path with 1 rank
        return run_distributed(a, rank, world, local)
    device = local
    nrow = ncol = a.size
    n = nrow * ncol
    d8_buf = _hip.synth_d8_device(nrow, ncol, device=device, **SYNTH)
    out_buf = _hip.DeviceBuffer(n * 4, device)

    for _ in range(a.warmup):
        one_step(d8_buf, out_buf, nrow, ncol, device)
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    t0 = time.perf_counter()
    timed = [one_step(d8_buf, out_buf, nrow, ncol, device, profile=True) for _ in range(a.steps)]
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    dt = time.perf_counter() - t0
    ms_per_step = dt / a.steps * 1e3
    value = n * a.steps / dt / 1e6
    # kernel durations: HIP events recorded live, inside the timed region, on the library's stream
    segs, info = mean_segments([t[0] for t in timed]), timed[-1][1]
    roofline = roofline_of(segs, n, a.size, ms_per_step)

    out = dict(metric="Mcells/s upstream_area on D8 raster", value=round(value, 2), unit="Mcells/s", n_gpus=1,
               steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="int32", data="synthetic",
               config=dict(workload=f"{nrow}x{ncol} synthetic D8 (river regime, seed 0), "
                                    "upstream_area(unit='cell') int32, decode+order+sweep per step",
                           n_valid=info["n_valid"], n_pits=info["n_pits"], n_levels=info["n_levels"],
                           parallelism="1 GPU"),
               roofline=roofline)

    # size-independent invariant (reference tests/test_streams_basins.py:24-27): the upstream areas of
    # the pits add up to the number of valid cells.  In the river regime every pit sits on the last row.
    last_codes = d8_buf.download(np.uint8, (ncol,), offset_bytes=(nrow - 1) * ncol)
    last_upa = out_buf.download(np.int32, (ncol,), offset_bytes=(nrow - 1) * ncol * 4)
    pits = last_codes == 0
    if int(pits.sum()) == info["n_pits"]:
        out["invariant_pit_sum_equals_n_valid"] = bool(int(last_upa[pits].astype(np.int64).sum()) == info["n_valid"])

    if not a.no_cpu_baseline:
        d8_host = d8_buf.download(np.uint8, (nrow, ncol))
        rows = a.cpu_rows or min(nrow, max(1, int(1.2e8 // ncol)))
        cpu, upa_cpu = cpu_baseline(d8_host, rows)
        out["cpu_baseline"] = cpu
        # parity spot check of the benchmarked result against the oracle on the sample's interior:
        # rows whose whole upstream area lies inside the sample are identical in both rasters
        got = out_buf.download(np.int32, (rows, ncol))
        if rows == nrow:
            out["parity_vs_oracle"] = bool(np.array_equal(got, upa_cpu))
        else:
            out["parity_vs_oracle"] = bool(np.array_equal(got, upa_cpu))  # flow is southwards: upstream = rows above
    print(json.dumps(out))


if __name__ == "__main__":
    main()
