#!/usr/bin/env python3
"""bench.py — headline benchmark: Mcells/s of FlwdirRaster.upstream_area("cell") on the synthetic
90000 x 90000 D8 raster of BASELINE.json configs[3] (8.1 Gcells; it fits one 288 GB MI355X), with the
HBM-roofline fraction of the dominant kernel and of the whole pass, the graph statistics SURVEY.md §8d
asks for, and the single-thread CPU baseline (the oracle restatement of the reference's serial
algorithm) timed on the same host on a bounded sample of the same raster.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--regime river|rough|meander]
                    [--no-cpu-baseline] [--no-secondary]

A "step" is one complete pass of the hot path over the raster: device-resident uint8 D8 codes in ->
device-resident int32 upstream cell counts out, INCLUDING the decode / pit rule / validation of the
raster and every structure the kernels need (a fresh raster handle per step; nothing is cached
between steps).  Inputs are generated in HBM by the device twin of the oracle's synthetic generator
before the timed region; the result stays in HBM.

N > 1: STRONG scaling — the same size x size raster is split into N row blocks (size/N rows each + one
halo row per inner edge), one block per rank/GPU, one RCCL all-gather of the boundary records per pass
(DESIGN.md §4.4).  `python bench.py --gpus N` starts its N ranks itself (plain processes, LOCAL_RANK = GPU
index); under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the
launcher's RANK / WORLD_SIZE / MASTER_* are used.  Either way the host side (barrier, max-reduce of the wall
time, rendezvous of the RCCL unique id) is the library's torch-free TCP group; no PyTorch is imported.

With N = 1 the JSON line also carries `secondary`: the 10000 x 10000 pass (configs[1]) and the
configs[2] operations (float32 accuflux + Strahler order at 30000 x 30000), each with its own
roofline object.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pyflwdir_amd import _hip  # noqa: E402

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per cell; the split of the 29 B of upstream_area("cell") over the
# phases that replace build / init / sweep is documented in DESIGN.md §5
B_ALG = {"upstream_area_cell": 29.0, "accuflux_f32": 33.0, "accuflux_f64": 45.0, "strahler": 18.0, "basins_u32": 18.0, "hand_f32": 35.0}
B_ALG_PHASE = {"tile_local": 8.0, "exit_graph": 4.0, "tile_final": 17.0}
# segment -> the kernel it times (names as rocprofv3 prints them); single-launch segments only
KERNEL_OF = {"tile_local": "void k_tile_local_fast<true, false>(TileArgs)", "tile_final": "void k_tile_final_fast<false, 256>(TileArgs)"}
B_FLOOR = 5.0  # absolute lower bound of upstream_area("cell"): 1 B/cell of codes in + 4 B/cell of counts out
# synthetic regimes (oracle/pfd_oracle.c orc_synth_d8 and its device twin): tilt >> noise gives long
# parallel rivers (max rank ~ nrow), small tilt a rough surface with many pits and meandering paths
REGIMES = {"river": dict(seed=0, tilt=1 << 26, white=2, nodata_pct=0),
           "rough": dict(seed=0, tilt=100000, white=2, nodata_pct=0),
           "meander": dict(seed=0, tilt=3000, white=2, nodata_pct=0)}


def _pmc_table():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def measured_traffic(kernel, nrow, ncol):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of this command at THIS raster size; tools/prof_pmc.sh
    writes the table).  None when the size was not measured: never a number from another size.
    kernel "_whole_pass": all kernels of one timed step together; "_op:<tag>": one warm call of an operation."""
    e = _pmc_table().get(f"{kernel}|{nrow}x{ncol}")
    return None if e is None else float(e["bytes_per_launch"])


def roofline_upa(segs, n, nrow, ncol, ms_per_step, regime="river"):
    """Roofline object of the tiled pass.  `frac` is the WHOLE-PASS fraction: SURVEY 8d's 29 algorithmic bytes per cell x
    cells / wall time of a step / 8 TB/s — a measurement, below 1 by construction.  The dominant kernel (the tile pass
    that takes longest; HIP events on the handle's stream inside the timed region) is named with the share of the model
    its phase replaces (`frac_model_share`: bookkeeping, the phases are not separable workloads) and, like the pass, with
    the bytes the PMC counters saw at THIS size (`frac_measured`, profiles/pmc_traffic.json) and the 5 B/cell no
    implementation can avoid (`frac_floor`)."""
    cand = [s for s in segs if s["name"] in KERNEL_OF] or segs
    dom = max(cand, key=lambda s: s["ms"])
    b_alg = B_ALG_PHASE.get(dom["name"], B_ALG["upstream_area_cell"])
    launches = max(1, dom["launches"])
    avg_ms = dom["ms"] / launches
    share = (b_alg * n / launches) / (avg_ms * 1e-3) / 1e9
    kname = KERNEL_OF.get(dom["name"], dom["name"])
    whole = B_ALG["upstream_area_cell"] * n / (ms_per_step * 1e-3) / 1e9
    river = regime == "river"  # (the PMC passes ran on the river raster)
    traffic = measured_traffic(kname, nrow, ncol) if river else None
    traffic_pass = measured_traffic("_whole_pass", nrow, ncol) if river else None
    floor = B_FLOOR * n / (ms_per_step * 1e-3) / 1e9
    return dict(bound="hbm", achieved=round(whole, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(whole / PEAK_HBM_GBS, 5),
                alg_bytes_per_cell=B_ALG["upstream_area_cell"], scope="whole pass (every kernel of a step)",
                traffic=traffic_pass,
                measured_bytes_per_cell=None if traffic_pass is None else round(traffic_pass / n, 3),
                frac_measured=None if traffic_pass is None else
                round(traffic_pass / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                floor_bytes_per_cell=B_FLOOR, frac_floor=round(floor / PEAK_HBM_GBS, 5),
                dominant_kernel=dict(kernel=kname, launches=dom["launches"], avg_launch_ms=round(avg_ms, 5),
                                     alg_bytes_per_cell_share=b_alg, frac_model_share=round(share / PEAK_HBM_GBS, 5),
                                     traffic=traffic,
                                     frac_measured=None if traffic is None else
                                     round(traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)),
                phases_ms={s["name"]: round(s["ms"], 3) for s in segs})


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=90000, help="the raster is size x size (split over the GPUs)")
    ap.add_argument("--regime", default="river", choices=sorted(REGIMES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the raster given to the CPU baseline (0=auto)")
    ap.add_argument("--op", choices=["upstream_area", "hand", "basins", "accuflux", "strahler"], default="upstream_area",
                    help="hand / basins: BASELINE configs[4] — the sharded collective on the 36000 x 72000 tile (rows x cols; "
                         "--rows / --cols override), one row block per GPU, boundary rows over RCCL send/recv; accuflux "
                         "(float32 cell areas per row = upstream_area in area units) / strahler: the seeded up-sweeps on the "
                         "same tile (incremental from the second exchange on)")
    ap.add_argument("--rows", type=int, default=36000)
    ap.add_argument("--cols", type=int, default=72000)
    ap.add_argument("--ops", choices=["c3", "c5"], default=None,
                    help="only the operation lines of configs[2] (30000^2) / configs[4] (36000x72000): what the PMC passes run")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="REHEARSAL on a box with fewer GPUs than ranks: the self-spawned ranks preload the test-only stand-in "
                         "for RCCL (tests/rccl_loopback) so that the world > 1 RCCL branches of the library run with ranks "
                         "sharing a GPU; the line names the binding (config.rccl_binding = 'loopback stand-in'); no timing "
                         "claim can rest on it")
    return ap.parse_args()


LOOPBACK_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "rccl_loopback", "librccl_loopback.so")


def rccl_binding():
    """What the library's ncclXxx calls are bound to in THIS process: 'rccl', or 'loopback stand-in' when the test-only
    interposer of tests/rccl_loopback is preloaded (it exports a marker symbol)."""
    import ctypes

    try:
        ctypes.CDLL(None).pfd_rccl_loopback_active
        return "loopback stand-in (tests/rccl_loopback, ranks share a GPU)"
    except (AttributeError, OSError):
        return "rccl"


def reserve_working_memory(device, gib):
    """One arena for the library's working buffers before anything is timed (pfd_reserve): hipMalloc of multi-GiB blocks
    returns in 0.2 ms or in seconds on this hardware, unpredictably — a benchmark (or a serving process) pays that once,
    up front.  A failure is not fatal: the class cache and hipMalloc remain."""
    if gib <= 0:
        return
    try:
        _hip.reserve(int(gib * 2**30), device)
    except Exception as exc:  # noqa: BLE001
        print(f"bench.py: pfd_reserve({gib} GiB) failed ({exc}); running without an arena", file=sys.stderr)


def rank_reserve_gib(whole_raster_gib, world):
    """Arena of one rank of a `world`-rank job: its share of what the whole raster needs (+50 %: per-rank fixed costs),
    divided again when several ranks share a GPU (test boxes); PFD_BENCH_RESERVE_GIB overrides the whole-raster figure."""
    whole = float(os.environ.get("PFD_BENCH_RESERVE_GIB", whole_raster_gib))
    share = whole if world == 1 else 1.5 * whole / world
    ranks_per_gpu = max(1, -(-world // max(1, _hip.device_count())))
    return min(share, 200.0 / ranks_per_gpu)


def one_step(d8_buf, out_buf, nrow, ncol, device, profile=False):
    """One pass of the hot path.  With profile=True the library brackets every phase with HIP
    events on its own stream (6 events per pass) and the phase times are returned."""
    h = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE, deferred=True)
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


def timed_steps(step, steps, warmup, device, after_warmup=None):
    """W untimed steps, then exactly K steps between two device synchronisations.  Every step ends
    with the library's own stream synchronisation, so the per-step wall times are exact too."""
    for _ in range(warmup):
        step(False)
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    if after_warmup is not None:
        after_warmup()
    marks = [time.perf_counter()]
    timed = []
    for _ in range(steps):
        timed.append(step(True))
        marks.append(time.perf_counter())
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    total = time.perf_counter() - marks[0]
    per = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
    return total, per, timed


def cpu_baseline(d8_host, rows, nrow):
    """Single-thread oracle (restatement of the reference's serial pipeline) on a bounded sample:
    the first `rows` rows of the same raster (a self-contained raster: the cut edge simply becomes an
    outlet row)."""
    from oracle import oracle as O

    sample = np.ascontiguousarray(d8_host[:rows])
    t0 = time.perf_counter()
    upa, tim, st = O.upstream_area_cell(sample)
    dt = time.perf_counter() - t0
    return dict(value=round(sample.size / dt / 1e6, 3), unit="Mcells/s", cores=1, kind="port",
                sample=f"first {rows} of {nrow} rows x {d8_host.shape[1]} cols of the same raster "
                       f"({sample.size / 1e6:.0f} Mcells, {dt:.1f} s; decode {tim['decode_s']:.2f} s, idxs_seq "
                       f"{tim['idxs_seq_s']:.2f} s, accuflux {tim['accuflux_s']:.2f} s)",
                host_cpus=os.cpu_count()), upa


def rhine_mosaic(min_edge=10000):
    """The reference's 682 x 997 Rhine sub-basin (tests/golden/rhine.npz, from /root/reference/tests/data) tiled
    to at least min_edge x min_edge cells.  Every copy gets a one-cell nodata frame: flow that left the fixture
    stays an outlet (the pit rule) instead of entering the neighbouring copy, so the mosaic is acyclic and keeps
    the real raster's path statistics."""
    d8 = np.load(os.path.join(ROOT, "tests", "golden", "rhine.npz"))["d8"]
    framed = np.full((d8.shape[0] + 2, d8.shape[1] + 2), 247, np.uint8)
    framed[1:-1, 1:-1] = d8
    reps = (-(-min_edge // framed.shape[0]), -(-min_edge // framed.shape[1]))
    return np.ascontiguousarray(np.tile(framed, reps))


def serpentine(nrow, ncol):
    """Worst case of the tile pass: every 64 x 64 tile is ONE boustrophedon path of 4096 cells ending in a pit
    (even rows flow east, odd rows west, the row ends flow south) — the longest in-tile path there is, so the
    pointer doubling needs all its 12-13 rounds in every tile."""
    t = np.empty((64, 64), np.uint8)
    t[0::2, :] = 1
    t[1::2, :] = 16
    t[0::2, 63] = 4
    t[1::2, 0] = 4
    t[63, 0] = 0
    return np.ascontiguousarray(np.tile(t, (-(-nrow // 64), -(-ncol // 64)))[:nrow, :ncol])


def filled_mosaic(min_edge=10000, base=2048, device=0):
    """The realistic variant SURVEY 8d asks for: the synthetic elevation (rough regime) of a base x base raster,
    depression-filled by the library's priority flood (fill_depressions, reference pyflwdir/dem.py:17-143) so that
    rivers run through the former pits to the raster edge, tiled like the Rhine mosaic (nodata frame per copy)."""
    from pyflwdir_amd import dem

    elev = _hip.synth_elev_device(base, base, device=device, **REGIMES["rough"])
    z = elev.download(np.float32, (base, base))
    elev.free()
    _, d8 = dem.fill_depressions(z)
    framed = np.full((base + 2, base + 2), 247, np.uint8)
    framed[1:-1, 1:-1] = d8
    reps = (-(-min_edge // framed.shape[0]), -(-min_edge // framed.shape[1]))
    return np.ascontiguousarray(np.tile(framed, reps))


def upa_line(nrow, ncol, regime, steps, warmup, device, cpu=True, cpu_rows=0, checks=True, invariant_checks=None):
    """The upstream_area("cell") pass on one GPU: returns the JSON fields of one bench line."""
    if regime == "rhine_mosaic_device":
        # the reference's Rhine raster (tests/golden/rhine.npz) tiled on the DEVICE: the base raster (0.7 MB) is uploaded,
        # pfd_synth_mosaic writes the nrow x ncol mosaic (one-cell nodata frame per copy) straight into HBM
        base = np.load(os.path.join(ROOT, "tests", "golden", "rhine.npz"))["d8"]
        d8_buf = _hip.synth_mosaic_device(base, nrow, ncol, device=device)
        synth = dict(seed=None, tilt=None)
        label = (f"{nrow}x{ncol} mosaic of the reference's Rhine sub-basin ({base.shape[0]}x{base.shape[1]} cells, nodata frame "
                 "per copy), built in HBM by pfd_synth_mosaic")
    elif regime in ("rhine_mosaic", "serpentine", "filled_mosaic"):
        host = (rhine_mosaic(min(nrow, ncol)) if regime == "rhine_mosaic" else
                filled_mosaic(min(nrow, ncol), device=device) if regime == "filled_mosaic" else serpentine(nrow, ncol))
        nrow, ncol = host.shape
        synth = dict(seed=None, tilt=None)
        d8_buf = _hip.DeviceBuffer(host.size, device)
        d8_buf.upload(host)
        label = (f"{nrow}x{ncol} mosaic of the reference's Rhine sub-basin (682x997 cells, nodata frame per copy)"
                 if regime == "rhine_mosaic" else
                 f"{nrow}x{ncol} mosaic of a depression-filled synthetic DEM (2048x2048, fill_depressions, nodata frame per copy)"
                 if regime == "filled_mosaic" else
                 f"{nrow}x{ncol} serpentine tiles (every 64x64 tile one 4096-cell path: worst case of the tile pass)")
    else:
        synth = REGIMES[regime]
        d8_buf = _hip.synth_d8_device(nrow, ncol, device=device, **synth)
        label = f"{nrow}x{ncol} synthetic D8 ({regime} regime, seed {synth['seed']}, tilt {synth['tilt']})"
    n = nrow * ncol
    out_buf = _hip.DeviceBuffer(n * 4, device)
    alloc0 = {}
    total, per, timed = timed_steps(lambda prof: one_step(d8_buf, out_buf, nrow, ncol, device, profile=prof), steps,
                                    warmup, device, after_warmup=lambda: alloc0.update(_hip.alloc_stats()))
    alloc1 = _hip.alloc_stats()
    ms_per_step = total / steps * 1e3
    segs, info = mean_segments([t[0] for t in timed]), timed[-1][1]
    out = dict(value=round(n * steps / total / 1e6, 2), ms_per_step=round(ms_per_step, 3),
               ms_per_step_median=round(statistics.median(per), 3), ms_per_step_min=round(min(per), 3),
               roofline=roofline_upa(segs, n, nrow, ncol, ms_per_step, regime))
    cfg = dict(workload=label + ", upstream_area(unit='cell') int32 on 1 GPU through the C-ABI (pfd_raster_create_deferred + "
                                "pfd_upstream_area_cell: what FlwdirRaster.upstream_area calls, device-resident in and out); a step = "
                                "decode + pit rule + validation + tile pass + exit-graph solve + final tile pass on a fresh handle",
               n_cells=n, n_valid=info["n_valid"], n_pits=info["n_pits"], parallelism="1 GPU")
    # the allocator inside the timed region: working buffers come from the reserved arena / the class cache; a hipMalloc
    # there can stall for seconds on this hardware (DESIGN.md "allocation"), so the line says how many there were
    out["allocator"] = dict(hipmalloc_calls_timed=alloc1["hipmalloc_calls"] - alloc0.get("hipmalloc_calls", 0),
                            arena_blocks_timed=alloc1["arena_blocks"] - alloc0.get("arena_blocks", 0),
                            reserved_GiB=round(alloc1["reserved_bytes"] / 2**30, 1),
                            step_max_over_median=round(max(per) / statistics.median(per), 3))
    if checks:
        # graph statistics (outside the timed region): longest flow path, in-degree histogram, and the pointer-
        # doubling rounds the tile passes needed (counted by one extra profiled pass)
        h = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE, deferred=True)
        h.set_profiling(2)
        h.upstream_area_cell(out=out_buf, memspace=_hip.PFD_DEVICE)
        st = h.graph_stats()
        h.close()
        cfg.update(max_rank=st["max_rank"], indegree_hist=st["indegree_hist"], tile_doubling_rounds=st["tile_rounds"])
    if checks if invariant_checks is None else invariant_checks:
        # size-independent invariants (reference tests/test_streams_basins.py:24-27): the upstream areas of the
        # pits add up to the number of valid cells; nodata cells hold -9999; upa == 1 + sum over the upstream cells
        out["invariants"] = invariants(d8_buf, out_buf, nrow, ncol, info, device)
    if cpu:
        rows = cpu_rows or min(nrow, max(1, int(8e8 // ncol)))  # (~12 s of one host core: the 10 - 30 s the contract asks for)
        d8_host = d8_buf.download(np.uint8, (rows + 1 if rows < nrow else rows, ncol))
        try:
            base, upa_cpu = cpu_baseline(d8_host, rows, nrow)
        except Exception as exc:  # noqa: BLE001 - (the oracle library is built by __graft_entry__.build(): say so, do not die)
            base, upa_cpu = None, None
            out["cpu_baseline"] = dict(value=None, unit="Mcells/s", cores=1, kind="port", sample="",
                                       error=f"{type(exc).__name__}: {exc}"[:300])
        if base is not None:
            out["cpu_baseline"] = base
            # parity of the benchmarked result with the oracle on the sample: flow never runs northwards in the
            # synthetic regimes' first rows only if every upstream cell lies in the sample — compare the cells whose
            # upstream area the oracle could see completely (all of them when no cell of the row below drains up)
            got = out_buf.download(np.int32, (rows, ncol))
            if rows < nrow:
                below = d8_host[rows]
                closed = not np.isin(below, (32, 64, 128)).any()  # no NW / N / NE pointer into the sample
            else:
                closed = True
            out["parity_vs_oracle"] = bool(np.array_equal(got, upa_cpu)) if closed else None
            out["parity_rows"] = rows
    d8_buf.free()
    out_buf.free()
    _hip.check(_hip.lib().pfd_trim(device))
    return out, cfg


def invariants(d8_buf, out_buf, nrow, ncol, info, device, samples=1_000_000):
    """Full-size checks that need no oracle (SURVEY.md §8d, C4 checks ii/iii):
    * device: every cell's local equation upa == 1 + sum over the cells draining into it, -9999 exactly on
      nodata, the pits' sum == n_valid (reference tests/test_streams_basins.py:24-27) — one streaming kernel
      that shares nothing with the engines (pfd_verify_upstream_area_cell); on an acyclic raster the
      equations have one solution, the reference's result;
    * host (numpy, independent of the device decode): the same equation at ~`samples` random cells."""
    h = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE, deferred=True)
    v = h.verify_upstream_area_cell(out_buf, memspace=_hip.PFD_DEVICE)
    h.close()
    res = dict(all_cells_upa_equals_1_plus_children=bool(v["bad_cells"] == 0 and v["n_valid"] == info["n_valid"]),
               nodata_is_minus_9999=bool(v["bad_nodata"] == 0),
               pit_sum_equals_n_valid=bool(v["pit_sum"] == info["n_valid"] and v["n_pits"] == info["n_pits"]),
               result_checksum=v["checksum"])
    rng = np.random.default_rng(12345)
    rows = np.unique(rng.integers(0, nrow, size=64))
    per_row = max(1, samples // len(rows))
    DR = {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0), 128: (-1, 1)}
    ok, checked = True, 0
    for r in rows:
        r = int(r)
        r0, r1 = max(0, r - 1), min(nrow, r + 2)
        d = d8_buf.download(np.uint8, (r1 - r0, ncol), offset_bytes=r0 * ncol)
        u = out_buf.download(np.int32, (r1 - r0, ncol), offset_bytes=r0 * ncol * 4)
        k = r - r0
        cols = rng.integers(0, ncol, size=min(per_row, ncol))
        exp = np.ones(cols.size, np.int64)
        for code, (dr, dc) in DR.items():
            rr, cc = k - dr, cols - dc  # the neighbour that drains into (k, cols) if it holds `code`
            if rr < 0 or rr >= d.shape[0]:
                continue
            inside = (cc >= 0) & (cc < ncol)
            ccc = np.clip(cc, 0, ncol - 1)
            hit = inside & (d[rr, ccc] == code)
            exp += np.where(hit, u[rr, ccc].astype(np.int64), 0)
        valid = d[k, cols] != 247
        ok &= bool(np.all((u[k, cols].astype(np.int64) == exp)[valid])) and bool(np.all(u[k, cols][~valid] == -9999))
        checked += int(cols.size)
    res["host_sample_upa_equals_1_plus_children"] = ok
    res["host_sample_cells"] = checked
    return res


# ---- secondary lines: BASELINE.json configs[2] (float32 accuflux + Strahler at 30000 x 30000) and configs[4]
#      (basins from 1000 outlets + HAND at a 72000 x 36000-cell tile) ----
OP_TAG = {"accuflux(float32, direction='up')": "accuflux_f32_up", "stream_order(type='strahler')": "strahler",
          "hand(drain, elevtn float32) -> float64": "hand_f32", "basins(1000 outlets) -> uint32": "basins_u32"}


def pick_outlets(h, nrow, ncol, device, k=1000, seed=5):
    """k outlets with large upstream areas: the maximum of k sampled rows (distinct cells, many nested in one
    another's basins — BASELINE configs[4]: "basins() delineation from 1000 outlets")."""
    upa = _hip.DeviceBuffer(nrow * ncol * 4, device)
    h.upstream_area_cell(out=upa, memspace=_hip.PFD_DEVICE)
    rng = np.random.default_rng(seed)
    outl = []
    for r in np.unique(rng.integers(0, nrow, k)):
        row = upa.download(np.int32, (ncol,), offset_bytes=int(r) * ncol * 4)
        outl.append(int(r) * ncol + int(np.argmax(row)))
    upa.free()
    outl = np.array(outl[:k], np.int64)
    return outl, np.arange(1, outl.size + 1, dtype=np.uint32)


def op_lines(nrow, ncol, synth, label, steps, device, ops=("accuflux", "strahler", "hand", "basins")):
    """Wall time of complete warm calls of the order-sensitive operations (bit-identical to the reference's serial
    loops) and of basins(), everything device-resident; the first call on the handle, which also builds the sweep
    plan, is reported separately with its own roofline (the reference orders its cells once per object too,
    flwdir.py:231-250: SURVEY 8d's byte model charges that build to every call — the warm call does not pay it)."""
    n = nrow * ncol
    d8_buf = _hip.synth_d8_device(nrow, ncol, device=device, **synth)
    # (a throw-away handle first: its plan build pays the process's cold hipMallocs of the GB-sized plan and sort
    #  buffers — seconds at this size — which the caching allocator then keeps; `first_call_on_handle_ms` below is the
    #  first order-sensitive call on a FRESH handle in a warm process: plan build + sweep)
    h0 = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE)
    tmp = _hip.DeviceBuffer(n, device)
    h0.strahler(None, out=tmp, memspace=_hip.PFD_DEVICE)
    h0.close()
    tmp.free()
    h = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE)
    lines, bufs = [], [d8_buf]

    def run(name, fn, b_alg, dtype, first_builds_plan):
        t0 = time.perf_counter()
        fn()  # first call on the handle: builds the plan of the exact-order engine (once per raster) + allocations
        first_ms = (time.perf_counter() - t0) * 1e3
        per, segs = [], None
        h.set_profiling(True)
        for _ in range(steps):
            t1 = time.perf_counter()
            fn()
            per.append((time.perf_counter() - t1) * 1e3)
            segs = h.last_timing()
        h.set_profiling(False)
        ms = statistics.median(per)
        sweep = [s for s in segs if s["name"].startswith(("sweep", "chain", "exact"))]
        achieved = b_alg * n / (ms * 1e-3) / 1e9
        traffic = measured_traffic("_op:" + OP_TAG[name], nrow, ncol) if synth == REGIMES["river"] or nrow != ncol else None
        line = dict(op=name, workload=f"{label}, {name}; warm call through the C-ABI on a handle whose sweep plan exists "
                                      "(built once per raster by the first order-sensitive call)",
                    dtype=dtype, ms_per_call=round(ms, 3), ms_per_call_min=round(min(per), 3),
                    value=round(n / ms / 1e3, 2), unit="Mcells/s",
                    roofline=dict(bound="hbm", achieved=round(achieved, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                  frac=round(achieved / PEAK_HBM_GBS, 5), traffic=traffic,
                                  frac_measured=None if traffic is None else round(traffic / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                  alg_bytes_per_cell=b_alg,
                                  phases_ms={s["name"]: round(s["ms"], 3) for s in segs},
                                  launches={s["name"]: s["launches"] for s in sweep}))
        if first_builds_plan:  # the same bytes against the call that also decodes, orders and plans
            fa = b_alg * n / (first_ms * 1e-3) / 1e9
            line["first_call_on_handle_ms"] = round(first_ms, 2)
            line["roofline_first_call"] = dict(bound="hbm", achieved=round(fa, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                               frac=round(fa / PEAK_HBM_GBS, 5), alg_bytes_per_cell=b_alg,
                                               note="first order-sensitive call on a fresh handle: plan build + sweep")
        lines.append(line)

    first = True
    out_b = _hip.DeviceBuffer(n, device)
    bufs.append(out_b)
    if "accuflux" in ops:
        w = _hip.synth_weights_device(n, seed=1, device=device)
        out_f = _hip.DeviceBuffer(n * 4, device)
        bufs += [w, out_f]
        run("accuflux(float32, direction='up')",
            lambda: h.accuflux(w, _hip.PFD_F32, nodata_f=-9999.0, has_nodata=1, direction=_hip.PFD_UP, out=out_f,
                               memspace=_hip.PFD_DEVICE), B_ALG["accuflux_f32"], "f32", first)
        first = False
    if "strahler" in ops or "hand" in ops:
        run("stream_order(type='strahler')", lambda: h.strahler(None, out=out_b, memspace=_hip.PFD_DEVICE),
            B_ALG["strahler"], "u8", first)
        first = False
    if "hand" in ops:
        # height above the nearest drain, float32 elevation -> float64 (reference dem.height_above_nearest_drain,
        # pyflwdir/dem.py:299-330); drain = the cells of Strahler order 1
        elev = _hip.synth_elev_device(nrow, ncol, device=device, **synth)
        out_d = _hip.DeviceBuffer(n * 8, device)
        bufs += [elev, out_d]
        run("hand(drain, elevtn float32) -> float64", lambda: h.hand(out_b, elev, _hip.PFD_F32, out=out_d, memspace=_hip.PFD_DEVICE),
            B_ALG["hand_f32"], "f64", first)
        first = False
    if "basins" in ops:
        outl, ids = pick_outlets(h, nrow, ncol, device)
        lab = _hip.DeviceBuffer(n * 4, device)
        bufs.append(lab)
        run("basins(1000 outlets) -> uint32", lambda: h.basins(outl, ids, out=lab, memspace=_hip.PFD_DEVICE),
            B_ALG["basins_u32"], "u32", False)
        v = h.verify_basins(outl, ids, lab, memspace=_hip.PFD_DEVICE)  # every cell's local equation, on the device
        lines[-1]["invariants"] = dict(all_cells_label_equals_downstream_label=bool(v["bad_cells"] == 0 and v["bad_nodata"] == 0),
                                       labelled_cells=v["n_labelled"], outlets=int(outl.size))
    h.close()
    for b in bufs:
        b.free()
    _hip.check(_hip.lib().pfd_trim(device))
    return lines


def km2_line(nrow, ncol, synth, label, steps, device):
    """`FlwdirRaster.upstream_area(unit="km2")` on a lat/lon grid — the reference's documented call (pyflwdir.py:770-801,
    gis_utils.py:379-412): float64 cell areas, one per raster row (pyflwdir_amd/gis.py area_rows), accumulated in the serial
    loop's operand order by the exact-order engine (pfd_accuflux_rows, what raster.py calls), nodata cells -9999.  The
    FIRST call on a fresh handle (plan build + sweep: what a one-shot user pays) and the warm call, device-resident."""
    from pyflwdir_amd import gis
    from pyflwdir_amd._affine import get_affine

    n = nrow * ncol
    res = 1.0 / 1200.0  # 3 arc-seconds, centred on 50 N
    rows = np.ascontiguousarray(gis.area_rows(get_affine()(res, 0.0, 5.0, 0.0, -res, 50.0 + nrow * res / 2), (nrow, ncol), True, unit="m2")
                                / gis.AREA_FACTORS["km2"])
    assert rows.dtype == np.float64
    d8_buf = _hip.synth_d8_device(nrow, ncol, device=device, **synth)
    out = _hip.DeviceBuffer(n * 8, device)
    h = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE)

    def call():
        h.accuflux_rows(rows, _hip.PFD_F64, nodata_i=-9999, nodata_f=-9999.0, has_nodata=1, direction=_hip.PFD_UP, mask_invalid=1,
                        out=out, memspace=_hip.PFD_DEVICE)

    _hip.check(_hip.lib().pfd_device_synchronize(device))
    t0 = time.perf_counter()
    call()
    first_ms = (time.perf_counter() - t0) * 1e3
    per = []
    h.set_profiling(True)
    for _ in range(steps):
        t1 = time.perf_counter()
        call()
        per.append((time.perf_counter() - t1) * 1e3)
    segs = h.last_timing()
    ms = statistics.median(per)
    b_alg = B_ALG["accuflux_f64"]
    # the result against the cell-count pass: on every cell  area[km2] >= count * (smallest cell area) and
    # <= count * (largest) — a size-independent sanity bound, not a parity claim (parity: uparea_km2_latlon goldens)
    ach, fa = b_alg * n / (ms * 1e-3) / 1e9, b_alg * n / (first_ms * 1e-3) / 1e9
    line = dict(op="upstream_area_km2(lat/lon grid, float64)", workload=f"{label}, upstream_area(unit='km2') on a 3-arc-second lat/lon "
                "grid: float64 row areas through pfd_accuflux_rows (exact-order engine); first call on a fresh handle and warm call",
                dtype="f64", ms_per_call=round(ms, 3), ms_per_call_min=round(min(per), 3), value=round(n / ms / 1e3, 2), unit="Mcells/s",
                first_call_on_handle_ms=round(first_ms, 2),
                roofline=dict(bound="hbm", achieved=round(ach, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 5),
                              traffic=None, frac_measured=None, alg_bytes_per_cell=b_alg,
                              phases_ms={s["name"]: round(s["ms"], 3) for s in segs}),
                roofline_first_call=dict(bound="hbm", achieved=round(fa, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                         frac=round(fa / PEAK_HBM_GBS, 5), alg_bytes_per_cell=b_alg,
                                         note="first call on a fresh handle: plan build + sweep"))
    # the opt-in tolerance mode (raster.py upstream_area(unit, exact=False); csrc/wide.h): the same areas in 64-bit fixed
    # point on the tiled engine — order-free, no plan.  First call on a fresh DEFERRED handle (decode + validation inside),
    # warm call, and the largest relative distance to the exact result on the first / last 1500 rows (the main stems end
    # in the last rows: the largest sums of the raster)
    lines = [line]
    band = min(1500, nrow)
    ex_top, ex_bot = out.download(np.float64, (band, ncol)), out.download(np.float64, (band, ncol), (nrow - band) * ncol * 8)
    hf = _hip.RasterHandle(d8_buf, nrow, ncol, device=device, memspace=_hip.PFD_DEVICE, deferred=True)
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    t0 = time.perf_counter()
    got, quantum = hf.upstream_area_rows_fixed(rows, out=out, memspace=_hip.PFD_DEVICE)
    first_f = (time.perf_counter() - t0) * 1e3
    if got is not None:
        perf = []
        hf.set_profiling(True)
        for _ in range(steps):
            t1 = time.perf_counter()
            hf.upstream_area_rows_fixed(rows, out=out, memspace=_hip.PFD_DEVICE)
            perf.append((time.perf_counter() - t1) * 1e3)
        segs_f = hf.last_timing()
        msf = statistics.median(perf)
        rel = 0.0
        for ex, off in ((ex_top, 0), (ex_bot, (nrow - band) * ncol * 8)):
            g = out.download(np.float64, (band, ncol), off)
            v = ex != -9999.0
            assert np.array_equal(g[~v], ex[~v])
            rel = max(rel, float(np.max(np.abs(g[v] - ex[v]) / ex[v], initial=0.0)))
        achf, faf = b_alg * n / (msf * 1e-3) / 1e9, b_alg * n / (first_f * 1e-3) / 1e9
        lines.append(dict(op="upstream_area_km2(lat/lon grid, float64), exact=False", workload=f"{label}, the opt-in tolerance mode: row areas "
                          "in 64-bit fixed point accumulated as integers on the LDS-tiled engine (pfd_upstream_area_rows_fixed): "
                          "order-free, no plan; first call on a fresh deferred handle and warm call", dtype="u64 fixed point -> f64",
                          ms_per_call=round(msf, 3), ms_per_call_min=round(min(perf), 3), value=round(n / msf / 1e3, 2), unit="Mcells/s",
                          first_call_on_handle_ms=round(first_f, 2), quantum_km2=quantum,
                          max_rel_diff_to_exact_sampled=rel, within_1e_9=bool(rel <= 1e-9),
                          roofline=dict(bound="hbm", achieved=round(achf, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                        frac=round(achf / PEAK_HBM_GBS, 5), traffic=None, frac_measured=None,
                                        alg_bytes_per_cell=b_alg, phases_ms={s_["name"]: round(s_["ms"], 3) for s_ in segs_f}),
                          roofline_first_call=dict(bound="hbm", achieved=round(faf, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                                   frac=round(faf / PEAK_HBM_GBS, 5), alg_bytes_per_cell=b_alg,
                                                   note="first call on a fresh deferred handle: no plan to build")))
    hf.close()
    h.close()
    d8_buf.free()
    out.free()
    return lines


N1_RECORD = os.path.join(ROOT, ".bench_n1.json")  # (git-ignored scratch: lets the N > 1 lines quote their speed-up)


def api_lines(nrow, ncol, synth, device, reps=3):
    """What a drop-in numpy caller sees (SURVEY 8d "report H2D/D2H separately"): host uint8 raster ->
    ``from_array`` -> ``upstream_area()`` -> host int32 raster through ``FlwdirRaster``, wall clock, with the upload, the
    device work and the download told apart (``pfd_transfer_stats``: the library's own clocks around its staging copies;
    compute = wall - h2d - d2h).  The process's default arena is whatever the front end reserved (bench.py's own arena
    exists already at this point, so the first-call figure here is NOT a cold process: tests/test_gpu_frontend.py
    measures that)."""
    import pyflwdir_amd as pyflwdir

    n = nrow * ncol
    buf = _hip.synth_d8_device(nrow, ncol, device=device, **synth)
    d8 = buf.download(np.uint8, (nrow, ncol))
    buf.free()
    rows = []
    for _ in range(reps + 1):  # (the first repetition is reported apart: first pageable copies, fresh result pages)
        _hip.transfer_stats(reset=True)
        t0 = time.perf_counter()
        flw = pyflwdir.from_array(d8, ftype="d8")
        t1 = time.perf_counter()
        tr_a = _hip.transfer_stats(reset=True)
        upa = flw.upstream_area()
        t2 = time.perf_counter()
        tr_b = _hip.transfer_stats(reset=True)
        ok = bool(upa.dtype == np.int32 and upa.shape == (nrow, ncol))
        rows.append(dict(from_array_ms=(t1 - t0) * 1e3, upstream_area_ms=(t2 - t1) * 1e3, h2d_ms=tr_a["h2d_ms"] + tr_b["h2d_ms"],
                         h2d_bytes=tr_a["h2d_bytes"] + tr_b["h2d_bytes"], d2h_ms=tr_a["d2h_ms"] + tr_b["d2h_ms"],
                         d2h_bytes=tr_a["d2h_bytes"] + tr_b["d2h_bytes"], prefault_ms=tr_b["prefault_ms"],
                         upa_d2h_ms=tr_b["d2h_ms"], upa_d2h_bytes=tr_b["d2h_bytes"], ok=ok, upa_max=int(upa.max())))
        del upa, flw
    first, warm = rows[0], rows[1:]

    def med(k):
        return statistics.median(r[k] for r in warm)

    total_ms = med("from_array_ms") + med("upstream_area_ms")
    line = dict(op="FlwdirRaster: from_array + upstream_area() (host numpy in, host numpy out)",
                workload=f"{nrow}x{ncol} synthetic D8, uint8 host array -> int32 host array", dtype="int32",
                value=round(n / total_ms / 1e3, 2), unit="Mcells/s", ms_per_call=round(total_ms, 3),
                from_array_ms=round(med("from_array_ms"), 3), upstream_area_ms=round(med("upstream_area_ms"), 3),
                h2d_ms=round(med("h2d_ms"), 3), d2h_ms=round(med("d2h_ms"), 3),
                compute_ms=round(total_ms - med("h2d_ms") - med("d2h_ms"), 3),
                h2d_GBps=round(med("h2d_bytes") / max(med("h2d_ms"), 1e-9) / 1e6, 2),
                d2h_GBps=round(med("d2h_bytes") / max(med("d2h_ms"), 1e-9) / 1e6, 2),
                upstream_area_d2h_GBps=round(med("upa_d2h_bytes") / max(med("upa_d2h_ms"), 1e-9) / 1e6, 2),
                prefault_ms=round(med("prefault_ms"), 3),
                first_call=dict(from_array_ms=round(first["from_array_ms"], 3), upstream_area_ms=round(first["upstream_area_ms"], 3),
                                h2d_ms=round(first["h2d_ms"], 3), d2h_ms=round(first["d2h_ms"], 3)),
                ok=all(r["ok"] for r in rows) and len({r["upa_max"] for r in rows}) == 1,
                pinned_pcie_GBps_measured=57.0)  # (tools/probes/pcie_probe.cpp on the test box: pinned D2H 56.8, H2D 57.2 GB/s)
    return [line]


def save_n1_record(a, out):
    try:
        with open(N1_RECORD, "w") as f:
            json.dump(dict(size=a.size, regime=a.regime, ms_per_step=out["ms_per_step"],
                           result_checksum=out.get("invariants", {}).get("result_checksum")), f)
    except OSError:
        pass


def load_n1_record(a):
    try:
        with open(N1_RECORD) as f:
            r = json.load(f)
    except (OSError, ValueError):
        return None
    return r if r.get("size") == a.size and r.get("regime") == a.regime else None


def run_distributed(a, rank, world, local):
    """N > 1: one rank per GPU, STRONG scaling — the size x size raster is split into N row blocks;
    every rank generates its own rows (+ one halo row per inner edge) directly in its HBM."""
    from pyflwdir_amd import dist as pdist
    from pyflwdir_amd import hostgroup

    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)  # (only missing for the single-process PFD_BENCH_FORCE_DIST run)
    # host-side group for barrier / max-reduce / unique-id rendezvous: the library's own torch-free TCP group
    # (MASTER_ADDR / MASTER_PORT + 23: next to, not on, the port of a launcher's store).  PFD_BENCH_GROUP=torch
    # uses the gloo group of torch.distributed instead (only meaningful under torch.distributed.run)
    if os.environ.get("PFD_BENCH_GROUP", "tcp") == "tcp":
        grp = hostgroup.HostGroup(rank, world)
    else:
        import torch.distributed as tdist

        tdist.init_process_group(backend="gloo")
        grp = hostgroup.TorchGroup()
    device = local % max(1, _hip.device_count())  # (one rank per GPU; the modulo only matters on test boxes)
    # (ranks that share a GPU share its HBM: the arena is sized per rank)
    reserve_working_memory(device, rank_reserve_gib(64.0, world))
    ncol = nrow_total = a.size
    r0, r1 = pdist.block_rows(nrow_total, world)[rank]
    own = r1 - r0
    top, bot = pdist.halo_of(rank, world)
    synth = REGIMES[a.regime]
    d8_buf = _hip.synth_d8_device(nrow_total, ncol, row0=r0 - top, nrows=own + top + bot, device=device, **synth)
    out_buf = _hip.DeviceBuffer(own * ncol * 4, device)
    # RCCL communicator (all-gather over xGMI); if it cannot be brought up on every rank the same protocol
    # runs with the records travelling through the host group (transport named in the output)
    probe = pdist.DistributedRaster(d8_buf, own, ncol, rank, world, device, memspace=_hip.PFD_DEVICE, group=grp,
                                    transport=os.environ.get("PFD_DIST_TRANSPORT", "auto"), deferred=True)
    comm, transport = probe.comm, probe.transport
    rccl_world = comm.info()["nranks"] if comm is not None else None  # (what RCCL itself reports)
    probe.handle.close()

    def step(profile=False):
        # a fresh handle per step, like the single-GPU bench: decode + local solve + exchange + final pass
        h = _hip.RasterHandle(d8_buf, own, ncol, device=device, memspace=_hip.PFD_DEVICE, halo=(top, bot),
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
    grp.barrier()
    t0 = time.perf_counter()
    timed, marks = [], [t0]
    for _ in range(a.steps):
        timed.append(step(profile=True))
        marks.append(time.perf_counter())  # (a step ends with the pass's own stream synchronisation and agreement)
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    grp.barrier()
    dt = grp.allreduce(time.perf_counter() - t0, "max")
    # every rank's own step times, so that a first hardware run explains itself: a collective step
    # costs what its slowest rank costs — per rank median / max and max over median, and per step the maximum over the ranks
    per = [(b - a_) * 1e3 for a_, b in zip(marks[:-1], marks[1:])]
    all_per = [json.loads(x.decode()) for x in grp.allgather(json.dumps([round(v, 3) for v in per]).encode())]
    rank_steps = dict(per_rank_median_ms=[round(statistics.median(p), 3) for p in all_per],
                      per_rank_max_ms=[round(max(p), 3) for p in all_per],
                      step_max_over_median=[round(max(p) / statistics.median(p), 3) for p in all_per],
                      per_step_max_over_ranks_ms=[round(max(p[i] for p in all_per), 3) for i in range(len(per))])
    segs, info = mean_segments([t[0] for t in timed]), timed[-1][1]
    # checksum of the whole result (the sum over the ranks must equal the 1-GPU run's
    # invariants.result_checksum: SURVEY §8d C4 check iii) and the pit-sum invariant (river regime: every
    # pit sits on the last row of the raster)
    csum = _hip.checksum_i32(out_buf, own * ncol, device)
    n_valid = grp.allreduce(int(info["n_valid"]), "sum")
    n_pits = grp.allreduce(int(info["n_pits"]), "sum")
    csum = grp.allreduce(int(csum), "sum")
    pit_sum = 0
    if rank == world - 1:
        last_codes = d8_buf.download(np.uint8, (ncol,), offset_bytes=(top + own - 1) * ncol)
        last_upa = out_buf.download(np.int32, (ncol,), offset_bytes=(own - 1) * ncol * 4)
        pit_sum = int(last_upa[last_codes == 0].astype(np.int64).sum())
    pit_sum = grp.allreduce(pit_sum, "sum")
    if rank == 0:
        n = nrow_total * ncol
        ms_per_step = dt / a.steps * 1e3
        roof = roofline_upa(segs, own * ncol, own, ncol, ms_per_step)
        per_gpu = B_ALG["upstream_area_cell"] * n / (ms_per_step * 1e-3) / 1e9 / world
        roof.update(per_gpu=True, achieved=round(per_gpu, 2), frac=round(per_gpu / PEAK_HBM_GBS, 5),
                    scope="whole pass, per GPU (rank 0's block)", frac_floor=round(B_FLOOR / B_ALG["upstream_area_cell"] * per_gpu / PEAK_HBM_GBS, 5))
        out = dict(metric="Mcells/s upstream_area on D8 raster", value=round(n * a.steps / dt / 1e6, 2), unit="Mcells/s",
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_per_step, 3),
                   higher_is_better=True, scaling="strong", vs_baseline=None, dtype="int32", data="synthetic",
                   config=dict(workload=f"{nrow_total}x{ncol} synthetic D8 ({a.regime} regime, seed {synth['seed']}) split "
                                        f"into {world} row blocks ({own} rows + halo on rank 0), one block per GPU, "
                                        "upstream_area(unit='cell') int32; a step = decode + local solve + all-gather of "
                                        "the boundary records + interface solve + final pass on fresh handles",
                               n_cells=n, n_valid=n_valid, n_pits=n_pits,
                               parallelism=f"{world} row blocks, 1 all-gather/pass", transport=transport,
                               rccl_world_size=rccl_world, rccl_binding=rccl_binding() if comm is not None else None,
                               host_group=type(grp).__name__,
                               launcher="self-spawned" if os.environ.get("PFD_BENCH_SPAWNED") else "external",
                               retried_with_host_transport=bool(os.environ.get("PFD_BENCH_RETRY")),
                               devices_visible=_hip.device_count()),
                   roofline=roof, rank_steps=rank_steps,
                   invariants=dict(result_checksum=csum,
                                   last_row_pit_sum_equals_n_valid=bool(pit_sum == n_valid) if a.regime == "river" else None))
        n1 = load_n1_record(a)
        if n1 is not None:  # the N = 1 run of the same raster on this box (bench.py writes it at N = 1)
            out["speedup_vs_n1"] = round(n1["ms_per_step"] / ms_per_step, 3)
            out["n1_ms_per_step"] = n1["ms_per_step"]
            if n1.get("result_checksum") is not None:
                out["invariants"]["result_checksum_equals_n1"] = bool(n1["result_checksum"] == csum)
        print(json.dumps(out))
    grp.barrier()
    if comm is not None:
        comm.close()
    grp.close()


C5_SYNTH = dict(seed=2, tilt=100000, white=2, nodata_pct=30)  # the configs[4]-shaped raster of the secondary lines


def run_distributed_op(a, rank, world, local):
    """`--op hand|basins` (BASELINE configs[4]): the sharded collective on a rows x cols tile cut into `world` row blocks,
    one per rank / GPU; boundary rows (hand) travel device to device over RCCL send/recv between neighbours, the basins
    records in one RCCL all-gather (host group when RCCL cannot come up: the line says which).  A step = one complete
    collective call on device-resident inputs (the block's sweep plan is built by the warm-up call); value = cells of the
    whole tile per second, max over the ranks."""
    from pyflwdir_amd import dist as pdist
    from pyflwdir_amd import hostgroup

    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    grp = hostgroup.HostGroup(rank, world)
    device = local % max(1, _hip.device_count())
    reserve_working_memory(device, rank_reserve_gib(96.0, world))
    nrow_total, ncol = a.rows, a.cols
    r0, r1 = pdist.block_rows(nrow_total, world)[rank]
    own = r1 - r0
    top, bot = pdist.halo_of(rank, world)
    ndev = own + top + bot
    d8_buf = _hip.synth_d8_device(nrow_total, ncol, row0=r0 - top, nrows=ndev, device=device, **C5_SYNTH)
    dr = pdist.DistributedRaster(d8_buf, own, ncol, rank, world, device, memspace=_hip.PFD_DEVICE, group=grp,
                                 transport=os.environ.get("PFD_DIST_TRANSPORT", "auto"))
    # upstream area of the own rows (collective), for the drain mask / the outlets
    upa = _hip.DeviceBuffer(own * ncol * 4, device)
    dr.upstream_area(out=upa, memspace=_hip.PFD_DEVICE)
    res = None
    if a.op == "hand":
        import ctypes as C

        elev = _hip.synth_elev_device(nrow_total, ncol, row0=r0 - top, nrows=ndev, device=device, **C5_SYNTH)
        drain = _hip.DeviceBuffer(ndev * ncol, device)
        _hip.check(_hip.lib().pfd_memcpy_h2d(device, C.c_void_p(drain.addr), _hip.ptr(np.zeros(ncol, np.uint8)), C.c_size_t(ncol)))
        _hip.check(_hip.lib().pfd_memcpy_h2d(device, C.c_void_p(drain.addr + (ndev - 1) * ncol), _hip.ptr(np.zeros(ncol, np.uint8)),
                                             C.c_size_t(ncol)))
        band = 2000
        for b0 in range(0, own, band):  # drain = more than 100 upstream cells (rivers), built band by band on the host
            rows = min(band, own - b0)
            u = upa.download(np.int32, (rows, ncol), offset_bytes=b0 * ncol * 4)
            _hip.check(_hip.lib().pfd_memcpy_h2d(device, C.c_void_p(drain.addr + (top + b0) * ncol),
                                                 _hip.ptr(np.ascontiguousarray(u > 100).view(np.uint8)), C.c_size_t(rows * ncol)))
        iters = [0]
        res = _hip.DeviceBuffer(ndev * ncol * 8, device)  # (the result stays resident: one buffer for every step)

        checked = [False]

        def step():  # (the elevations are checked for NaN / inf by the first, untimed call; the buffer does not change)
            _, iters[0] = dr.hand(drain, elev, elev_code=_hip.PFD_F32, out=res, check_finite=not checked[0])
            checked[0] = True

        def checksum():
            return _hip.checksum_i32(_hip.ptr(res).value + top * ncol * 8, own * ncol * 2, device)
        label, dtype, extra_bytes = "hand(drain = upstream cells > 100, elevtn float32) -> float64", "f64", 8
    elif a.op in ("accuflux", "strahler"):
        # the seeded up-sweeps: float32 accuflux of one cell area per ROW (upstream_area in area units on a lat/lon grid)
        # / the Strahler order; exchange-until-stable, every exchange after the first folds only the chains below a
        # halo value that changed (pfd_set_block_update)
        f32 = a.op == "accuflux"
        areas = (np.cos(np.linspace(-0.9, 0.9, nrow_total)) * 0.81).astype(np.float32)[r0 - top:r0 - top + ndev]
        res = _hip.DeviceBuffer(ndev * ncol * (4 if f32 else 1), device)
        iters = [0]

        def step():
            if f32:
                _, iters[0] = dr.accuflux(areas, (-9999, -9999.0, 1), by_row=True, out=res)
            else:
                _, iters[0] = dr.stream_order(out=res)

        def checksum():
            return _hip.checksum_i32(_hip.ptr(res).value + top * ncol * (4 if f32 else 1), own * ncol // (1 if f32 else 4), device)
        label, dtype, extra_bytes = ("accuflux(float32 cell area per row, direction='up') -> float32", "f32", 4) if f32 else \
                                    ("stream_order(type='strahler') -> uint8", "u8", 1)
    else:
        # 1000 outlets of the whole tile: the largest upstream area of sampled rows (every rank offers its rows' maxima)
        rng = np.random.default_rng(5)
        rows_g = np.unique(rng.integers(0, nrow_total, 1000))
        mine = []
        for r in rows_g[(rows_g >= r0) & (rows_g < r1)]:
            row = upa.download(np.int32, (ncol,), offset_bytes=int(r - r0) * ncol * 4)
            mine.append(int(r) * ncol + int(np.argmax(row)))
        parts = grp.allgather(np.array(mine, np.int64).tobytes())
        outl = np.concatenate([np.frombuffer(p, np.int64) for p in parts])
        ids = np.arange(1, outl.size + 1, dtype=np.uint32)
        res = _hip.DeviceBuffer(own * ncol * 4, device)
        iters = [1]

        def step():
            dr.basins(outl, ids, nrow_total, out=res, memspace=_hip.PFD_DEVICE)

        def checksum():
            return _hip.checksum_i32(res, own * ncol, device)
        label, dtype, extra_bytes = f"basins({outl.size} outlets) -> uint32", "u32", 4
    upa.free()
    for _ in range(max(1, a.warmup)):  # (the first call builds the block's sweep plan / pays the cold allocations)
        step()
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    dr.exchanges.clear()
    grp.barrier()
    t0 = time.perf_counter()
    per = []
    for _ in range(a.steps):
        t1 = time.perf_counter()
        step()
        per.append((time.perf_counter() - t1) * 1e3)
    _hip.check(_hip.lib().pfd_device_synchronize(device))
    grp.barrier()
    dt = grp.allreduce(time.perf_counter() - t0, "max")
    csum = grp.allreduce(int(checksum()), "sum") & ((1 << 64) - 1)
    n_ex = len(dr.exchanges) // a.steps
    kinds = sorted({k for k, _ in dr.exchanges})
    ex_bytes = sum(b for _, b in dr.exchanges) // a.steps
    if rank == 0:
        n = nrow_total * ncol
        ms = dt / a.steps * 1e3
        b_alg = B_ALG[{"hand": "hand_f32", "basins": "basins_u32", "accuflux": "accuflux_f32", "strahler": "strahler"}[a.op]]
        per_gpu = b_alg * n / (ms * 1e-3) / 1e9 / world
        out = dict(metric=f"Mcells/s {a.op} on D8 raster", value=round(n * a.steps / dt / 1e6, 2), unit="Mcells/s", n_gpus=world,
                   steps=a.steps, warmup=max(1, a.warmup), ms_per_step=round(ms, 3), ms_per_step_rank0=[round(x, 3) for x in per],
                   higher_is_better=True, scaling="strong", vs_baseline=None, dtype=dtype, data="synthetic",
                   config=dict(workload=f"{nrow_total}x{ncol} synthetic D8 (rough regime, 30 % nodata: BASELINE configs[4] shape) in "
                                        f"{world} row blocks, {label}; a step = one collective call on device-resident inputs, the "
                                        "block's sweep plan built by the warm-up call",
                               n_cells=n, parallelism=f"{world} row blocks", transport=dr.transport,
                               rccl_world_size=dr.comm.info()["nranks"] if dr.comm is not None else None,
                               rccl_binding=rccl_binding() if dr.comm is not None else None,
                               exchanges_per_step=n_ex, exchange_kinds=kinds, exchange_bytes_per_rank_per_step=ex_bytes,
                               iterations=iters[0], devices_visible=_hip.device_count(),
                               # (hand: the timed steps skip the non-finite scan of the unchanged elevation buffer — the
                               #  first, untimed call did it; the default API call pays it every time)
                               check_finite=False if a.op == "hand" else None,
                               launcher="self-spawned" if os.environ.get("PFD_BENCH_SPAWNED") else "external"),
                   roofline=dict(bound="hbm", achieved=round(per_gpu, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                 frac=round(per_gpu / PEAK_HBM_GBS, 5), alg_bytes_per_cell=b_alg, per_gpu=True, traffic=None),
                   invariants=dict(result_checksum=csum))
        rec_path = os.path.join(ROOT, f".bench_n1_{a.op}.json")
        if world == 1:
            try:
                with open(rec_path, "w") as f:
                    json.dump(dict(rows=nrow_total, cols=ncol, ms_per_step=ms, result_checksum=csum), f)
            except OSError:
                pass
        else:
            try:
                with open(rec_path) as f:
                    n1 = json.load(f)
                if n1.get("rows") == nrow_total and n1.get("cols") == ncol:
                    out["speedup_vs_n1"] = round(n1["ms_per_step"] / ms, 3)
                    out["n1_ms_per_step"] = round(n1["ms_per_step"], 3)
                    out["invariants"]["result_checksum_equals_n1"] = bool(n1["result_checksum"] == csum)
            except (OSError, ValueError):
                pass
        print(json.dumps(out))
    grp.barrier()
    if res is not None:
        res.free()
    dr.close()
    grp.close()


def free_port():
    """A TCP port on 127.0.0.1 with 64 free ports above it (MASTER_PORT; the host group listens on MASTER_PORT + 23)."""
    import socket

    for _ in range(64):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        if port + 64 >= 65535:
            continue
        try:
            t = socket.socket()
            t.bind(("127.0.0.1", port + 23))
            t.close()
            return port
        except OSError:
            continue
    raise RuntimeError("no free TCP port pair on 127.0.0.1")


def rank_environments(n, n_devices, port, base_env=None, loopback=False):
    """Environment of each of the n ranks `bench.py --gpus n` starts itself: the variables a launcher would set
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), the torch-free TCP host group, dmabuf IPC for RCCL.
    LOCAL_RANK is the GPU index; with fewer GPUs than ranks (test boxes) the ranks share GPUs and the boundary
    records travel through the host group, because RCCL refuses two ranks on one device."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base_env is None else base_env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 PFD_BENCH_SPAWNED="1")
        e.setdefault("PFD_BENCH_GROUP", "tcp")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if loopback:  # (--rccl-loopback: the stand-in is interposed in front of librccl.so, the RCCL transport is forced)
            e["LD_PRELOAD"] = LOOPBACK_SO + (":" + e["LD_PRELOAD"] if e.get("LD_PRELOAD") else "")
            e["PFD_DIST_TRANSPORT"] = "rccl"
        elif n_devices < n:
            e.setdefault("PFD_DIST_TRANSPORT", "host")
        envs.append(e)
    return envs


def spawn_ranks(a, argv=None, n_devices=None, timeout=3000.0, script=None, extra_env=None, out=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as plain processes (one per GPU), pass rank
    0's JSON line through on stdout (or write it to the file object ``out``), return the first non-zero exit code.
    A rank that dies takes the others down (each child is killed by its own PID) instead of leaving them waiting in a
    collective."""
    import subprocess

    n_devices = _hip.device_count() if n_devices is None else n_devices
    if n_devices < 1:
        raise SystemExit("bench.py: no HIP device visible")
    argv = sys.argv[1:] if argv is None else argv
    loopback = bool(getattr(a, "rccl_loopback", False))
    if loopback and not os.path.exists(LOOPBACK_SO):
        raise SystemExit("bench.py --rccl-loopback: build tests/rccl_loopback first (make -C tests/rccl_loopback)")
    envs = rank_environments(a.gpus, n_devices, free_port(), loopback=loopback)
    procs = []
    for r, e in enumerate(envs):
        e.update(extra_env or {})
        # rank 0 owns stdout (the one JSON line); whatever another rank prints goes to stderr
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=out if r == 0 else sys.stderr))
    deadline = time.time() + timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if live and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
            time.sleep(2.0 if rc != 124 else 0.0)  # (a failing rank's message travels with the agreement first)
            for p in live:
                if p.poll() is None:
                    p.kill()
            for p in live:
                p.wait()
            break
        time.sleep(0.02)
    return rc


def spawn_with_fallback(a, spawn=spawn_ranks, first_timeout=900.0):
    """The self-spawned N-GPU run, made robust for its first contact with real multi-GPU hardware: rank 0's line is
    collected in a file and only printed once the attempt has succeeded; if the attempt with the automatic transport
    (RCCL all-gather over xGMI) fails or does not finish in `first_timeout` seconds, the same job runs once more with
    the boundary records travelling through the host group (PFD_DIST_TRANSPORT=host; the line says which transport
    ran).  An explicitly chosen transport is not second-guessed."""
    import tempfile

    attempts = [({}, first_timeout)]
    if not os.environ.get("PFD_DIST_TRANSPORT") and not getattr(a, "rccl_loopback", False):
        attempts.append((dict(PFD_DIST_TRANSPORT="host", PFD_BENCH_RETRY="1"), 3000.0))
    rc = 1
    for extra, tmo in attempts:
        with tempfile.TemporaryFile(mode="w+") as f:
            rc = spawn(a, timeout=tmo, extra_env=extra, out=f)
            f.seek(0)
            text = f.read()
        if rc == 0 and text.strip():
            sys.stdout.write(text)
            sys.stdout.flush()
            return 0
        print(f"bench.py: the {a.gpus}-rank run failed (exit code {rc})"
              + ("; retrying with the host transport" if extra == {} and len(attempts) > 1 else ""), file=sys.stderr)
        rc = rc or 1
    return rc


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        # no launcher (this is how the driver calls it): bench.py starts its own ranks
        raise SystemExit(spawn_with_fallback(a))
    if a.op != "upstream_area":  # configs[4]: the sharded collectives (also with one rank: the N = 1 reference line)
        return run_distributed_op(a, rank, world, local)
    if world > 1 or os.environ.get("PFD_BENCH_FORCE_DIST"):  # the env knob runs the RCCL path with 1 rank
        return run_distributed(a, rank, world, local)
    device = local
    reserve_working_memory(device, rank_reserve_gib(64.0, 1))
    if a.ops:  # (tools/prof_pmc.sh: the operation lines alone, `steps` warm calls each)
        if a.ops == "c3":
            lines = op_lines(30000, 30000, REGIMES[a.regime], f"30000x30000 synthetic D8 ({a.regime} regime)", a.steps, device)
        else:
            lines = op_lines(36000, 72000, dict(seed=2, tilt=100000, white=2, nodata_pct=30),
                             "36000x72000 synthetic D8 (rough regime, 30 % nodata: BASELINE configs[4] shape)", a.steps, device,
                             ops=("hand", "basins"))
        print(json.dumps(dict(ops=lines)))
        return
    line, cfg = upa_line(a.size, a.size, a.regime, a.steps, a.warmup, device, cpu=not a.no_cpu_baseline,
                         cpu_rows=a.cpu_rows)
    out = dict(metric="Mcells/s upstream_area on D8 raster", value=line.pop("value"), unit="Mcells/s", n_gpus=1,
               steps=a.steps, warmup=a.warmup, ms_per_step=line.pop("ms_per_step"), higher_is_better=True,
               scaling="strong", vs_baseline=None, dtype="int32", data="synthetic", config=cfg)
    out.update(line)
    save_n1_record(a, out)
    if not a.no_secondary:
        try:
            out["secondary"] = secondary_lines(a, device)
        except Exception as exc:  # noqa: BLE001 - the headline line must get out whatever a side line does
            out["secondary"] = []
            out["secondary_error"] = f"{type(exc).__name__}: {exc}"[:400]
            print(f"bench.py: secondary lines failed: {out['secondary_error']}", file=sys.stderr)
    print(json.dumps(out))


def compact_row(line, tag):
    """One short row per side line — op, workload tag, ms, Mcells/s, roofline fractions, first-call ms — so that the whole
    JSON line fits a log tail; the full objects go to the file PFD_BENCH_DETAIL names (tools/prof_evidence.sh)."""
    roof = line.get("roofline", {})
    ms = line.get("ms_per_call", line.get("ms_per_step"))
    row = dict(op=line["op"].split("(")[0], tag=tag, dtype=line.get("dtype"), ms=ms, Mcells_s=line.get("value"),
               frac=roof.get("frac"), frac_measured=roof.get("frac_measured"), B_cell_model=roof.get("alg_bytes_per_cell"))
    if roof.get("traffic") and line.get("value") and ms:
        row["B_cell_measured"] = round(roof["traffic"] / (line["value"] * 1e3 * ms), 2)
    if "quantum_km2" in line:  # the opt-in tolerance mode of upstream_area(unit): told apart from the exact row
        row["mode"] = "exact=False: 64-bit fixed point, order-free"
        row["max_rel_diff_to_exact"] = line["max_rel_diff_to_exact_sampled"]
    if "first_call_on_handle_ms" in line:
        row["first_call_ms"] = line["first_call_on_handle_ms"]
        row["frac_first_call"] = line["roofline_first_call"]["frac"]
    if "invariants" in line:
        row["ok"] = bool(all(v for v in line["invariants"].values() if isinstance(v, bool)))
    for k in ("n_pits", "max_rank", "tile_doubling_rounds", "from_array_ms", "upstream_area_ms", "h2d_ms", "d2h_ms", "compute_ms",
              "h2d_GBps", "d2h_GBps", "upstream_area_d2h_GBps", "first_call"):
        if k in line:
            row[k] = line[k]
    if "ok" in line and "ok" not in row:
        row["ok"] = line["ok"]
    return row


def secondary_lines(a, device):
    """The side lines of the N = 1 run (see the module docstring), one compact row each."""
    full, sec = [], []

    def add(lines, tag):
        for ln in lines:
            full.append(dict(tag=tag, **ln))
            sec.append(compact_row(ln, tag))

    l2, c2 = upa_line(10000, 10000, a.regime, 20, 5, device, cpu=False, checks=False)
    add([dict(op="upstream_area(unit='cell')", workload=c2["workload"], value=l2["value"], unit="Mcells/s",
              ms_per_step=l2["ms_per_step"], ms_per_step_median=l2["ms_per_step_median"], dtype="int32",
              n_valid=c2["n_valid"], n_pits=c2["n_pits"], roofline=l2["roofline"])], f"C2 10000x10000 {a.regime}")
    add(op_lines(30000, 30000, REGIMES[a.regime], f"30000x30000 synthetic D8 ({a.regime} regime)", 3, device),
        f"C3 30000x30000 {a.regime}")
    add(km2_line(30000, 30000, REGIMES[a.regime], f"30000x30000 synthetic D8 ({a.regime} regime)", 3, device), f"C3 30000x30000 {a.regime}")
    # the path a numpy caller of the drop-in takes (host raster in, host raster out), upload / device / download apart
    add(api_lines(10000, 10000, REGIMES[a.regime], device), f"API C2 10000x10000 {a.regime}")
    add(api_lines(30000, 30000, REGIMES[a.regime], device), f"API C3 30000x30000 {a.regime}")
    # configs[4]'s shape: a MERIT-Hydro-like 3-arcsec tile, 72000 x 36000 cells (36000 rows), rough terrain, 30 % ocean
    add(op_lines(36000, 72000, C5_SYNTH, "36000x72000 synthetic D8 (rough regime, 30 % nodata: BASELINE configs[4] shape)", 2,
                 device, ops=("hand", "basins")), "C5 36000x72000 rough, 30 % nodata")
    # workload spread: the same pass on a rough surface, on a pit-riddled one, on a mosaic of the reference's real
    # Rhine raster and on the tile pass's worst case, with the graph statistics that explain the differences
    for reg in ("rough", "meander", "rhine_mosaic", "filled_mosaic", "serpentine"):
        if reg == a.regime:
            continue
        l3, c3 = upa_line(10000, 10000, reg, 10, 2, device, cpu=False, checks=True)
        add([dict(op="upstream_area(unit='cell')", **c3, **l3, unit="Mcells/s", dtype="int32")], f"10000x10000 {reg}")
    # the headline size on a second regime (SURVEY 7 hard part 6: realism next to every number): the reference's Rhine
    # sub-basin tiled over 90000 x 90000 cells by a device kernel (nodata frame per copy)
    if a.size >= 30000:
        l4, c4 = upa_line(a.size, a.size, "rhine_mosaic_device", 3, 1, device, cpu=False, checks=True)
        add([dict(op="upstream_area(unit='cell')", **c4, **l4, unit="Mcells/s", dtype="int32")], f"C4 {a.size}x{a.size} rhine mosaic")
    detail = os.environ.get("PFD_BENCH_DETAIL")
    if detail:
        with open(detail, "w") as f:
            json.dump(full, f)
    return sec


if __name__ == "__main__":
    main()
