#!/usr/bin/env python3
"""Fold a rocpd PMC summary (tools/rocpd_pmc_summary.py) into profiles/pmc_traffic.json: per kernel and raster size the
HBM bytes per launch = (FETCH_SIZE * fetch_factor + WRITE_SIZE) * 1024, plus
  "_whole_pass|SxS"   all kernels of ONE timed upstream_area step together (tile passes, exit graph, memsets),
  "_op:<tag>|RxC"     all kernels of ONE warm call of an operation (bench.py --ops c3|c5).

FETCH_SIZE / WRITE_SIZE are in KB.  gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM).  Round 4 calibrated it on known-bytes streaming kernels per access width
(tools/prof_calib.sh, `_calibration_widths` in the table): coalesced reads of 1, 4, 8 and 16 bytes per lane ALL report
exactly half their bytes (factor 2.00); a scattered 4-byte read reports one 64-byte request (its true size is 64 or
128 bytes: x2 is an upper bound there); WRITE_SIZE is exact (x1.00; x0.985 for byte stores).  The table therefore
doubles FETCH_SIZE.  (Rounds 1-3 used k_verify_upa as the known-bytes kernel and found x0.95: that kernel re-reads
neighbouring rows of the result from HBM, so its "known" 5 B/cell was too low — recorded below as a cross-check only.)

    python tools/pmc_traffic.py <pmc_fetch_write.csv> <size | c3 | c5>
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_KERNELS = ("k_tile", "k_exit_lists", "k_boundary_records", "k_super", "k_link3", "k_link4", "k_hyper", "k_sx_totals", "k_push4", "k_coarse_round",
                "k_check_saturated", "fillBuffer", "k_pass_clear", "k_l4_restart")
# kernels of one warm call, by a substring of their (templated) names; calls per bench.py --ops run = steps + 1
OPS = {"accuflux_f32_up": ("AccuUp<float",), "strahler": ("Strahler",), "hand_f32": ("Hand<float",),
       "basins_u32": ("k_path<1", "k_xround<1", "k_xinit<1", "k_labels_out", "k_seed", "k_flag_seed", "k_zero_seed")}
OP_SHAPE = {"c3": (30000, 30000), "c5": (36000, 72000)}


def main(path, what):
    rows, calls, sums = {}, {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.setdefault(r["kernel"], {})[r["counter"]] = float(r["avg_value"])
            calls[r["kernel"]] = int(r["dispatches"])
            sums.setdefault(r["kernel"], {})[r["counter"]] = float(r["sum_value"])
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tab = json.load(open(out_path))
    except (OSError, ValueError):
        tab = {}
    if what in OP_SHAPE:
        nrow, ncol = OP_SHAPE[what]
        cal = None
    else:
        nrow = ncol = int(what)
        n = nrow * ncol
        cal = None
        for k, v in rows.items():
            if "k_verify_upa" in k and "FETCH_SIZE" in v:
                cal = dict(kernel=k, known_bytes=5 * n, fetch_kb=v["FETCH_SIZE"],
                           fetch_factor=round(5 * n / (v["FETCH_SIZE"] * 1024), 4))
    widths = tab.get("_calibration_widths", {})
    factor_used = float(widths.get("read_16B", {}).get("factor", 2.0))  # (2.00 for every coalesced width measured)
    size = f"{nrow}x{ncol}"

    def total(k):
        v = sums[k]
        return (v.get("FETCH_SIZE", 0.0) * factor_used + v.get("WRITE_SIZE", 0.0)) * 1024

    for k, v in rows.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        tab[f"{k}|{size}"] = dict(fetch_kb_raw=v["FETCH_SIZE"], write_kb_raw=v["WRITE_SIZE"], fetch_factor=factor_used,
                                  dispatches=calls[k], bytes_per_launch=(v["FETCH_SIZE"] * factor_used + v["WRITE_SIZE"]) * 1024)
    if what in OP_SHAPE:
        for tag, pats in OPS.items():
            ks = [k for k in sums if any(p in k for p in pats)]
            if not ks:
                continue
            ncalls = 3  # bench.py --ops ... --steps 2: the first call + 2 warm calls run the same sweep kernels
            tab[f"_op:{tag}|{size}"] = dict(bytes_per_launch=sum(total(k) for k in ks) / ncalls, calls=ncalls, kernels=sorted(ks))
    else:
        passes = max([calls[k] for k in calls if "k_tile_final_fast" in k] or [0])
        if passes:
            ks = [k for k in sums if any(p in k for p in STEP_KERNELS)]
            tab[f"_whole_pass|{size}"] = dict(bytes_per_launch=sum(total(k) for k in ks) / passes, passes=passes,
                                              kernels=sorted(ks))
        tab[f"_calibration|{size}"] = cal
    json.dump(tab, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "kernels"})
                      for k, v in tab.items() if k.endswith(size) and (k[0] == "_" or "k_tile" in k)}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
