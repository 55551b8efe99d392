#!/usr/bin/env python3
"""Fold a rocpd PMC summary (tools/rocpd_pmc_summary.py) into gpurun_out/pmc_traffic.json / profiles/pmc_traffic.json:
per kernel and raster size the HBM bytes per launch = (FETCH_SIZE * fetch_factor + WRITE_SIZE) * 1024.

FETCH_SIZE / WRITE_SIZE are in KB.  gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes for wide
streaming reads (MI355X_MICROARCH.md, HBM); the factor for THIS access pattern is calibrated in the same
run on a kernel with a known read volume: k_verify_upa streams the int32 result once (4 B/cell) plus the
codes (1 B/cell, neighbours from cache) — 5 bytes per cell.  WRITE_SIZE was calibrated in round 1 (x1.00
for coalesced dword/16-byte stores, x1.25 for byte stores; profiles/README.md).

    python tools/pmc_traffic.py <pmc_fetch_write.csv> <size>
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path, size):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.setdefault(r["kernel"], {})[r["counter"]] = float(r["avg_value"])
    n = size * size
    cal = None
    for k, v in rows.items():
        if "k_verify_upa" in k and "FETCH_SIZE" in v:
            cal = dict(kernel=k, known_bytes=5 * n, fetch_kb=v["FETCH_SIZE"],
                       fetch_factor=round(5 * n / (v["FETCH_SIZE"] * 1024), 4))
    factor = cal["fetch_factor"] if cal else 1.0
    # the factor is a property of the request width: 1 (narrow requests) or 2 (128-byte requests counted as 64)
    factor_used = 2.0 if factor > 1.5 else 1.0
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tab = json.load(open(out_path))
    except (OSError, ValueError):
        tab = {}
    for k, v in rows.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        tab[f"{k}|{size}x{size}"] = dict(fetch_kb_raw=v["FETCH_SIZE"], write_kb_raw=v["WRITE_SIZE"],
                                         fetch_factor=factor_used,
                                         bytes_per_launch=(v["FETCH_SIZE"] * factor_used + v["WRITE_SIZE"]) * 1024)
    tab[f"_calibration|{size}x{size}"] = cal
    json.dump(tab, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in tab.items() if k.endswith(f"{size}x{size}") and ("k_tile" in k or k[0] == "_")},
                     indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
