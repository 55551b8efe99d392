"""short per-kernel table of a tools/rocpd_summary.py csv:  python tools/kstats.py <kernel_stats.csv> [substr ...]"""
import csv, sys
keys = sys.argv[2:] or ["k_x", "k_plan", "k_fill", "rocclr", "k_path", "k_xround", "k_tile", "k_super", "k_hyper", "k_coarse"]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["kernel"]
    if any(k in n for k in keys):
        print(f"{n.split('(')[0][:64]:66s} calls {r['calls']:>5s} total {float(r['total_us'])/1e3:9.2f} ms avg {float(r['avg_us'])/1e3:9.3f} ms")
