#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default ROCm 7.2 output) into the per-kernel
table `rocprofv3 --stats` prints: name, calls, total ns, average ns, percentage.

    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, tot, avg, pct in rows:
        print(f"\"{name}\",{calls},{tot:.3f},{avg:.3f},{pct:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
