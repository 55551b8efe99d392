#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases.

    python tools/rocpd_pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db
"""
import sqlite3
import sys

print("kernel,counter,dispatches,avg_value,sum_value")
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
         "group by kernel_name, counter_name order by sum(value) desc")
    for name, ctr, n, avg, tot in c.execute(q):
        print(f"\"{name}\",{ctr},{n},{avg:.3f},{tot:.3f}")
