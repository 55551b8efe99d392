# the slowest HIP API calls of a command, with what the GPU was doing around them — to NAME a one-off stall:
#   API_CMD="python tools/bench_blocks_isolated.py 11250 8 90000" API_MIN_MS=4 bash tools/prof_api.sh > gpurun_out/api.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/api
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d $O/st -o s -- bash -c "cd $R && $API_CMD" > $O/cmd.txt 2>&1
cd $R
grep -v "^W2026\|^E2026" $O/cmd.txt | tail -8
python - <<PY
import sqlite3, os
c = sqlite3.connect("$O/st/s_results.db")
cols = [r[1] for r in c.execute("pragma table_info(regions)")]
minms = float(os.environ.get("API_MIN_MS", "4"))
rows = list(c.execute("select name, start, end from regions order by start"))
t0 = rows[0][1]
slow = [(n, s, e) for n, s, e in rows if (e - s) / 1e6 >= minms]
print(f"{len(rows)} HIP API calls; {len(slow)} of them took >= {minms} ms:")
kern = list(c.execute("select name, start, end from kernels order by start"))
for n, s, e in slow:
    inside = [(k, ks, ke) for k, ks, ke in kern if ke > s and ks < e]
    busy = sum(min(ke, e) - max(ks, s) for _, ks, ke in inside) / 1e6
    prevk = [k for k, ks, ke in kern if ke <= s][-1:] or ["-"]
    print(f"  t = {(s - t0) / 1e6:10.2f} ms  {(e - s) / 1e6:9.2f} ms  {n:28s} GPU busy inside {busy:7.2f} ms ({len(inside)} kernels); last kernel before: {prevk[0].split('(')[0][:40]}")
PY
rm -rf $O/st
