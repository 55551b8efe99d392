# per-dispatch timeline (start offset, duration, gap to the previous kernel) of the LAST `TL_N` kernels of a command:
#   TL_CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary" TL_N=60 bash tools/prof_timeline.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/st -o s -- bash -c "cd $R && $TL_CMD" > $O/cmd.txt 2>&1
cd $R
python - <<PY
import sqlite3, os
c=sqlite3.connect("$O/st/s_results.db")
rows=list(c.execute("select name, start, end from kernels order by start"))
n=int(os.environ.get("TL_N","60"))
rows=rows[-n:]
t0=rows[0][1]; prev=None
for name,s,e in rows:
    gap=0 if prev is None else (s-prev)/1e3
    print(f"{(s-t0)/1e3:10.1f} us  dur {(e-s)/1e3:9.1f}  gap {gap:7.1f}  {name.split('(')[0][:60]}")
    prev=e
PY
