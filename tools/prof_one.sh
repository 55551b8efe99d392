# kernel trace of one warm operation:  bash tools/prof_one.sh NROW NCOL TAG -> gpurun_out/TAG/kernels.txt (per-dispatch durations of the exact engine)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${3:-one}
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/st -o s -- python $R/tools/accuflux_probe.py $1 $2 > $O/probe.txt 2>&1
cd $R
python - <<PY
import sqlite3
c=sqlite3.connect("$O/st/s_results.db")
rows=list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
out=open("$O/kernels.txt","w")
for n,s,e,g,w in rows[-int(__import__("os").environ.get("PROBE_TAIL","45")):]:
    out.write(f"{n.split('(')[0][:50]:52s} wg {g//max(w,1):8d} {(e-s)/1e3:10.1f} us\n")
PY
tail -45 $O/kernels.txt; grep -v "^W2026\|^E2026" $O/probe.txt | tail -4
