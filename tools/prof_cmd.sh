# rocprofv3 kernel statistics of any command: PROF_CMD (required), $1 = lines of the summary to print (short names)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc -o s -- bash -c "cd $R && $PROF_CMD" > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/pc/s_results.db > gpurun_out/pc_summary.csv
python - <<PY | head -${1:-30}
import csv
for r in csv.DictReader(open("gpurun_out/pc_summary.csv")):
    print(f"{r['kernel'].split('(')[0][:56]:58s} calls {int(r['calls']):5d} total {float(r['total_us'])/1e3:9.2f} ms avg {float(r['avg_us'])/1e3:8.3f} ms")
PY
rm -rf $R/gpurun_out/pc
