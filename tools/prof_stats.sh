# rocprofv3 kernel statistics of one bench.py configuration: PROF_ARGS (default: the 90000^2 headline, no secondary lines)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/st
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/st -o s -- python $R/bench.py ${PROF_ARGS:---steps 5 --warmup 1 --no-cpu-baseline --no-secondary} > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/st/s_results.db | head -${1:-24}
