export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/st
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/st -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/st/s_results.db | head -${1:-12}
