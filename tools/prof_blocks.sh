# rocprofv3 kernel statistics of the in-process row-block run (tools/bench_blocks.py ARGS): where a block's time goes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pb
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pb -o s -- python $R/tools/bench_blocks.py "$@" > $R/gpurun_out/pb_run.txt 2>&1
cd $R
tail -n 3 gpurun_out/pb_run.txt
python tools/rocpd_summary.py gpurun_out/pb/s_results.db | head -40
rm -rf gpurun_out/pb
