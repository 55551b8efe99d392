# rocprofv3 kernel statistics of any command: PROF_CMD (required), $1 = lines of the summary to print
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc -o s -- bash -c "cd $R && $PROF_CMD" > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/pc/s_results.db | head -${1:-30}
