# kernel statistics of tools/bench_ops.py (the non-headline operations):  bash tools/prof_ops.sh NROW NCOL [nodata tilt] -> gpurun_out/<tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${5:-prof_ops}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/st -o s -- python $R/tools/bench_ops.py $1 $2 ${3:-0} ${4:-67108864} > $O/ops.txt 2>&1
cd $R
python tools/rocpd_summary.py $O/st/s_results.db > $O/kernel_stats.csv
head -40 $O/kernel_stats.csv
tail -12 $O/ops.txt
