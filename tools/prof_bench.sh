# kernel statistics of the default bench.py run:  bash tools/prof_bench.sh [tag] [bench args...]
#   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-prof_bench}
shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/st -o s -- python $R/bench.py "$@" > $O/bench.out 2>&1
cd $R
grep '^{"metric"' $O/bench.out | tail -1 > $O/bench.json
python tools/rocpd_summary.py $O/st/s_results.db > $O/kernel_stats.csv
python tools/kstats.py $O/kernel_stats.csv k_tile k_super k_hyper k_coarse k_link k_x k_push k_decode | head -40
