# HBM traffic per kernel at the benchmarked sizes: separate --pmc passes (FETCH_SIZE costs 3 of the 4 TCC slots,
# WRITE_SIZE 2: they cannot share a pass), kernel trace only.
#   bash tools/prof_pmc.sh [size] [tag]          the upstream_area pass at size x size
#   bash tools/prof_pmc.sh c3|c5 [tag]           the operation lines (bench.py --ops c3|c5, 2 warm calls each)
# -> gpurun_out/<tag>/pmc_fetch_write.csv, profiles/pmc_traffic.json updated
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-90000}
T=${2:-pmc_$S}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
if [ "$S" = "c3" ] || [ "$S" = "c5" ]; then
  B="python $R/bench.py --ops $S --steps 2"
else
  B="python $R/bench.py --size $S --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- $B > $O/write.log 2>&1
cd $R
python tools/rocpd_pmc_summary.py $O/fetch/f_results.db $O/write/w_results.db > $O/pmc_fetch_write.csv
python tools/pmc_traffic.py $O/pmc_fetch_write.csv $S
