# 20 consecutive runs (fresh process each) of the two block benchmarks with one reserved arena: per-phase max / median.
#   bash tools/stability.sh TAG [RUNS]   -> gpurun_out/TAG_stability.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}; RUNS=${2:-20}
O=gpurun_out/${TAG}_stability.txt
: > $O
export PFD_TOOL_RESERVE_GIB=${PFD_TOOL_RESERVE_GIB:-120}
for i in $(seq 1 $RUNS); do
  echo "== run $i: bench_blocks_isolated 11250 8 90000" >> $O
  python tools/bench_blocks_isolated.py 11250 8 90000 2>&1 | grep -v "^W2026\|^E2026" >> $O
  echo "== run $i: bench_hand_blocks 36000 72000 4" >> $O
  python tools/bench_hand_blocks.py 36000 72000 4 2>&1 | grep -v "^W2026\|^E2026" >> $O
done
python tools/stability_summary.py $O | tee -a $O
