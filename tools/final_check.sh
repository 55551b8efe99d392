# the last check of a round on its final build: the whole GPU suite, the default bench line, the operation lines
#   bash tools/final_check.sh TAG  -> gpurun_out/TAG/
cd $GRAFT_REPO_ROOT
T=${1:-final}
O=gpurun_out/$T
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep 'passed\|failed\|rror' | tail -5 > $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
PFD_BENCH_DETAIL=$O/bench_secondary_full.json python bench.py > $O/bench_default.json 2> $O/bench_default.err
export PFD_TOOL_RESERVE_GIB=120
bash tools/run_ops.sh ${T}_ops > /dev/null 2>&1; cp gpurun_out/${T}_ops/*.txt $O/; rm -rf gpurun_out/${T}_ops
bash tools/prof_bench.sh ${T}_benchc3 --ops c3 --steps 2 > /dev/null 2>&1; cp gpurun_out/${T}_benchc3/kernel_stats.csv $O/ops30k_kernel_stats.csv; rm -rf gpurun_out/${T}_benchc3
find gpurun_out -name '*_results.db' -delete
cat $O/gpu_tests.txt $O/smoke.txt; tail -c 600 $O/bench_default.json
