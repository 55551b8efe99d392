cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof_ops.sh 30000 30000 0 67108864 r02n/ops30k > /dev/null 2>&1
bash tools/prof_ops.sh 10000 10000 0 67108864 r02n/ops10k > /dev/null 2>&1
cat $O/pytest.log; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
