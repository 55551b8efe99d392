cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02a/pytest.log
timeout 900 python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
tail -c 3000 gpurun_out/r02a/bench_default.json; tail -5 gpurun_out/r02a/bench_default.err; cat gpurun_out/r02a/pytest.log
