cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t
timeout 1200 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -40 > gpurun_out/t/pytest.log
cat gpurun_out/t/pytest.log
