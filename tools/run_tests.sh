# the GPU test suite on the box: full log under gpurun_out/t/, the summary lines on stdout
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t
timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q "$@" > gpurun_out/t/pytest.log 2>&1
echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|error" gpurun_out/t/pytest.log | tail -8
