cd $GRAFT_REPO_ROOT
for v in A B A B; do
  cp abtmp/$v.so pyflwdir_amd/libpfd_hip.so
  echo "== $v"
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('phases_ms'))"
done
cp abtmp/B.so pyflwdir_amd/libpfd_hip.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py tests/test_gpu_large.py tests/test_gpu_fuzz.py tests/test_gpu_deferred.py -x -q -m gpu 2>&1 | tail -3
