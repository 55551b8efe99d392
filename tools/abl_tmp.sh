python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_large.py tests/test_gpu_blocks.py tests/test_gpu_scale.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
for a in 0; do echo "ablate $a"; PFD_ENABLE_KNOBS=1 PFD_SUPER_ABLATE=$a PROF_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary" bash tools/prof_stats.sh 60 > gpurun_out/ks_$a.csv 2>&1; python tools/kstats.py gpurun_out/ks_$a.csv k_tile k_super k_hyper k_coarse k_link k_push k_exit; done
rm -rf gpurun_out/st
python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('phases_ms'))"
