# same-box A/B of a plan knob on the operation lines of tools/bench_ops.py: A = knob set (the previous form), B = default
#   AB_KNOB=PFD_ROUNDS_LATE [AB_ARGS="30000 30000 0"] [AB_TESTS=1] bash tools/ab_tails.sh
cd $GRAFT_REPO_ROOT
export PFD_TOOL_RESERVE_GIB=${PFD_TOOL_RESERVE_GIB:-100}
if [ -n "$AB_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_blocks.py -x -q -m gpu 2>&1 | tail -3; fi
for v in A B A B; do
  if [ $v = A ]; then echo "== ${AB_KNOB}=1"; env PFD_ENABLE_KNOBS=1 ${AB_KNOB}=1 python tools/bench_ops.py ${AB_ARGS:-30000 30000 0} 2>&1 | grep "^accuflux\|^strahler\|^hand\|exact_plan"
  else echo "== default"; python tools/bench_ops.py ${AB_ARGS:-30000 30000 0} 2>&1 | grep "^accuflux\|^strahler\|^hand\|exact_plan"; fi
done
