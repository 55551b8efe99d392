# first order-sensitive call of a FRESH process against the warm-allocator first call (tools/plan_time.py), with and
# without a reserved arena, at the configs[4] shape and at 30000^2:  bash tools/first_call.sh TAG -> gpurun_out/TAG_first_call.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05}_first_call.txt
: > $O
for rep in 1 2 3; do
  for gib in "" 150; do
    echo "== fresh process $rep, PFD_TOOL_RESERVE_GIB='$gib'" >> $O
    PFD_TOOL_RESERVE_GIB=$gib python tools/plan_time.py 36000 72000 2 30 100000 2>&1 | grep -v "^W2026\|^E2026" >> $O
    PFD_TOOL_RESERVE_GIB=$gib python tools/plan_time.py 30000 30000 2 2>&1 | grep -v "^W2026\|^E2026" >> $O
  done
done
cat $O
