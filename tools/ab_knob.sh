# same-box A/B of a library knob (needs PFD_ENABLE_KNOBS=1): AB_KNOB=NAME, values AB_A / AB_B alternate A B A B; AB_CMD is
# the command whose output is compared (default: the 90000^2 headline without its side lines; the phases are printed)
cd $GRAFT_REPO_ROOT
export PFD_ENABLE_KNOBS=1
for v in A B A B; do
  if [ $v = A ]; then val=${AB_A:-0}; else val=${AB_B:-1}; fi
  echo "== ${AB_KNOB}=$val"
  env ${AB_KNOB}=$val bash -c "${AB_CMD:-python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python tools/ab_line.py}"
done
