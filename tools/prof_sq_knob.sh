# SQ counters of the tile kernels at 10000^2 with a library knob off / on: SQ_KNOB=NAME (values 0 and 1), $1 = kernel filter
export PFD_ENABLE_KNOBS=1
for v in 0 1; do
  echo "== ${SQ_KNOB}=$v"
  env ${SQ_KNOB}=$v bash $GRAFT_REPO_ROOT/tools/prof_sq.sh "${1:-k_tile_local}"
done
