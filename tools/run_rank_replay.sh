# N = 1 step and the per-rank passes of the 8-block split on the same box: gpurun_out/TAG_rank_replay.txt
#   bash tools/run_rank_replay.sh [TAG] [SIZE] [RANKS]
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}; SIZE=${2:-90000}; RANKS=${3:-8}
O=gpurun_out/${TAG}_rank_replay.txt
python bench.py --size $SIZE --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_n1.json 2> gpurun_out/${TAG}_n1.err
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_n1.json'))['ms_per_step'])")
CS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_n1.json'))['invariants']['result_checksum'])")
echo "N = 1: $MS ms per step, checksum $CS" > $O
python tools/bench_rank_replay.py --size $SIZE --gpus $RANKS --steps ${STEPS:-100} --reserve-gib ${RESERVE_GIB:-24} --n1-ms $MS --checksum $CS >> $O 2>&1
cat $O
