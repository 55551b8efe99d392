# round-1 evidence run (profiles/r01g_*): kernel stats, PMC traffic, other sizes and operations
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01g
mkdir -p $O
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_profiled.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- $B > /dev/null 2>&1
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --size 30000 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_30000.json 2>&1
python bench.py --size 90000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_90000.json 2>&1
python tools/bench_ops.py 10000 10000 0 > $O/ops_10000.txt 2>&1
python tools/bench_ops.py 30000 30000 0 > $O/ops_30000.txt 2>&1
python tools/bench_ops.py 36000 72000 30 100000 > $O/ops_c5.txt 2>&1
python tools/bench_blocks.py 10000 4 > $O/blocks_4.txt 2>&1
tail -1 $O/bench_default.json | cut -c1-200
