# the evidence run of a round:  bash tools/prof_round.sh TAG   -> gpurun_out/TAG/ (summaries only: the rocpd databases
# are deleted once summarised, gpurun copies at most 64 MiB back); copy what is to be judged to profiles/
T=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/$T
mkdir -p $O
clean() { find gpurun_out -name '*_results.db' -delete; }
export PFD_TOOL_RESERVE_GIB=${PFD_TOOL_RESERVE_GIB:-120}   # (tools/*.py: one arena for the working buffers; bench.py reserves its own)
bash tools/prof_calib.sh ${T}_calib > $O/calib.txt 2>&1; cp gpurun_out/${T}_calib/calib.csv $O/pmc_calibration_widths.csv  # (first: the FETCH_SIZE factor the folds below use)
python bench.py > $O/bench_default_before_pmc.json 2> $O/bench_default.err
bash tools/prof_pmc.sh 90000 ${T}_pmc > $O/pmc90k.txt 2>&1; cp gpurun_out/${T}_pmc/pmc_fetch_write.csv $O/pmc_fetch_write_90000.csv; clean
bash tools/prof_pmc.sh 10000 ${T}_pmc10k > $O/pmc10k.txt 2>&1; cp gpurun_out/${T}_pmc10k/pmc_fetch_write.csv $O/pmc_fetch_write_10000.csv; clean
bash tools/prof_pmc.sh c3 ${T}_pmc_c3 > $O/pmc_c3.txt 2>&1; cp gpurun_out/${T}_pmc_c3/pmc_fetch_write.csv $O/pmc_fetch_write_ops30k.csv; clean
bash tools/prof_pmc.sh c5 ${T}_pmc_c5 > $O/pmc_c5.txt 2>&1; cp gpurun_out/${T}_pmc_c5/pmc_fetch_write.csv $O/pmc_fetch_write_ops_c5.csv; clean
cp profiles/pmc_traffic.json $O/
PFD_BENCH_DETAIL=$O/bench_secondary_full.json python bench.py > $O/bench_default.json 2>/dev/null   # (now with the traffic of THIS build)
bash tools/prof_bench.sh ${T}_bench90k --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/kstats90k.txt 2>&1
cp gpurun_out/${T}_bench90k/kernel_stats.csv $O/bench_90000_kernel_stats.csv; clean
bash tools/prof_bench.sh ${T}_bench10k --size 10000 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary > $O/kstats10k.txt 2>&1
cp gpurun_out/${T}_bench10k/kernel_stats.csv $O/bench_10000_kernel_stats.csv; clean
bash tools/prof_bench.sh ${T}_benchc3 --ops c3 --steps 2 > $O/kstats_c3.txt 2>&1
cp gpurun_out/${T}_benchc3/kernel_stats.csv $O/ops30k_kernel_stats.csv; clean
SQ_SIZE=10000 bash tools/prof_sq.sh "k_tile|k_super" > $O/sq_counters.csv 2>&1; clean
bash tools/run_ops.sh ${T}_ops > /dev/null 2>&1; cp gpurun_out/${T}_ops/*.txt $O/
python tools/bench_blocks.py 11250 8 90000 > $O/blocks8_c4.txt 2>&1
PFD_BLOCK_PHASES=1 python tools/bench_blocks_isolated.py 11250 8 90000 > $O/blocks8_isolated.txt 2>&1
bash tools/run_rank_replay.sh ${T} > /dev/null 2>&1; cp gpurun_out/${T}_rank_replay.txt $O/rank_replay.txt
python bench.py --gpus 8 --rccl-loopback --steps 5 --warmup 1 > $O/bench_gpus8_loopback.json 2> $O/bench_gpus8_loopback.err
for op in hand basins accuflux strahler; do for g in 1 4; do python bench.py --gpus $g --op $op --steps 2 --warmup 1 > $O/op_${op}_n$g.json 2>/dev/null; done; done
python tools/bench_blocks.py 10000 4 > $O/blocks4.txt 2>&1
python tools/bench_hand_blocks.py 36000 72000 4 > $O/hand_blocks_c5.txt 2>&1
python tools/bench_hand_blocks.py 30000 30000 4 0 67108864 100 > $O/hand_blocks_30k.txt 2>&1
python tools/bench_up_blocks.py 30000 30000 4 > $O/up_blocks_30k.txt 2>&1
PFD_UP_FULL=1 python tools/bench_up_blocks.py 30000 30000 4 > $O/up_blocks_30k_full_sweeps.txt 2>&1
python tools/bench_up_blocks.py 36000 72000 4 30 100000 > $O/up_blocks_c5.txt 2>&1
(python tools/plan_time.py 30000 30000 3; python tools/plan_time.py 36000 72000 2 30 100000; python tools/xplan_info.py) > $O/plan_time.txt 2>&1
python tools/wide_probe.py 10000 30000 > $O/wide_probe.txt 2>&1
PROF_CMD="python tools/wide_probe.py 30000" bash tools/prof_cmd.sh 80 > $O/wide_kernels_30k.txt 2>&1
for r in rough meander; do python bench.py --regime $r --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_90000_$r.json 2>/dev/null; done
rm -rf gpurun_out/${T}_calib gpurun_out/${T}_pmc gpurun_out/${T}_pmc10k gpurun_out/${T}_pmc_c3 gpurun_out/${T}_pmc_c5 gpurun_out/${T}_bench90k gpurun_out/${T}_bench10k gpurun_out/${T}_benchc3 gpurun_out/sq1 gpurun_out/sq2 gpurun_out/${T}_ops
du -sh gpurun_out
for f in hand_blocks_c5 hand_blocks_30k blocks8_c4; do tail -n 4 $O/$f.txt; done
