# the evidence run of a round:  bash tools/prof_round.sh TAG   -> gpurun_out/TAG_* (copy the summaries to profiles/)
T=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_bench.sh ${T}_bench > /dev/null 2>&1
bash tools/prof_pmc.sh 90000 ${T}_pmc > /dev/null 2>&1
bash tools/prof_pmc.sh 10000 ${T}_pmc10k > /dev/null 2>&1
cp profiles/pmc_traffic.json gpurun_out/${T}_pmc10k/
python bench.py > gpurun_out/${T}_bench_default.json 2>/dev/null
bash tools/run_ops.sh ${T}_ops > /dev/null 2>&1
SQ_SIZE=10000 bash tools/prof_sq.sh "k_tile<" > gpurun_out/${T}_sq.csv 2>&1
bash tools/prof_ops.sh 30000 30000 0 67108864 ${T}_ops30k > /dev/null 2>&1
ls gpurun_out | grep ${T}_
