export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --size ${SQ_SIZE:-10000} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rm -rf $R/gpurun_out/sq1 $R/gpurun_out/sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/sq1 -o q -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --kernel-trace -d $R/gpurun_out/sq2 -o q -- $B > /dev/null 2>&1
cd $R
python tools/rocpd_pmc_summary.py gpurun_out/sq1/q_results.db gpurun_out/sq2/q_results.db | grep -E "${1:-k_tile}" | sort
