set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01d_stats -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r01d_bench.json 2> $R/gpurun_out/r01d_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r01d_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r01d_write -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/r01d_sq1 -o q -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d $R/gpurun_out/r01d_sq2 -o q -- $B > /dev/null 2>&1
cd $R
find gpurun_out -name "*.db" | head -20
tail -1 gpurun_out/r01d_bench.json | cut -c1-300
