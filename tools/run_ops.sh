cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ops}
mkdir -p $O
export PFD_DEBUG=1
python tools/bench_ops.py 10000 10000 0 > $O/ops_10000.txt 2>&1
python tools/bench_ops.py 30000 30000 0 > $O/ops_30000.txt 2>&1
python tools/bench_ops.py 36000 72000 30 100000 > $O/ops_c5.txt 2>&1
python tools/bench_ops.py 10000 10000 0 3000 > $O/ops_meander.txt 2>&1
tail -n 20 $O/ops_10000.txt $O/ops_30000.txt $O/ops_c5.txt $O/ops_meander.txt
