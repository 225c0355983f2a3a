set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "ping_pong" > $O/t_pp.log 2>&1 || (tail -40 $O/t_pp.log; exit 1)
tail -2 $O/t_pp.log
timeout -k 10 600 python tools/gemm_vs_vendor.py > $O/gemm_vs_vendor_a.txt 2>&1 || (tail -20 $O/gemm_vs_vendor_a.txt; exit 1)
cat $O/gemm_vs_vendor_a.txt
