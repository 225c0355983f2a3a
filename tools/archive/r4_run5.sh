set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "ping_pong" > $O/t_pp2.log 2>&1 || (tail -40 $O/t_pp2.log; exit 1)
tail -2 $O/t_pp2.log
timeout -k 10 400 python tools/gemm_vs_vendor.py > $O/gemm_vs_vendor_b.txt 2>&1 || (tail -20 $O/gemm_vs_vendor_b.txt; exit 1)
cat $O/gemm_vs_vendor_b.txt
for k in qkvpart fc1part outpart fc2part; do
  echo "== $k default: $(timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  echo "== $k pingpong: $(GAVA_PP=1 timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
done
