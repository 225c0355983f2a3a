#!/bin/bash
# round 3, step 1: the two-workgroups-per-CU producer kernel - parity, then same-box A/B against the 256^2 kernel
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "pair or named or without_the_row_stats or folded_into" > $O/pair_tests.log 2>&1 || { tail -40 $O/pair_tests.log; exit 1; }
tail -3 $O/pair_tests.log
for r in 1 2 3; do
  for k in outpart fc2part out fc2; do
    echo "== pair $k $(timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
    echo "== v3   $k $(GAVA_GEMM_VARIANT=3 timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  done
done 2>&1 | tee $O/pair_ab.log
