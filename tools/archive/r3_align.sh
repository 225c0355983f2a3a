#!/bin/bash
# round 3: aligned tile walk of the fp32-output 256^2 kernels (rounds of an XCD = super-tile blocks) against the plain walk
# (GAVA_TILE_ALIGN=0), same box: parity first, then per-kernel and whole-forward A/B, then the fabric counters
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or fold or layernorm_folding" > $O/align_tests.log 2>&1 || { tail -30 $O/align_tests.log; exit 1; }
tail -1 $O/align_tests.log
{
for r in 1 2 3; do
  for k in outpart fc2part out fc2; do
    echo "== aligned $k $(python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
    echo "== plain   $k $(GAVA_TILE_ALIGN=0 python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  done
done
python tools/ab_env.py "aligned:" "plain:GAVA_TILE_ALIGN=0" --rounds 3 2>&1 | grep "=="
python tools/ab_env.py "aligned:" "plain:GAVA_TILE_ALIGN=0" --rounds 2 --config c5 2>&1 | grep "=="
} 2>&1 | tee $O/align_ab.log
