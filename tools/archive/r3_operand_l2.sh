#!/bin/bash
# round 3: what the 256^2 k-loops would run at if an operand never missed the L2 (experiment build, results wrong:
# tools/ab_build.sh opl2 -DGAVA_EXP_OPERAND_L2 -DGAVA_ENABLE_ABLATE; GAVA_OPERAND_L2 bit 1: every tile reads the same A rows,
# bit 2: the same W rows; GAVA_GEMM_ABLATE=4: no epilogue)
O=gpurun_out/r3; mkdir -p $O
L=gava_clip_amd/libgava_hip_opl2.so
run() { echo "== $1 :: $(env $2 GAVA_HIP_LIB=$L timeout -k 10 120 python tools/kernel_bench.py $3 --iters 30 2>/dev/null | tail -1)"; }
{
for k in fc1part qkvpart fc2part; do
  for abl in 0 4; do
    for m in 0 1 2 3; do
      run "ablate $abl operand_l2 $m" "GAVA_GEMM_ABLATE=$abl GAVA_OPERAND_L2=$m" $k
    done
  done
done
} 2>&1 | tee $O/operand_l2.log
