#!/bin/bash
# round 3: the 256^2 kernel (one workgroup per CU) with the roles split over CUs: even slots only k-loops, odd slots only epilogues
O=gpurun_out/r3; mkdir -p $O
L=gava_clip_amd/libgava_hip_abl.so
run() { echo "== $1 :: $(env $2 GAVA_HIP_LIB=$L timeout -k 10 120 python tools/kernel_bench.py $3 --iters 30 2>/dev/null | tail -1)"; }
{
for k in outpart fc2part; do
  run "v3 base" "GAVA_GEMM_ABLATE=0" $k
  run "v3 no epilogue (all CUs)" "GAVA_GEMM_ABLATE=4" $k
  run "v3 roles split over CUs" "GAVA_GEMM_ABLATE=8" $k
  run "v3 k-loop half alone" "GAVA_GEMM_ABLATE=24" $k
  run "v3 epilogue half alone" "GAVA_GEMM_ABLATE=40" $k
done
} 2>&1 | tee $O/v3_roles.log
