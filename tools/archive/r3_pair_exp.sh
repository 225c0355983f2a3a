#!/bin/bash
# round 3: what limits the two-workgroups-per-CU producer kernel?  (experiment build: tools/ab_build.sh abl -DGAVA_ENABLE_ABLATE)
# GAVA_PAIR_MODE bits: 1 delay the CU's second workgroup by GAVA_PAIR_DELAY x 10 ns, 2 "second" = by hardware wave slot (else by
# dispatch order), 4 / 32 issue priority to the first / second workgroup, 8 k-loop only, 16 epilogue only
O=gpurun_out/r3; mkdir -p $O
L=gava_clip_amd/libgava_hip_abl.so
run() { echo "== $1 :: $(env $2 GAVA_HIP_LIB=$L timeout -k 10 120 python tools/kernel_bench.py $3 --iters 30 2>/dev/null | tail -1)"; }
{
for k in outpart fc2part; do
  run "pair base" "GAVA_PAIR_MODE=0" $k
  run "pair k-loop only" "GAVA_PAIR_MODE=8" $k
  run "pair epilogue only" "GAVA_PAIR_MODE=16" $k
  run "v3 base" "GAVA_GEMM_VARIANT=3" $k
  run "v3 no epilogue" "GAVA_GEMM_VARIANT=3 GAVA_GEMM_ABLATE=4" $k
  run "pair prio first" "GAVA_PAIR_MODE=4" $k
  run "pair prio second" "GAVA_PAIR_MODE=32" $k
  run "pair prio first (hw slot)" "GAVA_PAIR_MODE=6" $k
done
for d in 500 1000 2000 3000 4000 6000; do
  run "pair delay $d order" "GAVA_PAIR_MODE=1 GAVA_PAIR_DELAY=$d" outpart
  run "pair delay $d hwslot" "GAVA_PAIR_MODE=3 GAVA_PAIR_DELAY=$d" outpart
done
for d in 2000 4000 7000 10000; do
  run "pair delay $d order" "GAVA_PAIR_MODE=1 GAVA_PAIR_DELAY=$d" fc2part
  run "pair delay $d hwslot" "GAVA_PAIR_MODE=3 GAVA_PAIR_DELAY=$d" fc2part
done
run "pair base again" "GAVA_PAIR_MODE=0" outpart
} 2>&1 | tee $O/pair_exp1.log
