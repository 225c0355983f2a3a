#!/bin/bash
# round 3: the 256^2 producers with part of the workgroups started late (a phase offset between CUs so that one group's
# HBM-bound epilogues meet the other group's k-loops).  Experiment build: tools/ab_build.sh stg -DGAVA_EXP_STAGGER (start
# delay only, nothing inside the loops).  GAVA_STAGGER_MODE 0: odd slots of every XCD, 1: quarters, 2: XCDs 4-7.
O=gpurun_out/r3; mkdir -p $O
L=gava_clip_amd/libgava_hip_stg.so
run() { echo "== $1 :: $(env $2 timeout -k 10 120 python tools/kernel_bench.py $3 --iters 30 2>/dev/null | tail -1)"; }
{
for k in outpart fc2part; do
run "product lib" "GAVA_HIP_LIB=gava_clip_amd/libgava_hip.so" $k
run "stagger build, no delay" "GAVA_HIP_LIB=$L GAVA_PAIR_DELAY=0" $k
for m in 0 1 2; do
for d in 1000 2000 3000 5000; do
  run "mode $m delay $d" "GAVA_HIP_LIB=$L GAVA_STAGGER_MODE=$m GAVA_PAIR_DELAY=$d" $k
done; done; done
} 2>&1 | tee $O/v3_stagger.log
