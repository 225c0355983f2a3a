#!/bin/bash
# round 3: can a k-loop and an epilogue overlap at all?  A CU's first workgroup runs only k-loops, its second only epilogues
# (GAVA_PAIR_MODE 64; +2: "second" by hardware wave slot instead of dispatch order; +128 / +256: one role alone)
O=gpurun_out/r3; mkdir -p $O
L=gava_clip_amd/libgava_hip_abl.so
run() { echo "== $1 :: $(env $2 GAVA_HIP_LIB=$L timeout -k 10 120 python tools/kernel_bench.py $3 --iters 30 2>/dev/null | tail -1)"; }
{
for k in outpart fc2part; do
  run "both roles, by order" "GAVA_PAIR_MODE=64" $k
  run "k-loop half alone, by order" "GAVA_PAIR_MODE=192" $k
  run "epilogue half alone, by order" "GAVA_PAIR_MODE=320" $k
  run "both roles, by hw slot" "GAVA_PAIR_MODE=66" $k
  run "k-loop half alone, by hw slot" "GAVA_PAIR_MODE=194" $k
  run "epilogue half alone, by hw slot" "GAVA_PAIR_MODE=322" $k
done
} 2>&1 | tee $O/pair_exp2.log
