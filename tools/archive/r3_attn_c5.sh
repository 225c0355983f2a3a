#!/bin/bash
# round 3: the 320-key attention class (ViT-L/14, T=32): one-tile form for the odd 17th query tile (experiment build:
# tools/ab_build.sh odd -DGAVA_ATTN_ODD_TILE) against the product build, same box
O=gpurun_out/r3; mkdir -p $O
for r in 1 2 3; do
  echo "one-tile form $(GAVA_HIP_LIB=gava_clip_amd/libgava_hip_odd.so python tools/kernel_bench.py attn --cfg VIT_L14_T32 --B 32 --iters 20 2>/dev/null | tail -1)"
  echo "product       $(python tools/kernel_bench.py attn --cfg VIT_L14_T32 --B 32 --iters 20 2>/dev/null | tail -1)"
done | tee $O/attn_c5.log
