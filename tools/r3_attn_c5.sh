#!/bin/bash
# round 3: the 320-key attention class (ViT-L/14, T=32): one-tile form for the odd 17th query tile against the build before it
O=gpurun_out/r3; mkdir -p $O
for r in 1 2 3; do
  echo "new $(python tools/kernel_bench.py attn --cfg VIT_L14_T32 --B 32 --iters 20 2>/dev/null | tail -1)"
  echo "old $(GAVA_HIP_LIB=gava_clip_amd/libgava_hip_attnold.so python tools/kernel_bench.py attn --cfg VIT_L14_T32 --B 32 --iters 20 2>/dev/null | tail -1)"
done | tee $O/attn_c5.log
