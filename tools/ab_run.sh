#!/bin/bash
# same-box A/B: tools/ab_run.sh "base prio p4prio" "fc1 qkv fc2 out"   (base = the product library)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in $1; do
  lib=gava_clip_amd/libgava_hip_$v.so; [ "$v" = base ] && lib=gava_clip_amd/libgava_hip.so
  for k in $2; do
    echo -n "$v: "; GAVA_HIP_LIB=$PWD/$lib GAVA_GEMM_VARIANT=${GAVA_GEMM_VARIANT:-3} timeout -k 10 120 python tools/kernel_bench.py $k --iters 20 2>/dev/null | tail -1
  done
done
done
