#!/bin/bash
# Same-box A/B of per-layer kernels between the product library and experiment builds (tools/ab_build.sh NAME ...):
#   tools/ab_kernels.sh "fc1part qkvpart" NAME1 NAME2 ...     (kernel names: tools/kernel_bench.py)
# interleaved rounds, so that the box's clock drift lands on every variant alike
kernels=$1; shift
for r in 1 2 3; do
for lib in new "$@"; do
  L=gava_clip_amd/libgava_hip_$lib.so; [ $lib = new ] && L=gava_clip_amd/libgava_hip.so
  for k in $kernels; do
  echo "== $lib $k $(GAVA_HIP_LIB=$L python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  done
done; done
