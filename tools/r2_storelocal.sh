#!/bin/bash
# consumer GEMMs: build variants (names after libgava_hip_, built with tools/ab_build.sh) against the product, same box.
# Used in round 2 for the store experiments of DESIGN "Round 2": stores redirected to an L2-resident 128 KiB per workgroup
# (-DGAVA_ABL_STORE_LOCAL, temporary patch), non-temporal stores of the 16-bit outputs, of the fp32 stream, ...
for r in 1 2; do
for lib in new "$@"; do
  L=gava_clip_amd/libgava_hip_$lib.so; [ $lib = new ] && L=gava_clip_amd/libgava_hip.so
  for k in fc1part qkvpart; do
  echo "== $lib $k $(GAVA_HIP_LIB=$L python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  done
done; done
