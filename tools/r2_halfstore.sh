#!/bin/bash
# consumer GEMMs (fc1 / qkv with the LayerNorm fold, partials mode): build variants given as arguments vs the product, same box
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_ops.py -q -x -k "fold or fused or gemm" > gpurun_out/r2/fold_tests.log 2>&1 || { tail -30 gpurun_out/r2/fold_tests.log; exit 1; }
tail -1 gpurun_out/r2/fold_tests.log
for r in 1 2 3; do
for lib in new "$@"; do
  L=gava_clip_amd/libgava_hip_$lib.so; [ $lib = new ] && L=gava_clip_amd/libgava_hip.so
  for k in fc1part qkvpart; do
  echo "== $lib $k $(GAVA_HIP_LIB=$L python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  done
done; done
