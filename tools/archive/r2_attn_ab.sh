#!/bin/bash
# attention variants, same box, interleaved: libs given as arguments (names after libgava_hip_, "new" = product)
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_ops.py -q -x -k attention > gpurun_out/r2/attn_tests.log 2>&1 || { tail -30 gpurun_out/r2/attn_tests.log; exit 1; }
tail -1 gpurun_out/r2/attn_tests.log
for l in stamps np_stamps; do [ -f gava_clip_amd/libgava_hip_$l.so ] && GAVA_HIP_LIB=gava_clip_amd/libgava_hip_$l.so python tools/attn_stamps.py 512 2>&1 | grep "cycles per problem"; done
for r in 1 2 3; do
for lib in "$@"; do
  L=gava_clip_amd/libgava_hip_$lib.so; [ $lib = new ] && L=gava_clip_amd/libgava_hip.so
  echo "== $lib c2 $(GAVA_HIP_LIB=$L python tools/kernel_bench.py attn --iters 50 2>/dev/null | tail -1)"
  echo "== $lib c2np $(GAVA_ATTN_PERSIST=0 GAVA_HIP_LIB=$L python tools/kernel_bench.py attn --iters 50 2>/dev/null | tail -1)"
  echo "== $lib c5 $(GAVA_HIP_LIB=$L python tools/kernel_bench.py attn --iters 20 --B 32 --cfg VIT_L14_T32 2>/dev/null| tail -1)"
done; done
