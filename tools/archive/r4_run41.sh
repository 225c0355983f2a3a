cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4gelu
export L=gava_clip_amd/libgava_hip_base.so
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r4gelu/ops.log 2>&1; rc=$?; echo "ops rc $rc"; tail -2 gpurun_out/r4gelu/ops.log
[ $rc -eq 0 ] || exit 1
for lib in $L "" $L ""; do echo "== fc1part lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 120 python tools/kernel_bench.py fc1part --iters 30 2>&1 | tail -1; done
timeout -k 10 500 python tools/ab_env.py "base:GAVA_HIP_LIB=$L" "gelu8:" --rounds 4 > gpurun_out/r4gelu/ab.log 2>&1; tail -3 gpurun_out/r4gelu/ab.log
