cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4addr
export L=gava_clip_amd/libgava_hip_base.so
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_stream_as_16_bit_pair or ping_pong" > gpurun_out/r4addr/ops2.log 2>&1; rc=$?; echo "ops rc $rc"; tail -2 gpurun_out/r4addr/ops2.log
[ $rc -eq 0 ] || exit 1
for k in outpair fc2pair outpair fc2pair; do for lib in $L ""; do echo "== $k lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>&1 | tail -1; done; done
timeout -k 10 500 python tools/ab_env.py "base:GAVA_HIP_LIB=$L" "resaddr:" --rounds 4 > gpurun_out/r4addr/ab2.log 2>&1; tail -3 gpurun_out/r4addr/ab2.log
