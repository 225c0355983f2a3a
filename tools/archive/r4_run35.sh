cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4pp2
export L=gava_clip_amd/libgava_hip_pp2.so
GAVA_HIP_LIB=$L GAVA_PP=1 timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ping_pong or residual_stream_as_16_bit_pair or folded" > gpurun_out/r4pp2/ops2.log 2>&1; rc=$?; echo "ops rc $rc"; tail -3 gpurun_out/r4pp2/ops2.log
[ $rc -eq 0 ] || exit 1
for k in fc1part qkvpart; do for v in "GAVA_PP=2" "GAVA_PP=1"; do echo "== $k $v"; env $v GAVA_HIP_LIB=$L timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>&1 | tail -1; done; done
timeout -k 10 600 python tools/ab_env.py "pp4:" "pp2prod:GAVA_HIP_LIB=$L" "pp2all:GAVA_HIP_LIB=$L,GAVA_PP=1" --rounds 4 > gpurun_out/r4pp2/ab2.log 2>&1; tail -4 gpurun_out/r4pp2/ab2.log
