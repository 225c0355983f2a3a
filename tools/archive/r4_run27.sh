cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4hl
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_stream_as_16_bit_pair" > gpurun_out/r4hl/ops3.log 2>&1; echo "ops rc $?"; tail -3 gpurun_out/r4hl/ops3.log
timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -q -s -m gpu -k "residual_stream_as_16_bit_pair or c2_full_batch_vs" > gpurun_out/r4hl/fwd3.log 2>&1; echo "fwd rc $?"; grep "^\[" gpurun_out/r4hl/fwd3.log; tail -2 gpurun_out/r4hl/fwd3.log
for k in outpair fc2pair; do for lib in "" gava_clip_amd/libgava_hip_seg64.so; do echo "== $k lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 120 python tools/kernel_bench.py $k --iters 30 2>&1 | tail -1; done; done
timeout -k 10 400 python tools/ab_env.py "lines:" "seg64:GAVA_HIP_LIB=gava_clip_amd/libgava_hip_seg64.so" --rounds 4 > gpurun_out/r4hl/ab_lines.log 2>&1; tail -3 gpurun_out/r4hl/ab_lines.log
