cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4hl
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_stream_as_16_bit_pair or layernorm_and_join_rows" > gpurun_out/r4hl/ops.log 2>&1; echo "ops rc $?"; tail -5 gpurun_out/r4hl/ops.log
timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -x -q -s -m gpu -k "residual_stream_as_16_bit_pair or c2_full or c3_c5_full_batch" > gpurun_out/r4hl/fwd.log 2>&1; echo "fwd rc $?"; grep "^\[" gpurun_out/r4hl/fwd.log; tail -4 gpurun_out/r4hl/fwd.log
timeout -k 10 300 python tools/ab_env.py "pair:" "fp32:GAVA_PAIR_STREAM=0" --rounds 4 > gpurun_out/r4hl/ab.log 2>&1; tail -6 gpurun_out/r4hl/ab.log
