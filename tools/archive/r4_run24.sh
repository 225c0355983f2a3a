cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4hl
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_stream_as_16_bit_pair or layernorm_and_join_rows" > gpurun_out/r4hl/ops.log 2>&1; echo "ops rc $?"; tail -3 gpurun_out/r4hl/ops.log
timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -x -q -s -m gpu -k "residual_stream_as_16_bit_pair" > gpurun_out/r4hl/fwd.log 2>&1; echo "fwd rc $?"; grep "^\[" gpurun_out/r4hl/fwd.log; tail -3 gpurun_out/r4hl/fwd.log
S=$(date +%s); timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4hl/all.log 2>&1; echo "all rc $? $(( $(date +%s) - S )) s"; tail -4 gpurun_out/r4hl/all.log
