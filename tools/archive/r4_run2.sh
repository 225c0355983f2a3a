set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_lo" > gpurun_out/r4/t_wlo8_ops.log 2>&1 || (tail -40 gpurun_out/r4/t_wlo8_ops.log; exit 1)
tail -3 gpurun_out/r4/t_wlo8_ops.log
timeout -k 10 600 python tools/wlo_modes.py --modes fp16,fp16+wlo,fp16+wlo8 > gpurun_out/r4/wlo_modes_b.log 2>&1 || (tail -30 gpurun_out/r4/wlo_modes_b.log; exit 1)
cat gpurun_out/r4/wlo_modes_b.log
