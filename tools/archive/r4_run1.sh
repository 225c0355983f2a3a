set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_ops.py -x -q -k "weight_lo" > gpurun_out/r4/t_wlo_ops.log 2>&1 || (tail -30 gpurun_out/r4/t_wlo_ops.log; exit 1)
tail -3 gpurun_out/r4/t_wlo_ops.log
python tools/wlo_modes.py --modes fp16,fp16+wlo > gpurun_out/r4/wlo_modes_a.log 2>&1 || (tail -30 gpurun_out/r4/wlo_modes_a.log; exit 1)
cat gpurun_out/r4/wlo_modes_a.log
