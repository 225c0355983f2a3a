cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python tools/wlo_modes.py --no-time --modes fp16,fp16+wlo,fp16+wlo8 > gpurun_out/r4/wlo_modes_24.log 2>&1 || tail -20 gpurun_out/r4/wlo_modes_24.log
tail -30 gpurun_out/r4/wlo_modes_24.log | cut -c1-230
