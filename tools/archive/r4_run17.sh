cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py -x -q -k "c1_other_seeds or weight_lo_modes_meet" > gpurun_out/r4/t_24seeds.log 2>&1; echo rc=$?; tail -3 gpurun_out/r4/t_24seeds.log
