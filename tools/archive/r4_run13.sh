cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_boundary.py -x -q > $O/t_fwd_early.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_fwd_early.log
timeout -k 10 900 python tools/ab_env.py "late:GAVA_SIDE_EARLY=0" "early:GAVA_SIDE_EARLY=1" --rounds 3 2>&1 | grep "=="
timeout -k 10 900 python tools/ab_env.py "late:GAVA_SIDE_EARLY=0" "early:GAVA_SIDE_EARLY=1" --rounds 2 --config c5 2>&1 | grep "=="
