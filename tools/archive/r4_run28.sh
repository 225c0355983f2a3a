cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4dyn
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "claimed_tiles" > gpurun_out/r4dyn/ops.log 2>&1; rc=$?; echo "ops rc $rc"; tail -3 gpurun_out/r4dyn/ops.log
[ $rc -eq 0 ] && timeout -k 10 300 python -m pytest tests/test_gpu_forward.py -x -q -s -m gpu -k "c2_full or hipgraph" > gpurun_out/r4dyn/fwd.log 2>&1; echo "fwd rc $?"; grep "^\[" gpurun_out/r4dyn/fwd.log; tail -2 gpurun_out/r4dyn/fwd.log
[ $rc -eq 0 ] && timeout -k 10 400 python tools/ab_env.py "helper:" "nohelper:GAVA_QKV_HELPER=0" --rounds 4 > gpurun_out/r4dyn/ab.log 2>&1; tail -3 gpurun_out/r4dyn/ab.log
