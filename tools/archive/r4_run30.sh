cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4dyn
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "claimed_tiles" > gpurun_out/r4dyn/ops2.log 2>&1; rc=$?; echo "ops rc $rc"; tail -3 gpurun_out/r4dyn/ops2.log
[ $rc -eq 0 ] && timeout -k 10 500 python tools/ab_env.py "helper:" "nohelper:GAVA_QKV_HELPER=0" "claimsonly:GAVA_QKV_HELPER=2" --rounds 3 > gpurun_out/r4dyn/ab3.log 2>&1; tail -4 gpurun_out/r4dyn/ab3.log
