cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4text
timeout -k 10 300 python -m pytest tests/test_gpu_forward.py -x -q -s -m gpu -k "really_runs_beside or hipgraph or deterministic" > gpurun_out/r4text/t.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r4text/t.log
timeout -k 10 400 python tools/text_cost.py --rounds 6 > gpurun_out/r4text/cost.log 2>&1; tail -14 gpurun_out/r4text/cost.log
