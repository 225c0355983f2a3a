cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "race_screen" 2>&1 | tail -5
