cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -x -q -s -k "small_shapes" 2>&1 | tail -15
