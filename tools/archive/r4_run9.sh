cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests_a.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gpu_tests_a.log
echo "pp2 train: $(timeout -k 10 200 python tools/train_bench.py --iters 3 2>&1 | tail -1)"
echo "pp0 train: $(GAVA_PP=0 timeout -k 10 200 python tools/train_bench.py --iters 3 2>&1 | tail -1)"
