cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4/gpu_tests_b.log
timeout -k 10 600 python tools/ab_env.py "nst2:GAVA_SMALL_NST=2" "nst4:GAVA_SMALL_NST=4" --rounds 2 --config c5 2>&1 | grep "==\|FAILED"
timeout -k 10 600 python tools/ab_env.py "nst2:GAVA_SMALL_NST=2" "nst4:GAVA_SMALL_NST=4" --rounds 2 --config c3 2>&1 | grep "==\|FAILED"
echo "nst4 train: $(timeout -k 10 200 python tools/train_bench.py --iters 3 2>&1 | tail -1)"
echo "nst2 train: $(GAVA_SMALL_NST=2 timeout -k 10 200 python tools/train_bench.py --iters 3 2>&1 | tail -1)"
