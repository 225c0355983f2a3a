mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_backward.py -q -x > gpurun_out/r2/bwd_tests.log 2>&1 || { tail -30 gpurun_out/r2/bwd_tests.log; exit 1; }
tail -2 gpurun_out/r2/bwd_tests.log
for r in 1 2; do
echo "fold:   $(python tools/train_bench.py --B 64 2>/dev/null | grep 'step 3')"
echo "nofold: $(GAVA_TRAIN_FOLD=0 python tools/train_bench.py --B 64 2>/dev/null | grep 'step 3')"
done
