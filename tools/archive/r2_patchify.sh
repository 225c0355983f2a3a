mkdir -p gpurun_out/r2
python -m pytest tests/test_preprocess.py tests/test_gpu_forward.py -m gpu -q -x > gpurun_out/r2/patchify_tests.log 2>&1 || { tail -30 gpurun_out/r2/patchify_tests.log; exit 1; }
tail -1 gpurun_out/r2/patchify_tests.log
for k in patch patchu8 patchify patchifyu8 patchA; do python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1; done
python tools/ab_env.py "direct:GAVA_PATCH_DIRECT=1" "twopass:GAVA_PATCH_DIRECT=0" --rounds 3 --config c2 2>&1 | tail -2
