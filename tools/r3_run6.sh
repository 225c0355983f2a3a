mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r3/gpu_tests_3.log 2>&1; tail -4 gpurun_out/r3/gpu_tests_3.log
timeout -k 10 500 python bench.py > gpurun_out/r3/bench_c2_a.json 2> gpurun_out/r3/bench_c2_a.log; tail -3 gpurun_out/r3/bench_c2_a.log; head -c 1500 gpurun_out/r3/bench_c2_a.json
