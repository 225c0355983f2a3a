mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "attention_f32 or c1_other or c1_vit or text or c3_full or c5_full or boundary or tiny_forward" > gpurun_out/r3/gpu_tests_2.log 2>&1; tail -5 gpurun_out/r3/gpu_tests_2.log; grep "c1_b16_s\|c3 full\|c5 full\|\[c1/" gpurun_out/r3/gpu_tests_2.log
