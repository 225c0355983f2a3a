cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4coop
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > gpurun_out/r4coop/ops.log 2>&1; rc=$?; echo "ops rc $rc"; tail -3 gpurun_out/r4coop/ops.log
[ $rc -eq 0 ] || exit 1
for lib in gava_clip_amd/libgava_hip_nocoop.so "" gava_clip_amd/libgava_hip_nocoop.so ""; do echo "== attn c5 lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 200 python tools/kernel_bench.py attn --cfg VIT_L14_T32 --B 32 --iters 20 2>&1 | tail -1; done
