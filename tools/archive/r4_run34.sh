cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4pp2
export L=gava_clip_amd/libgava_hip_pp2.so
for lib in "" $L "" $L; do echo "== train lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 200 python tools/train_bench.py --B 64 --iters 5 2>&1 | tail -2; done
timeout -k 10 500 python tools/ab_env.py "pp4:" "pp2:GAVA_HIP_LIB=$L" --rounds 2 --config c5 > gpurun_out/r4pp2/ab_c5.log 2>&1; tail -3 gpurun_out/r4pp2/ab_c5.log
for lib in "" $L; do echo "== vendor table lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_vs_vendor.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
