cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4pp2
export L=gava_clip_amd/libgava_hip_pp2.so
timeout -k 10 700 python tools/ab_env.py "pp4:" "pp2prod:GAVA_HIP_LIB=$L" "pp2all:GAVA_HIP_LIB=$L,GAVA_PP=1" --rounds 2 --config c5 > gpurun_out/r4pp2/ab_c5b.log 2>&1; tail -4 gpurun_out/r4pp2/ab_c5b.log
for v in "GAVA_PP=2" "GAVA_PP=1" "GAVA_PP=2" "GAVA_PP=1"; do echo "== train $v"; env $v GAVA_HIP_LIB=$L timeout -k 10 200 python tools/train_bench.py --B 64 --iters 5 2>&1 | tail -1; done
timeout -k 10 400 python tools/ab_env.py "pp2prod:GAVA_HIP_LIB=$L" "pp2all:GAVA_HIP_LIB=$L,GAVA_PP=1" --rounds 4 > gpurun_out/r4pp2/ab3.log 2>&1; tail -3 gpurun_out/r4pp2/ab3.log
