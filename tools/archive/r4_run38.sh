cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4pp3
export L=gava_clip_amd/libgava_hip_base.so
S=$(date +%s); timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4pp3/all.log 2>&1; echo "all rc $? $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r4pp3/all.log
for lib in $L "" $L ""; do echo "== train lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 200 python tools/train_bench.py --B 64 --iters 5 2>&1 | tail -1; done
for lib in $L ""; do echo "== wlo lib=$lib"; GAVA_HIP_LIB=$lib timeout -k 10 300 python tools/wlo_modes.py --modes fp16+wlo,fp16+wlo8 2>&1 | tail -3; done
