cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
for r in 1 2; do
echo "direct (default):       $(timeout -k 10 100 python tools/kernel_bench.py patch --iters 30 2>/dev/null | tail -1)"
echo "patchify pass:          $(timeout -k 10 100 python tools/kernel_bench.py patchify --iters 30 2>/dev/null | tail -1)"
echo "two-pass GEMM on 128^2: $(timeout -k 10 100 python tools/kernel_bench.py patchA --iters 30 2>/dev/null | tail -1)"
echo "two-pass GEMM on 256^2: $(GAVA_PATCH_256=1 timeout -k 10 100 python tools/kernel_bench.py patchA --iters 30 2>/dev/null | tail -1)"
done
timeout -k 10 600 python tools/ab_env.py "direct:" "twopass128:GAVA_PATCH_DIRECT=0" "twopass256:GAVA_PATCH_DIRECT=0 GAVA_PATCH_256=1" --rounds 3 2>&1 | grep "=="
