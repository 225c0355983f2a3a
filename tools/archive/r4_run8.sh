cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
for lib in "" noglds noreads noboth; do
  L=gava_clip_amd/libgava_hip.so; [ -n "$lib" ] && L=gava_clip_amd/libgava_hip_$lib.so
  echo "=== ${lib:-product}"
  GAVA_HIP_LIB=$PWD/$L timeout -k 10 300 python tools/gemm_vs_vendor.py 2>&1 | grep "M=4096\|M=100864 N=3072\|M=100864 N=768 K=3072" | sed 's/max diff.*//'
done > $O/pp_ablate.txt 2>&1
cat $O/pp_ablate.txt
