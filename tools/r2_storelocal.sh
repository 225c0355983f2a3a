for r in 1 2; do
for lib in new ntstore; do
  L=gava_clip_amd/libgava_hip_$lib.so; [ $lib = new ] && L=gava_clip_amd/libgava_hip.so
  for k in fc1part qkvpart; do
  echo "== $lib $k $(GAVA_HIP_LIB=$L python tools/kernel_bench.py $k --iters 30 2>/dev/null | tail -1)"
  done
done; done
