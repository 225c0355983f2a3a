#!/bin/bash
# Build an experiment variant of the library next to the product one (same-box A/B timing):
#   tools/ab_build.sh NAME -DGAVA_V3_P0=4 ...   ->  gava_clip_amd/libgava_hip_NAME.so
# use with  GAVA_HIP_LIB=gava_clip_amd/libgava_hip_NAME.so python tools/kernel_bench.py fc1
set -e
cd "$(dirname "$0")/.."
name=$1; shift
abi=$(python -c "from gava_clip_amd.build import abi_hash; print(abi_hash())")
mkdir -p gava_clip_amd/build/$name
rm -f gava_clip_amd/build/$name/*.o
for f in gemm attention rowops forward preprocess backward attention_bwd; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGAVA_ABI_HASH=$abi "$@" -c gava_clip_amd/csrc/$f.hip -o gava_clip_amd/build/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gava_clip_amd/libgava_hip_$name.so gava_clip_amd/build/$name/*.o
echo gava_clip_amd/libgava_hip_$name.so
