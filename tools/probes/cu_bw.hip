// Per-CU streaming bandwidth vs number of active CUs (round-2 probe): does a lone CU read / write HBM faster than
// 1/256 of the chip?  One 512-thread workgroup per CU, each streams its own 32 MiB slice with 16-byte accesses,
// 16 in flight per lane.  hipcc --offload-arch=gfx950 -O3 cu_bw.hip -o cu_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(512) void rd(const uint4* p, size_t per_wg, uint4* sink) {
  const uint4* b = p + (size_t)blockIdx.x * per_wg;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = threadIdx.x; i + 15 * 512 < per_wg; i += 16 * 512) {
    uint4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = b[i + k * 512];
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
  }
  if (acc.x == 0x12345678) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(512) void wr(uint4* p, size_t per_wg) {
  uint4* b = p + (size_t)blockIdx.x * per_wg;
  const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
  for (size_t i = threadIdx.x; i + 15 * 512 < per_wg; i += 16 * 512) {
#pragma unroll
    for (int k = 0; k < 16; ++k) b[i + k * 512] = v;
  }
}
int main() {
  const size_t per_wg = (32u << 20) / 16;   // 32 MiB per workgroup, in uint4
  uint4 *buf, *sink;
  hipMalloc(&buf, 256 * per_wg * 16); hipMalloc(&sink, 4096);
  hipMemset(buf, 1, 256 * per_wg * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int g : {1, 8, 32, 64, 128, 256}) {
      float best = 1e9;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(rd, dim3(g), dim3(512), 0, 0, buf, per_wg, sink);
        else hipLaunchKernelGGL(wr, dim3(g), dim3(512), 0, 0, buf, per_wg);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double gb = (double)g * per_wg * 16 / 1e9;
      printf("%s %3d workgroups: %.3f ms, %.1f GB/s per CU, %.2f TB/s total (%.1f B/clk/CU at 2.1 GHz)\n", mode ? "write" : "read ", g, best,
             gb / g / (best * 1e-3), gb / (best * 1e-3) / 1e3, gb / g / (best * 1e-3) / 2.1);
    }
  return 0;
}
