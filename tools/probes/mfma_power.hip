// Pure-MFMA probe: sustained rate of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 with the board at its power cap
// (no LDS, no memory in the loop; operands rotate over 4 register sets filled with random fp16 data).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(const f16x8* src, float* out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 1023]; b[i] = src[(threadIdx.x + 64 * i + 333) & 1023]; }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[u], c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
  } else {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[u], c[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 16;
  const double secs = argc > 2 ? atof(argv[2]) : 2.0;
  std::vector<_Float16> h(1024 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  f16x8* src; float* out;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 2 * 512 * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 20000;
  const double flops_per_launch = shape == 16 ? 256.0 * 2 * 8 * iters * 32 * (2.0 * 16 * 16 * 32) : 256.0 * 2 * 8 * iters * 16 * (2.0 * 32 * 32 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {   // warm-up
    if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(512), dim3(512), 0, 0, src, out, iters);
    else hipLaunchKernelGGL(k<32>, dim3(512), dim3(512), 0, 0, src, out, iters);
  }
  hipDeviceSynchronize();
  double total = 0; int n = 0;
  while (total < secs * 1e3) {
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) {
      if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(512), dim3(512), 0, 0, src, out, iters);
      else hipLaunchKernelGGL(k<32>, dim3(512), dim3(512), 0, 0, src, out, iters);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total += ms; n += 10;
    printf("shape %dx%d: %.1f TFLOP/s (%.3f ms per launch)\n", shape, shape, flops_per_launch * 10 / ms / 1e9, ms / 10);
    fflush(stdout);
  }
  return 0;
}
