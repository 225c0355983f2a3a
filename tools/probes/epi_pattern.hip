// Round-3 probe: what a CU's write / read path sustains for the access shapes of the residual GEMMs' epilogue, with 128 or 256
// CUs active.  One 512-thread workgroup per CU; every wave moves 1 KiB per instruction (16 B per lane), rows 3072 B apart
// (fp32 residual stream, D = 768):
//   seg64  : 16 rows x 64 contiguous bytes per instruction  (the epilogue's fp32 stores / residual loads: lane (fr, fg))
//   seg128 :  8 rows x 128 contiguous bytes per instruction (whole cache lines)
//   contig : 1 KiB contiguous
// modes: w = stores only, r = loads only, m = 6 stores + 4 loads interleaved (the epilogue's mix), 80 operations per wave
// and "tile", no wait inside a tile, vmcnt(0) between tiles.
// hipcc --offload-arch=gfx950 -O3 epi_pattern.hip -o epi_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr long PITCH = 3072;           // bytes between rows
template <int SHAPE, int MODE>
__global__ __launch_bounds__(512) void k(char* base, long wg_bytes, int tiles, float4* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* b = base + (size_t)blockIdx.x * wg_bytes;
  // lane -> (row, byte offset in the row) of one 1 KiB instruction
  long off;
  int rows_per_inst;
  if (SHAPE == 0) { off = (long)(lane & 15) * PITCH + (lane >> 4) * 16; rows_per_inst = 16; }        // 16 rows x 64 B
  else if (SHAPE == 1) { off = (long)(lane >> 3) * PITCH + (lane & 7) * 16; rows_per_inst = 8; }     // 8 rows x 128 B
  else { off = (long)lane * 16; rows_per_inst = 0; }
  float4 acc = make_float4(0, 0, 0, 0);
  const float4 v = make_float4(lane, 1, 2, 3);
  for (int t = 0; t < tiles; ++t) {
    // a wave's region of the tile: 32 KiB of stores (+16 KiB) and 32 KiB of loads, in 1 KiB instructions
    char* tb = b + ((size_t)t * 8 + wave) * (SHAPE == 2 ? 80 * 1024 : 80 * 1024);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int op = g * 10 + i;
        char* p = SHAPE == 2 ? tb + op * 1024 + off
                             : tb + (long)(op / 2) * rows_per_inst * 0 + (long)op * (SHAPE == 0 ? 64 : 128) % PITCH + (long)(op * (SHAPE == 0 ? 64 : 128) / PITCH) * rows_per_inst * PITCH + off;
        const bool is_load = MODE == 1 || (MODE == 2 && i >= 6);
        if (is_load) { const float4 x = *reinterpret_cast<const float4*>(p); acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w; }
        else *reinterpret_cast<float4*>(p) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc.x == 12345.678f) sink[blockIdx.x] = acc;
}
template <int SHAPE, int MODE> float run(char* buf, long wg_bytes, int tiles, int g, float4* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(g), dim3(512), 0, 0, buf, wg_bytes, tiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}
int main() {
  const int tiles = 24;
  const long wg_bytes = (long)tiles * 8 * 80 * 1024 + 64 * PITCH;   // per workgroup (~15 MiB)
  char* buf; float4* sink;
  hipMalloc(&buf, 256 * wg_bytes); hipMalloc(&sink, 4096 * 16);
  hipMemset(buf, 1, 256 * wg_bytes);
  const char* sn[3] = {"seg64 ", "seg128", "contig"};
  const char* mn[3] = {"stores", "loads ", "mix6:4"};
  for (int g : {128, 256})
    for (int s = 0; s < 3; ++s)
      for (int m = 0; m < 3; ++m) {
        float ms = 0;
#define R(S, M) if (s == S && m == M) ms = run<S, M>(buf, wg_bytes, tiles, g, sink)
        R(0, 0); R(0, 1); R(0, 2); R(1, 0); R(1, 1); R(1, 2); R(2, 0); R(2, 1); R(2, 2);
        const double gb = (double)g * tiles * 8 * 80 * 1024 / 1e9;
        printf("%3d CUs %s %s: %.3f ms, %.1f GB/s per CU (%.1f B/clk at 2.1 GHz), %.2f TB/s total, %.1f us per 640 KiB tile\n", g, sn[s], mn[m], ms,
               gb / g / (ms * 1e-3), gb / g / (ms * 1e-3) / 2.1, gb / (ms * 1e-3) / 1e3, ms * 1e3 / tiles);
      }
  return 0;
}
