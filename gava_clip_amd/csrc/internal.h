// internal.h — launchers shared between translation units of libgava_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
namespace gava {
int side_ln(const float* gp, const float* lp, const float* cp, const float* summ, const float* gamma,
            const float* beta, void* out, int G, int T, int BT, int D, int prec, hipStream_t s);
int patch_gather(const float* x, void* out, int B, int T, int S, int P, int Kp, int prec, hipStream_t s);
int cls_embed(float* X, const float* cls, const float* pos, const float* time, int BT, int T, int D,
              long frame_stride, hipStream_t s);
int mean_rows(const float* in, float* out, int B, int T, int D, hipStream_t s);
int copy_rows(const float* in, long in_stride, float* out, int rows, int D, hipStream_t s);
int text_embed(const float* emb, const float* pos, const float* ctx, const int* tok, float* X,
               int n_prompts, int L, int W, int n_ctx, hipStream_t s);
unsigned long long* debug_buffer();
// CUs the persistent GEMM leaves free on its next launches (so a concurrent stream can run small kernels)
void set_gemm_cu_reserve(int n);
int gemm_cu_reserve();   // set by gava_debug_set_buffer; nullptr = stamps off
}  // namespace gava
