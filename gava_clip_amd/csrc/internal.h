// internal.h — launchers shared between translation units of libgava_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gava_hip.h"
namespace gava {
int side_ln(const float* gp, const float* lp, const float* cp, const float* summ, const float* gamma,
            const float* beta, void* out, int G, int T, int BT, int D, int prec, hipStream_t s);
int cls_embed(float* X, const float* cls, const float* pos, const float* time, int BT, int T, int D,
              long frame_stride, hipStream_t s);
int mean_rows(const float* in, float* out, int B, int T, int D, hipStream_t s);
int copy_rows(const float* in, long in_stride, float* out, int rows, int D, hipStream_t s);
// fp32 rows = hi + lo of the residual stream's 16-bit pair (rows in_stride elements apart)
int join_rows(const void* hi, const void* lo, long in_stride, float* out, long out_stride, int rows, int D, int prec, hipStream_t s);
// gemm.hip: does gava_gemm take an EPI_F32 GEMM of this shape with the residual as a 16-bit pair (gava_gemm_args.resid16)?
bool gemm_takes_pair(int M, int N, int K, long lda, long ldw);
int text_embed(const float* emb, const float* pos, const float* ctx, const int* tok, float* X,
               int n_prompts, int L, int W, int n_ctx, hipStream_t s);
unsigned long long* debug_buffer();   // set by gava_debug_set_buffer; nullptr = stamps off

// MFMA attention backward (attention_bwd.hip), launched by gava_attention_backward (backward.hip)
struct AttnBwdMfmaParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; long ld_qkv;
  long ld_q, ld_dq; int q_rows;  // query-side layout: rows per frame in q / dout / dq and their strides
  const unsigned short* sk; const unsigned short* sv; long ld_side;
  const unsigned short* dout; long ld_dout;
  unsigned short* dq; unsigned short* dk; unsigned short* dv; long ld_dqkv;
  float* dsk; float* dsv; long ld_dside;
  float* stats;                 // [batch*heads][q_pad][2]: log2-sum-exp and delta per query
  int batch, heads, n_q, n_kmain, n_g, T, has_summary, n_keys, q_pad;
  float q_scale;
};
int attention_bwd_mfma(const AttnBwdMfmaParams& p, int prec, int act_prec, int causal, hipStream_t s);
}  // namespace gava
