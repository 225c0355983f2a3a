// backward.hip — gradient kernels for the trainable subset (training/train.py:441-490 calls
// loss.backward() through VitaCLIP.forward; every transformer weight is frozen, VitaCLIP_model.py:230-239, so the
// path needs dgrad only: gradients flow THROUGH the frozen GEMMs to the prompt parameters).
//   * dgrad GEMMs are gava_gemm on transposed weight copies (dX = dY . W  ==  gemm(A = dY, W = W^T [in][out])).
//   * this file: LayerNorm backward, QuickGELU backward and the entry point of the attention backward (kernels in
//     attention_bwd.hip).
// Gradient operands are bf16 (fp32 range: no loss scaling needed inside the library), accumulation fp32.
#include "common.h"
#include "internal.h"

namespace {

constexpr int MAXV = 4;  // float4 per lane: D <= 1024

// ---------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row.  y = xhat * gamma + beta, xhat = (x - mean) * rstd (eps 1e-5):
//   g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dgamma += dy * xhat;  dbeta += dy
struct LnBwdParams {
  const float* x; long x_stride; const int* x_idx;      // forward input rows (optionally gathered)
  const float* gamma;
  const float* dy; long dy_stride;
  float* dx; long dx_stride; const int* dx_idx;         // optionally scattered
  float* dgamma; float* dbeta;                          // optional, fp32 [D], accumulated with atomics
  unsigned short* dx16; long dx16_stride;               // optional h16 copy of the (accumulated) dx: next dgrad's operand
  int rows, D, accumulate;
};

template <class P>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const LnBwdParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* xr = p.x + (p.x_idx ? (long)p.x_idx[row] : (long)row) * p.x_stride;
  const float* dyr = p.dy + (long)row * p.dy_stride;
  float4 v[MAXV], g[MAXV], d[MAXV];
  bool act[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    act[i] = c < p.D;
    if (act[i]) {
      v[i] = *reinterpret_cast<const float4*>(xr + c);
      d[i] = *reinterpret_cast<const float4*>(dyr + c);
      g[i] = *reinterpret_cast<const float4*>(p.gamma + c);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float inv = 1.f / (float)p.D;
  const float mean = wave_sum(s) * inv;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * inv + 1e-5f);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;   // xhat
      if (p.dgamma) {
        const int c = (lane + 64 * i) * 4;
        atomicAdd(p.dgamma + c, d[i].x * v[i].x); atomicAdd(p.dgamma + c + 1, d[i].y * v[i].y);
        atomicAdd(p.dgamma + c + 2, d[i].z * v[i].z); atomicAdd(p.dgamma + c + 3, d[i].w * v[i].w);
        atomicAdd(p.dbeta + c, d[i].x); atomicAdd(p.dbeta + c + 1, d[i].y);
        atomicAdd(p.dbeta + c + 2, d[i].z); atomicAdd(p.dbeta + c + 3, d[i].w);
      }
      g[i].x *= d[i].x; g[i].y *= d[i].y; g[i].z *= d[i].z; g[i].w *= d[i].w;   // g = dy * gamma
      sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
  const float mg = wave_sum(sg) * inv, mgx = wave_sum(sgx) * inv;
  float* dxr = p.dx + (p.dx_idx ? (long)p.dx_idx[row] : (long)row) * p.dx_stride;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) {
      const int c = (lane + 64 * i) * 4;
      float4 o;
      o.x = rstd * (g[i].x - mg - v[i].x * mgx); o.y = rstd * (g[i].y - mg - v[i].y * mgx);
      o.z = rstd * (g[i].z - mg - v[i].z * mgx); o.w = rstd * (g[i].w - mg - v[i].w * mgx);
      if (p.accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dxr + c);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(dxr + c) = o;
      if (p.dx16) *reinterpret_cast<uint2*>(p.dx16 + (long)row * p.dx16_stride + c) = pack4<P>(o.x, o.y, o.z, o.w);
    }
}

// ---------------------------------------------------------------------------------------------
// QuickGELU backward: y = x * s, s = sigmoid(1.702 x)  =>  dy/dx = s * (1 + 1.702 x (1 - s))
template <class P>
__global__ __launch_bounds__(256) void qgelu_bwd_kernel(const unsigned short* pre, const unsigned short* dh,
                                                         unsigned short* dpre, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint2 a = reinterpret_cast<const uint2*>(pre)[i];
    const uint2 b = reinterpret_cast<const uint2*>(dh)[i];
    float x[4] = {P::up((unsigned short)a.x), P::up((unsigned short)(a.x >> 16)), P::up((unsigned short)a.y), P::up((unsigned short)(a.y >> 16))};
    float d[4] = {P::up((unsigned short)b.x), P::up((unsigned short)(b.x >> 16)), P::up((unsigned short)b.y), P::up((unsigned short)(b.y >> 16))};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = 1.f / (1.f + __expf(-1.702f * x[e]));
      d[e] *= s * (1.f + 1.702f * x[e] * (1.f - s));
    }
    reinterpret_cast<uint2*>(dpre)[i] = pack4<P>(d[0], d[1], d[2], d[3]);
  }
}

}  // namespace

extern "C" int gava_layernorm_backward(const gava_layernorm_bwd_args* a, gava_stream_t stream) {
  if (!a || !a->x || !a->gamma || !a->dy || !a->dx) return GAVA_EINVAL;
  if (a->rows <= 0 || a->D <= 0 || a->D % 4 || a->D > MAXV * 256) return GAVA_EINVAL;
  if ((a->dgamma == nullptr) != (a->dbeta == nullptr)) return GAVA_EINVAL;
  if (a->x_stride % 4 || a->dy_stride % 4 || a->dx_stride % 4) return GAVA_EINVAL;
  if (a->dx16 && (a->dx_row_index || a->dx16_stride % 4 || (a->prec != GAVA_PREC_F16 && a->prec != GAVA_PREC_BF16))) return GAVA_EINVAL;
  LnBwdParams p{a->x, a->x_stride, a->x_row_index, a->gamma, a->dy, a->dy_stride, a->dx, a->dx_stride,
                a->dx_row_index, a->dgamma, a->dbeta, (unsigned short*)a->dx16, a->dx16_stride, a->rows, a->D, a->accumulate};
  dim3 grid((a->rows + 3) / 4), block(256);
  if (a->dx16 && a->prec == GAVA_PREC_F16) hipLaunchKernelGGL(layernorm_bwd_kernel<PrecF16>, grid, block, 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<PrecBF16>, grid, block, 0, (hipStream_t)stream, p);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

extern "C" int gava_qgelu_backward(const void* pre, const void* dh, void* dpre, size_t n, int prec, gava_stream_t stream) {
  if (!pre || !dh || !dpre || n % 4) return GAVA_EINVAL;
  if (n == 0) return GAVA_OK;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  if (prec == GAVA_PREC_F16)
    hipLaunchKernelGGL(qgelu_bwd_kernel<PrecF16>, dim3(blocks), dim3(256), 0, s, (const unsigned short*)pre, (const unsigned short*)dh, (unsigned short*)dpre, n4);
  else if (prec == GAVA_PREC_BF16)
    hipLaunchKernelGGL(qgelu_bwd_kernel<PrecBF16>, dim3(blocks), dim3(256), 0, s, (const unsigned short*)pre, (const unsigned short*)dh, (unsigned short*)dpre, n4);
  else return GAVA_EINVAL;
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

extern "C" int gava_attention_backward(const gava_attention_bwd_args* a, gava_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->dout || !a->dq || !a->dk || !a->dv) return GAVA_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n <= 0) return GAVA_EINVAL;
  if (a->prec != GAVA_PREC_F16 && a->prec != GAVA_PREC_BF16) return GAVA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int n_side = a->side_k ? a->n_g + a->T + (a->has_summary ? 1 : 0) : 0;
  // n main rows per sequence / frame, the first n_q of them query (0 = all), prompt rows (vision blocks) gathered as in
  // gava_attention; causal only without prompt rows (text tower): MFMA kernels (attention_bwd.hip)
  if (!a->workspace || (a->causal && (n_side || a->n_q))) return GAVA_EINVAL;
  if (n_side && (!a->side_v || !a->dside_k || !a->dside_v || a->n_g < 0 || a->T <= 0 || a->batch % a->T)) return GAVA_EINVAL;
  if ((a->ld_qkv | a->ld_side | a->ld_dout) % 8) return GAVA_EINVAL;
  if ((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->dout | (uintptr_t)a->side_k | (uintptr_t)a->side_v |
        (uintptr_t)a->dside_k | (uintptr_t)a->dside_v) & 15) || (a->ld_dqkv % 4) || (a->ld_dside % 4)) return GAVA_EINVAL;
  if ((((uintptr_t)a->dq | (uintptr_t)a->dk | (uintptr_t)a->dv) & 7)) return GAVA_EINVAL;
  gava::AttnBwdMfmaParams p;
  p.q = (const unsigned short*)a->q; p.k = (const unsigned short*)a->k; p.v = (const unsigned short*)a->v; p.ld_qkv = a->ld_qkv;
  p.sk = (const unsigned short*)a->side_k; p.sv = (const unsigned short*)a->side_v; p.ld_side = a->ld_side;
  p.dout = (const unsigned short*)a->dout; p.ld_dout = a->ld_dout;
  p.dq = (unsigned short*)a->dq; p.dk = (unsigned short*)a->dk; p.dv = (unsigned short*)a->dv; p.ld_dqkv = a->ld_dqkv;
  // separate query-side buffers (the CLS-only last block): q_batch_rows rows per frame, own strides
  p.q_rows = a->q_batch_rows ? a->q_batch_rows : a->n;
  p.ld_q = a->q_batch_rows ? a->ld_q : a->ld_qkv;
  p.ld_dq = a->q_batch_rows ? a->ld_dq : a->ld_dqkv;
  if (a->q_batch_rows && (a->ld_q % 8 || a->ld_dq % 4 || a->q_batch_rows < (a->n_q ? a->n_q : a->n))) return GAVA_EINVAL;
  p.dsk = a->dside_k; p.dsv = a->dside_v; p.ld_dside = a->ld_dside;
  p.stats = (float*)a->workspace;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q ? a->n_q : a->n; p.n_kmain = a->n;
  p.n_g = a->n_g; p.T = n_side ? a->T : 1; p.has_summary = a->has_summary; p.n_keys = a->n + n_side;
  p.q_pad = ((p.n_q + 15) / 16 + 1) / 2 * 32;
  p.q_scale = a->q_scale;
  if (p.n_keys > 320 || p.n_q > a->n || p.n_q > 288) return GAVA_EINVAL;
  return gava::attention_bwd_mfma(p, a->prec, a->act_prec_set ? a->act_prec : a->prec, a->causal, s);
}

extern "C" size_t gava_attention_backward_workspace_bytes(int batch, int heads, int n_q) {
  if (batch <= 0 || heads <= 0 || n_q <= 0) return 0;
  return (size_t)batch * heads * (((n_q + 15) / 16 + 1) / 2 * 32) * 2 * sizeof(float);
}
