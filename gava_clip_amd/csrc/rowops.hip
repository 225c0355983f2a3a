// rowops.hip — HBM-bound row kernels: LayerNorm, conversions, token/prompt assembly, pooling,
// similarity head.  One 64-lane wave owns one row (float4 per lane, fully coalesced 1 KiB per
// wave instruction); statistics are reduced with cross-lane shuffles, no LDS.
#include <cstdlib>
#include "common.h"
#include "internal.h"

namespace {

constexpr int MAXV = 4;  // float4 per lane: D <= 1024

template <class P>
static __device__ __forceinline__ void store_h16x4(unsigned short* dst, float4 v) {
  *reinterpret_cast<uint2*>(dst) = pack4<P>(v.x, v.y, v.z, v.w);
}

// normalise a row held as v[0..nv) float4 per lane; eps = 1e-5, two-pass statistics
static __device__ __forceinline__ void ln_row(float4 (&v)[MAXV], int nv_lane_count, int D,
                                              const bool (&act)[MAXV]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) { v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd; }
  (void)nv_lane_count;
}

struct LnParams {
  const float* in; long in_stride; const int* idx;
  const float* gamma; const float* beta;
  unsigned short* out16; long o16_stride;
  float* out32; long o32_stride;
  int rows, D, split;
  const float* gamma2; const float* beta2;
  unsigned short* out_hi; unsigned short* out_lo; long ohl_stride;   // 16-bit pair of what out32 receives
};

template <class P>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const long src_row = p.idx ? (long)p.idx[row] : (long)row;
  const float* src = p.in + src_row * p.in_stride;
  float4 v[MAXV];
  bool act[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    act[i] = c < p.D;
    if (act[i]) v[i] = *reinterpret_cast<const float4*>(src + c);
  }
  if (p.gamma) {
    ln_row(v, 0, p.D, act);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (act[i]) {
        const int c = (lane + 64 * i) * 4;
        const float4 g = *reinterpret_cast<const float4*>(p.gamma + c);
        const float4 b = *reinterpret_cast<const float4*>(p.beta + c);
        v[i].x = v[i].x * g.x + b.x; v[i].y = v[i].y * g.y + b.y;
        v[i].z = v[i].z * g.z + b.z; v[i].w = v[i].w * g.w + b.w;
      }
  }
  if (p.gamma2) {
    // second LayerNorm on the first one's result: out32 gets the first, out16 the second
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (act[i]) {
        const int c = (lane + 64 * i) * 4;
        if (p.out32) *reinterpret_cast<float4*>(p.out32 + (long)row * p.o32_stride + c) = v[i];
        if (p.out_hi) {
          uint2 hi, lo;
          split4_lo16<P>(v[i].x, v[i].y, v[i].z, v[i].w, hi, lo);
          *reinterpret_cast<uint2*>(p.out_hi + (long)row * p.ohl_stride + c) = hi;
          *reinterpret_cast<uint2*>(p.out_lo + (long)row * p.ohl_stride + c) = lo;
        }
      }
    ln_row(v, 0, p.D, act);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (act[i]) {
        const int c = (lane + 64 * i) * 4;
        const float4 g = *reinterpret_cast<const float4*>(p.gamma2 + c);
        const float4 b = *reinterpret_cast<const float4*>(p.beta2 + c);
        v[i].x = v[i].x * g.x + b.x; v[i].y = v[i].y * g.y + b.y;
        v[i].z = v[i].z * g.z + b.z; v[i].w = v[i].w * g.w + b.w;
      }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) {
      const int c = (lane + 64 * i) * 4;
      if (p.out16) {
        unsigned short* o = p.out16 + (long)row * p.o16_stride + c;
        if (p.split) {
          uint2 hi, lo;
          split4<P>(v[i].x, v[i].y, v[i].z, v[i].w, hi, lo);
          *reinterpret_cast<uint2*>(o) = hi;
          *reinterpret_cast<uint2*>(o + p.D) = lo;
          *reinterpret_cast<uint2*>(o + 2 * p.D) = hi;
        } else {
          store_h16x4<P>(o, v[i]);
        }
      }
      if (p.out32 && !p.gamma2) *reinterpret_cast<float4*>(p.out32 + (long)row * p.o32_stride + c) = v[i];
      if (p.out_hi && !p.gamma2) {
        uint2 hi, lo;
        split4_lo16<P>(v[i].x, v[i].y, v[i].z, v[i].w, hi, lo);
        *reinterpret_cast<uint2*>(p.out_hi + (long)row * p.ohl_stride + c) = hi;
        *reinterpret_cast<uint2*>(p.out_lo + (long)row * p.ohl_stride + c) = lo;
      }
    }
}

// ---- vision "side" rows: assemble [global prompts | local prompts + cls_proj | summary] and
// apply norm1 -> h16 (the K/V-only tokens of VitaCLIP_vision_encoder_utils.py:171-190).
struct SideParams {
  const float* gp;      // [G][D]
  const float* lp;      // [T][D]
  const float* cp;      // [BT][D] cls_proj output
  const float* summ;    // [BT][D] summary token
  const float* gamma; const float* beta;
  unsigned short* out;  // [G + 2*BT][D]
  int G, T, BT, D;
};

template <class P>
__global__ __launch_bounds__(256) void side_ln_kernel(const SideParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.G + 2 * p.BT) return;
  float4 v[MAXV];
  bool act[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    act[i] = c < p.D;
    if (!act[i]) continue;
    if (row < p.G) {
      v[i] = *reinterpret_cast<const float4*>(p.gp + (long)row * p.D + c);
    } else if (row < p.G + p.BT) {
      const int f = row - p.G;
      const float4 a = *reinterpret_cast<const float4*>(p.lp + (long)(f % p.T) * p.D + c);
      const float4 b = *reinterpret_cast<const float4*>(p.cp + (long)f * p.D + c);
      v[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    } else {
      v[i] = *reinterpret_cast<const float4*>(p.summ + (long)(row - p.G - p.BT) * p.D + c);
    }
  }
  ln_row(v, 0, p.D, act);
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (act[i]) {
      const int c = (lane + 64 * i) * 4;
      const float4 g = *reinterpret_cast<const float4*>(p.gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(p.beta + c);
      store_h16x4<P>(p.out + (long)row * p.D + c,
                     make_float4(v[i].x * g.x + b.x, v[i].y * g.y + b.y, v[i].z * g.z + b.z, v[i].w * g.w + b.w));
    }
}

// cls rows of the embedding: X[frame*(n+1)] = cls_token + pos[0] + time[frame % T]
__global__ void cls_embed_kernel(float* X, const float* cls, const float* pos, const float* time,
                                 int BT, int T, int D, long frame_stride) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)BT * D) return;
  const int c = (int)(id % D);
  const long f = id / D;
  X[f * frame_stride + c] = cls[c] + pos[c] + time[(f % T) * D + c];
}

// out[b][c] = mean_t in[(b*T+t)][c]
__global__ void mean_rows_kernel(const float* in, float* out, int B, int T, int D) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * D) return;
  const int c = (int)(id % D);
  const long b = id / D;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += in[(b * T + t) * D + c];
  out[id] = s / (float)T;
}

// fp32 rows out of the stream's 16-bit pair (gava_gemm_args.resid16): out = hi + lo; the B*T CLS rows the last block works on, debug taps
template <class P>
__global__ void join_rows_kernel(const unsigned short* hi, const unsigned short* lo, long in_stride, float* out, long out_stride,
                                 int rows, int D) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)rows * D) return;
  const long r = id / D, c = id % D;
  out[r * out_stride + c] = P::up(hi[r * in_stride + c]) + PrecF16::up(lo[r * in_stride + c]);
}

// copy strided rows (debug taps)
__global__ void copy_rows_kernel(const float* in, long in_stride, float* out, int rows, int D) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)rows * D) return;
  out[id] = in[(id / D) * in_stride + id % D];
}

// text prompt assembly + positional embedding (VitaCLIP_text_encoder.py:323-332,155):
// X[n][l] = (l==0 ? emb[tok[n][0]] : l<=n_ctx ? ctx[n][l-1] : emb[tok[n][l]]) + pos[l]
__global__ void text_embed_kernel(const float* emb, const float* pos, const float* ctx, const int* tok,
                                  float* X, int n_prompts, int L, int W, int n_ctx) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)n_prompts * L * W) return;
  const int c = (int)(id % W);
  const long row = id / W;
  const int l = (int)(row % L);
  const long n = row / L;
  float v;
  if (!tok) v = ctx[id];                        // direct mode: ctx holds the whole (n, L, W) prompt embeddings
  else if (l >= 1 && l <= n_ctx) v = ctx[(n * n_ctx + (l - 1)) * W + c];
  else v = emb[(long)tok[n * L + l] * W + c];
  X[id] = v + pos[(long)l * W + c];
}

template <class P>
__global__ void convert_kernel(const float* in, unsigned short* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = P::cvt(in[i]);
}

// ---- similarity head, all fp32 (VitaCLIP_model.py:255,287-293).  One wave per output.
__global__ void l2norm_rows_kernel(const float* in, float* out, int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < E; c += 64) { const float v = in[(long)row * E + c]; s += v * v; }
  const float inv = 1.0f / sqrtf(wave_sum(s));
  for (int c = lane; c < E; c += 64) out[(long)row * E + c] = in[(long)row * E + c] * inv;
}

// logits = exp(logit_scale) * V T^T (+ bias) as ONE fp32 MFMA GEMM: v_mfma_f32_16x16x4_f32 is bit-for-bit an
// fp32 fma chain (exact f32, no reduced-precision path), which the similarity head needs.  One wave per
// 16x16 output tile; lane l feeds A[i=l&15][k=l>>4] = V[b0+i][k0+k] and B[k=l>>4][j=l&15] = T[c0+j][k0+k];
// D: column = lane&15 (class), row = 4*(lane>>4)+r (clip).
__global__ __launch_bounds__(64) void logits_mfma_kernel(const float* vn, const float* tn, const float* logit_scale,
                                                         const float* logit_bias, int B, int C, int E, float* logits) {
  const int lane = threadIdx.x;
  const int tiles_c = (C + 15) / 16;
  const int b0 = (blockIdx.x / tiles_c) * 16, c0 = (blockIdx.x % tiles_c) * 16;
  const int i = lane & 15, kq = lane >> 4;
  const int vb = b0 + i < B ? b0 + i : B - 1, tc = c0 + i < C ? c0 + i : C - 1;
  const float* vp = vn + (long)vb * E + kq;
  const float* tp = tn + (long)tc * E + kq;
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < E; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[k0], tp[k0], acc, 0, 0, 0);
  const float ls = expf(logit_scale[0]);
  const float lb = logit_bias ? logit_bias[0] : 0.f;
  const int c = c0 + i;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    if (b < B && c < C) logits[(long)b * C + c] = ls * acc[r] + lb;
  }
}

// m[c] = mean_k normalise(t[c][k]): the class mean of the unit prompt features.  By linearity
// mean_k <v, normalise(t_ck)> = <v, m_c>, so the logits need one GEMM against m (VitaCLIP_model.py:288-289).
__global__ void class_mean_kernel(const float* t, float* m, int C, int n_kv, int E) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  for (int e = lane; e < E; e += 64) m[(long)c * E + e] = 0.f;
  for (int k = 0; k < n_kv; ++k) {
    const float* r = t + ((long)c * n_kv + k) * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += r[e] * r[e];
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int e = lane; e < E; e += 64) m[(long)c * E + e] += r[e] * inv;
  }
  for (int e = lane; e < E; e += 64) m[(long)c * E + e] /= (float)n_kv;
}

// text_features[c] = normalise(mean_k tn[c][k])
__global__ void text_feature_kernel(const float* tn, float* tf, int C, int n_kv, int E) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float s = 0.f;
  for (int e = lane; e < E; e += 64) {
    float m = 0.f;
    for (int k = 0; k < n_kv; ++k) m += tn[((long)c * n_kv + k) * E + e];
    m /= (float)n_kv;
    s += m * m;
  }
  const float inv = 1.0f / sqrtf(wave_sum(s));
  for (int e = lane; e < E; e += 64) {
    float m = 0.f;
    for (int k = 0; k < n_kv; ++k) m += tn[((long)c * n_kv + k) * E + e];
    tf[(long)c * E + e] = (m / (float)n_kv) * inv;
  }
}

// (mean, rstd) of each row from the GEMM epilogues' per-64-column partial sums, fixed summation order.  A block owns 64
// consecutive rows = one contiguous run of 64 * slots float2: it is loaded coalesced into LDS (a thread per row would read
// with a stride of slots * 8 bytes), then one thread per row sums its slots in order.
constexpr int RS_ROWS = 64, RS_MAX_SLOTS = 32;
__global__ __launch_bounds__(256) void row_stats_kernel(const float2* part, int slots, float inv_d, int rows, float2* stats) {
  __shared__ float2 buf[RS_ROWS * RS_MAX_SLOTS];
  const int r0 = blockIdx.x * RS_ROWS;
  const int n = min(RS_ROWS, rows - r0) * slots;
  const float2* src = part + (long)r0 * slots;
  for (int i = threadIdx.x; i < n; i += 256) buf[i] = src[i];
  __syncthreads();
  const int r = threadIdx.x;
  if (r >= RS_ROWS || r0 + r >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < slots; ++k) { const float2 v = buf[r * slots + k]; s1 += v.x; s2 += v.y; }
  const float mean = s1 * inv_d;
  const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
  stats[r0 + r] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
}

}  // namespace

extern "C" int gava_row_stats(const float* rowsum, int slots, int D, int rows, float* stats, gava_stream_t stream) {
  if (!rowsum || !stats || slots <= 0 || slots > RS_MAX_SLOTS || D <= 0 || rows <= 0) return GAVA_EINVAL;
  hipLaunchKernelGGL(row_stats_kernel, dim3((rows + RS_ROWS - 1) / RS_ROWS), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)rowsum, slots, 1.0f / (float)D, rows, (float2*)stats);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int gava_layernorm(const gava_layernorm_args* a, gava_stream_t stream) {
  if (!a || !a->in || a->rows <= 0) return GAVA_EINVAL;
  if (a->D % 4 || a->D > 256 * MAXV || a->D <= 0) return GAVA_EINVAL;
  if (!a->out16 && !a->out32 && !a->out_hi) return GAVA_EINVAL;
  if ((a->out_hi != nullptr) != (a->out_lo != nullptr) || (a->out_hi && (a->out_hl_stride % 4 || a->out_hl_stride < a->D))) return GAVA_EINVAL;
  if (a->split_out && (!a->out16 || a->out16_stride < 3 * (int64_t)a->D)) return GAVA_EINVAL;
  if (a->gamma && !a->beta) return GAVA_EINVAL;
  if ((a->gamma2 != nullptr) != (a->beta2 != nullptr) || (a->gamma2 && (!a->gamma || !a->out16 || (!a->out32 && !a->out_hi)))) return GAVA_EINVAL;
  if (a->in_stride % 4 || (a->out16 && a->out16_stride % 4) || (a->out32 && a->out32_stride % 4)) return GAVA_EINVAL;
  LnParams p{a->in, a->in_stride, a->in_row_index, a->gamma, a->beta, (unsigned short*)a->out16,
             a->out16_stride, a->out32, a->out32_stride, a->rows, a->D, a->split_out, a->gamma2, a->beta2,
             (unsigned short*)a->out_hi, (unsigned short*)a->out_lo, a->out_hl_stride};
  dim3 grid((a->rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (a->prec == GAVA_PREC_F16) hipLaunchKernelGGL(layernorm_kernel<PrecF16>, grid, block, 0, s, p);
  else if (a->prec == GAVA_PREC_BF16) hipLaunchKernelGGL(layernorm_kernel<PrecBF16>, grid, block, 0, s, p);
  else return GAVA_EINVAL;
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

extern "C" int gava_convert_h16(const float* in, void* out, size_t n, int prec, gava_stream_t stream) {
  if (!in || !out) return GAVA_EINVAL;
  if (n == 0) return GAVA_OK;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (prec == GAVA_PREC_F16) hipLaunchKernelGGL(convert_kernel<PrecF16>, dim3(blocks), dim3(256), 0, s, in, (unsigned short*)out, n);
  else if (prec == GAVA_PREC_BF16) hipLaunchKernelGGL(convert_kernel<PrecBF16>, dim3(blocks), dim3(256), 0, s, in, (unsigned short*)out, n);
  else return GAVA_EINVAL;
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

extern "C" int gava_similarity_head(const float* video, const float* text, const float* logit_scale,
                                    const float* logit_bias, int B, int C, int n_kv, int E, float* logits,
                                    float* text_features, float* video_norm, gava_stream_t stream) {
  // video_norm receives the normalised video rows; text_features first holds the class means of the unit prompt
  // features (the GEMM's second operand), then is re-normalised in place (VitaCLIP_model.py:290-291; with n_kv == 1
  // that re-normalises an already unit row exactly as upstream does).
  if (!video || !text || !logit_scale || !logits || !text_features || !video_norm) return GAVA_EINVAL;
  if (B <= 0 || C <= 0 || E <= 0 || n_kv <= 0 || E % 4) return GAVA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, s, video, video_norm, B, E);
  if (n_kv == 1) hipLaunchKernelGGL(l2norm_rows_kernel, dim3((C + 3) / 4), dim3(256), 0, s, text, text_features, C, E);
  else hipLaunchKernelGGL(class_mean_kernel, dim3((C + 3) / 4), dim3(256), 0, s, text, text_features, C, n_kv, E);
  hipLaunchKernelGGL(logits_mfma_kernel, dim3(((B + 15) / 16) * ((C + 15) / 16)), dim3(64), 0, s, video_norm,
                     text_features, logit_scale, logit_bias, B, C, E, logits);
  hipLaunchKernelGGL(text_feature_kernel, dim3((C + 3) / 4), dim3(256), 0, s, text_features, text_features, C, 1, E);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

static unsigned long long* g_debug_buffer = nullptr;
extern "C" int gava_debug_set_buffer(void* dev_u64) {
  g_debug_buffer = (unsigned long long*)dev_u64;
  return GAVA_OK;
}
namespace gava {
unsigned long long* debug_buffer() { return g_debug_buffer; }   // diagnostics only (tools/gemm_stamps.py); nullptr = stamps off
}

// ---- internal launchers used by the fused drivers -------------------------------------------
namespace gava {

int side_ln(const float* gp, const float* lp, const float* cp, const float* summ, const float* gamma,
            const float* beta, void* out, int G, int T, int BT, int D, int prec, hipStream_t s) {
  SideParams p{gp, lp, cp, summ, gamma, beta, (unsigned short*)out, G, T, BT, D};
  const int rows = G + 2 * BT;
  if (prec == GAVA_PREC_F16) hipLaunchKernelGGL(side_ln_kernel<PrecF16>, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(side_ln_kernel<PrecBF16>, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

int cls_embed(float* X, const float* cls, const float* pos, const float* time, int BT, int T, int D,
              long frame_stride, hipStream_t s) {
  const long n = (long)BT * D;
  hipLaunchKernelGGL(cls_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, cls, pos, time, BT, T, D, frame_stride);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

int mean_rows(const float* in, float* out, int B, int T, int D, hipStream_t s) {
  const long n = (long)B * D;
  hipLaunchKernelGGL(mean_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, B, T, D);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

int copy_rows(const float* in, long in_stride, float* out, int rows, int D, hipStream_t s) {
  const long n = (long)rows * D;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, in_stride, out, rows, D);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

int join_rows(const void* hi, const void* lo, long in_stride, float* out, long out_stride, int rows, int D, int prec, hipStream_t s) {
  const long n = (long)rows * D;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (prec == GAVA_PREC_F16)
    hipLaunchKernelGGL(join_rows_kernel<PrecF16>, grid, block, 0, s, (const unsigned short*)hi, (const unsigned short*)lo, in_stride, out, out_stride, rows, D);
  else if (prec == GAVA_PREC_BF16)
    hipLaunchKernelGGL(join_rows_kernel<PrecBF16>, grid, block, 0, s, (const unsigned short*)hi, (const unsigned short*)lo, in_stride, out, out_stride, rows, D);
  else return GAVA_EINVAL;
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

int text_embed(const float* emb, const float* pos, const float* ctx, const int* tok, float* X,
               int n_prompts, int L, int W, int n_ctx, hipStream_t s) {
  const long n = (long)n_prompts * L * W;
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, emb, pos, ctx, tok, X, n_prompts, L, W, n_ctx);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

}  // namespace gava
