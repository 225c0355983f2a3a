// backward.hip — gradient kernels for the trainable subset (training/train.py:441-490 calls
// loss.backward() through VitaCLIP.forward; every transformer weight is frozen, VitaCLIP_model.py:230-239, so the
// path needs dgrad only: gradients flow THROUGH the frozen GEMMs to the prompt parameters).
//   * dgrad GEMMs are gava_gemm on transposed weight copies (dX = dY . W  ==  gemm(A = dY, W = W^T [in][out])).
//   * this file: LayerNorm backward, QuickGELU backward, softmax-attention backward.
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
  int rows, D, accumulate;
};

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

// ---------------------------------------------------------------------------------------------
// Attention backward, short sequences (n <= 88 keys; the text tower's 77).  One workgroup per (sequence, head):
//   S = Q K^T (Q already carries 1/sqrt(dh)), P = softmax(S + mask), O = P V
//   dV = P^T dO;  dP = dO V^T;  dS = P * (dP - rowsum(P * dP));  dQ = q_scale * dS K;  dK = dS^T Q
// Everything fp32 in LDS/registers: 5 products of n x n x 64 are ~2 MFLOP per head, not worth MFMA tiles;
// the vision-side (n = 214, 6144 heads per layer) version is MFMA work of its own.
constexpr int ATT_BWD_MAXN = 88, DH = 64;   // 154 KiB of LDS at n = 88

struct AttnBwdParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; long ld_qkv;
  const unsigned short* dout; long ld_dout;
  unsigned short* dq; unsigned short* dk; unsigned short* dv; long ld_dqkv;
  int heads, n, causal;
  float q_scale;
};

template <class P>
__global__ __launch_bounds__(256) void attention_bwd_small_kernel(const AttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int n = p.n, tid = threadIdx.x;
  const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
  float* Q = reinterpret_cast<float*>(smem_raw);         // [n][DH+1] each (padded: conflict-free column walks)
  constexpr int LDT = DH + 1;
  float* K = Q + n * LDT;
  float* V = K + n * LDT;
  float* dO = V + n * LDT;
  float* Pm = dO + n * LDT;                               // [n][n+1]  P, then dS
  float* dP = Pm + n * (n + 1);                           // [n][n+1]
  const int LDP = n + 1;
  const long row0 = (long)b * n;
  for (int e = tid; e < n * DH; e += 256) {
    const int i = e / DH, d = e % DH;
    const long r = row0 + i;
    Q[i * LDT + d] = P::up(p.q[r * p.ld_qkv + h * DH + d]);
    K[i * LDT + d] = P::up(p.k[r * p.ld_qkv + h * DH + d]);
    V[i * LDT + d] = P::up(p.v[r * p.ld_qkv + h * DH + d]);
    dO[i * LDT + d] = P::up(p.dout[r * p.ld_dout + h * DH + d]);
  }
  __syncthreads();
  // S and dP
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    float s = 0.f, t = 0.f;
    if (!p.causal || j <= i) {
#pragma unroll 16
      for (int d = 0; d < DH; ++d) { s += Q[i * LDT + d] * K[j * LDT + d]; t += dO[i * LDT + d] * V[j * LDT + d]; }
    } else {
      s = -INFINITY;
    }
    Pm[i * LDP + j] = s; dP[i * LDP + j] = t;
  }
  __syncthreads();
  // row softmax and dS, one wave per row
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < n; i += 4) {
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, Pm[i * LDP + j]);
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) { const float e = __expf(Pm[i * LDP + j] - m); Pm[i * LDP + j] = e; sum += e; }
    sum = wave_sum(sum);
    const float r = 1.f / sum;
    float delta = 0.f;
    for (int j = lane; j < n; j += 64) { const float pj = Pm[i * LDP + j] * r; Pm[i * LDP + j] = pj; delta += pj * dP[i * LDP + j]; }
    delta = wave_sum(delta);
    // dP row becomes dS; the P row is kept for dV
    for (int j = lane; j < n; j += 64) dP[i * LDP + j] = Pm[i * LDP + j] * (dP[i * LDP + j] - delta);
  }
  __syncthreads();
  // dQ[i][d] = q_scale * sum_j dS[i][j] K[j][d];  dK[j][d] = sum_i dS[i][j] Q[i][d];  dV[j][d] = sum_i P[i][j] dO[i][d]
  for (int e = tid; e < n * DH; e += 256) {
    const int i = e / DH, d = e % DH;
    float aq = 0.f, ak = 0.f, av = 0.f;
    for (int j = 0; j < n; ++j) {
      aq += dP[i * LDP + j] * K[j * LDT + d];
      ak += dP[j * LDP + i] * Q[j * LDT + d];
      av += Pm[j * LDP + i] * dO[j * LDT + d];
    }
    const long r = row0 + i;
    p.dq[r * p.ld_dqkv + h * DH + d] = P::cvt(aq * p.q_scale);
    p.dk[r * p.ld_dqkv + h * DH + d] = P::cvt(ak);
    p.dv[r * p.ld_dqkv + h * DH + d] = P::cvt(av);
  }
}

// ---------------------------------------------------------------------------------------------
// Attention backward for the vision blocks: one workgroup per (frame, head), 197 queries x 214 keys, the keys
// being the frame's own rows plus the gathered prompt ("side") rows exactly as in attention.hip.
// First correct version, VALU arithmetic with fp32 accumulation (an MFMA version is the next step):
//   * a thread pair owns key j (32 head dims each): K[j], V[j] and the dK[j], dV[j] accumulators live in registers, so
//     S = Q K^T, dP = dO V^T, dV += P^T dO and dK += dS^T Q need no cross-thread traffic at all - queries are
//     processed in blocks of 8 rows that every thread reads as LDS broadcasts;
//   * softmax statistics and delta = rowsum(P*dP) are wave reductions + an 8-entry LDS exchange per row;
//   * dQ = dS K is the one product that runs across keys: dS block and K go through LDS, thread = (d, row).
// Main-row gradients are written as h16 into a [rows][3D] buffer (the operand of the in-projection dgrad);
// prompt-row gradients are shared by many workgroups (global prompts: every frame): each workgroup stores its
// partial into dside [frame][G + T + 1][2D] and the caller reduces over the frames that share a row.
constexpr int AB_QB = 8, AB_MAXK = 256, AB_HALF = DH / 2;   // 8 query rows per block: 8 x 64 dQ outputs = one per thread

struct AttnBwdMainParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; long ld_qkv;
  const unsigned short* sk; const unsigned short* sv; long ld_side;
  const unsigned short* dout; long ld_dout;
  unsigned short* dq; unsigned short* dk; unsigned short* dv; long ld_dqkv;
  float* dsk; float* dsv; long ld_dside;
  int batch, heads, n_q, n_kmain, n_g, T, has_summary, n_keys;
  float q_scale;
};

template <class P>
__global__ __launch_bounds__(512) void attention_bwd_main_kernel(const AttnBwdMainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x / p.heads, h = blockIdx.x % p.heads;   // frame, head
  const int nk = p.n_keys;
  constexpr int LDK = DH + 1;
  float* Kf = reinterpret_cast<float*>(smem_raw);      // [nk][65] fp32 K for the dQ product
  float* Qf = Kf + nk * LDK;                            // [QB][64] fp32 (dK += dS^T Q)
  float* Of = Qf + AB_QB * DH;                          // [QB][64] fp32 (dV += P^T dO)
  const int LDS_ = nk + 1;
  float* Sb = Of + AB_QB * DH;                          // [QB][nk+1]  dS block
  float* Pb = Sb + AB_QB * LDS_;                        // [QB][nk+1]  P block
  float* red = Pb + AB_QB * LDS_;                       // [QB][8]
  unsigned short* Qh = reinterpret_cast<unsigned short*>(red + AB_QB * 8);   // [QB][64] h16 as stored (dot2 operands)
  unsigned short* Oh = Qh + AB_QB * DH;
  const long row0 = (long)n * p.n_kmain;                // main rows of this frame (queries are its first n_q rows)

  // ---- a thread PAIR owns key j (adjacent lanes: 32 of the 64 head dims each): K, V packed h16 and the fp32
  //      dK, dV halves in registers
  const int j = tid >> 1, hf = tid & 1, d0 = hf * AB_HALF;
  const bool has_key = j < nk;
  const bool lead = has_key && hf == 0;                 // one thread of the pair speaks in the row reductions
  unsigned kp2[AB_HALF / 2], vp2[AB_HALF / 2];
  float dkr[AB_HALF], dvr[AB_HALF];
  long side_row = -1;
#pragma unroll
  for (int d = 0; d < AB_HALF / 2; ++d) { kp2[d] = 0u; vp2[d] = 0u; }
#pragma unroll
  for (int d = 0; d < AB_HALF; ++d) { dkr[d] = 0.f; dvr[d] = 0.f; }
  if (has_key) {
    const unsigned short* kp; const unsigned short* vp;
    if (j < p.n_kmain) {
      kp = p.k + (row0 + j) * p.ld_qkv + h * DH + d0; vp = p.v + (row0 + j) * p.ld_qkv + h * DH + d0;
    } else {
      const int sidx = j - p.n_kmain;
      side_row = sidx < p.n_g ? sidx
               : sidx < p.n_g + p.T ? p.n_g + (long)(n / p.T) * p.T + (sidx - p.n_g)
                                    : (long)p.n_g + p.batch + n;
      kp = p.sk + side_row * p.ld_side + h * DH + d0; vp = p.sv + side_row * p.ld_side + h * DH + d0;
    }
#pragma unroll
    for (int c = 0; c < AB_HALF / 8; ++c) {
      const uint4 a = *reinterpret_cast<const uint4*>(kp + 8 * c);
      const uint4 b = *reinterpret_cast<const uint4*>(vp + 8 * c);
      kp2[4 * c] = a.x; kp2[4 * c + 1] = a.y; kp2[4 * c + 2] = a.z; kp2[4 * c + 3] = a.w;
      vp2[4 * c] = b.x; vp2[4 * c + 1] = b.y; vp2[4 * c + 2] = b.z; vp2[4 * c + 3] = b.w;
    }
#pragma unroll
    for (int d = 0; d < AB_HALF / 2; ++d) {
      Kf[j * LDK + d0 + 2 * d] = P::up((unsigned short)kp2[d]);
      Kf[j * LDK + d0 + 2 * d + 1] = P::up((unsigned short)(kp2[d] >> 16));
    }
  }

  const int nblk = (p.n_q + AB_QB - 1) / AB_QB;
  for (int blk = 0; blk < nblk; ++blk) {
    const int i0 = blk * AB_QB;
    __syncthreads();   // previous block's readers of the row blocks are done (and Kf is complete before its first use)
    {                  // ---- stage the block's query rows of Q and dO: 8 x 64 = one element per thread
      const int i = tid >> 6, d = tid & 63;
      const bool ok = i0 + i < p.n_q;
      const unsigned short qv = ok ? p.q[(row0 + i0 + i) * p.ld_qkv + h * DH + d] : (unsigned short)0;
      const unsigned short ov = ok ? p.dout[(row0 + i0 + i) * p.ld_dout + h * DH + d] : (unsigned short)0;
      Qh[tid] = qv; Oh[tid] = ov;
      Qf[tid] = P::up(qv); Of[tid] = P::up(ov);
    }
    __syncthreads();
    // ---- S = Q K^T and dP = dO V^T for this key: packed dot products (v_dot2c), pair-summed over the two halves
    float S[AB_QB], dP[AB_QB];
#pragma unroll
    for (int i = 0; i < AB_QB; ++i) {
      float s = 0.f, t = 0.f;
#pragma unroll
      for (int c = 0; c < AB_HALF / 8; ++c) {
        const uint4 qv = *reinterpret_cast<const uint4*>(Qh + i * DH + d0 + 8 * c);
        const uint4 ov = *reinterpret_cast<const uint4*>(Oh + i * DH + d0 + 8 * c);
        s = P::dot2(qv.x, kp2[4 * c], s); s = P::dot2(qv.y, kp2[4 * c + 1], s);
        s = P::dot2(qv.z, kp2[4 * c + 2], s); s = P::dot2(qv.w, kp2[4 * c + 3], s);
        t = P::dot2(ov.x, vp2[4 * c], t); t = P::dot2(ov.y, vp2[4 * c + 1], t);
        t = P::dot2(ov.z, vp2[4 * c + 2], t); t = P::dot2(ov.w, vp2[4 * c + 3], t);
      }
      s += __shfl_xor(s, 1, 64);
      t += __shfl_xor(t, 1, 64);
      S[i] = has_key ? s : -INFINITY;
      dP[i] = t;
    }
    // ---- softmax over keys (across the workgroup), then dS = P * (dP - rowsum(P * dP))
    auto reduce_rows = [&](float (&x)[AB_QB], bool is_max) {
#pragma unroll
      for (int i = 0; i < AB_QB; ++i) {
        float v = x[i];
#pragma unroll
        for (int o = 32; o; o >>= 1) { const float w = __shfl_xor(v, o, 64); v = is_max ? fmaxf(v, w) : v + w; }
        if (lane == 0) red[i * 8 + wave] = v;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < AB_QB; ++i) {
        const float4 r0 = *reinterpret_cast<const float4*>(red + i * 8);
        const float4 r1 = *reinterpret_cast<const float4*>(red + i * 8 + 4);
        x[i] = is_max ? fmaxf(fmaxf(fmaxf(r0.x, r0.y), fmaxf(r0.z, r0.w)), fmaxf(fmaxf(r1.x, r1.y), fmaxf(r1.z, r1.w)))
                      : ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
      }
      __syncthreads();
    };
    float m[AB_QB];
#pragma unroll
    for (int i = 0; i < AB_QB; ++i) m[i] = S[i];
    reduce_rows(m, true);
#pragma unroll
    for (int i = 0; i < AB_QB; ++i) { S[i] = has_key ? __expf(S[i] - m[i]) : 0.f; m[i] = lead ? S[i] : 0.f; }
    reduce_rows(m, false);
#pragma unroll
    for (int i = 0; i < AB_QB; ++i) { S[i] = S[i] / m[i]; m[i] = lead ? S[i] * dP[i] : 0.f; }   // S is P now
    reduce_rows(m, false);
#pragma unroll
    for (int i = 0; i < AB_QB; ++i) {
      const bool ok = has_key && i0 + i < p.n_q;
      S[i] = ok ? S[i] : 0.f;
      dP[i] = ok ? S[i] * (dP[i] - m[i]) : 0.f;                                      // dP is dS now
      if (lead) { Sb[i * LDS_ + j] = dP[i]; Pb[i * LDS_ + j] = S[i]; }
    }
    __syncthreads();
    // ---- dV += P^T dO, dK += dS^T Q (own key, own half, registers).  A real loop over the rows, P / dS re-read from
    //      LDS: fully unrolled, the compiler hoists all 8 rows of Q and dO into registers at once and spills.
    if (has_key) {
#pragma unroll 1
      for (int i = 0; i < AB_QB; ++i) {
        const float ds = Sb[i * LDS_ + j], pj = Pb[i * LDS_ + j];
#pragma unroll
        for (int d = 0; d < AB_HALF; d += 4) {
          const float4 qv = *reinterpret_cast<const float4*>(Qf + i * DH + d0 + d);
          dkr[d] += ds * qv.x; dkr[d + 1] += ds * qv.y; dkr[d + 2] += ds * qv.z; dkr[d + 3] += ds * qv.w;
        }
#pragma unroll
        for (int d = 0; d < AB_HALF; d += 4) {
          const float4 ov = *reinterpret_cast<const float4*>(Of + i * DH + d0 + d);
          dvr[d] += pj * ov.x; dvr[d + 1] += pj * ov.y; dvr[d + 2] += pj * ov.z; dvr[d + 3] += pj * ov.w;
        }
      }
    }
    // ---- dQ[i][d] = q_scale * sum_j dS[i][j] K[j][d]: thread = (d = lane, row = wave)
    {
      const int d = lane, i = i0 + wave;
      float a0 = 0.f, a1 = 0.f;
      int jj = 0;
      for (; jj + 1 < nk; jj += 2) {
        a0 += Sb[wave * LDS_ + jj] * Kf[jj * LDK + d];
        a1 += Sb[wave * LDS_ + jj + 1] * Kf[(jj + 1) * LDK + d];
      }
      if (jj < nk) a0 += Sb[wave * LDS_ + jj] * Kf[jj * LDK + d];
      if (i < p.n_q) p.dq[(row0 + i) * p.ld_dqkv + h * DH + d] = P::cvt((a0 + a1) * p.q_scale);
    }
  }
  // ---- key gradients: own rows as h16, prompt rows by fp32 atomics
  if (has_key) {
    if (j < p.n_kmain) {
      unsigned short* ok = p.dk + (row0 + j) * p.ld_dqkv + h * DH + d0;
      unsigned short* ov = p.dv + (row0 + j) * p.ld_dqkv + h * DH + d0;
#pragma unroll
      for (int d = 0; d < AB_HALF; d += 4) {
        *reinterpret_cast<uint2*>(ok + d) = pack4<P>(dkr[d], dkr[d + 1], dkr[d + 2], dkr[d + 3]);
        *reinterpret_cast<uint2*>(ov + d) = pack4<P>(dvr[d], dvr[d + 1], dvr[d + 2], dvr[d + 3]);
      }
    } else {
      // per-frame partial of a shared prompt row: plain stores, the caller sums over the frames that share the row
      // (atomics into the shared rows cost 30x the arithmetic: 512 frames x 12 heads contend for the global prompts)
      const long pr = (long)n * (nk - p.n_kmain) + (j - p.n_kmain);
      float* ok = p.dsk + pr * p.ld_dside + h * DH + d0;
      float* ov = p.dsv + pr * p.ld_dside + h * DH + d0;
#pragma unroll
      for (int d = 0; d < AB_HALF; d += 4) {
        *reinterpret_cast<float4*>(ok + d) = make_float4(dkr[d], dkr[d + 1], dkr[d + 2], dkr[d + 3]);
        *reinterpret_cast<float4*>(ov + d) = make_float4(dvr[d], dvr[d + 1], dvr[d + 2], dvr[d + 3]);
      }
    }
  }
}

}  // namespace

extern "C" int gava_layernorm_backward(const gava_layernorm_bwd_args* a, gava_stream_t stream) {
  if (!a || !a->x || !a->gamma || !a->dy || !a->dx) return GAVA_EINVAL;
  if (a->rows <= 0 || a->D <= 0 || a->D % 4 || a->D > MAXV * 256) return GAVA_EINVAL;
  if ((a->dgamma == nullptr) != (a->dbeta == nullptr)) return GAVA_EINVAL;
  if (a->x_stride % 4 || a->dy_stride % 4 || a->dx_stride % 4) return GAVA_EINVAL;
  LnBwdParams p{a->x, a->x_stride, a->x_row_index, a->gamma, a->dy, a->dy_stride, a->dx, a->dx_stride,
                a->dx_row_index, a->dgamma, a->dbeta, a->rows, a->D, a->accumulate};
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((a->rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
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
  if (n_side == 0 && a->n_q == 0 && a->n <= ATT_BWD_MAXN) {
    AttnBwdParams p{(const unsigned short*)a->q, (const unsigned short*)a->k, (const unsigned short*)a->v, a->ld_qkv,
                    (const unsigned short*)a->dout, a->ld_dout, (unsigned short*)a->dq, (unsigned short*)a->dk,
                    (unsigned short*)a->dv, a->ld_dqkv, a->heads, a->n, a->causal, a->q_scale};
    const size_t lds = (size_t)(4 * a->n * (DH + 1) + 2 * a->n * (a->n + 1)) * sizeof(float);
    if (lds > 160 * 1024) return GAVA_EINVAL;
    dim3 grid(a->batch * a->heads), block(256);
    if (a->prec == GAVA_PREC_F16) {
      if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attention_bwd_small_kernel<PrecF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GAVA_ELAUNCH;
      hipLaunchKernelGGL(attention_bwd_small_kernel<PrecF16>, grid, block, lds, s, p);
    } else {
      if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attention_bwd_small_kernel<PrecBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GAVA_ELAUNCH;
      hipLaunchKernelGGL(attention_bwd_small_kernel<PrecBF16>, grid, block, lds, s, p);
    }
    GAVA_CHECK_LAUNCH();
    return GAVA_OK;
  }
  // vision blocks: n main rows per frame, the first n_q of them query (0 = all), prompt rows gathered as in gava_attention
  if (a->causal) return GAVA_EINVAL;
  if (n_side && (!a->side_v || !a->dside_k || !a->dside_v || a->n_g < 0 || a->T <= 0 || a->batch % a->T)) return GAVA_EINVAL;
  if ((a->ld_qkv | a->ld_side) % 8) return GAVA_EINVAL;
  if ((((uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->side_k | (uintptr_t)a->side_v | (uintptr_t)a->dside_k | (uintptr_t)a->dside_v) & 15) ||
      (a->ld_dqkv % 4) || (a->ld_dside % 4)) return GAVA_EINVAL;
  AttnBwdMainParams p;
  p.q = (const unsigned short*)a->q; p.k = (const unsigned short*)a->k; p.v = (const unsigned short*)a->v; p.ld_qkv = a->ld_qkv;
  p.sk = (const unsigned short*)a->side_k; p.sv = (const unsigned short*)a->side_v; p.ld_side = a->ld_side;
  p.dout = (const unsigned short*)a->dout; p.ld_dout = a->ld_dout;
  p.dq = (unsigned short*)a->dq; p.dk = (unsigned short*)a->dk; p.dv = (unsigned short*)a->dv; p.ld_dqkv = a->ld_dqkv;
  p.dsk = a->dside_k; p.dsv = a->dside_v; p.ld_dside = a->ld_dside;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q ? a->n_q : a->n; p.n_kmain = a->n;
  p.n_g = a->n_g; p.T = n_side ? a->T : 1; p.has_summary = a->has_summary; p.n_keys = a->n + n_side;
  p.q_scale = a->q_scale;
  if (p.n_keys > AB_MAXK || p.n_q > a->n) return GAVA_EINVAL;
  const size_t lds = (size_t)(p.n_keys * (DH + 1) + 2 * AB_QB * DH + 2 * AB_QB * (p.n_keys + 1) + AB_QB * 8) * sizeof(float) + 2 * AB_QB * DH * sizeof(unsigned short);
  dim3 grid(a->batch * a->heads), block(512);
  if (a->prec == GAVA_PREC_F16) {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attention_bwd_main_kernel<PrecF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GAVA_ELAUNCH;
    hipLaunchKernelGGL(attention_bwd_main_kernel<PrecF16>, grid, block, lds, s, p);
  } else {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attention_bwd_main_kernel<PrecBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GAVA_ELAUNCH;
    hipLaunchKernelGGL(attention_bwd_main_kernel<PrecBF16>, grid, block, lds, s, p);
  }
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}
