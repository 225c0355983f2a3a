// attention_bwd.hip — MFMA backward of softmax(Q K^T) V, head dim 64: the vision-block attention (the frame's rows +
// the gathered prompt rows), the T-token summary attention and the causal text attention, built from the same pieces as
// attention.hip:
//   S^T = K Q^T accumulators ARE the B operand of the next product, transposed operands come from the
//   hardware-transposing LDS read, 160-byte padded LDS rows, one workgroup (4 waves) per (frame, head).
// Two kernels (the two contraction directions need the score tile in the two orientations):
//   1. attn_bwd_dq_kernel   per 16-query tile: S^T = K Q^T -> P^T; dP^T = V dO^T; delta = rowsum(P * dP);
//                           dS^T = P^T * (dP^T - delta); dQ^T = K^T dS^T.  Also writes the row statistics
//                           L2[q] = log2(sum_k exp(s_qk)) (+ max) and delta[q] for the second kernel.
//   2. attn_bwd_dkv_kernel  per 16-key tile, looping over query tiles: S = Q K^T, dP = dO V^T (query on the MFMA
//                           row), P = exp2(S*log2e - L2), dS = P * (dP - delta); dV^T += dO^T P, dK^T += Q^T dS with
//                           P / dS again used straight from the accumulators as B operands (contraction over queries).
// q is expected pre-scaled by 1/sqrt(dh) (as the forward's QKV GEMM writes it); dq is multiplied by q_scale.
#include "common.h"
#include "internal.h"
#include <type_traits>

namespace {

constexpr int LDS_ROW = 160;  // bytes per row in LDS (64 x 2 B + 32 B pad): conflict-free for b128 and tr_b64 reads
constexpr float LOG2E = 1.4426950408889634f;

// Activations kept from the forward may be stored in another 16-bit type (PA, e.g. fp16) than the one the gradients
// and the MFMAs use (P, bf16): convert 8 packed values on load.  PA == P compiles to nothing.
template <class PA, class P>
static __device__ __forceinline__ uint4 to_p(uint4 v) {
  if constexpr (std::is_same<PA, P>::value) {
    return v;
  } else {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = P::cvt2(PA::up((unsigned short)w[e]), PA::up((unsigned short)(w[e] >> 16)));
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
}
template <class PA, class P>
static __device__ __forceinline__ s16x8_t load_act8(const unsigned short* p) {
  return __builtin_bit_cast(s16x8_t, to_p<PA, P>(*reinterpret_cast<const uint4*>(p)));
}

// key index -> row of the gathered prompt matrix (same as attention.hip)
static __device__ __forceinline__ long side_row_of(const gava::AttnBwdMfmaParams& p, int frame, int sidx) {
  return sidx < p.n_g ? sidx
       : sidx < p.n_g + p.T ? p.n_g + (long)(frame / p.T) * p.T + (sidx - p.n_g)
                            : (long)p.n_g + p.batch + frame;
}

template <class P, class PA, int NKT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const gava::AttnBwdMfmaParams p) {
  constexpr int KP = NKT * 16;
  constexpr int NIT = (KP * 8 + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * KP * LDS_ROW];
  char* Ks = smem;
  char* Vs = smem + KP * LDS_ROW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x / p.heads, h = blockIdx.x - n * p.heads;
  const int fr = lane & 15, fg = lane >> 4;
  const int n_qt = (p.n_q + 15) >> 4;
  const long row0 = (long)n * p.n_kmain;

  // ---- stage K, V (prompt rows gathered), zero rows beyond n_keys
  {
    uint4 kv[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id >> 3, chunk = id & 7;
      const int rowc = row < p.n_keys ? row : 0;
      const bool is_main = rowc < p.n_kmain;
      const long off = (is_main ? (row0 + rowc) * p.ld_qkv : side_row_of(p, n, rowc - p.n_kmain) * p.ld_side) + h * 64 + chunk * 8;
      kv[it] = *reinterpret_cast<const uint4*>((is_main ? p.k : p.sk) + off);
      vv[it] = *reinterpret_cast<const uint4*>((is_main ? p.v : p.sv) + off);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id >> 3, chunk = id & 7;
      if (id < KP * 8) {
        const bool ok = row < p.n_keys;
        *reinterpret_cast<uint4*>(Ks + row * LDS_ROW + chunk * 16) = ok ? to_p<PA, P>(kv[it]) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Vs + row * LDS_ROW + chunk * 16) = ok ? to_p<PA, P>(vv[it]) : make_uint4(0, 0, 0, 0);
      }
    }
  }
  __syncthreads();
  const int tr_off = (4 * fg + (fr >> 2)) * LDS_ROW + (fr & 3) * 8;

  for (int qt = wave; qt < n_qt; qt += 4) {
    const int qi = qt * 16 + fr;
    const int qrow = qi < p.n_q ? qi : p.n_q - 1;
    const long qrow0 = (long)n * p.q_rows;   // query-side buffers (q, dout, dq) may hold fewer rows per frame
    const unsigned short* qp = p.q + (qrow0 + qrow) * p.ld_q + h * 64 + 8 * fg;
    const unsigned short* op = p.dout + (qrow0 + qrow) * p.ld_dout + h * 64 + 8 * fg;
    const s16x8_t q0 = load_act8<PA, P>(qp), q1 = load_act8<PA, P>(qp + 32);
    const s16x8_t g0 = *reinterpret_cast<const s16x8_t*>(op), g1 = *reinterpret_cast<const s16x8_t*>(op + 32);

    // S^T = K Q^T and dP^T = V dO^T: lane holds, for its query fr, keys kt*16 + 4*fg + r
    f32x4_t s[NKT], dp[NKT];
    constexpr int QCH = NKT <= 7 ? NKT : (NKT % 7 == 0 ? 7 : (NKT % 5 == 0 ? 5 : 2));
    // two passes (K then V) so that only one chunk of fragments is live beside the 2 x NKT accumulators
#pragma unroll
    for (int c0 = 0; c0 < NKT; c0 += QCH) {
      s16x8_t kf[QCH][2];
#pragma unroll
      for (int t = 0; t < QCH; ++t) {
        const int ro = ((c0 + t) * 16 + fr) * LDS_ROW + fg * 16;
        kf[t][0] = *reinterpret_cast<const s16x8_t*>(Ks + ro);
        kf[t][1] = *reinterpret_cast<const s16x8_t*>(Ks + ro + 64);
      }
#pragma unroll
      for (int t = 0; t < QCH; ++t) {
        f32x4_t a = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        a = P::mfma(kf[t][0], q0, a);
        a = P::mfma(kf[t][1], q1, a);
        s[c0 + t] = a;
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * QCH, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * QCH, 0);
    }
#pragma unroll
    for (int c0 = 0; c0 < NKT; c0 += QCH) {
      s16x8_t vf[QCH][2];
#pragma unroll
      for (int t = 0; t < QCH; ++t) {
        const int ro = ((c0 + t) * 16 + fr) * LDS_ROW + fg * 16;
        vf[t][0] = *reinterpret_cast<const s16x8_t*>(Vs + ro);
        vf[t][1] = *reinterpret_cast<const s16x8_t*>(Vs + ro + 64);
      }
#pragma unroll
      for (int t = 0; t < QCH; ++t) {
        f32x4_t b = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        b = P::mfma(vf[t][0], g0, b);
        b = P::mfma(vf[t][1], g1, b);
        dp[c0 + t] = b;
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * QCH, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * QCH, 1);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt * 16 + 16 > p.n_keys || (CAUSAL && kt * 16 + 15 > qt * 16)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * fg + r;
          s[kt][r] = (key < p.n_keys && (!CAUSAL || key <= qi)) ? s[kt][r] : -INFINITY;
        }
      }
      mx = fmaxf(fmaxf(mx, s[kt][0]), fmaxf(s[kt][1], fmaxf(s[kt][2], s[kt][3])));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mneg = -mx * LOG2E;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], LOG2E, mneg));
        s[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kt][r] *= inv;                       // P
        delta += s[kt][r] * dp[kt][r];
      }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (fg == 0 && qi < p.n_q) {
      float* st = p.stats + ((long)blockIdx.x * p.q_pad + qi) * 2;
      st[0] = mx * LOG2E + __builtin_amdgcn_logf(sum);   // log2(sum_k exp(s_k)): P = exp2(s*log2e - L2)
      st[1] = delta;
    }
    // dS^T = P^T * (dP^T - delta), then dQ^T = K^T dS^T (exactly the forward's O^T = V^T P^T with K for V)
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    constexpr int NC2 = NKT / 2;
    constexpr int PCH = NC2 % 7 == 0 ? 7 : (NC2 % 5 == 0 ? 5 : (NC2 % 3 == 0 ? 3 : 1));
#pragma unroll
    for (int b0 = 0; b0 < NC2; b0 += PCH) {
      s16x4_t t0[PCH][4], t1[PCH][4];
#pragma unroll
      for (int c = 0; c < PCH; ++c) {
        const char* kb = Ks + (b0 + c) * 32 * LDS_ROW + tr_off;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          t0[c][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, kb + dt * 32));
          t1[c][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, kb + 16 * LDS_ROW + dt * 32));
        }
      }
#pragma unroll
      for (int c = 0; c < PCH; ++c) {
        const int cc = b0 + c;
        float d[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[r] = s[2 * cc][r] * (dp[2 * cc][r] - delta);
          d[4 + r] = s[2 * cc + 1][r] * (dp[2 * cc + 1][r] - delta);
        }
        const uint2 lo = pack4<P>(d[0], d[1], d[2], d[3]);
        const uint2 hi = pack4<P>(d[4], d[5], d[6], d[7]);
        const s16x8_t df = __builtin_bit_cast(s16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x8_t kT = __builtin_shufflevector(t0[c][dt], t1[c][dt], 0, 1, 2, 3, 4, 5, 6, 7);
          o[dt] = P::mfma(kT, df, o[dt]);
        }
      }
    }
    if (qi < p.n_q) {
      unsigned short* dq = p.dq + (qrow0 + qi) * p.ld_dq + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(dq + dt * 16) = pack4<P>(o[dt][0] * p.q_scale, o[dt][1] * p.q_scale, o[dt][2] * p.q_scale, o[dt][3] * p.q_scale);
    }
  }
}

template <class P, class PA, int NQT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const gava::AttnBwdMfmaParams p) {
  static_assert(NQT % 2 == 0, "query tiles are consumed in pairs (32-deep MFMA contraction)");
  constexpr int QP = NQT * 16;
  constexpr int NIT = (QP * 8 + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * QP * LDS_ROW + 2 * QP * sizeof(float)];
  char* Qs = smem;
  char* Os = smem + QP * LDS_ROW;
  float* L2s = reinterpret_cast<float*>(smem + 2 * QP * LDS_ROW);
  float* Dls = L2s + QP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x / p.heads, h = blockIdx.x - n * p.heads;
  const int fr = lane & 15, fg = lane >> 4;
  const long row0 = (long)n * p.n_kmain;
  const int n_kt = (p.n_keys + 15) >> 4;
  const int n_side = p.n_keys - p.n_kmain;

  // ---- stage Q, dO (zero rows beyond n_q) and the row statistics (P = 0 for padded queries)
  {
    uint4 qv[NIT], ov[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id >> 3, chunk = id & 7;
      const int rowc = row < p.n_q ? row : 0;
      qv[it] = *reinterpret_cast<const uint4*>(p.q + ((long)n * p.q_rows + rowc) * p.ld_q + h * 64 + chunk * 8);
      ov[it] = *reinterpret_cast<const uint4*>(p.dout + ((long)n * p.q_rows + rowc) * p.ld_dout + h * 64 + chunk * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id >> 3, chunk = id & 7;
      if (id < QP * 8) {
        const bool ok = row < p.n_q;
        *reinterpret_cast<uint4*>(Qs + row * LDS_ROW + chunk * 16) = ok ? to_p<PA, P>(qv[it]) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Os + row * LDS_ROW + chunk * 16) = ok ? ov[it] : make_uint4(0, 0, 0, 0);
      }
    }
    for (int qx = tid; qx < QP; qx += 256) {
      const bool ok = qx < p.n_q;
      const float* st = p.stats + ((long)blockIdx.x * p.q_pad + (ok ? qx : 0)) * 2;
      L2s[qx] = ok ? st[0] : INFINITY;
      Dls[qx] = ok ? st[1] : 0.f;
    }
  }
  __syncthreads();
  const int tr_off = (4 * fg + (fr >> 2)) * LDS_ROW + (fr & 3) * 8;

  for (int kt = wave; kt < n_kt; kt += 4) {
    const int key = kt * 16 + fr;
    const bool key_ok = key < p.n_keys;
    const int keyc = key_ok ? key : 0;
    const bool is_main = keyc < p.n_kmain;
    const long koff = (is_main ? (row0 + keyc) * p.ld_qkv : side_row_of(p, n, keyc - p.n_kmain) * p.ld_side) + h * 64 + 8 * fg;
    const unsigned short* kp = (is_main ? p.k : p.sk) + koff;
    const unsigned short* vp = (is_main ? p.v : p.sv) + koff;
    // B operands: this lane's key, head dims 8*fg.. and 32 + 8*fg..
    const s16x8_t kb0 = load_act8<PA, P>(kp), kb1 = load_act8<PA, P>(kp + 32);
    const s16x8_t vb0 = load_act8<PA, P>(vp), vb1 = load_act8<PA, P>(vp + 32);
    f32x4_t dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dk[dt] = dv[dt]; }

#pragma unroll 1
    for (int c = 0; c < NQT / 2; ++c) {
      // S = Q K^T, dP = dO V^T for query tiles 2c, 2c+1: lane holds key fr, queries t*16 + 4*fg + r
      f32x4_t st[2], pt[2];
      s16x4_t oT0[4], oT1[4], qT0[4], qT1[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ro = ((2 * c + t) * 16 + fr) * LDS_ROW + fg * 16;
        const s16x8_t qa0 = *reinterpret_cast<const s16x8_t*>(Qs + ro), qa1 = *reinterpret_cast<const s16x8_t*>(Qs + ro + 64);
        const s16x8_t oa0 = *reinterpret_cast<const s16x8_t*>(Os + ro), oa1 = *reinterpret_cast<const s16x8_t*>(Os + ro + 64);
        f32x4_t a = (f32x4_t){0.f, 0.f, 0.f, 0.f}, b = a;
        a = P::mfma(qa0, kb0, a);
        a = P::mfma(qa1, kb1, a);
        b = P::mfma(oa0, vb0, b);
        b = P::mfma(oa1, vb1, b);
        st[t] = a;
        pt[t] = b;
      }
      {
        const char* ob = Os + c * 32 * LDS_ROW + tr_off;
        const char* qb = Qs + c * 32 * LDS_ROW + tr_off;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          oT0[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, ob + dt * 32));
          oT1[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, ob + 16 * LDS_ROW + dt * 32));
          qT0[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, qb + dt * 32));
          qT1[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, qb + 16 * LDS_ROW + dt * 32));
        }
      }
      float pv[8], dsv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 l2 = *reinterpret_cast<const float4*>(L2s + (2 * c + t) * 16 + 4 * fg);
        const float4 dl = *reinterpret_cast<const float4*>(Dls + (2 * c + t) * 16 + 4 * fg);
        const float l2a[4] = {l2.x, l2.y, l2.z, l2.w}, dla[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool vis = key_ok && (!CAUSAL || key <= (2 * c + t) * 16 + 4 * fg + r);
          const float pr = vis ? __builtin_amdgcn_exp2f(fmaf(st[t][r], LOG2E, -l2a[r])) : 0.f;
          pv[4 * t + r] = pr;
          dsv[4 * t + r] = pr * (pt[t][r] - dla[r]);
        }
      }
      const uint2 plo = pack4<P>(pv[0], pv[1], pv[2], pv[3]), phi = pack4<P>(pv[4], pv[5], pv[6], pv[7]);
      const uint2 dlo = pack4<P>(dsv[0], dsv[1], dsv[2], dsv[3]), dhi = pack4<P>(dsv[4], dsv[5], dsv[6], dsv[7]);
      const s16x8_t pf = __builtin_bit_cast(s16x8_t, make_uint4(plo.x, plo.y, phi.x, phi.y));
      const s16x8_t df = __builtin_bit_cast(s16x8_t, make_uint4(dlo.x, dlo.y, dhi.x, dhi.y));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = P::mfma(__builtin_shufflevector(oT0[dt], oT1[dt], 0, 1, 2, 3, 4, 5, 6, 7), pf, dv[dt]);
        dk[dt] = P::mfma(__builtin_shufflevector(qT0[dt], qT1[dt], 0, 1, 2, 3, 4, 5, 6, 7), df, dk[dt]);
      }
    }
    // dv[dt][r] = dV[key fr][d = dt*16 + 4*fg + r]
    if (key_ok) {
      if (is_main) {
        unsigned short* ok_ = p.dk + (row0 + key) * p.ld_dqkv + h * 64 + 4 * fg;
        unsigned short* ov_ = p.dv + (row0 + key) * p.ld_dqkv + h * 64 + 4 * fg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          *reinterpret_cast<uint2*>(ok_ + dt * 16) = pack4<P>(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
          *reinterpret_cast<uint2*>(ov_ + dt * 16) = pack4<P>(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
        }
      } else {
        // per-frame partial of a shared prompt row (the caller sums over the frames sharing it)
        const long pr = (long)n * n_side + (key - p.n_kmain);
        float* ok_ = p.dsk + pr * p.ld_dside + h * 64 + 4 * fg;
        float* ov_ = p.dsv + pr * p.ld_dside + h * 64 + 4 * fg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          *reinterpret_cast<float4*>(ok_ + dt * 16) = make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
          *reinterpret_cast<float4*>(ov_ + dt * 16) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
        }
      }
    }
  }
}

template <class P, class PA, bool CAUSAL>
int launch(const gava::AttnBwdMfmaParams& p, hipStream_t s) {
  dim3 grid(p.batch * p.heads), block(256);
  const int kt = (p.n_keys + 15) / 16, qt2 = ((p.n_q + 15) / 16 + 1) / 2 * 2;
  if (kt <= 2) hipLaunchKernelGGL((attn_bwd_dq_kernel<P, PA, 2, CAUSAL>), grid, block, 0, s, p);
  else if (kt <= 6) hipLaunchKernelGGL((attn_bwd_dq_kernel<P, PA, 6, CAUSAL>), grid, block, 0, s, p);
  else if (kt <= 14) hipLaunchKernelGGL((attn_bwd_dq_kernel<P, PA, 14, CAUSAL>), grid, block, 0, s, p);
  else if (kt <= 20) hipLaunchKernelGGL((attn_bwd_dq_kernel<P, PA, 20, CAUSAL>), grid, block, 0, s, p);
  else return GAVA_EINVAL;
  if (qt2 <= 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<P, PA, 2, CAUSAL>), grid, block, 0, s, p);
  else if (qt2 <= 6) hipLaunchKernelGGL((attn_bwd_dkv_kernel<P, PA, 6, CAUSAL>), grid, block, 0, s, p);
  else if (qt2 <= 14) hipLaunchKernelGGL((attn_bwd_dkv_kernel<P, PA, 14, CAUSAL>), grid, block, 0, s, p);
  else if (qt2 <= 18) hipLaunchKernelGGL((attn_bwd_dkv_kernel<P, PA, 18, CAUSAL>), grid, block, 0, s, p);
  else return GAVA_EINVAL;
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

}  // namespace

namespace gava {
int attention_bwd_mfma(const AttnBwdMfmaParams& p, int prec, int act_prec, int causal, hipStream_t s) {
  if (causal) {   // the text tower (77 tokens, no prompt rows): same-precision only
    if (p.n_keys != p.n_kmain || prec != act_prec) return GAVA_EINVAL;
    if (prec == GAVA_PREC_F16) return launch<PrecF16, PrecF16, true>(p, s);
    if (prec == GAVA_PREC_BF16) return launch<PrecBF16, PrecBF16, true>(p, s);
    return GAVA_EINVAL;
  }
  if (prec == GAVA_PREC_F16 && act_prec == GAVA_PREC_F16) return launch<PrecF16, PrecF16, false>(p, s);
  if (prec == GAVA_PREC_BF16 && act_prec == GAVA_PREC_BF16) return launch<PrecBF16, PrecBF16, false>(p, s);
  if (prec == GAVA_PREC_BF16 && act_prec == GAVA_PREC_F16) return launch<PrecBF16, PrecF16, false>(p, s);   // fp16 forward, bf16 gradients
  return GAVA_EINVAL;
}
}  // namespace gava
