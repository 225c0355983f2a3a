// common.h — device-side building blocks shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gava_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-bit MFMA operand precisions.  Storage is always 2 bytes; fragments travel as 8 x 16-bit.
struct PrecF16 {
  typedef _Float16 T;
  static __device__ __forceinline__ f32x4_t mfma(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned short cvt(float x) {
    return __builtin_bit_cast(unsigned short, (_Float16)x);
  }
  static __device__ __forceinline__ float up(unsigned short u) {
    return (float)__builtin_bit_cast(_Float16, u);
  }
  // two values -> one dword in ONE instruction (v_cvt_pk_f16_f32, round-to-nearest-even)
  static __device__ __forceinline__ unsigned cvt2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
  }
  // c + a.lo*b.lo + a.hi*b.hi on packed pairs (v_dot2c_f32_f16), fp32 accumulate
  static __device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
  }
};
struct PrecBF16 {
  typedef __bf16 T;
  static __device__ __forceinline__ f32x4_t mfma(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned short cvt(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);
  }
  static __device__ __forceinline__ float up(unsigned short u) {
    return __builtin_bit_cast(float, (unsigned)u << 16);
  }
  static __device__ __forceinline__ unsigned cvt2(float a, float b) {   // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
  }
  static __device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {   // v_dot2c_f32_bf16
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
  }
};

template <class P>
static __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = P::cvt2(a, b);
  r.y = P::cvt2(c, d);
  return r;
}

// hi/lo split of 4 values: hi = h16(v), lo = h16(v - hi)
template <class P>
static __device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  hi.x = P::cvt2(a, b);
  hi.y = P::cvt2(c, d);
  lo = pack4<P>(a - P::up((unsigned short)hi.x), b - P::up((unsigned short)(hi.x >> 16)),
                c - P::up((unsigned short)hi.y), d - P::up((unsigned short)(hi.y >> 16)));
}

// the residual stream's pair: hi in the operand type, lo = fp16 of the remainder whatever the operand type (2^-22 of |x| with
// fp16 operands, 2^-20 with bf16 ones)
template <class P>
static __device__ __forceinline__ void split4_lo16(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  hi.x = P::cvt2(a, b);
  hi.y = P::cvt2(c, d);
  lo = pack4<PrecF16>(a - P::up((unsigned short)hi.x), b - P::up((unsigned short)(hi.x >> 16)),
                      c - P::up((unsigned short)hi.y), d - P::up((unsigned short)(hi.y >> 16)));
}

// x[l] + x[l ^ 16] + x[l ^ 32] + x[l ^ 48] in every lane, on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap
// instead of two dependent ds_bpermute round trips through the LDS pipe
static __device__ __forceinline__ float sum_across_lane_groups(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

static __device__ __forceinline__ float max_across_lane_groups(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static __device__ __forceinline__ float quick_gelu(float x) {
  // x * sigmoid(1.702 x)  (VitaCLIP_vision_encoder_utils.py:18-20); exp(-1.702 x) = exp2(x * (-1.702 log2 e)):
  // one multiply in front of v_exp_f32 instead of the two that __expf(-1.702f * x) compiles to
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.4554669595930157f));
}

// two values at a time: the multiplies and the add become packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32),
// which matters in the GEMM epilogues - they are VALU-issue-bound (finding 6)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void quick_gelu2(float& a, float& b) {
#ifdef GAVA_QGELU_SCALAR   // A/B builds: the previous scalar form
  a = a * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * a));
  b = b * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * b));
  return;
#endif
  const f32x2_t x = {a, b};
  const f32x2_t t = x * (f32x2_t){-2.4554669595930157f, -2.4554669595930157f};
  const f32x2_t d = (f32x2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (f32x2_t){1.0f, 1.0f};
  const f32x2_t y = x * (f32x2_t){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  a = y.x; b = y.y;
}

// Eight values at a time, stage by stage (all multiplies, all exps, all adds, all reciprocals, all multiplies): in the pairwise form
// hipcc ran the sixteen values of a row group through ONE pair of temporaries - multiply, exp, add, rcp, multiply, each waiting for
// the one before, an s_nop after every step (264 per tile in the fc1 kernel).  Stage-wise the eight chains advance side by side.
static __device__ __forceinline__ void quick_gelu8(float (&v)[16], int base) {
  f32x2_t t[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = (f32x2_t){v[base + 2 * q], v[base + 2 * q + 1]} * (f32x2_t){-2.4554669595930157f, -2.4554669595930157f};
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = (f32x2_t){__builtin_amdgcn_exp2f(t[q].x), __builtin_amdgcn_exp2f(t[q].y)};
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = t[q] + (f32x2_t){1.0f, 1.0f};
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = (f32x2_t){__builtin_amdgcn_rcpf(t[q].x), __builtin_amdgcn_rcpf(t[q].y)};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t y = (f32x2_t){v[base + 2 * q], v[base + 2 * q + 1]} * t[q];
    v[base + 2 * q] = y.x; v[base + 2 * q + 1] = y.y;
  }
}

static __device__ __forceinline__ float quick_gelu_grad(float x) {
  // d/dx [x * s(1.702 x)] = s * (1 + 1.702 x (1 - s))
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
  return s * (1.0f + 1.702f * x * (1.0f - s));
}

#define GAVA_CHECK_LAUNCH()                                   \
  do {                                                        \
    if (hipGetLastError() != hipSuccess) return GAVA_ELAUNCH; \
  } while (0)
