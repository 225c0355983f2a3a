// clip_pixel.h — one pixel of the evaluation data path of video_dataset/dataset.py:117-139, from decoded uint8 frames:
// (u8/255 - mean)/std, bilinear short-side resize (torch upsample_bilinear2d, align_corners=False, every step in fp32 with
// torch's own contraction pattern), centre crop.  Shared by the stand-alone preprocessing kernel and by the patch-embedding GEMM's uint8
// A-tile loader, so that both produce the same bits.
#pragma once
#include "common.h"

// a product the compiler may NOT fuse into a following add: __fmul_rn / __fadd_rn are plain operators to hipcc (it turned
// scale * (dst + 0.5) - 0.5 into one v_fma in one kernel and not in the other), an empty asm makes the value opaque
static __device__ __forceinline__ float cp_mul(float a, float b) {
  float r = a * b;
  asm volatile("" : "+v"(r));
  return r;
}

struct ClipGeom {          // the fields of gava_clip_desc the device code reads
  const unsigned char* frames; int n_frames, height, width, t_st, rate, h_st, w_st; float scale_h, scale_w;
};

// normalised value of a byte: from the caller's table when there is one (exact, and the same bits in every kernel), else the
// two true divisions of the reference's expression
static __device__ __forceinline__ float clip_norm(const float* lut, int c, unsigned char v, float mean, float stdv) {
  return lut ? lut[c * 256 + v] : __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.f), mean), stdv);
}

// Output pixel (y, x) of the size x size crop of source frame f, channel c.
// Arithmetic = torch's CPU upsample_bilinear2d (align_corners=False) bit for bit, found by trying the contraction patterns
// against the oracle (tools history, round 2): the source index is ONE fma, scale * (dst + 0.5) - 0.5, and each lerp is
// fma(w0, p0, w1 * p1) - x86 builds of torch contract exactly these.  Explicit fmaf / opaque products instead of plain
// operators: left to itself hipcc contracted the same source differently in the two kernels that use it.
// With every rounding spelled out the function can be inlined anywhere: the stand-alone preprocessing kernel and the
// patch-embedding GEMM's uint8 loader produce the same bits (tests/test_preprocess.py), and the loader's eight pixels per
// task have their 32 byte loads in flight together.
// lut: fp32 [3][256] normalised byte values, or NULL (then the reference's two true divisions with mean / stdv).
static __device__ __forceinline__ float clip_pixel1(const unsigned char* frames, int height, int width, int h_st, int w_st,
                                                              float scale_h, float scale_w, const float* lut, float mean,
                                                              float stdv, int f, int c, int y, int x) {
  const float sy = fmaxf(__builtin_fmaf(scale_h, (float)(y + h_st) + 0.5f, -0.5f), 0.f);
  const float sx = fmaxf(__builtin_fmaf(scale_w, (float)(x + w_st) + 0.5f, -0.5f), 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + 1 < height ? y0 + 1 : height - 1;
  const int x1 = x0 + 1 < width ? x0 + 1 : width - 1;
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const unsigned char* fr = frames + (size_t)f * height * width * 3;
  const unsigned char* r0 = fr + (size_t)y0 * width * 3;
  const unsigned char* r1 = fr + (size_t)y1 * width * 3;
  const float v00 = clip_norm(lut, c, r0[x0 * 3 + c], mean, stdv), v01 = clip_norm(lut, c, r0[x1 * 3 + c], mean, stdv);
  const float v10 = clip_norm(lut, c, r1[x0 * 3 + c], mean, stdv), v11 = clip_norm(lut, c, r1[x1 * 3 + c], mean, stdv);
  const float top = __builtin_fmaf(lx0, v00, cp_mul(lx1, v01)), bot = __builtin_fmaf(lx0, v10, cp_mul(lx1, v11));
  return __builtin_fmaf(ly0, top, cp_mul(ly1, bot));
}
