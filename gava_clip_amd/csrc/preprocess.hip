// Clip preprocessing on the GPU: the evaluation data path of the reference
// (video_dataset/dataset.py:117-139) from decoded uint8 RGB frames to the fp32 model input.
// One thread per output pixel and frame, all three channels: the 4 bilinear taps are 4 x 3 adjacent bytes, the
// three stores are coalesced planes.  HBM-bound: 3 B in (at most 12 touched) and 12 B out per pixel.
#include "common.h"

namespace {

struct PrepParams {
  const unsigned char* frames;
  float* out;
  long out_stride_c, out_stride_t;
  int n_frames, height, width, T, rate, size;
  int new_h, new_w, h_st, w_st, t_st;
  float scale_h, scale_w;
  float mean[3], std[3];
};

__global__ __launch_bounds__(256) void preprocess_kernel(const PrepParams p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, t = blockIdx.z;
  if (x >= p.size) return;
  int f = p.t_st + t * p.rate;
  f = f < p.n_frames ? f : p.n_frames - 1;
  // torch upsample_bilinear2d, align_corners=False (area_pixel_compute_source_index): all in fp32
  const float sy = fmaxf(p.scale_h * ((float)(y + p.h_st) + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(p.scale_w * ((float)(x + p.w_st) + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + 1 < p.height ? y0 + 1 : p.height - 1;
  const int x1 = x0 + 1 < p.width ? x0 + 1 : p.width - 1;
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const unsigned char* fr = p.frames + (size_t)f * p.height * p.width * 3;
  const unsigned char* r0 = fr + (size_t)y0 * p.width * 3;
  const unsigned char* r1 = fr + (size_t)y1 * p.width * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // (u8 / 255 - mean) / std exactly as the reference spells it (true divisions, fp32)
    const float v00 = ((float)r0[x0 * 3 + c] / 255.f - p.mean[c]) / p.std[c];
    const float v01 = ((float)r0[x1 * 3 + c] / 255.f - p.mean[c]) / p.std[c];
    const float v10 = ((float)r1[x0 * 3 + c] / 255.f - p.mean[c]) / p.std[c];
    const float v11 = ((float)r1[x1 * 3 + c] / 255.f - p.mean[c]) / p.std[c];
    const float top = __fadd_rn(__fmul_rn(lx0, v00), __fmul_rn(lx1, v01));
    const float bot = __fadd_rn(__fmul_rn(lx0, v10), __fmul_rn(lx1, v11));
    const float o = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
    p.out[c * p.out_stride_c + t * p.out_stride_t + (long)y * p.size + x] = o;
  }
}

}  // namespace

extern "C" int gava_preprocess_clip(const gava_preprocess_args* a, gava_stream_t stream) {
  if (!a || !a->frames || !a->out) return GAVA_EINVAL;
  if (a->n_frames <= 0 || a->height <= 0 || a->width <= 0 || a->T <= 0 || a->rate <= 0 || a->size <= 0) return GAVA_EINVAL;
  for (int c = 0; c < 3; ++c)
    if (!(a->std[c] > 0.f)) return GAVA_EINVAL;
  PrepParams p;
  p.frames = a->frames; p.out = a->out;
  p.out_stride_c = a->out_stride_c; p.out_stride_t = a->out_stride_t;
  p.n_frames = a->n_frames; p.height = a->height; p.width = a->width;
  p.T = a->T; p.rate = a->rate; p.size = a->size;
  // dataset.py:124-129 (integer arithmetic)
  if (a->height < a->width) { p.new_w = (int)((long)a->width * a->size / a->height); p.new_h = a->size; }
  else { p.new_h = (int)((long)a->height * a->size / a->width); p.new_w = a->size; }
  if (p.new_h < a->size || p.new_w < a->size) return GAVA_EINVAL;   // dataset.py:182 asserts the same
  p.h_st = (p.new_h - a->size) / 2; p.w_st = (p.new_w - a->size) / 2;
  if (a->first_spatial_view) {   // dataset.py:188-199: three crops along the long side, the first at offset 0
    if (p.new_h != a->size && p.new_w != a->size) return GAVA_EINVAL;   // upstream asserts min side == size
    p.h_st = 0; p.w_st = 0;
  }
  const int seg = (a->T - 1) * a->rate + 1;
  p.t_st = (a->n_frames > seg && !a->first_temporal_view) ? (a->n_frames - seg) / 2 : 0;
  p.scale_h = (float)a->height / (float)p.new_h;
  p.scale_w = (float)a->width / (float)p.new_w;
  for (int c = 0; c < 3; ++c) { p.mean[c] = a->mean[c]; p.std[c] = a->std[c]; }
  dim3 block(256), grid((a->size + 255) / 256, a->size, a->T);
  hipLaunchKernelGGL(preprocess_kernel, grid, block, 0, (hipStream_t)stream, p);
  if (hipGetLastError() != hipSuccess) return GAVA_ELAUNCH;
  return GAVA_OK;
}
