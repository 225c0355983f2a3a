// Clip preprocessing on the GPU: the evaluation data path of the reference
// (video_dataset/dataset.py:117-139) from decoded uint8 RGB frames to the fp32 model input.
// One thread per output pixel and frame, all three channels: the 4 bilinear taps are 4 x 3 adjacent bytes, the
// three stores are coalesced planes.  HBM-bound: 3 B in (at most 12 touched) and 12 B out per pixel.
#include "common.h"
#include "clip_pixel.h"

namespace {

struct PrepParams {
  ClipGeom g;
  float* out;
  long out_stride_c, out_stride_t;
  int T, size;
  float mean[3], std[3];
  const float* lut;
};

__global__ __launch_bounds__(256) void preprocess_kernel(const PrepParams p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, t = blockIdx.z;
  if (x >= p.size) return;
  int f = p.g.t_st + t * p.g.rate;
  f = f < p.g.n_frames ? f : p.g.n_frames - 1;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    p.out[c * p.out_stride_c + t * p.out_stride_t + (long)y * p.size + x] =
        clip_pixel1(p.g.frames, p.g.height, p.g.width, p.g.h_st, p.g.w_st, p.g.scale_h, p.g.scale_w, p.lut, p.mean[c], p.std[c], f, c, y, x);
}

}  // namespace

extern "C" int gava_clip_geometry(gava_clip_desc* d, int T, int rate, int size, int first_temporal_view, int first_spatial_view) {
  if (!d || d->n_frames <= 0 || d->height <= 0 || d->width <= 0 || T <= 0 || rate <= 0 || size <= 0) return GAVA_EINVAL;
  int new_h, new_w;
  // dataset.py:124-129 (integer arithmetic)
  if (d->height < d->width) { new_w = (int)((long)d->width * size / d->height); new_h = size; }
  else { new_h = (int)((long)d->height * size / d->width); new_w = size; }
  if (new_h < size || new_w < size) return GAVA_EINVAL;   // dataset.py:182 asserts the same
  d->h_st = (new_h - size) / 2; d->w_st = (new_w - size) / 2;
  if (first_spatial_view) {   // dataset.py:188-199: three crops along the long side, the first at offset 0
    if (new_h != size && new_w != size) return GAVA_EINVAL;   // upstream asserts min side == size
    d->h_st = 0; d->w_st = 0;
  }
  const int seg = (T - 1) * rate + 1;
  d->t_st = (d->n_frames > seg && !first_temporal_view) ? (d->n_frames - seg) / 2 : 0;
  d->rate = rate;
  d->scale_h = (float)d->height / (float)new_h;
  d->scale_w = (float)d->width / (float)new_w;
  return GAVA_OK;
}

extern "C" int gava_preprocess_clip(const gava_preprocess_args* a, gava_stream_t stream) {
  if (!a || !a->frames || !a->out) return GAVA_EINVAL;
  if (a->T <= 0 || a->size <= 0) return GAVA_EINVAL;
  for (int c = 0; c < 3; ++c)
    if (!(a->std[c] > 0.f)) return GAVA_EINVAL;
  gava_clip_desc d{};
  d.frames = a->frames; d.n_frames = a->n_frames; d.height = a->height; d.width = a->width;
  const int e = gava_clip_geometry(&d, a->T, a->rate, a->size, a->first_temporal_view, a->first_spatial_view);
  if (e != GAVA_OK) return e;
  PrepParams p;
  p.g = ClipGeom{d.frames, d.n_frames, d.height, d.width, d.t_st, d.rate, d.h_st, d.w_st, d.scale_h, d.scale_w};
  p.out = a->out; p.out_stride_c = a->out_stride_c; p.out_stride_t = a->out_stride_t;
  p.T = a->T; p.size = a->size;
  for (int c = 0; c < 3; ++c) { p.mean[c] = a->mean[c]; p.std[c] = a->std[c]; }
  p.lut = a->lut;
  dim3 block(256), grid((a->size + 255) / 256, a->size, a->T);
  hipLaunchKernelGGL(preprocess_kernel, grid, block, 0, (hipStream_t)stream, p);
  if (hipGetLastError() != hipSuccess) return GAVA_ELAUNCH;
  return GAVA_OK;
}
