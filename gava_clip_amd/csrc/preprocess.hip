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

// Patch matrix (see gava_patchify_args): one workgroup per (patch row gy, channel c, frame f): P image rows of `size`
// pixels in, g segments of P x P 16-bit values out.  VEC consecutive pixels per thread (4 when P % 4 == 0: 16-byte fp32
// loads, 8-byte stores; 2 for P = 14): a vector never straddles a patch.
struct PatchifyParams {
  const float* x; const gava_clip_desc* clips; const float* lut;
  unsigned short* out; long ldo;
  int T, size, P, g;
};

template <class Pr, int VEC>
__global__ __launch_bounds__(256) void patchify_kernel(const PatchifyParams p) {
  const int gy = blockIdx.x, c = blockIdx.y, f = blockIdx.z;
  const int b = f / p.T, t = f - b * p.T;
  const int n_vec = p.P * p.size / VEC, PP = p.P * p.P;
  const float* plane = nullptr;
  ClipGeom cg{};
  int fsrc = 0;
  if (p.x) plane = p.x + (((long)b * 3 + c) * p.T + t) * p.size * p.size + (long)gy * p.P * p.size;
  else {
    const gava_clip_desc d = p.clips[b];
    cg = ClipGeom{d.frames, d.n_frames, d.height, d.width, d.t_st, d.rate, d.h_st, d.w_st, d.scale_h, d.scale_w};
    fsrc = d.t_st + t * d.rate;
    fsrc = fsrc < d.n_frames ? fsrc : d.n_frames - 1;
  }
  unsigned short* orow0 = p.out + ((long)f * p.g * p.g + (long)gy * p.g) * p.ldo + c * PP;
  for (int i = threadIdx.x; i < n_vec; i += 256) {
    const int e0 = i * VEC, iy = e0 / p.size, x0 = e0 - iy * p.size;
    float v[VEC];
    if (p.x) {
      if (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(plane + e0);
        v[0] = q.x; v[1] = q.y; v[VEC - 2] = q.z; v[VEC - 1] = q.w;
      } else if (VEC == 2) {
        const float2 q = *reinterpret_cast<const float2*>(plane + e0);
        v[0] = q.x; v[1] = q.y;
      } else {
        v[0] = plane[e0];      // VEC == 1: a clip tensor at an odd storage offset (no alignment requirement at all)
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        v[e] = clip_pixel1(cg.frames, cg.height, cg.width, cg.h_st, cg.w_st, cg.scale_h, cg.scale_w, p.lut, 0.f, 1.f, fsrc, c,
                           gy * p.P + iy, x0 + e);
    }
    const int px = x0 / p.P, ix = x0 - px * p.P;
    unsigned short* o = orow0 + (long)px * p.ldo + iy * p.P + ix;
    if (VEC == 4) *reinterpret_cast<uint2*>(o) = pack4<Pr>(v[0], v[1], v[VEC - 2], v[VEC - 1]);
    else if (VEC == 2) *reinterpret_cast<unsigned*>(o) = Pr::cvt2(v[0], v[VEC - 1]);
    else *o = Pr::cvt(v[0]);
  }
  // zero the K padding of this patch row once (the c == 0 workgroup)
  const int pad = (int)p.ldo - 3 * PP;
  if (c == 0 && pad > 0) {
    for (int i = threadIdx.x; i < p.g * pad; i += 256) {
      const int px = i / pad, k = i - px * pad;
      p.out[((long)f * p.g * p.g + (long)gy * p.g + px) * p.ldo + 3 * PP + k] = 0;
    }
  }
}

}  // namespace

extern "C" int gava_patchify(const gava_patchify_args* a, gava_stream_t stream) {
  if (!a || !a->out || (!a->x && !a->clips) || (a->x && a->clips)) return GAVA_EINVAL;
  if (a->clips && !a->clip_lut) return GAVA_EINVAL;
  if (a->B <= 0 || a->T <= 0 || a->size <= 0 || a->patch <= 0 || a->size % a->patch || a->patch % 2) return GAVA_EINVAL;
  if (a->ldo < 3 * a->patch * a->patch || a->ldo % 8 || ((uintptr_t)a->out & 15)) return GAVA_EINVAL;
  if (a->x && ((uintptr_t)a->x & 3)) return GAVA_EINVAL;
  PatchifyParams p;
  p.x = a->x; p.clips = a->clips; p.lut = a->clip_lut;
  p.out = (unsigned short*)a->out; p.ldo = a->ldo;
  p.T = a->T; p.size = a->size; p.P = a->patch; p.g = a->size / a->patch;
  dim3 grid(p.g, 3, a->B * a->T), block(256);
  // vector width of the fp32 loads: 4 / 2 floats when the patch size, the row length and the clips' address allow it, else
  // one float at a time (a clip tensor that is a view at an odd offset of a larger storage: slower, never rejected)
  const bool v4 = a->patch % 4 == 0 && (!a->x || (((uintptr_t)a->x & 15) == 0 && a->size % 4 == 0));
  const bool v2 = !a->x || (((uintptr_t)a->x & 7) == 0 && a->size % 2 == 0);
  hipStream_t s = (hipStream_t)stream;
  if (a->prec == GAVA_PREC_F16) {
    if (v4) hipLaunchKernelGGL((patchify_kernel<PrecF16, 4>), grid, block, 0, s, p);
    else if (v2) hipLaunchKernelGGL((patchify_kernel<PrecF16, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((patchify_kernel<PrecF16, 1>), grid, block, 0, s, p);
  } else if (a->prec == GAVA_PREC_BF16) {
    if (v4) hipLaunchKernelGGL((patchify_kernel<PrecBF16, 4>), grid, block, 0, s, p);
    else if (v2) hipLaunchKernelGGL((patchify_kernel<PrecBF16, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((patchify_kernel<PrecBF16, 1>), grid, block, 0, s, p);
  } else return GAVA_EINVAL;
  if (hipGetLastError() != hipSuccess) return GAVA_ELAUNCH;
  return GAVA_OK;
}

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
