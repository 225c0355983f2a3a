// gemm.hip — C[M,N] = A[M,K]·W[N,K]^T, 16-bit operands (fp16|bf16), fp32 accumulate, fused epilogue.
//
// gfx950 design (v1, "2-phase"):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
//     4x4 MFMA 16x16x32 accumulators), BK = 64, two LDS stages of (16 KiB A + 16 KiB W) = 64 KiB
//     -> two workgroups per CU, one k-tile of prefetch in flight behind the MFMAs.
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging).  The LDS image
//     is lane-linear per wave instruction, so the bank-conflict swizzle (16-byte chunk index
//     XOR ((row>>1)&7)) is applied to the per-lane SOURCE address and again on the ds_read_b128.
//   * MFMA operands are swapped (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends with 4 consecutive output COLUMNS of one row: 16-byte fp32 / 8-byte h16 stores
//     and one float4 bias load per accumulator.
//   * workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of
//     tiles, n-tile fastest: the A row-panel is fetched from HBM once per XCD and W stays in L2.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W

struct GemmParams {
  const unsigned short* A; long lda;
  const unsigned short* W; long ldw;
  const float* bias;
  void* out; long ldo;
  const float* resid; long ldr;
  int M, N, K;
  int scale_cols; float scale;
  const float* pos; const float* time; int n_patches; int T;
  int tiles_n, n_tiles;
  int split_out;
};

// issue the 4+4 global_load_lds_dwordx4 of this wave for one k-tile
static __device__ __forceinline__ void stage_tile(const unsigned short* const (&srcA)[4],
                                                  const unsigned short* const (&srcW)[4],
                                                  long koff, char* lds_stage, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    char* dstA = lds_stage + (wave * 4 + i) * 1024;
    char* dstW = dstA + TILE_BYTES;
    __builtin_amdgcn_global_load_lds(GLB_PTR(srcA[i] + koff), LDS_PTR(void, dstA), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(srcW[i] + koff), LDS_PTR(void, dstW), 16, 0, 0);
  }
}

template <class P, int EPI, bool RES, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware, bijective block -> tile map (8 XCDs, blocks dealt round-robin)
  const int nwg = p.n_tiles;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int nt = wg % p.tiles_n, mt = wg / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  // per-lane source pointers for the staging loads (swizzle on the source side)
  const unsigned short* srcA[4];
  const unsigned short* srcW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    srcA[i] = p.A + (long)gm * p.lda + chunk * 8;
    srcW[i] = p.W + (long)(n0 + row) * p.ldw + chunk * 8;
  }

  // per-lane fragment read offsets (bytes) inside a tile
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr >> 1;
  const int offk0 = ((fg ^ swz) << 4), offk1 = (((4 + fg) ^ swz) << 4);
  const int a_row_off = (wr * 64 + fr) * 128;
  const int w_row_off = (wc * 64 + fr) * 128 + TILE_BYTES;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile(srcA, srcW, 0, smem, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    if (kt + 1 < nk) stage_tile(srcA, srcW, (long)(kt + 1) * BK, smem + ((kt + 1) & 1) * STAGE_BYTES, wave);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int offk = kk ? offk1 : offk0;
      s16x8_t af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const s16x8_t*>(cur + a_row_off + i * 2048 + offk);
        wf[i] = *reinterpret_cast<const s16x8_t*>(cur + w_row_off + i * 2048 + offk);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(wf[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane holds out[m][n .. n+3], m = m0+wr*64+i*16+fr, n = n0+wc*64+j*16+4*fg
  const int nbase = n0 + wc * 64 + 4 * fg;
  float4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bj[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nbase + j * 16) : make_float4(0, 0, 0, 0);

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + fr;
    if (m >= p.M) continue;
    long orow = m;
    const float* posr = nullptr;
    const float* timr = nullptr;
    if (EPI == GAVA_EPI_F32_PATCH) {
      const int frame = m / p.n_patches, pp = m - frame * p.n_patches;
      orow = (long)frame * (p.n_patches + 1) + 1 + pp;
      posr = p.pos + (long)(1 + pp) * p.N;
      timr = p.time + (long)(frame % p.T) * p.N;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nbase + j * 16;
      float v0 = acc[i][j][0] + bj[j].x, v1 = acc[i][j][1] + bj[j].y;
      float v2 = acc[i][j][2] + bj[j].z, v3 = acc[i][j][3] + bj[j].w;
      if (EPI == GAVA_EPI_H16 || EPI == GAVA_EPI_H16_QGELU) {
        if (EPI == GAVA_EPI_H16) {
          if (n < p.scale_cols) { v0 *= p.scale; v1 *= p.scale; v2 *= p.scale; v3 *= p.scale; }
        } else {
          v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3);
        }
        unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + orow * p.ldo + n;
        if (SPLIT) {
          uint2 hi, lo;
          split4<P>(v0, v1, v2, v3, hi, lo);
          *reinterpret_cast<uint2*>(o) = hi;
          *reinterpret_cast<uint2*>(o + p.N) = lo;
          *reinterpret_cast<uint2*>(o + 2 * p.N) = hi;
        } else {
          *reinterpret_cast<uint2*>(o) = pack4<P>(v0, v1, v2, v3);
        }
      } else if (EPI == GAVA_EPI_F32) {
        if (RES) {
          const float4 rr = *reinterpret_cast<const float4*>(p.resid + orow * p.ldr + n);
          v0 += rr.x; v1 += rr.y; v2 += rr.z; v3 += rr.w;
        }
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) =
            make_float4(v0, v1, v2, v3);
      } else {  // GAVA_EPI_F32_PATCH
        const float4 pr = *reinterpret_cast<const float4*>(posr + n);
        const float4 tr = *reinterpret_cast<const float4*>(timr + n);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) =
            make_float4(v0 + pr.x + tr.x, v1 + pr.y + tr.y, v2 + pr.z + tr.z, v3 + pr.w + tr.w);
      }
    }
  }
}

template <class P>
int launch_prec(const GemmParams& gp, int epi, hipStream_t s) {
  dim3 grid(gp.n_tiles), block(256);
  switch (epi) {
    case GAVA_EPI_H16:
      if (gp.split_out) hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_H16, false, true>), grid, block, 0, s, gp);
      else hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_H16, false, false>), grid, block, 0, s, gp);
      break;
    case GAVA_EPI_H16_QGELU:
      if (gp.split_out) hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_H16_QGELU, false, true>), grid, block, 0, s, gp);
      else hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_H16_QGELU, false, false>), grid, block, 0, s, gp);
      break;
    case GAVA_EPI_F32:
      if (gp.resid) hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_F32, true, false>), grid, block, 0, s, gp);
      else hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_F32, false, false>), grid, block, 0, s, gp);
      break;
    case GAVA_EPI_F32_PATCH: hipLaunchKernelGGL((gemm_kernel<P, GAVA_EPI_F32_PATCH, false, false>), grid, block, 0, s, gp); break;
    default: return GAVA_EINVAL;
  }
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

}  // namespace

extern "C" int gava_gemm(const gava_gemm_args* a, gava_stream_t stream) {
  if (!a || !a->A || !a->W || !a->out) return GAVA_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return GAVA_EINVAL;
  if (a->N % BN || a->K % BK) return GAVA_EINVAL;
  if (a->lda % 8 || a->ldw % 8 || a->lda < a->K || a->ldw < a->K) return GAVA_EINVAL;
  if (((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->out) & 15) return GAVA_EINVAL;
  if (a->bias && ((uintptr_t)a->bias & 15)) return GAVA_EINVAL;
  if (a->ldo % 4 || a->ldo < (a->split_out ? 3 : 1) * (int64_t)a->N) return GAVA_EINVAL;
  if (a->split_out && a->epilogue != GAVA_EPI_H16 && a->epilogue != GAVA_EPI_H16_QGELU) return GAVA_EINVAL;
  if (a->epilogue == GAVA_EPI_F32 && a->resid && (a->ldr % 4 || ((uintptr_t)a->resid & 15))) return GAVA_EINVAL;
  if (a->epilogue == GAVA_EPI_F32_PATCH &&
      (!a->pos || !a->time || a->n_patches <= 0 || a->T <= 0 || a->M % a->n_patches)) return GAVA_EINVAL;
  GemmParams gp;
  gp.A = (const unsigned short*)a->A; gp.lda = a->lda;
  gp.W = (const unsigned short*)a->W; gp.ldw = a->ldw;
  gp.bias = a->bias; gp.out = a->out; gp.ldo = a->ldo;
  gp.resid = a->resid; gp.ldr = a->ldr;
  gp.M = a->M; gp.N = a->N; gp.K = a->K;
  gp.scale_cols = a->scale_cols; gp.scale = a->scale;
  gp.pos = a->pos; gp.time = a->time; gp.n_patches = a->n_patches; gp.T = a->T;
  gp.split_out = a->split_out;
  gp.tiles_n = a->N / BN;
  gp.n_tiles = gp.tiles_n * ((a->M + BM - 1) / BM);
  hipStream_t s = (hipStream_t)stream;
  if (a->prec == GAVA_PREC_F16) return launch_prec<PrecF16>(gp, a->epilogue, s);
  if (a->prec == GAVA_PREC_BF16) return launch_prec<PrecBF16>(gp, a->epilogue, s);
  return GAVA_EINVAL;
}
