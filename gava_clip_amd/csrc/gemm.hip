// gemm.hip — C[M,N] = A[M,K]·W[N,K]^T, 16-bit operands (fp16|bf16), fp32 accumulate, fused epilogue.
//
// gfx950 design (v2):
//   * BM x BN output tile per workgroup, one wave per 64x64 sub-tile (4x4 MFMA 16x16x32
//     accumulators, 64 VGPRs).  Default 256x128 -> 8 waves, two per SIMD, one workgroup per CU.
//   * operands go L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging) into an NST-deep ring
//     of BK = 64 stages (3 x 48 KiB).  Loads run NST-1 k-tiles ahead of the MFMAs behind a
//     COUNTED s_waitcnt vmcnt(N) and a raw s_barrier (one barrier per k-tile; __syncthreads()
//     would drain the ring with vmcnt(0)).
//   * the LDS image of a glds is lane-linear, so the bank-conflict swizzle (16-byte chunk index
//     XOR ((row>>1)&7), measured SQ_LDS_BANK_CONFLICT = 0) sits on the per-lane SOURCE address and
//     again on the ds_read_b128.
//   * MFMA operands are swapped (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends with 4 consecutive output COLUMNS of one row: 16-byte fp32 / 8-byte h16 stores
//     and one float4 bias load per accumulator.
//   * workgroup -> tile map: each XCD (private 4 MiB L2) walks a contiguous range of tiles, and
//     inside it tiles are ordered in SM x SN super-tiles whose operand panels fit that L2.
#include "common.h"
#include "clip_pixel.h"
#include "internal.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

// 1: set_src's lane-dependent terms are recomputed per tile instead of hoisted out of the tile loop (35 VGPRs less: the
// FOLD kernels need that to stay spill-free; measured +1 % on fc2, neutral on fc1 / qkv).  0 = hoisted, for A/B builds.
#ifndef GAVA_V3_RECOMPUTE_SRC
#define GAVA_V3_RECOMPUTE_SRC 1
#endif
#ifndef GAVA_V3_NAT
#define GAVA_V3_NAT 1   // 0: permuted column order also for the fp32-output kernels (A/B builds)
#endif
#ifndef GAVA_V3_PRIO
#define GAVA_V3_PRIO 1
#endif

#ifndef GAVA_L8_SCHED
#define GAVA_L8_SCHED 1
#endif
namespace {

#ifdef GAVA_ENABLE_ABLATE
#define ABLATE p.ablate
#else
#define ABLATE 0
#endif
constexpr int BK = 64;
// PP2 (default): the ping-pong loop in two 32-MFMA phases per k-tile (half the barriers) instead of four 16-MFMA ones - plain GEMMs
// 8192^3 1368 -> 1445 TF/s, the fc1 shape 1022 -> 1076 (vendor 1065), K = 3072 1151 -> 1221 (vendor 1067); c2 forward -1.1 %, c5 -3.1 %
// (profiles/r04_pingpong.txt).  -DGAVA_PP2=0: the four-phase loop (A/B builds)
#ifndef GAVA_PP2
#define GAVA_PP2 1
#endif
// s_waitcnt immediate for "vmcnt(n) only" on gfx9/CDNA: vmcnt[3:0] | expcnt 7 | lgkmcnt 15 | vmcnt[5:4] << 14

struct GemmParams {
  const unsigned short* A; long lda;
  const unsigned short* W; long ldw;
  const float* bias;
  void* out; long ldo;
  const float* resid; long ldr;
  const unsigned short* aux;   // EPI_H16_QGELU_BWD: pre-activations, laid out as out
  int aux_f16;                 // ... stored as fp16 (else bf16), independent of the operand precision
  unsigned short* aux_out;     // EPI_H16_QGELU: optional copy of the pre-activation (training forward)
  // LayerNorm folding: producer (EPI_F32) extras and consumer (EPI_H16*) inputs, see gava_hip.h
  unsigned short* x16; long ldx16; float2* rowsum;
  const float2* fstats; const float* fs; const float* ft;
  int rowsum_reduced;          // producer (persistent kernel): rowsum is [rows][4] float2, one slot per 256-column tile
  const float2* fpart;         // consumer (persistent FOLD kernel): those partials; it makes (mean, rstd) itself
  int fold_slots; float fold_inv_d;
  int M, N, K;
  int scale_cols; float scale;
  const float* pos; const float* time; int n_patches; int T;
  const float* frames; int fsize; int patch;   // im2col-free patch A operand (EPI_F32_PATCH, A == nullptr)
  const gava_clip_desc* clips; const float* clut;   // ... from decoded uint8 frames (gava_clip_desc, fp32 [3][256] table)
  int tiles_m, tiles_n, n_tiles;
  int split_out;
  int sm, sn;   // super-tile shape (in tiles)
  unsigned long long* dbg;  // debug only: per-wave segment cycle sums (gava_debug_set_buffer)
  int cu_reserve;   // persistent kernels: CUs left out of the grid (gava_gemm_args.cu_reserve)
  int kernel;       // gava_gemm_args.kernel
  // experiment builds only (each switch has its own field; all 0 in the product library)
  int pair_delay;   // -DGAVA_ENABLE_ABLATE / -DGAVA_EXP_STAGGER: start delay of a CU's second workgroup / of the late slots, 10 ns ticks
  int pair_sleep;   // -DGAVA_ENABLE_ABLATE: s_sleep units after every epilogue chunk of the pair kernel
  int stagger_mode; // -DGAVA_EXP_STAGGER: which workgroups start late (0 odd slots, 1 (slot & 3) * delay / 2, 2 XCDs 4-7)
  int operand_l2;   // -DGAVA_EXP_OPERAND_L2: bit 0 every tile reads the same A rows, bit 1 the same W rows (results wrong)
  int ablate;   // timing probes, always 0 unless built with -DGAVA_ENABLE_ABLATE: 1 = no staging loads after the prologue,
                // 2 = no LDS reads/MFMA, 4 = no epilogue
  // weight-lo pass (gava_gemm_args.w_lo).  1: W rows are [W_hi | W_lo], the k-loop runs 2 K / BK stages and the A operand
  // wraps after nka = K / BK of them
  int w_lo, nka;
  // 2 (persistent 256^2 kernel, L8 instantiations): after the 16-bit stages nk8 = K / 128 stages of the 8-bit lo product:
  // A8 = bf8 rows of A (2 lda bytes apart), W8 = e4m3 rows of 2^w8_exp W_lo (2 ldw bytes apart); scale8 = the E8M0 byte
  // 127 - w8_exp in every byte (the block-scaled MFMA multiplies the W8 operand by 2^-w8_exp)
  const unsigned char* A8; const unsigned char* W8; int nk8; unsigned scale8;
  unsigned char* out8; long ldo8;   // L8, 16-bit-output epilogues: bf8 copy of the output rows
  unsigned char* x8; long ldx8;     // L8, producers: bf8 copy of x16
  // HL instantiations (gava_gemm_args.resid16): the residual stream as a 16-bit pair - r16 / rlo in (pitch ldr), x16 / xlo out
  // (pitch ldx16); hi in the operand type, lo = fp16(x - hi)
  const unsigned short* r16; const unsigned short* rlo; unsigned short* xlo;
};

static __device__ __forceinline__ float aux_up(unsigned short u, int f16) {
  return f16 ? PrecF16::up(u) : PrecBF16::up(u);
}

template <class P, int EPI, bool RES, bool SPLIT, int BM, int BN, int NST>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, 2)
void gemm_kernel(const GemmParams p) {
  constexpr int NW = (BM / 64) * (BN / 64);       // waves
  constexpr int WN = BN / 64;                     // waves along n
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int PIECES = (BM + BN) / 8;           // 1 KiB glds pieces (8 rows x 128 B) per stage
  constexpr int PPW = PIECES / NW;                // pieces per wave
  static_assert(PIECES % NW == 0, "pieces must divide over the waves");
  static_assert(NST >= 2 && NST <= 4, "ring depth");
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;

  // ---- block -> tile: XCD-contiguous (bijective for any grid), then SM x SN super-tiles
  int mt, nt;
  {
    const int nwg = p.n_tiles, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = p.sm * p.tiles_n;
    const int g = wg / per_group;
    const int first_m = g * p.sm;
    const int sm = min(p.sm, p.tiles_m - first_m);
    const int w = wg - g * per_group;
    const int chunk = w / (sm * p.sn), rr = w - chunk * (sm * p.sn);
    mt = first_m + rr % sm;
    nt = chunk * p.sn + rr / sm;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const bool direct = EPI == GAVA_EPI_F32_PATCH && (p.frames != nullptr || p.clips != nullptr);
  // ---- per-lane source pointers of this wave's staging pieces (swizzle on the source side)
  const unsigned short* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wave + i * NW;
    const int row = piece * 8 + (lane >> 3);          // row in the stacked [A;W] tile
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (piece < BM / 8) {
      int gm = m0 + row;
      gm = gm < p.M ? gm : p.M - 1;
      src[i] = direct ? p.W : p.A + (long)gm * p.lda + chunk * 8;   // unused in direct mode
    } else {
      src[i] = p.W + (long)(n0 + row - BM) * p.ldw + chunk * 8;
    }
  }
  auto stage = [&](int kt) {
    char* base = smem + (kt % NST) * STAGE;
    const int kta = kt >= p.nka ? kt - p.nka : kt;   // w_lo: the second half of the k-loop re-reads A
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + i * NW;
      if (direct && piece < BM / 8) continue;     // A tile comes from stage_patch_a()
      __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + (long)(piece < BM / 8 ? kta : kt) * BK), LDS_PTR(void, base + piece * 1024), 16, 0, 0);
    }
  };
  // im2col-free patch operand: task = (row of the A tile, 16-byte chunk of 8 consecutive k).  k = (c,ky,kx);
  // for P % 8 == 0 a chunk is 8 consecutive pixels of one image row: two coalesced float4 loads.
  // Split staging (issue early / write late): patch_load() puts the global loads in flight before the MFMAs
  // of the current k-tile, patch_write() converts and writes the LDS tile after them.
  constexpr int NTASK = BM * 8 / (NW * 64);
  float pe[NTASK][8];
  auto patch_load = [&](int kt_) {
    const int kt = kt_ >= p.nka ? kt_ - p.nka : kt_;
    const int P2 = p.patch * p.patch, Kreal = 3 * P2, g = p.fsize / p.patch;
#pragma unroll
    for (int ti = 0; ti < NTASK; ++ti) {
      const int task = tid + ti * NW * 64;
      const int row = task >> 3, chunk = task & 7;
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      const int frame = m / p.n_patches, pp = m - frame * p.n_patches;
      const int b = frame / p.T, t = frame - b * p.T;
      const int py = pp / g, px = pp - py * g;
      const int k = kt * BK + chunk * 8;
      if (EPI == GAVA_EPI_F32_PATCH && p.clips) {
        // uint8 source: every element is one pixel of the (virtual) preprocessed clip - temporal crop, normalisation,
        // bilinear resize and centre crop evaluated here (clip_pixel.h), nothing but the decoded frames is read from HBM
        const gava_clip_desc& cd = p.clips[b];
        int f = cd.t_st + t * cd.rate;
        f = f < cd.n_frames ? f : cd.n_frames - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kq = k + q;
          float v = 0.f;
          if (kq < Kreal) {
            const int c = kq / P2, rem = kq - c * P2, ky = rem / p.patch, kx = rem - ky * p.patch;
            v = clip_pixel1(cd.frames, cd.height, cd.width, cd.h_st, cd.w_st, cd.scale_h, cd.scale_w, p.clut, 0.f, 1.f, f, c,
                            py * p.patch + ky, px * p.patch + kx);
          }
          pe[ti][q] = v;
        }
      } else if ((p.patch & 7) == 0 && k + 8 <= Kreal) {
        const int c = k / P2, rem = k - c * P2, ky = rem / p.patch, kx = rem - ky * p.patch;
        const float* sp = p.frames + ((((long)b * 3 + c) * p.T + t) * p.fsize + (py * p.patch + ky)) * p.fsize + px * p.patch + kx;
        const float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 4);
        pe[ti][0] = lo.x; pe[ti][1] = lo.y; pe[ti][2] = lo.z; pe[ti][3] = lo.w;
        pe[ti][4] = hi.x; pe[ti][5] = hi.y; pe[ti][6] = hi.z; pe[ti][7] = hi.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kq = k + q;
          float v = 0.f;
          if (kq < Kreal) {
            const int c = kq / P2, rem = kq - c * P2, ky = rem / p.patch, kx = rem - ky * p.patch;
            v = p.frames[((((long)b * 3 + c) * p.T + t) * p.fsize + (py * p.patch + ky)) * p.fsize + px * p.patch + kx];
          }
          pe[ti][q] = v;
        }
      }
    }
  };
  auto patch_write = [&](int kt) {
    char* base = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int ti = 0; ti < NTASK; ++ti) {
      const int task = tid + ti * NW * 64;
      const int row = task >> 3, chunk = task & 7;
      const uint2 x = pack4<P>(pe[ti][0], pe[ti][1], pe[ti][2], pe[ti][3]), y = pack4<P>(pe[ti][4], pe[ti][5], pe[ti][6], pe[ti][7]);
      *reinterpret_cast<uint4*>(base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = make_uint4(x.x, x.y, y.x, y.y);
    }
  };

  // per-lane fragment read offsets (bytes) inside a stage
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr >> 1;
  const int offk0 = ((fg ^ swz) << 4), offk1 = (((4 + fg) ^ swz) << 4);
  const int a_row_off = (wr * 64 + fr) * 128;
  const int w_row_off = (wc * 64 + fr) * 128 + A_BYTES;

  const int nk = p.w_lo == 1 ? 2 * p.nka : p.nka;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nk) { stage(t); if (direct) { patch_load(t); patch_write(t); } }

  // Accumulators start at the residual tile (out = resid + A W^T accumulates in fp32 on top of it): the
  // residual read is in flight while the first operand stage lands, instead of being a serialised
  // load -> add -> store chain in the epilogue when every register still holds an accumulator.
  f32x4_t acc[4][4];
  if (EPI == GAVA_EPI_F32 && RES) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + wr * 64 + i * 16 + fr;
      m = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = *reinterpret_cast<const f32x4_t*>(p.resid + (long)m * p.ldr + n0 + wc * 64 + 4 * fg + j * 16);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; the tiles issued after it (up to NST-2 of them) may stay in flight
    const int ahead = min(nk, kt + NST - 1) - (kt + 1);
    if (NST == 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (NST >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (direct) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's A-tile ds_writes are done
    __builtin_amdgcn_s_barrier();
    const bool refill = kt + NST - 1 < nk && !(ABLATE & 1);   // refills the slot every wave finished reading at kt-1
    if (refill) {
      stage(kt + NST - 1);
      if (direct) patch_load(kt + NST - 1);
    }
    const char* cur = smem + (kt % NST) * STAGE;
    if (ABLATE & 2) continue;
    s16x8_t af[2][4], wf[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int offk = kk ? offk1 : offk0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[kk][i] = *reinterpret_cast<const s16x8_t*>(cur + a_row_off + i * 2048 + offk);
        wf[kk][i] = *reinterpret_cast<const s16x8_t*>(cur + w_row_off + i * 2048 + offk);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(wf[kk][j], af[kk][i], acc[i][j]);
    // all 16 LDS reads are issued before the first MFMA: the second half's latency hides under the
    // first half's MFMAs (the compiler emits counted lgkmcnt waits in issue order)
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
    if (direct && refill) patch_write(kt + NST - 1);
  }

  if (ABLATE & 4) return;
  // ---- epilogue: lane holds out[m][n .. n+3], m = m0+wr*64+i*16+fr, n = n0+wc*64+j*16+4*fg
  const int nbase = n0 + wc * 64 + 4 * fg;
  float4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bj[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nbase + j * 16) : make_float4(0, 0, 0, 0);
  // one explicit use on the common path, so that the bias is waited for once and not again (with vmcnt(0),
  // i.e. for the previous row's stores) in every `m < M` block below
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(bj[j].x), "v"(bj[j].y), "v"(bj[j].z), "v"(bj[j].w));
  // folded LayerNorm (consumer): out = rstd * (acc - mean * s_n) + t_n; t_n arrives in the bias registers
  constexpr bool CAN_FOLD = EPI == GAVA_EPI_H16 || EPI == GAVA_EPI_H16_QGELU;
  const bool fold = CAN_FOLD && p.fstats != nullptr;
  float4 sj[4];
  float2 rs[4];
  if (fold) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sj[j] = *reinterpret_cast<const float4*>(p.fs + nbase + j * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + fr;
      rs[i] = p.fstats[m < p.M ? m : p.M - 1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" ::"v"(sj[j].x), "v"(sj[j].y), "v"(sj[j].z), "v"(sj[j].w));
      asm volatile("" ::"v"(rs[j].x), "v"(rs[j].y));
    }
  }

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
    float ps1 = 0.f, ps2 = 0.f;   // producer side of the folding: this lane's share of sum x, sum x^2 of the row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nbase + j * 16;
      float v0 = acc[i][j][0] + bj[j].x, v1 = acc[i][j][1] + bj[j].y;
      float v2 = acc[i][j][2] + bj[j].z, v3 = acc[i][j][3] + bj[j].w;
      if (fold) {
        const float r = rs[i].y, mr = -rs[i].x * rs[i].y;
        v0 = fmaf(acc[i][j][0], r, fmaf(mr, sj[j].x, bj[j].x)); v1 = fmaf(acc[i][j][1], r, fmaf(mr, sj[j].y, bj[j].y));
        v2 = fmaf(acc[i][j][2], r, fmaf(mr, sj[j].z, bj[j].z)); v3 = fmaf(acc[i][j][3], r, fmaf(mr, sj[j].w, bj[j].w));
      }
      if (EPI == GAVA_EPI_H16 || EPI == GAVA_EPI_H16_QGELU || EPI == GAVA_EPI_H16_QGELU_BWD) {
        if (EPI == GAVA_EPI_H16) {
          if (n < p.scale_cols) { v0 *= p.scale; v1 *= p.scale; v2 *= p.scale; v3 *= p.scale; }
        } else if (EPI == GAVA_EPI_H16_QGELU) {
          if (p.aux_out) *reinterpret_cast<uint2*>(p.aux_out + orow * p.ldo + n) = pack4<P>(v0, v1, v2, v3);
          quick_gelu2(v0, v1); quick_gelu2(v2, v3);
        } else {
          const uint2 ax = *reinterpret_cast<const uint2*>(p.aux + orow * p.ldo + n);
          v0 *= quick_gelu_grad(aux_up((unsigned short)ax.x, p.aux_f16)); v1 *= quick_gelu_grad(aux_up((unsigned short)(ax.x >> 16), p.aux_f16));
          v2 *= quick_gelu_grad(aux_up((unsigned short)ax.y, p.aux_f16)); v3 *= quick_gelu_grad(aux_up((unsigned short)(ax.y >> 16), p.aux_f16));
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
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) =
            make_float4(v0, v1, v2, v3);
        if (p.x16) {
          *reinterpret_cast<uint2*>(p.x16 + orow * p.ldx16 + n) = pack4<P>(v0, v1, v2, v3);
          ps1 += (v0 + v1) + (v2 + v3);
          ps2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
      } else {  // GAVA_EPI_F32_PATCH
        const float4 pr = *reinterpret_cast<const float4*>(posr + n);
        const float4 tr = *reinterpret_cast<const float4*>(timr + n);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) =
            make_float4(v0 + pr.x + tr.x, v1 + pr.y + tr.y, v2 + pr.z + tr.z, v3 + pr.w + tr.w);
      }
    }
    if (EPI == GAVA_EPI_F32 && p.x16) {
      // the 4 lanes that share this row (same fr, fg = 0..3) cover the wave's 64 columns: one partial per row and wave
      ps1 = sum_across_lane_groups(ps1); ps2 = sum_across_lane_groups(ps2);
      if (fg == 0) p.rowsum[orow * (p.N / 64) + (n0 + wc * 64) / 64] = make_float2(ps1, ps2);
    }
  }
}

template <class P, int BM, int BN, int NST>
int launch_tile(GemmParams gp, int epi, hipStream_t s) {
  gp.tiles_m = (gp.M + BM - 1) / BM;
  gp.tiles_n = gp.N / BN;
  gp.n_tiles = gp.tiles_m * gp.tiles_n;
  // super-tile: ~32 concurrent tiles per XCD; keep (sm A-panels + sn W-panels) inside the 4 MiB L2
  gp.sn = gp.tiles_n < 8 ? gp.tiles_n : 8;
  gp.sm = 32 / gp.sn < 1 ? 1 : 32 / gp.sn;
  if (gp.sm > gp.tiles_m) gp.sm = gp.tiles_m;
  dim3 grid(gp.n_tiles), block((BM / 64) * (BN / 64) * 64);
#define GAVA_LAUNCH(EPI, RES, SPLIT) \
  hipLaunchKernelGGL((gemm_kernel<P, EPI, RES, SPLIT, BM, BN, NST>), grid, block, 0, s, gp)
  switch (epi) {
    case GAVA_EPI_H16:
      if (gp.split_out) GAVA_LAUNCH(GAVA_EPI_H16, false, true); else GAVA_LAUNCH(GAVA_EPI_H16, false, false);
      break;
    case GAVA_EPI_H16_QGELU:
      if (gp.split_out) GAVA_LAUNCH(GAVA_EPI_H16_QGELU, false, true); else GAVA_LAUNCH(GAVA_EPI_H16_QGELU, false, false);
      break;
    case GAVA_EPI_F32:
      if (gp.resid) GAVA_LAUNCH(GAVA_EPI_F32, true, false); else GAVA_LAUNCH(GAVA_EPI_F32, false, false);
      break;
    case GAVA_EPI_F32_PATCH: GAVA_LAUNCH(GAVA_EPI_F32_PATCH, false, false); break;
    case GAVA_EPI_H16_QGELU_BWD: GAVA_LAUNCH(GAVA_EPI_H16_QGELU_BWD, false, false); break;
    default: return GAVA_EINVAL;
  }
#undef GAVA_LAUNCH
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}


// ---------------------------------------------------------------------------------------------
// v3: persistent 256x256 kernel for the big-M GEMMs.
//   * 8 waves as 2(M) x 4(N), 128x64 outputs per wave (8x4 accumulators = 128 VGPRs); BK = 64,
//     2 LDS stages of 64 KiB.  Per FLOP it pulls half the operand bytes of the 128^2 tile through
//     L2->LDS (the measured limiter of v2) and needs 0.375 instead of 0.5 ds_read_b128 per MFMA.
//   * persistent: gridDim = #CUs; each workgroup walks its XCD's tile range with the k-loop
//     flattened across tiles, so the first stage of the next tile is already in flight during
//     the epilogue, and the epilogue's stores drain behind a COUNTED vmcnt while the next tile
//     computes (nothing waits on them until one k-tile later).
//   * W rows are read from LDS in a permuted order (row = 16*(fr>>2) + 4*j + (fr&3)) so that the
//     4 accumulators of an output row hold 16 CONSECUTIVE columns per lane: 32-byte h16 /
//     64-byte fp32 contiguous per lane, whole 128-byte lines per 4 lanes, no cross-lane shuffles.
//     The W tile uses its own bank swizzle (bits 1,4,5 of the row) that is conflict-free for
//     that read pattern.
// ALIGN (round 3; the fp32-output instantiations): rounds of an XCD coincide with super-tiles.  An XCD's workgroups then
// always work on ONE sm x sn block of tiles at a time (slot s = tile s of the block), whole blocks are dealt to the XCDs, and
// every A panel is in flight once per block instead of drifting across rounds (per_xcd = 32 slots against 30-tile blocks
// at N = 768: the plain walk fetches fc2's 620 MB A operand 1.5 times).  The launcher picks it only when it costs no extra round.
// (Round 3 also tried a two-stage "touch" of the streaming A operand on top of it - waves 0-3 load one dword per 128-byte A line
// of stage g+2 behind the LDS-DMA of stage g+1 and wait with vmcnt(1): fc2 0.586 vs 0.570 ms, forward 22.02 vs 21.75 ms, slower;
// removed.  Its first version aborted the process: an asm load whose destination the compiler considers dead lands later in a
// register that holds something else by then - an asynchronous destination needs a register live for as long as loads fly.)
// L8 (round 4, fp16 operands): the weight-lo product at 8 bits.  After the nk 16-bit stages of a tile come nk8 = K / 128 stages
// whose LDS image has the SAME shape (256 rows x 128 bytes per operand, same swizzles, same fragment reads: a lane's two 16-byte
// chunks are now 32 k-values of an 8-bit row instead of 2 x 8 of a 16-bit one - the k order inside an MFMA is free as long as
// both operands use the same one), multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (32 per stage at 8 passes = the matrix-pipe
// time of a 16-bit stage for twice the k) into the SAME accumulators: the W8 operand (e4m3) carries 2^w8_exp W_lo, the
// instruction's E8M0 scale takes the factor back.  The epilogues also write bf8 copies of their outputs (out8 / x8) for the next
// L8 GEMM.  L8 = 1: only those copies (a producer whose own lo product is 16-bit); L8 = 2: copies + the 8-bit stages, nk8 >= 1
// (the tile switch then only happens in the 8-bit loop: one copy of that code per wave half instead of two).
// PP (round 4): the "ping-pong" k-loop.  A 64-deep k-tile is cut into FOUR phases of 16 MFMAs (one quadrant of
// the wave's 128 x 64 outputs x K = 64), each { fragment reads of the quadrant | 2 LDS-DMA pieces (one half-tile per phase and
// workgroup) | counted vmcnt | s_barrier | lgkmcnt(0) | s_setprio 1 | 16 MFMA | s_setprio 0 | s_barrier }, and the two wave
// groups (waves 0-3 / 4-7: the SIMD partners) run ONE BARRIER APART: while one group's MFMAs hold the matrix pipe its partner
// issues fragment reads and LDS-DMA (the default loop interleaves loads and MFMAs inside every wave and leaves the SIMD
// arbitration to the hardware: 64 MFMAs per barrier, a stage of 4.1 k cycles for 2.05 k cycles of MFMA).  A k-tile buffer holds
// four half-tiles in the order they are consumed, [A0 | W0 | W1 | A1] (128 rows x 128 B each): A half mi = rows wr*128 + mi*64 ..
// of every row half, W half ni = the rows of every wave's quadrant ni, so that quadrant (mi, ni) reads A half mi and W half ni
// only, a half-tile is dead after one phase and can be restaged two phases later; half-tiles are staged SIX ahead, one per phase,
// through the tile switch.  The remap sits on the staging SOURCE rows: the accumulator layout - and with it every epilogue - is
// the default loop's.  Measured (tools/gemm_vs_vendor.py, non-persistent first version): 4096^3 1264 vs 1122 TF/s, K = 3072 at
// M = 100864 1165 vs 1108 (vendor 1100).  K % 128 == 0, K >= 256.
template <class P, int EPI, bool RES, bool SPLIT, bool FOLD = false, bool ALIGN = false, int L8 = 0, bool PP = false, bool HL = false>
__global__ __launch_bounds__(512, 2)
void gemm256_kernel(const GemmParams p) {
  static_assert(!(PP && L8 == 2), "the ping-pong loop has no 8-bit stages");
  constexpr bool PP2 = GAVA_PP2 && PP;
  static_assert(!HL || (PP && RES && EPI == GAVA_EPI_F32 && !L8 && !FOLD && !SPLIT), "HL: the residual producers on the ping-pong loop");
  constexpr int BM = 256, BN = 256, NW = 8;
  constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;   // 32 KiB, 64 KiB
  constexpr int PPW = (BM + BN) / 8 / NW;                            // 8 glds per wave per stage
  constexpr int NSTORE = (EPI == GAVA_EPI_F32 || EPI == GAVA_EPI_F32_PATCH) ? 32 : (SPLIT ? 48 : 16);
  // accumulators start at the residual tile.  (Round 2 tried the other order - accumulators from zero, the residual
  // row groups fetched by LDS-DMA during stages 1..8 and added one stage later, so that the epilogue is left with its
  // stores: out_proj 0.277 vs 0.279 ms, fc2 0.591 vs 0.587 ms, plain out / fc2 4 % slower.  The epilogue's time is its
  // stores, not the residual loads; removed.)
  constexpr bool ACC_RES = EPI == GAVA_EPI_F32 && RES;
  // Column layout of the accumulators.  16-bit outputs: W rows are read in permuted order so that a lane ends with 16
  // consecutive columns (32 contiguous bytes, two b128 stores).  fp32 outputs (NAT): natural order, lane (fr, fg) holds
  // columns 16*jj + 4*fg + r, so that the four lanes of a row write / read 64 contiguous bytes per instruction - with the
  // permuted order every fp32 store and residual load touched 64 separate 16-byte pieces and the fc2 epilogue took
  // 30-50 k cycles per tile (tools/gemm_stamps.py fc2), four times the QuickGELU epilogue of fc1.
  // HL (the stream as a 16-bit hi / lo pair): nothing leaves as fp32 - the permuted order of the 16-bit outputs, a lane's 16 consecutive
  // columns are 32 contiguous bytes of hi and 32 of lo, the four lanes of a row fill a 128-byte line of each with two instructions
  // (its own column order - lane (fr, fg) holds columns 32 (jj >> 1) + 8 fg + 4 (jj & 1) + r of its wave's 64: the fragment reads
  // of the permuted order on W rows staged in natural order - so that the four lanes of a row move 64 CONTIGUOUS bytes of hi, or
  // of lo, per instruction, like the fp32 form does; with the permuted order's 16 consecutive columns per lane an instruction
  // touched four 16-byte pieces 32 bytes apart in every row)
  constexpr bool NAT = GAVA_V3_NAT && (EPI == GAVA_EPI_F32 || EPI == GAVA_EPI_F32_PATCH) && !HL;
  constexpr int CJ = NAT ? 16 : 4;        // column step between a lane's accumulators jj and jj+1
  constexpr int CF = NAT ? 4 : 16;        // column step between the lane groups fg and fg+1
  constexpr int WJ = NAT ? 2048 : 512;    // LDS byte step between the W fragments jj and jj+1
  // folded LayerNorm (consumer side, see gava_hip.h).  Every wave fetches, with the first operand stage of a tile, the
  // (mean, rstd) pairs of ITS 128 rows (1 KiB) and s_n of ITS 64 columns (256 B) by LDS-DMA into a private, tile-parity
  // indexed block behind the ring: no barrier is needed before it reads them back, only its own vmcnt.
  constexpr bool CAN_FOLD = FOLD;
  constexpr int FOLD_WAVE = 1536, FOLD_BYTES = NW * FOLD_WAVE;   // per wave: 1 KiB of row statistics / partials, 256 B of s_n, 256 B of t_n
  // producers of the folding (EPI_F32 + x16_out): the 16-bit copy of a row group goes through a wave-private LDS block
  // (16 rows x 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7: conflict-free for the 8-byte writes and the 16-byte
  // reads) so that it leaves as 128 contiguous bytes per row - 2 store instructions touching 16 lines instead of 4
  // touching 64
  constexpr bool X16_STAGE = EPI == GAVA_EPI_F32 && !HL;
  constexpr bool ROWSUM = EPI == GAVA_EPI_F32;       // (sum, sum^2) partials of the row groups: the PS block below
  constexpr int XS_PITCH = 128, XS_WAVE = 16 * XS_PITCH;
  // consumers in "partials" mode (p.fpart): (mean, rstd) pairs of the tile's 256 rows, shared by the workgroup, by tile parity
  constexpr int PAIRS_OFF = 2 * STAGE + 2 * FOLD_BYTES, PAIRS_BYTES = 2 * 256 * 8;
  // producers with p.rowsum_reduced: the four waves of a row half leave their (sum, sum^2) per row here [wr][row][wc]
  constexpr int PS_OFF = 2 * STAGE + (X16_STAGE ? NW * XS_WAVE : 0), PS_BYTES = 2 * 128 * 4 * 8;
  constexpr int TAIL_LDS = (CAN_FOLD ? 2 * FOLD_BYTES + PAIRS_BYTES : 0) + (X16_STAGE ? NW * XS_WAVE : 0) + (ROWSUM ? PS_BYTES : 0);
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + TAIL_LDS];
  constexpr bool fold = FOLD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- this workgroup's tiles: XCD-contiguous range, strided by the workgroups of the XCD
  const int nwg = p.n_tiles, nb = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = nb >> 3;
  // ALIGN: the units dealt to the XCDs are super-tile blocks (sm m-tiles x sn n-tiles, all chunks of an m-range in a row)
  const int n_chunks = ALIGN ? p.tiles_n / p.sn : 1;
  const int n_units = ALIGN ? ((p.tiles_m + p.sm - 1) / p.sm) * n_chunks : nwg;
  const int q = n_units >> 3, r = n_units & 7;
  const int x_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int x_count = xcd < r ? q + 1 : q;
  int my_tiles_ = slot < x_count ? (x_count - slot + per_xcd - 1) / per_xcd : 0;
  if (ALIGN) {
    // slot s is tile s of every block; the blocks of the last (short) m-range are the last n_chunks units overall
    const int sm_last = p.tiles_m - ((p.tiles_m + p.sm - 1) / p.sm - 1) * p.sm;
    my_tiles_ = slot >= p.sm * p.sn ? 0 : (slot >= sm_last * p.sn ? max(0, min(x_count, n_units - n_chunks - x_first)) : x_count);
  }
  const int my_tiles = my_tiles_;
  if (my_tiles == 0) return;
  const int nk = p.w_lo == 1 ? 2 * p.nka : p.nka;
  const int nk8 = L8 == 2 ? p.nk8 : 0, NT = nk + nk8;    // stages per tile
  const int G = my_tiles * NT;
  auto tile_coords = [&](int j, int& m0, int& n0) {
    if (ALIGN) {
      // (taking an XCD's blocks last-to-first, so that the rows the previous kernel wrote last - what the Infinity Cache still holds
      // of fc2's 620 MB operand - are read first: measured, 21.62 vs 21.57 ms per c2 forward, no gain; not kept)
      const int unit = x_first + j, mr = unit / n_chunks, ch = unit - mr * n_chunks;
      const int first_m = mr * p.sm, sm = min(p.sm, p.tiles_m - first_m);
      m0 = (first_m + slot % sm) * BM;
      n0 = (ch * p.sn + slot / sm) * BN;
      return;
    }
    const int wg = x_first + slot + j * per_xcd;
    const int per_group = p.sm * p.tiles_n;
    const int g = wg / per_group, first_m = g * p.sm;
    const int sm = min(p.sm, p.tiles_m - first_m);
    const int w = wg - g * per_group;
    const int chunk = w / (sm * p.sn), rr = w - chunk * (sm * p.sn);
    m0 = (first_m + rr % sm) * BM;
    n0 = (chunk * p.sn + rr / sm) * BN;
  };

  unsigned src[PPW];   // 32-bit BYTE offsets from p.A / p.W (host guarantees they fit): base + zext(offset) is the LDS-DMA's
                       // scalar-base + 32-bit-lane-offset addressing form - no 64-bit address arithmetic, no address register pairs
  auto set_src = [&](int m0, int n0) {
    // the lane-dependent terms are recomputed per tile (a few dozen VALU) instead of being hoisted out of the tile loop:
    // hoisted they cost 35 VGPRs, the FOLD kernels then spill, and a spill reload waits on vmcnt, i.e. on the whole
    // operand prefetch in flight (measured: +9 % per stage)
    int ln = lane;
    if (FOLD || GAVA_V3_RECOMPUTE_SRC) asm volatile("" : "+v"(ln));
    if (PP) {
      // Half-tile slot s (0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1), piece (wave + 8 i), i = 0 / 1; lr = the piece's row in
      // the half-tile.  A half mi holds tile rows (lr >> 6) * 128 + mi * 64 + (lr & 63); W half ni holds, per wave column block (lr >> 5),
      // the 32 W rows whose fragments quadrant ni multiplies: natural order ni * 32 + (lr & 31), or - the 16-bit outputs' permuted order,
      // 16 consecutive columns per lane - 16 * ((lr >> 3) & 3) + 8 * ni + (lr & 7).  The eight lane offsets differ by UNIFORM amounts
      // (i adds 128 rows, mi 64, ni 32 or 8; the swizzle terms do not change), so three registers stand for them - src[0] = A piece
      // (slot 0, i 0), src[1] = the lane's offset in the last valid A row (rows past M read that row: a v_min at the use instead of a
      // clamp per entry), src[2] = W piece (slot 1, i 0) - and stage_half adds the uniform part.  (Eight registers before: the LayerNorm-folded
      // kernels on the two-phase loop then spilled 20 B/lane, and every reload of a spill waits on vmcnt(0), i.e. on the operand prefetch.)
      const int lr = wave * 8 + (ln >> 3);
      const unsigned chunk_a = (unsigned)(((ln & 7) ^ ((lr >> 1) & 7)) * 8);
      src[0] = ((unsigned)(m0 + (lr & 63)) * (unsigned)p.lda + chunk_a) * 2u;
      src[1] = ((unsigned)(p.M - 1) * (unsigned)p.lda + chunk_a) * 2u;
      const int wrow = (NAT || HL) ? (lr >> 5) * 64 + (lr & 31) : (lr >> 5) * 64 + 16 * ((lr >> 3) & 3) + (lr & 7);
      const int sw = NAT ? ((lr >> 1) & 7) : (((lr >> 1) & 1) | (((lr >> 3) & 3) << 1));
      src[2] = ((unsigned)(n0 + wrow) * (unsigned)p.ldw + (unsigned)(((ln & 7) ^ sw) * 8)) * 2u;
      return;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + i * NW;                 // 0..31: A rows, 32..63: W rows
      const int row = (piece & 31) * 8 + (ln >> 3);
      if (i < PPW / 2) {
        const int chunk = (ln & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
#ifdef GAVA_EXP_OPERAND_L2   // experiment builds (results WRONG): every tile reads the SAME A rows (1) and / or W rows (2), so that
        if (p.operand_l2 & 1) gm = row;   // operand stays in L2 - what the k-loop would run at without fabric misses on it
#endif
        src[i] = ((unsigned)gm * (unsigned)p.lda + chunk * 8) * 2u;
      } else {
        const int chunk = (ln & 7) ^ (NAT ? ((row >> 1) & 7) : (((row >> 1) & 1) | (((row >> 4) & 3) << 1)));
        src[i] = ((unsigned)(n0 + row) * (unsigned)p.ldw + chunk * 8) * 2u;
#ifdef GAVA_EXP_OPERAND_L2
        if (p.operand_l2 & 2) src[i] = ((unsigned)row * (unsigned)p.ldw + chunk * 8) * 2u;
#endif
      }
    }
  };
  auto piece = [&](int slot, int kt, int i) {
    if (i < PPW / 2 && kt >= p.nka) kt -= p.nka;    // w_lo: the second half of the k-loop re-reads A
    __builtin_amdgcn_global_load_lds(GLB_PTR(reinterpret_cast<const char*>(i < PPW / 2 ? p.A : p.W) + (size_t)(src[i] + (unsigned)(kt * BK * 2))),
                                     LDS_PTR(void, smem + slot * STAGE + (wave + i * NW) * 1024), 16, 0, 0);
  };
  auto stage = [&](int g, int kt) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece(g & 1, kt, i);
  };
  // L8: the same pieces of the 8-bit operands.  Their rows are 2 lda / 2 ldw BYTES apart - the byte pitch of the 16-bit
  // operands (host contract) - so src[] serves both phases unchanged: piece i of 8-bit stage k8 sits at the byte offset of
  // the 16-bit piece of stage k8 from the other base pointer
  auto stage8 = [&](int g, int k8) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR((i < PPW / 2 ? p.A8 : p.W8) + (size_t)(src[i] + (unsigned)(k8 * 128))),
                                       LDS_PTR(void, smem + (g & 1) * STAGE + (wave + i * NW) * 1024), 16, 0, 0);
  };
  // fold block of tile `tj` at (mm0, nn0); fold_stats holds tiles_m*256 rows (host contract)
  auto fold_fetch = [&](int tj, int mm0, int nn0) {
    if (CAN_FOLD) {
      // lane offsets recomputed at every use (the empty asm hides them from loop-invariant hoisting: hoisted, they are
      // spilled, and a spill reload waits on vmcnt, i.e. on the whole operand prefetch)
      unsigned l16 = lane * 16u, l4 = lane * 4u;
      asm volatile("" : "+v"(l16), "+v"(l4));
      char* fb = smem + 2 * STAGE + (tj & 1) * FOLD_BYTES + wave * FOLD_WAVE;
      // (mean, rstd) pairs of this wave's 128 rows, or - partials mode - the raw [4] float2 partials of ITS 32 of them
      const char* st = p.fpart ? reinterpret_cast<const char*>(p.fpart) + (size_t)(mm0 + wr * 128 + wc * 32) * 32
                               : reinterpret_cast<const char*>(p.fstats) + (size_t)(mm0 + wr * 128) * 8;
      const char* ss = reinterpret_cast<const char*>(p.fs) + (size_t)(nn0 + wc * 64) * 4;
      __builtin_amdgcn_global_load_lds(GLB_PTR(st + l16), LDS_PTR(void, fb), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(ss + l4), LDS_PTR(void, fb + 1024), 4, 0, 0);
      // t_n (the folded bias) of the same 64 columns: read back at the top of the epilogue instead of a global load there,
      // whose wait (vmcnt(0): the operand prefetch of the next tile is in flight) cost the older wave group 230 cycles per
      // stage in the qkv GEMM (tools/gemm_stamps.py qkvpart)
      const char* tt = reinterpret_cast<const char*>(p.ft) + (size_t)(nn0 + wc * 64) * 4;
      __builtin_amdgcn_global_load_lds(GLB_PTR(tt + l4), LDS_PTR(void, fb + 1280), 4, 0, 0);
    }
  };
  // fragment read offsets.  A rows: wr*128 + i*16 + fr, swizzle (row>>1)&7 = fr>>1.
  // W rows: wc*64 + 16*(fr>>2) + 4*j + (fr&3), swizzle ((row>>1)&1) | (((row>>4)&3)<<1) = ((fr>>1)&1) | ((fr>>2)<<1)
  const int swa = fr >> 1;
  const int swb = NAT ? (fr >> 1) : (((fr >> 1) & 1) | ((fr >> 2) << 1));
  const int a_off = (wr * 128 + fr) * 128;
  const int w_off = A_BYTES + (NAT ? (wc * 64 + fr) : (wc * 64 + 16 * (fr >> 2) + (fr & 3))) * 128;
  const int a_k0 = (fg ^ swa) << 4, a_k1 = ((4 + fg) ^ swa) << 4;
  const int w_k0 = (fg ^ swb) << 4, w_k1 = ((4 + fg) ^ swb) << 4;
  // PP: half-tile (k-tile kt of the tile src[] points at, slot sl) into k-tile buffer b; fragment rows inside a half-tile:
  // A quadrant mi: wr*64 + i'*16 + fr of slot (mi ? 3 : 0); W quadrant ni: wc*32 + jj'*16 + fr (natural) or
  // wc*32 + 8*(fr>>2) + 4*jj' + (fr&3) (permuted) of slot 1 + ni; every read swizzled by fr >> 1
  constexpr int HALF = 128 * BK * 2;
  auto stage_half = [&](int kt, int sl, int b) {
    const bool is_a = sl == 0 || sl == 3;
    if (is_a && kt >= p.nka) kt -= p.nka;           // w_lo = 1: the second half of the k-loop re-reads A
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned off;
      if (is_a) {
        const unsigned rows = (unsigned)(i * 128 + (sl == 3 ? 64 : 0));
        off = src[0] + rows * (unsigned)p.lda * 2u;
        off = off < src[1] ? off : src[1];          // rows past M: the last valid row (their results are never stored)
      } else {
        const unsigned rows = (unsigned)(i * 128 + (sl == 2 ? ((NAT || HL) ? 32 : 8) : 0));
        off = src[2] + rows * (unsigned)p.ldw * 2u;
      }
      __builtin_amdgcn_global_load_lds(GLB_PTR(reinterpret_cast<const char*>(is_a ? p.A : p.W) + (size_t)(off + (unsigned)(kt * BK * 2))),
                                       LDS_PTR(void, smem + b * STAGE + sl * HALF + (wave + 8 * i) * 1024), 16, 0, 0);
    }
  };
  const int pp_a = (wr * 64 + fr) * 128, pp_w = (NAT ? (wc * 32 + fr) : (wc * 32 + 8 * (fr >> 2) + (fr & 3))) * 128;
  const int pp_k0 = (fg ^ (fr >> 1)) << 4, pp_k1 = ((4 + fg) ^ (fr >> 1)) << 4;
  constexpr int PP_WJ = NAT ? 2048 : 512;           // LDS byte step between the two W fragments of a quadrant

  f32x4_t acc[8][4];
  // residual tile (rows mm0+wr*128+i*16+fr, columns nn0+wc*64+16*fg+4*jj..) straight into accumulator i
  auto load_resid = [&](int i, int mm0, int nn0) {
    int m = mm0 + wr * 128 + i * 16 + fr;
    m = m < p.M ? m : p.M - 1;
    if (HL) {
      // the raw pair: 8 registers of hi, 8 of lo, parked in the 16 accumulator registers they will become (resid_up)
      const long off = (long)m * p.ldr + nn0 + wc * 64 + 8 * fg;       // columns 8 fg .. + 7 and 32 + 8 fg .. + 7 of the wave's 64
      acc[i][0] = *reinterpret_cast<const f32x4_t*>(p.r16 + off);
      acc[i][1] = *reinterpret_cast<const f32x4_t*>(p.r16 + off + 32);
      acc[i][2] = *reinterpret_cast<const f32x4_t*>(p.rlo + off);
      acc[i][3] = *reinterpret_cast<const f32x4_t*>(p.rlo + off + 32);
      return;
    }
    const float* rp = p.resid + (long)m * p.ldr + nn0 + wc * 64 + CF * fg;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = *reinterpret_cast<const f32x4_t*>(rp + CJ * jj);
  };
  // HL: the parked pairs -> fp32 accumulators, acc[i][jj][r] = hi + lo of column 32 (jj >> 1) + 8 fg + 4 (jj & 1) + r; called once per tile, when the
  // loads have landed (the first wait of the tile) and before its first MFMA
  auto resid_up = [&]() {
    if (HL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t h0 = __builtin_bit_cast(u32x4_t, acc[i][0]), h1 = __builtin_bit_cast(u32x4_t, acc[i][1]);
        const u32x4_t l0 = __builtin_bit_cast(u32x4_t, acc[i][2]), l1 = __builtin_bit_cast(u32x4_t, acc[i][3]);
        const unsigned hw[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const unsigned lw[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[i][jj] = (f32x4_t){P::up((unsigned short)hw[2 * jj]) + PrecF16::up((unsigned short)lw[2 * jj]),
                                 P::up((unsigned short)(hw[2 * jj] >> 16)) + PrecF16::up((unsigned short)(lw[2 * jj] >> 16)),
                                 P::up((unsigned short)hw[2 * jj + 1]) + PrecF16::up((unsigned short)lw[2 * jj + 1]),
                                 P::up((unsigned short)(hw[2 * jj + 1] >> 16)) + PrecF16::up((unsigned short)(lw[2 * jj + 1] >> 16))};
      }
    }
  };

  // accumulators of tile `tj` start at -mean_m * s_n: the MFMAs then leave x.W' - mean * s, the epilogue scales by rstd.
  // LDS reads by hand: the compiler would fence plain ones with vmcnt(0) against the LDS-DMA in flight.
  // partials mode: (mean, rstd) of this wave's 32 rows from their (up to 4) partial sums, fixed order, into the shared
  // pairs area of tile `tj`; a workgroup barrier has to lie between this and fold_init(tj) / the epilogue of tile tj
  auto fold_reduce = [&](int tj) {
    if (CAN_FOLD) {
      const unsigned lb = (unsigned)(size_t)LDS_PTR(char, smem);
      unsigned a = (lane & 31) * 32u;
      asm volatile("" : "+v"(a));
      unsigned pa = a >> 2;                                                       // 8 bytes per row
      a += lb + 2 * STAGE + (tj & 1) * FOLD_BYTES + wave * FOLD_WAVE;
      pa += lb + PAIRS_OFF + (tj & 1) * (PAIRS_BYTES / 2) + (wr * 128 + wc * 32) * 8;
      f32x4_t q0, q1;
      asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(a));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(q1) : "v"(a));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(q0), "+v"(q1));
      float s1 = (q0[0] + q0[2]) + q1[0], s2 = (q0[1] + q0[3]) + q1[1];
      if (p.fold_slots > 3) { s1 += q1[2]; s2 += q1[3]; }
      const float mean = s1 * p.fold_inv_d;
      const float var = fmaxf(s2 * p.fold_inv_d - mean * mean, 0.f);
      const f32x2_t pr = {mean, 1.0f / sqrtf(var + 1e-5f)};
      if (lane < 32) asm volatile("ds_write_b64 %0, %1" ::"v"(pa), "v"(pr) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  auto pairs_base = [&](int tj) -> unsigned {   // LDS address of the (mean, rstd) pair of this wave's row 0 of tile tj
    const unsigned lb = (unsigned)(size_t)LDS_PTR(char, smem);
    return p.fpart ? lb + PAIRS_OFF + (tj & 1) * (PAIRS_BYTES / 2) + wr * 1024
                   : lb + 2 * STAGE + (tj & 1) * FOLD_BYTES + wave * FOLD_WAVE;
  };
  auto fold_init = [&](int tj) {
    if (CAN_FOLD) {
      const unsigned fb = (unsigned)(size_t)LDS_PTR(char, smem) + 2 * STAGE + (tj & 1) * FOLD_BYTES + wave * FOLD_WAVE;
      unsigned a_s = (lane >> 4) * 64u, a_m = (lane & 15) * 8u;
      asm volatile("" : "+v"(a_s), "+v"(a_m));   // see fold_fetch
      a_s += fb; a_m += pairs_base(tj);
      f32x4_t sj[4];
      float mu[8];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sj[jj]) : "v"(a_s), "n"(1024 + jj * 16));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(mu[i]) : "v"(a_m), "n"(i * 128));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) asm volatile("" : "+v"(sj[jj]));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(mu[i]));
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[i][jj] = (f32x4_t){-mu[i] * sj[jj][0], -mu[i] * sj[jj][1], -mu[i] * sj[jj][2], -mu[i] * sj[jj][3]};
    }
  };

  // producers with p.rowsum_reduced: (sum, sum^2) of a row over this tile's 256 columns = the four waves' partials in a
  // fixed order, one float2 per row and 256-column tile: rowsum[row][4]
  auto flush_rowsum = [&](int mm0, int nn0) {
    if (ROWSUM && tid < 256) {
      const float4* q = reinterpret_cast<const float4*>(smem + PS_OFF + tid * 32);
      const float4 a = q[0], b = q[1];
      const int m = mm0 + tid;
      if (m < p.M) p.rowsum[(long)m * 4 + nn0 / 256] = make_float2((a.x + a.z) + (b.x + b.z), (a.y + a.w) + (b.y + b.w));
    }
  };

  int m0, n0, m0n, n0n, m0p = 0, n0p = 0;
  tile_coords(0, m0, n0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (ACC_RES) {
      load_resid(i, m0, n0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
  set_src(m0, n0);
  if (PP) {
#pragma unroll
    for (int S = 0; S < 6; ++S) stage_half(S >> 2, S & 3, (S >> 2) & 1);     // six half-tiles ahead: k-tile 0 and half of k-tile 1
  } else {
    stage(0, 0);
  }
  if (CAN_FOLD) {
    fold_fetch(0, m0, n0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.fpart) {
      fold_reduce(0);
      __builtin_amdgcn_s_barrier();
    }
    fold_init(0);
  }
  if (PP) {
    // half-tiles 0, 1 (and the residual tile, older) have landed; the barrier publishes them; then group 1 falls one barrier behind
    // (PP2: its first phase reads half-tiles 0, 1 and 2)
    if (PP2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (!CAN_FOLD) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    resid_up();
    if (wr == 1) __builtin_amdgcn_s_barrier();
  }
  int counted = 0;   // the next wait may leave this wave's NSTORE (1) or 2*NSTORE (2: pre-activation copy) epilogue stores in flight
  // diagnostic stamps (gava_debug_set_buffer; tools/gemm_stamps.py): cycles in the vmcnt wait, the barrier,
  // the stage body and the epilogue
#ifdef GAVA_STAMPS      // diagnostic builds only (tools/ab_build.sh stamps -DGAVA_STAMPS): the stamps cost scalar registers
  const bool stamp = p.dbg != nullptr;
#else
  constexpr bool stamp = false;
#endif
  unsigned long long ts = 0, tW = 0, tB = 0, tC = 0, tE = 0, tE1 = 0, tE2 = 0, te = 0;   // tE1 / tE2: epilogue until the bias is there / its row groups
  bool in_epi = false;
  if (stamp) ts = clock64();
  // Static priority for the second-dispatched half: at equal priority waves 4-7 lose every issue arbitration to
  // their older SIMD partners and trail them by ~800 cycles per stage (stamps: stage 4.6k -> 4.1k cycles).
  if (GAVA_V3_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
#ifdef GAVA_EXP_STAGGER   // experiment builds (-DGAVA_EXP_STAGGER, nothing inside the loops): a start offset between workgroups,
  {                       // GAVA_PAIR_DELAY x 10 ns; GAVA_STAGGER_MODE 0: odd slots of every XCD late, 1: (slot & 3) * delay / 2, 2: XCDs 4-7 late
    const unsigned long long d = p.stagger_mode == 1 ? (unsigned long long)(slot & 3) * p.pair_delay / 2
                               : p.stagger_mode == 2 ? (xcd >= 4 ? p.pair_delay : 0) : ((slot & 1) ? p.pair_delay : 0);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(32);
  }
#endif

  for (int j = 0; j < my_tiles; ++j) {
    // what every stage starts with: this wave's operand pieces (and what else it has in flight) have landed, the barrier makes
    // the other waves' pieces visible and frees the other ring slot
    auto stage_head = [&](int kt) {
      if (stamp) { const unsigned long long t = clock64(); if (in_epi) tE += t - ts; else tC += t - ts; ts = t; in_epi = false; }
      if (counted == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE + 8) : "memory");    // L8: + one out8 store per row group
      else if (counted == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
      else if (counted == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSTORE <= 63 ? 2 * NSTORE : 0) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      counted = 0;
      if (stamp) { const unsigned long long t = clock64(); tW += t - ts; ts = t; }
      __builtin_amdgcn_s_barrier();
      if (stamp) { const unsigned long long t = clock64(); tB += t - ts; ts = t; }
#ifdef GAVA_TIMELINE   // diagnostic builds (tools/gemm_timeline.py): 10 ns wall-clock stamps per tile from waves 0 and 4
      if (kt == 0 && p.dbg && lane == 0 && (wave & 3) == 0 && j < 10) p.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 32 + j * 3 + 0] = wall_clock64();
#endif
      // every wave has left the previous tile's epilogue: its row-sum partials are complete in LDS
      if (ROWSUM && p.rowsum_reduced && kt == 0 && j > 0) flush_rowsum(m0p, n0p);
    };
    // LDS-DMA issue is expensive (~100+ cycles per 1 KiB piece beside running MFMAs): the two waves
    // that share a SIMD (w and w+4) issue their 8 pieces half an iteration apart, so one of them
    // always feeds the matrix pipe.  Waves 0-3 issue at the top of a stage, waves 4-7 after the second MFMA group.
    // kt: stage of this tile, 0 .. NT-1 (L8: the 8-bit stages follow the nk 16-bit ones)
    auto issue_next = [&](int kt, bool in_hi) {
      const int g = j * NT + kt;
      if (g + 1 < G) {
        if ((L8 == 2 && in_hi) || kt + 1 < NT) {
          if (L8 != 2 || (in_hi && kt + 1 < nk)) {
            stage(g + 1, kt + 1);
          } else {
            stage8(g + 1, kt + 1 - nk);
          }
          // partials mode: the next tile's partials are fetched three stages before the end of this tile, reduced one
          // stage later (below) and published by the barrier of the last stage - before this tile's epilogue starts
          if (CAN_FOLD && p.fpart && kt == NT - 3 && j + 1 < my_tiles) {
            tile_coords(j + 1, m0n, n0n);
            fold_fetch(j + 1, m0n, n0n);
          }
        } else {
          tile_coords(j + 1, m0n, n0n);
          set_src(m0n, n0n);
          stage(g + 1, 0);
          if (!(CAN_FOLD && p.fpart)) fold_fetch(j + 1, m0n, n0n);
        }
      }
    };
    if (PP) {
      // vmcnt(n) for the handful of counts the schedule uses (the immediate must be a literal)
      auto wait_vm = [&](int n) {
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        else if (n == 8 + NSTORE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NSTORE <= 63 ? 8 + NSTORE : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + 2 * NSTORE <= 63 ? 8 + 2 * NSTORE : 0) : "memory");
      };
      const bool has_next = j + 1 < my_tiles;
      s16x8_t af[4][2], w0f[2][2], w1f[2][2];      // A fragments of the quadrant row in work, W fragments of quadrant column 0 / 1
      auto pp_ktile = [&](int kt, auto Bc) {
        constexpr int B = decltype(Bc)::value;
        const char* buf = smem + B * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // ---- load section: the fragments of this phase's quadrant ...
#ifdef GAVA_PP_NOREADS      // experiment builds (results WRONG): the loop without its fragment reads
          if (kt < 0) {
#else
          if (q == 0) {
#endif
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              w0f[jj][0] = *reinterpret_cast<const s16x8_t*>(buf + 1 * HALF + pp_w + jj * PP_WJ + pp_k0);
              w0f[jj][1] = *reinterpret_cast<const s16x8_t*>(buf + 1 * HALF + pp_w + jj * PP_WJ + pp_k1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              af[i][0] = *reinterpret_cast<const s16x8_t*>(buf + 0 * HALF + pp_a + i * 2048 + pp_k0);
              af[i][1] = *reinterpret_cast<const s16x8_t*>(buf + 0 * HALF + pp_a + i * 2048 + pp_k1);
            }
#ifdef GAVA_PP_NOREADS
          } else if (kt < 0) {
#else
          } else if (q == 1) {
#endif
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              w1f[jj][0] = *reinterpret_cast<const s16x8_t*>(buf + 2 * HALF + pp_w + jj * PP_WJ + pp_k0);
              w1f[jj][1] = *reinterpret_cast<const s16x8_t*>(buf + 2 * HALF + pp_w + jj * PP_WJ + pp_k1);
            }
#ifdef GAVA_PP_NOREADS
          } else if (kt < 0) {
#else
          } else if (q == 2) {
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              af[i][0] = *reinterpret_cast<const s16x8_t*>(buf + 3 * HALF + pp_a + i * 2048 + pp_k0);
              af[i][1] = *reinterpret_cast<const s16x8_t*>(buf + 3 * HALF + pp_a + i * 2048 + pp_k1);
            }
          }
          // ... the half-tile six ahead (slots 2, 3 of the next k-tile in phases 0, 1: the other buffer; slots 0, 1 of the
          // one after in phases 2, 3: this buffer, last read two / three phases ago).  Two k-tiles before the end of the
          // tile the staging crosses into the next tile: its coordinates and fold block at phase 0, its sources at phase 2.
          if (kt == nk - 2 && has_next) {
            if (q == 0) {
              tile_coords(j + 1, m0n, n0n);
              if (CAN_FOLD) fold_fetch(j + 1, m0n, n0n);      // 3 more LDS-DMA loads in flight during this k-tile's four phases
            }
            if (q == 2) set_src(m0n, n0n);
          }
          const int ktt = kt + (q < 2 ? 1 : 2);
          const bool more = ktt < nk || has_next;
#ifndef GAVA_PP_NOGLDS      // experiment builds (results WRONG): the loop without its LDS-DMA
          if (more) stage_half(ktt < nk ? ktt : ktt - nk, (q + 2) & 3, q < 2 ? (B ^ 1) : B);
#endif
          // ... and the wait that retires what the NEXT phase reads: half-tiles up to "this phase + 2"; the four younger ones
          // (8 pieces of this wave) stay in flight, plus the fold block's 3 loads while they are younger than the half-tile
          // waited for, plus - in the first k-tile after an epilogue - that epilogue's stores (counted: exactly NSTORE / 2 NSTORE per wave)
          if (!more) wait_vm(0);
          else if (kt == 0 && j > 0 && counted == 0) wait_vm(q == 0 ? 0 : 8);
          else if (kt == 0 && j > 0) wait_vm(counted == 2 ? 8 + 2 * NSTORE : 8 + NSTORE);
          else if (CAN_FOLD && kt == nk - 2 && has_next) wait_vm(11);
          else wait_vm(8);
          __builtin_amdgcn_s_barrier();
          // every wave of both groups has left the previous tile's epilogue (they ran it side by side): its row-sum partials are complete
          if (ROWSUM && p.rowsum_reduced && kt == 0 && q == 0 && j > 0) flush_rowsum(m0p, n0p);
          if (HL && B == 0 && q == 0 && kt == 0 && j > 0) resid_up();      // the wait above was vmcnt(0): the next tile's pairs are in
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          // ---- MFMA section: quadrant (mi, ni) = (0,0) (0,1) (1,1) (1,0)
          constexpr int dummy = 0; (void)dummy;
          const int mi = q >> 1, ni = (q == 1 || q == 2) ? 1 : 0;
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj)
                acc[mi * 4 + i][ni * 2 + jj] = P::mfma(ni ? w1f[jj][kk] : w0f[jj][kk], af[i][kk], acc[mi * 4 + i][ni * 2 + jj]);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (CAN_FOLD && p.fpart && kt == nk - 1 && q == 1 && has_next) fold_reduce(j + 1);   // fetched a k-tile ago, landed by now
          __builtin_amdgcn_s_barrier();
        }
      };
      // PP2: a k-tile in TWO phases of 32 MFMAs.  Phase 0 = the quadrants (0,0) (0,1) - it reads A half 0 and both W halves - and stages
      // the half-tiles 2, 3 of the next k-tile (other buffer); phase 1 = (1,1) (1,0) - it reads A half 1 - and stages the half-tiles 0, 1 of
      // the k-tile after next (this buffer: last read a phase ago, the partner group is one barrier behind and past its reads).  Four
      // LDS-DMA pieces per wave and phase; the wait at the end of a load section retires what the NEXT phase reads: phase 0 leaves the
      // two younger phases' pieces in flight (8), phase 1 - whose successor reads three half-tiles - all but the youngest half-tile and
      // its own (6).  Same k order per accumulator as the four-phase loop: bit-identical results.
      auto pp2_ktile = [&](int kt, auto Bc) {
        constexpr int B = decltype(Bc)::value;
        const char* buf = smem + B * STAGE;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          if (ph == 0) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              w0f[jj][0] = *reinterpret_cast<const s16x8_t*>(buf + 1 * HALF + pp_w + jj * PP_WJ + pp_k0);
              w0f[jj][1] = *reinterpret_cast<const s16x8_t*>(buf + 1 * HALF + pp_w + jj * PP_WJ + pp_k1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              af[i][0] = *reinterpret_cast<const s16x8_t*>(buf + 0 * HALF + pp_a + i * 2048 + pp_k0);
              af[i][1] = *reinterpret_cast<const s16x8_t*>(buf + 0 * HALF + pp_a + i * 2048 + pp_k1);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              w1f[jj][0] = *reinterpret_cast<const s16x8_t*>(buf + 2 * HALF + pp_w + jj * PP_WJ + pp_k0);
              w1f[jj][1] = *reinterpret_cast<const s16x8_t*>(buf + 2 * HALF + pp_w + jj * PP_WJ + pp_k1);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              af[i][0] = *reinterpret_cast<const s16x8_t*>(buf + 3 * HALF + pp_a + i * 2048 + pp_k0);
              af[i][1] = *reinterpret_cast<const s16x8_t*>(buf + 3 * HALF + pp_a + i * 2048 + pp_k1);
            }
          }
          // two k-tiles before the end of the tile the staging crosses into the next tile: its coordinates at phase 0, its sources at
          // phase 1 (before that phase stages the next tile's first half-tiles)
          if (kt == nk - 2 && has_next) {
            if (ph == 0) {
              tile_coords(j + 1, m0n, n0n);
              if (CAN_FOLD) fold_fetch(j + 1, m0n, n0n);      // 3 more LDS-DMA loads, older than this phase's pieces
            } else set_src(m0n, n0n);
          }
          const int ktt = kt + (ph == 0 ? 1 : 2);
          const bool more = ktt < nk || has_next;
          if (more) {
            stage_half(ktt < nk ? ktt : ktt - nk, ph == 0 ? 2 : 0, ph == 0 ? (B ^ 1) : B);
            stage_half(ktt < nk ? ktt : ktt - nk, ph == 0 ? 3 : 1, ph == 0 ? (B ^ 1) : B);
          }
          if (!more) wait_vm(0);
          else if (kt == 0 && j > 0 && ph == 0) wait_vm(counted == 0 ? 0 : (counted == 2 ? 8 + 2 * NSTORE : 8 + NSTORE));
          else if (CAN_FOLD && kt == nk - 2 && has_next && ph == 0) wait_vm(11);      // + the fold block (retired by phase 1's wait)
          else wait_vm(ph == 0 ? 8 : 6);
          __builtin_amdgcn_s_barrier();
          if (ROWSUM && p.rowsum_reduced && kt == 0 && ph == 0 && j > 0) flush_rowsum(m0p, n0p);
          if (HL && B == 0 && ph == 0 && kt == 0 && j > 0) resid_up();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                acc[ph * 4 + i][jj] = P::mfma(w0f[jj][kk], af[i][kk], acc[ph * 4 + i][jj]);
                acc[ph * 4 + i][2 + jj] = P::mfma(w1f[jj][kk], af[i][kk], acc[ph * 4 + i][2 + jj]);
              }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (CAN_FOLD && p.fpart && kt == nk - 1 && ph == 0 && has_next) fold_reduce(j + 1);   // fetched a k-tile ago, landed by now
          __builtin_amdgcn_s_barrier();
        }
      };
      for (int kt = 0; kt < nk; kt += 2) {      // K % 128 == 0: the two k-tile buffers alternate statically
        if (PP2) {
          pp2_ktile(kt, std::integral_constant<int, 0>{});
          pp2_ktile(kt + 1, std::integral_constant<int, 1>{});
        } else {
          pp_ktile(kt, std::integral_constant<int, 0>{});
          pp_ktile(kt + 1, std::integral_constant<int, 1>{});
        }
      }
      // The epilogues of the two groups must run side by side, not one after the other (a group that is a barrier ahead would
      // wait at its next barrier for the other group's whole epilogue, and vice versa): group 0 waits here for group 1's last
      // phase, both run the epilogue, and group 1 falls a barrier behind again at its end (below).
      if (wr == 0) __builtin_amdgcn_s_barrier();
    } else
    for (int kt = 0; kt < nk; ++kt) {
      const int g = j * NT + kt;
      stage_head(kt);
      if (wave < 4) issue_next(kt, true);
      const char* cur = smem + (g & 1) * STAGE;
      s16x8_t wf0[4], wf1[4], a00[4], a01[4], a10[4], a11[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) wf0[jj] = *reinterpret_cast<const s16x8_t*>(cur + w_off + jj * WJ + w_k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) a00[i] = *reinterpret_cast<const s16x8_t*>(cur + a_off + i * 2048 + a_k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) a01[i] = *reinterpret_cast<const s16x8_t*>(cur + a_off + (4 + i) * 2048 + a_k0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = P::mfma(wf0[jj], a00[i], acc[i][jj]);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) wf1[jj] = *reinterpret_cast<const s16x8_t*>(cur + w_off + jj * WJ + w_k1);
#pragma unroll
      for (int i = 0; i < 4; ++i) a10[i] = *reinterpret_cast<const s16x8_t*>(cur + a_off + i * 2048 + a_k1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[4 + i][jj] = P::mfma(wf0[jj], a01[i], acc[4 + i][jj]);
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);  // wf0, a00, a01
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // a00 x wf0
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // wf1, a10
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // a01 x wf0
      if (wave >= 4) issue_next(kt, true);
#pragma unroll
      for (int i = 0; i < 4; ++i) a11[i] = *reinterpret_cast<const s16x8_t*>(cur + a_off + (4 + i) * 2048 + a_k1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = P::mfma(wf1[jj], a10[i], acc[i][jj]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[4 + i][jj] = P::mfma(wf1[jj], a11[i], acc[4 + i][jj]);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);   // a11
      __builtin_amdgcn_sched_group_barrier(0x008, 32, 1);  // a10 x wf1, a11 x wf1
      if (CAN_FOLD && p.fpart && kt == NT - 2 && j + 1 < my_tiles) fold_reduce(j + 1);   // its fetch was waited for at this stage's start
    }
    if (L8 == 2) {
      // the 8-bit lo stages (a loop of their own: one loop with both bodies made hipcc rotate the accumulators through scratch).
      // 128 k-values per row; lane (fr, fg) multiplies chunks fg and 4 + fg of its rows (the same LDS reads as the two k-halves
      // of a 16-bit stage), W8 (e4m3, scaled) as the MFMA's A operand, A8 (bf8) as its B operand
      for (int kt = nk; kt < NT; ++kt) {
        const int g = j * NT + kt;
        stage_head(kt);
        if (wave < 4) issue_next(kt, false);
        const char* cur = smem + (g & 1) * STAGE;
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        typedef int i32x8_t __attribute__((ext_vector_type(8)));
        i32x8_t w8[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(cur + w_off + jj * WJ + w_k0);
          const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(cur + w_off + jj * WJ + w_k1);
          w8[jj] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(cur + a_off + i * 2048 + a_k0);
          const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(cur + a_off + i * 2048 + a_k1);
          const i32x8_t a8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8[jj], a8, acc[i][jj], 0 /* e4m3 */, 1 /* bf8 */, 0,
                                                                          (int)p.scale8, 0, 0x7f7f7f7f);
          if (i == 3) {
            // pin the first 16 MFMAs in front of the branch below: hipcc otherwise sinks them behind it (their results are next
            // used an iteration later), next to the other 16 - with all eight A fragments (64 registers) live at once
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                              "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),
                              "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
#if GAVA_L8_SCHED
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);   // w8, a8 of row group 0
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
            if (wave >= 4) issue_next(kt, false);
          }
        }
#if GAVA_L8_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
#endif
        if (CAN_FOLD && p.fpart && kt == NT - 2 && j + 1 < my_tiles) fold_reduce(j + 1);
      }
    }

    if (stamp) { const unsigned long long t = clock64(); tC += t - ts; ts = t; in_epi = true; }
#ifdef GAVA_TIMELINE
    if (p.dbg && lane == 0 && (wave & 3) == 0 && j < 10) p.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 32 + j * 3 + 1] = wall_clock64();
#endif
    // ---- epilogue of tile j: lane holds out[m][n .. n+15], m = m0+wr*128+i*16+fr,
    //      n = n0 + wc*64 + 16*fg + 4*jj + r
    // (the ping-pong kernels near the register limit: the lane-derived constants of the epilogue are derived HERE from a lane id the compiler cannot see through - kept live
    // across the k-loop they were spilled (16 B/lane), and a spill reload is a scratch load: its s_waitcnt vmcnt(0) waited for the next
    // tile's whole operand prefetch at the top of every epilogue and for this tile's stores at its end)
    int lane_e = lane;
    constexpr bool LAUNDER = (PP && (FOLD || (ACC_RES && !HL))) || (L8 == 2 && FOLD);      // the instantiations that spilled without it
    if (LAUNDER) asm volatile("" : "+v"(lane_e));
    const int fr_e = LAUNDER ? (lane_e & 15) : fr, fg_e = LAUNDER ? (lane_e >> 4) : fg;
    const int nb0 = n0 + wc * 64 + (HL ? 8 : CF) * fg_e;
    float4 bj[4];
    if (CAN_FOLD) {   // t_n from this tile's fold block (LDS, by hand: see fold_init)
      unsigned a_t = (lane_e >> 4) * 64u;
      asm volatile("" : "+v"(a_t));
      a_t += (unsigned)(size_t)LDS_PTR(char, smem) + 2 * STAGE + (j & 1) * FOLD_BYTES + wave * FOLD_WAVE;
      f32x4_t tj[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tj[jj]) : "v"(a_t), "n"(1280 + jj * 16));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        asm volatile("" : "+v"(tj[jj]));
        bj[jj] = make_float4(tj[jj][0], tj[jj][1], tj[jj][2], tj[jj][3]);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        bj[jj] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb0 + (HL ? 32 * (jj >> 1) + 4 * (jj & 1) : CJ * jj)) : make_float4(0, 0, 0, 0);
    }
    // One explicit use on the common path: the compiler waits for the bias HERE, once.  Without it every row
    // below (a basic block of its own behind `m < M`) re-waits with vmcnt(0), i.e. for the previous row's stores.
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) asm volatile("" ::"v"(bj[jj].x), "v"(bj[jj].y), "v"(bj[jj].z), "v"(bj[jj].w));
    if (stamp) { te = clock64(); tE1 += te - ts; }
    const bool full = m0 + BM <= p.M;
    // folded LayerNorm: out = rstd_m * acc + t_n (acc already holds x.W' - mean_m * s_n; t_n came in as the bias)
    // rstd of row i+1 is fetched (LDS, by hand: see fold_init) while row i is processed: two registers, not eight
    float r_cur = 0.f, r_next = 0.f;
    unsigned a_r = (lane_e & 15) * 8u + 4u;
    if (CAN_FOLD) {
      asm volatile("" : "+v"(a_r));
      a_r += pairs_base(j);
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r_cur) : "v"(a_r) : "memory");
    }
    const long row0_ = (long)(m0 + wr * 128 + fr_e);
    // (pinned here: left to itself hipcc sinks the multiplies back into each row group's `m < M` branch)
    auto pinned = [](long x) {
      if (L8) return x;        // (the 8-bit kernels sit at the register limit: two more live registers there are spilled)
      unsigned lo = (unsigned)x, hi = (unsigned)((unsigned long)x >> 32);
      asm volatile("" : "+v"(lo), "+v"(hi));
      return (long)(((unsigned long)hi << 32) | lo);
    };
    const long row0_o = pinned(row0_ * p.ldo);
    const long row0_x16 = (EPI == GAVA_EPI_F32) ? pinned(row0_ * p.ldx16) : 0;
    const long row0_o8 = L8 ? pinned(row0_ * p.ldo8) : 0;
    const long res_next = HL ? pinned((long)(m0n + wr * 128 + fr_e) * p.ldr) : 0;      // (m0n: meaningful when there is a next tile)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + fr_e;
      if (CAN_FOLD && i < 7) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r_next) : "v"(a_r), "n"((i + 1) * 128));
      if (m < p.M && !(ABLATE & 4)) {
        long orow = m;
        const float* posr = nullptr;
        const float* timr = nullptr;
        if (EPI == GAVA_EPI_F32_PATCH) {
          const int frame = m / p.n_patches, pp = m - frame * p.n_patches;
          orow = (long)frame * (p.n_patches + 1) + 1 + pp;
          posr = p.pos + (long)(1 + pp) * p.N + nb0;
          timr = p.time + (long)(frame % p.T) * p.N + nb0;
        }
        // element offsets of this lane's output row: one 64-bit multiply per TILE (row group 0) + a uniform step per row group, instead of
        // a quarter-rate v_mul_lo_u32 / v_mad_u64_u32 set per row group and output array (51 of them per tile in the fc1 kernel)
        const long off_o = EPI == GAVA_EPI_F32_PATCH ? orow * p.ldo : row0_o + (long)(i * 16) * p.ldo;
        const long off_x16 = row0_x16 + (long)(i * 16) * p.ldx16;
        const long off_o8 = L8 ? row0_o8 + (long)(i * 16) * p.ldo8 : 0;
        (void)off_x16; (void)off_o8;
        float v[16];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          v[4 * jj + 0] = acc[i][jj][0] + bj[jj].x; v[4 * jj + 1] = acc[i][jj][1] + bj[jj].y;
          v[4 * jj + 2] = acc[i][jj][2] + bj[jj].z; v[4 * jj + 3] = acc[i][jj][3] + bj[jj].w;
          if (CAN_FOLD && fold) {
            const float r = r_cur;
            v[4 * jj + 0] = fmaf(acc[i][jj][0], r, bj[jj].x); v[4 * jj + 1] = fmaf(acc[i][jj][1], r, bj[jj].y);
            v[4 * jj + 2] = fmaf(acc[i][jj][2], r, bj[jj].z); v[4 * jj + 3] = fmaf(acc[i][jj][3], r, bj[jj].w);
          }
        }
        if (EPI == GAVA_EPI_H16 || EPI == GAVA_EPI_H16_QGELU || EPI == GAVA_EPI_H16_QGELU_BWD) {
          if (EPI == GAVA_EPI_H16) {
            if (nb0 < p.scale_cols) {
              // (the factor in a VECTOR register: as a scalar operand hipcc made eight copies of the {scale, scale} pair for its eight
              // packed multiplies, spilled them, and read each back with two v_readlane + s_nop - 144 v_readlane per tile)
              float sc;
              asm volatile("v_mov_b32 %0, %1" : "=v"(sc) : "s"(p.scale));
#pragma unroll
              for (int e = 0; e < 16; ++e) v[e] *= sc;
            }
          } else if (EPI == GAVA_EPI_H16_QGELU) {
            if (p.aux_out) {
              unsigned short* ao = p.aux_out + off_o + nb0;
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const uint2 x = pack4<P>(v[8 * hh], v[8 * hh + 1], v[8 * hh + 2], v[8 * hh + 3]);
                const uint2 y = pack4<P>(v[8 * hh + 4], v[8 * hh + 5], v[8 * hh + 6], v[8 * hh + 7]);
                *reinterpret_cast<uint4*>(ao + 8 * hh) = make_uint4(x.x, x.y, y.x, y.y);
              }
            }
#pragma unroll
            for (int e = 0; e < 16; e += 8) quick_gelu8(v, e);
          } else {
            const unsigned short* axp = p.aux + off_o + nb0;
            const uint4 a0 = *reinterpret_cast<const uint4*>(axp), a1 = *reinterpret_cast<const uint4*>(axp + 8);
            const unsigned aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[2 * e] *= quick_gelu_grad(aux_up((unsigned short)aw[e], p.aux_f16));
              v[2 * e + 1] *= quick_gelu_grad(aux_up((unsigned short)(aw[e] >> 16), p.aux_f16));
            }
          }
          unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + off_o + nb0;
          if (SPLIT) {
            uint2 hi[4], lo[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) split4<P>(v[4 * jj], v[4 * jj + 1], v[4 * jj + 2], v[4 * jj + 3], hi[jj], lo[jj]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint4 H = make_uint4(hi[2 * h].x, hi[2 * h].y, hi[2 * h + 1].x, hi[2 * h + 1].y);
              const uint4 L = make_uint4(lo[2 * h].x, lo[2 * h].y, lo[2 * h + 1].x, lo[2 * h + 1].y);
              *reinterpret_cast<uint4*>(o + 8 * h) = H;
              *reinterpret_cast<uint4*>(o + p.N + 8 * h) = L;
              *reinterpret_cast<uint4*>(o + 2 * p.N + 8 * h) = H;
            }
          } else {
            if (L8 && p.out8) {   // bf8 copy of the row segment: the A8 operand of the next GEMM's 8-bit lo product
              unsigned r8[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                r8[q] = __builtin_amdgcn_cvt_pk_bf8_f32(v[4 * q], v[4 * q + 1], 0u, false);
                r8[q] = __builtin_amdgcn_cvt_pk_bf8_f32(v[4 * q + 2], v[4 * q + 3], r8[q], true);
              }
              *reinterpret_cast<uint4*>(p.out8 + off_o8 + nb0) = make_uint4(r8[0], r8[1], r8[2], r8[3]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint2 x = pack4<P>(v[8 * h], v[8 * h + 1], v[8 * h + 2], v[8 * h + 3]);
              const uint2 y = pack4<P>(v[8 * h + 4], v[8 * h + 5], v[8 * h + 6], v[8 * h + 7]);
              if (EPI == GAVA_EPI_H16) {
                // the Q/K/V rows (and the other plain 16-bit outputs of this kernel) leave as non-temporal stores: same-box
                // forward 21.58 -> 21.46 ms; the QuickGELU output (read back by fc2 right away) and the fp32 stream lose with them
                typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
                const nt_u4 dv = {x.x, x.y, y.x, y.y};
                __builtin_nontemporal_store(dv, reinterpret_cast<nt_u4*>(o + 8 * h));
              } else {
                *reinterpret_cast<uint4*>(o + 8 * h) = make_uint4(x.x, x.y, y.x, y.y);
              }
            }
          }
        } else if (HL) {
          // the stream leaves as the pair: hi = h16(v), the operand of the GEMM that consumes the fold, and lo = fp16(v - hi)
          unsigned short* ho = p.x16 + off_x16 + nb0;
          unsigned short* lp = p.xlo + off_x16 + nb0;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint2 ha, la, hb, lb;
            split4_lo16<P>(v[8 * h], v[8 * h + 1], v[8 * h + 2], v[8 * h + 3], ha, la);
            split4_lo16<P>(v[8 * h + 4], v[8 * h + 5], v[8 * h + 6], v[8 * h + 7], hb, lb);
            *reinterpret_cast<uint4*>(ho + 32 * h) = make_uint4(ha.x, ha.y, hb.x, hb.y);
            *reinterpret_cast<uint4*>(lp + 32 * h) = make_uint4(la.x, la.y, lb.x, lb.y);
          }
          float ps1 = 0.f, ps2 = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) { ps1 += v[e]; ps2 += v[e] * v[e]; }
          ps1 = sum_across_lane_groups(ps1); ps2 = sum_across_lane_groups(ps2);
          if (p.rowsum_reduced) {
            if (fg == 0) *reinterpret_cast<float2*>(smem + PS_OFF + ((wr * 128 + i * 16 + fr) * 4 + wc) * 8) = make_float2(ps1, ps2);
          } else if (fg == 0) p.rowsum[orow * (p.N / 64) + (n0 + wc * 64) / 64] = make_float2(ps1, ps2);
        } else {
          float* o = reinterpret_cast<float*>(p.out) + off_o + nb0;
          if (EPI == GAVA_EPI_F32_PATCH) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float4 pr = *reinterpret_cast<const float4*>(posr + CJ * jj);
              const float4 tr = *reinterpret_cast<const float4*>(timr + CJ * jj);
              v[4 * jj] += pr.x + tr.x; v[4 * jj + 1] += pr.y + tr.y; v[4 * jj + 2] += pr.z + tr.z; v[4 * jj + 3] += pr.w + tr.w;
            }
          }
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            *reinterpret_cast<float4*>(o + CJ * jj) = make_float4(v[4 * jj], v[4 * jj + 1], v[4 * jj + 2], v[4 * jj + 3]);
          if (EPI == GAVA_EPI_F32 && p.x16) {
            // producer side of the LayerNorm folding: 16-bit copy of the row segment + its (sum x, sum x^2)
            unsigned short* xo = p.x16 + off_x16 + nb0;
            float ps1 = 0.f, ps2 = 0.f;
            (void)xo;
            if (X16_STAGE) {
              char* xs = smem + 2 * STAGE + wave * XS_WAVE;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
                *reinterpret_cast<uint2*>(xs + fr_e * XS_PITCH + (((2 * jj + (fg_e >> 1)) ^ ((fr_e >> 1) & 7)) << 4) + (fg_e & 1) * 8) =
                    pack4<P>(v[4 * jj], v[4 * jj + 1], v[4 * jj + 2], v[4 * jj + 3]);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) { ps1 += v[e]; ps2 += v[e] * v[e]; }
            ps1 = sum_across_lane_groups(ps1); ps2 = sum_across_lane_groups(ps2);
            if (p.rowsum_reduced) {
              if (fg_e == 0) *reinterpret_cast<float2*>(smem + PS_OFF + ((wr * 128 + i * 16 + fr_e) * 4 + wc) * 8) = make_float2(ps1, ps2);
            } else if (fg_e == 0) p.rowsum[orow * (p.N / 64) + (n0 + wc * 64) / 64] = make_float2(ps1, ps2);
          }
        }
      }
      if (X16_STAGE && p.x16 && !(ABLATE & 4)) {
        // lane l: row (l >> 3) (+8 in the second pass), 16-byte chunk (l & 7) of the 128-byte row segment
        const char* xs = smem + 2 * STAGE + wave * XS_WAVE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = (lane_e >> 3) + 8 * h;
          const uint4 d = *reinterpret_cast<const uint4*>(xs + row * XS_PITCH + (((lane_e & 7) ^ ((row >> 1) & 7)) << 4));
          const int mm = m0 + wr * 128 + i * 16 + row;
          if (mm < p.M) *reinterpret_cast<uint4*>(p.x16 + (long)mm * p.ldx16 + n0 + wc * 64 + (lane_e & 7) * 8) = d;
          if (L8 && p.x8 && mm < p.M) {   // bf8 copy of the same 8 values (fp16 operands only)
            const unsigned dw[4] = {d.x, d.y, d.z, d.w};
            unsigned r8[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              r8[q] = __builtin_amdgcn_cvt_pk_bf8_f32(P::up((unsigned short)dw[2 * q]), P::up((unsigned short)(dw[2 * q] >> 16)), 0u, false);
              r8[q] = __builtin_amdgcn_cvt_pk_bf8_f32(P::up((unsigned short)dw[2 * q + 1]), P::up((unsigned short)(dw[2 * q + 1] >> 16)), r8[q], true);
            }
            *reinterpret_cast<uint2*>(p.x8 + (long)mm * p.ldx8 + n0 + wc * 64 + (lane & 7) * 8) = make_uint2(r8[0], r8[1]);
          }
        }
      }
      // next tile: its residual rows go straight into the accumulators just freed (all 32 loads in flight
      // together, no temporaries, no wait inside the epilogue)
      if (HL && j + 1 < my_tiles) {
        // (the pair of the next tile, addressed like the stores above: one row-offset multiply per tile - res_next - a uniform step per row
        // group, rows past M clamped by a 64-bit min against the last row's offset)
        long ro = res_next + (long)(i * 16) * p.ldr;
        const long rmax = (long)(p.M - 1) * p.ldr;
        ro = ro < rmax ? ro : rmax;
        const long off = ro + n0n + wc * 64 + 8 * fg_e;
        acc[i][0] = *reinterpret_cast<const f32x4_t*>(p.r16 + off);
        acc[i][1] = *reinterpret_cast<const f32x4_t*>(p.r16 + off + 32);
        acc[i][2] = *reinterpret_cast<const f32x4_t*>(p.rlo + off);
        acc[i][3] = *reinterpret_cast<const f32x4_t*>(p.rlo + off + 32);
      } else if (ACC_RES && j + 1 < my_tiles) {
        load_resid(i, m0n, n0n);
      } else if (!CAN_FOLD) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
      if (CAN_FOLD && i < 7) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("v_mov_b32 %0, %1" : "=v"(r_cur) : "v"(r_next));
      }
    }
    if (stamp) { const unsigned long long t = clock64(); tE2 += t - te; }
#ifdef GAVA_TIMELINE
    if (p.dbg && lane == 0 && (wave & 3) == 0 && j < 10) p.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 32 + j * 3 + 2] = wall_clock64();
#endif
    if (CAN_FOLD && j + 1 < my_tiles) {
      // the next tile's fold block was issued before this epilogue's stores: it has landed once at most those are in flight
      // (partials mode: it landed, and was reduced, two stages ago)
      if (!p.fpart && !PP) {     // (PP: the fold block was retired by the counted waits of the k-tile it was fetched in)
        if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE <= 63 ? NSTORE : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      fold_init(j + 1);
    }
    // a full tile issued exactly NSTORE stores per wave after the in-flight stage: they may stay in flight
    // over the next wait (ragged tiles store fewer, and the accumulator-residual loads add to the count:
    // those cases fall back to vmcnt(0)).
    counted = (full && !ACC_RES && !(ABLATE & 4) && !(EPI == GAVA_EPI_F32 && p.x16)) ? ((EPI == GAVA_EPI_H16_QGELU && p.aux_out) ? 2 : 1) : 0;
    if (L8 && p.out8 && counted) counted = counted == 1 ? 3 : 0;
    if (PP && wr == 1 && j + 1 < my_tiles) __builtin_amdgcn_s_barrier();     // group 1 one barrier behind again
    m0p = m0; n0p = n0;
    m0 = m0n; n0 = n0n;
  }
  if (ROWSUM && p.rowsum_reduced) {
    __syncthreads();
    flush_rowsum(m0p, n0p);
  }
#ifdef GAVA_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.dbg && lane == 0 && (wave & 3) == 0) { p.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 32 + 30] = wall_clock64(); p.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 32 + 31] = my_tiles; }
#endif
  if (stamp && lane == 0) {
    const unsigned long long t = clock64();
    tE += t - ts;
    unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 8 + wave) * 8;
    d[0] = tW; d[1] = tB; d[2] = tC; d[3] = 0; d[4] = tE; d[5] = (unsigned long long)G; d[6] = tE1; d[7] = tE2;
  }
}


// The aligned walk's super-tile height (in m-tiles) for an fp32-output GEMM of tiles_m x tiles_n tiles on `blocks` workgroups, or
// 0 when the plain walk is taken: sn must divide the n-tiles, every CU of the grid is used and the block count per XCD does
// not exceed the rounds of the plain walk.  GAVA_TILE_ALIGN=0: never (A/B).
int aligned_walk_sm(int tiles_m, int tiles_n, int sn, int blocks, int avail) {
  static const bool align_ok = !(getenv("GAVA_TILE_ALIGN") && getenv("GAVA_TILE_ALIGN")[0] == '0');
  if (!align_ok || blocks != avail || tiles_n % sn) return 0;
  const int per_xcd = blocks / 8, a_sm = per_xcd / sn;
  if (a_sm < 1) return 0;
  const int units = ((tiles_m + a_sm - 1) / a_sm) * (tiles_n / sn);
  const int rounds_aligned = (units + 7) / 8, rounds_plain = ((tiles_m * tiles_n + 7) / 8 + per_xcd - 1) / per_xcd;
  return rounds_aligned <= rounds_plain ? a_sm : 0;
}

// the two-pass patch embedding's GEMM (A = the 16-bit patch matrix) on the persistent 256^2 kernel (ping-pong loop) instead of the
// 128^2 tile kernel when the batch gives it enough tiles; GAVA_PATCH_256=0: the 128^2 kernel (A/B)
bool patch_on_256() {
  static const int v = getenv("GAVA_PATCH_256") ? atoi(getenv("GAVA_PATCH_256")) : 1;
  return v != 0;
}

// where AUTO takes the ping-pong loop: GAVA_PP = 0 nowhere, 1 (default) wherever an instantiation exists, 2 the fp32-output GEMMs
// only (the residual producers out_proj / fc2 and the deep-K dgrad GEMMs of the backward).  With the four-phase loop the K = 768
// LayerNorm-folded consumers were 3-4 % slower on it and 2 was the default; on the two-phase loop they are level stand-alone and
// the forward gains with them (c2 -0.5 %, c5 -1.5 % on top of the producers': profiles/r04_pingpong.txt)
int pp_mode() {
  static const int v = getenv("GAVA_PP") ? atoi(getenv("GAVA_PP")) : 1;
  return v;
}

// CUs of the current device in whole XCD multiples, and the grid of a persistent launch
int persistent_cus() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu = prop.multiProcessorCount / 8 * 8;
    if (n_cu <= 0) n_cu = 8;
  }
  return n_cu;
}

template <class P, int KERN>
int launch_256(GemmParams gp, int epi, hipStream_t s) {
  gp.tiles_m = (gp.M + 255) / 256;
  gp.tiles_n = gp.N / 256;
  gp.n_tiles = gp.tiles_m * gp.tiles_n;
  gp.sn = gp.tiles_n < 4 ? gp.tiles_n : 4;
  gp.sm = 32 / gp.sn;
  if (gp.sm > gp.tiles_m) gp.sm = gp.tiles_m;
  const int n_cu = persistent_cus();
  if (n_cu < 0) return GAVA_ELAUNCH;
  // GAVA_CU_RESERVE (diagnostics, tools/side_probe.py) overrides what the caller asks for
  static const int forced = getenv("GAVA_CU_RESERVE") ? atoi(getenv("GAVA_CU_RESERVE")) : -1;
  const int reserve = forced >= 0 ? forced : (gp.cu_reserve > 0 ? gp.cu_reserve : 0);
  const int avail = n_cu - reserve > 8 ? (n_cu - reserve) / 8 * 8 : 8;
  const int blocks = gp.n_tiles < avail ? (gp.n_tiles + 7) / 8 * 8 : avail;
  // (A W-stationary tile walk - an XCD stays on one chunk of weight panels and walks down the M-groups - was measured in
  // round 2, tools/archive/r2_walk.sh: fabric fetch unchanged (FETCH_SIZE 421 -> 432 MB per fc1 launch), 3-6 % slower; removed.)
  dim3 grid(blocks), block(512);
  // aligned walk (fp32-output kernels): blocks of sm x sn tiles with sm * sn <= workgroups per XCD, taken when sn divides the
  // n-tiles, every XCD is full and the block count per XCD does not exceed the rounds of the plain walk; GAVA_TILE_ALIGN=0: off (A/B)
  bool align = false;
  // (the LayerNorm-folded consumers were measured with it too: fc1 0.5313 vs 0.5322 ms, no gain - their A operand is served by
  // the Infinity Cache whatever the walk - so only the fp32-output kernels carry the instantiation)
  if (epi == GAVA_EPI_F32) {
    const int a_sm = aligned_walk_sm(gp.tiles_m, gp.tiles_n, gp.sn, blocks, avail);
    if (a_sm) { align = true; gp.sm = a_sm; }
  }
  // the residual stream as a 16-bit pair (HL instantiations): the ping-pong loop only
  if (gp.r16) {
    const bool can_pp = gp.K % 128 == 0 && (gp.w_lo == 1 ? 2 : 1) * gp.K >= 256 && !gp.split_out && !gp.aux_out && !gp.aux;
    if (epi != GAVA_EPI_F32 || !can_pp || KERN != 3 || gp.w_lo == 2 || gp.out8 || gp.x8) return GAVA_EINVAL;
    if (align) hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, true, 0, true, true>), grid, block, 0, s, gp);
    else hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, false, 0, true, true>), grid, block, 0, s, gp);
    GAVA_CHECK_LAUNCH();
    return GAVA_OK;
  }
  // the 8-bit lo product / the bf8 copies: fp16 operands, the four per-block GEMMs of the vision tower in the forms the
  // inference driver launches them (LayerNorm-folded consumers, residual producers); anything else is rejected
  if (gp.w_lo == 2 || gp.out8 || gp.x8) {
    if constexpr (std::is_same<P, PrecF16>::value) {
      if (KERN != 3 || gp.split_out) return GAVA_EINVAL;
      const bool lo8 = gp.w_lo == 2;
      if (epi == GAVA_EPI_H16 && (gp.fstats || gp.fpart) && !gp.x8 && lo8)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16, false, false, true, false, 2>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_H16_QGELU && (gp.fstats || gp.fpart) && !gp.x8 && !gp.aux_out && lo8)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16_QGELU, false, false, true, false, 2>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && gp.resid && !gp.out8 && align && lo8)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, true, 2>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && gp.resid && !gp.out8 && align && !lo8)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, true, 1>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && gp.resid && !gp.out8 && lo8)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, false, 2>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && gp.resid && !gp.out8)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, false, 1>), grid, block, 0, s, gp);
      else
        return GAVA_EINVAL;
      GAVA_CHECK_LAUNCH();
      return GAVA_OK;
    } else {
      return GAVA_EINVAL;
    }
  }
  // the ping-pong k-loop (template flag PP): named by gava_gemm_args.kernel = GAVA_KERNEL_PP, or taken for the forward's four
  // per-block GEMMs when GAVA_PP != 0 (default: see pp_default()); K % 128 == 0 and at least four k-tiles
  {
    const bool can_pp = gp.K % 128 == 0 && (gp.w_lo == 1 ? 2 : 1) * gp.K >= 256 && !gp.split_out && !gp.aux_out && !gp.aux;
    const bool want_pp = gp.kernel == GAVA_KERNEL_PP ||
                         (gp.kernel == GAVA_KERNEL_AUTO && (pp_mode() == 1 || (pp_mode() == 2 && epi == GAVA_EPI_F32 && (gp.resid || gp.K >= 2048))));
    if (gp.kernel == GAVA_KERNEL_PP && !can_pp) return GAVA_EINVAL;
    if (want_pp && can_pp) {
      bool done = true;
      if (epi == GAVA_EPI_H16 && (gp.fstats || gp.fpart))
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16, false, false, true, false, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_H16_QGELU && (gp.fstats || gp.fpart))
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16_QGELU, false, false, true, false, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_H16)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16, false, false, false, false, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && gp.resid && align)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, true, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && gp.resid)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, false, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32 && align)          // the dgrad GEMMs of the backward (fp32 gradient accumulator, no residual)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, false, false, false, true, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, false, false, false, false, 0, true>), grid, block, 0, s, gp);
      else if (epi == GAVA_EPI_F32_PATCH && gp.A)     // the two-pass patch embedding's GEMM (A = the 16-bit patch matrix)
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32_PATCH, false, false, false, false, 0, true>), grid, block, 0, s, gp);
      else
        done = false;
      if (done) { GAVA_CHECK_LAUNCH(); return GAVA_OK; }
      if (gp.kernel == GAVA_KERNEL_PP) return GAVA_EINVAL;      // a named kernel that does not take the call is rejected, never replaced
    }
  }
#define GAVA_LAUNCH(EPI, RES, SPLIT) hipLaunchKernelGGL((gemm256_kernel<P, EPI, RES, SPLIT>), grid, block, 0, s, gp)
  switch (epi) {
    case GAVA_EPI_H16:
      if (gp.fstats || gp.fpart) {   // folded LayerNorm: its own instantiation, the plain kernels stay as they were
        if (KERN != 3 || gp.split_out) return GAVA_EINVAL;
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16, false, false, true>), grid, block, 0, s, gp);
      } else if (gp.split_out) GAVA_LAUNCH(GAVA_EPI_H16, false, true); else GAVA_LAUNCH(GAVA_EPI_H16, false, false);
      break;
    case GAVA_EPI_H16_QGELU:
      if (gp.fstats || gp.fpart) {
        if (KERN != 3 || gp.split_out) return GAVA_EINVAL;
        hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16_QGELU, false, false, true>), grid, block, 0, s, gp);
      } else if (gp.split_out) GAVA_LAUNCH(GAVA_EPI_H16_QGELU, false, true); else GAVA_LAUNCH(GAVA_EPI_H16_QGELU, false, false);
      break;
    case GAVA_EPI_F32:
      if (align) {
        if (gp.resid) hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, true, false, false, true>), grid, block, 0, s, gp);
        else hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_F32, false, false, false, true>), grid, block, 0, s, gp);
      } else if (gp.resid) GAVA_LAUNCH(GAVA_EPI_F32, true, false); else GAVA_LAUNCH(GAVA_EPI_F32, false, false);
      break;
    case GAVA_EPI_F32_PATCH: GAVA_LAUNCH(GAVA_EPI_F32_PATCH, false, false); break;
    case GAVA_EPI_H16_QGELU_BWD:
      if (KERN != 3) return GAVA_EINVAL;
      hipLaunchKernelGGL((gemm256_kernel<P, GAVA_EPI_H16_QGELU_BWD, false, false>), grid, block, 0, s, gp);
      break;
    default: return GAVA_EINVAL;
  }
#undef GAVA_LAUNCH
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}


// ---------------------------------------------------------------------------------------------
// v4 "pair": persistent 128 x 256 kernel for the fp32-output GEMMs with a heavy epilogue (out_proj, fc2 and the dgrad
// GEMMs: fp32 residual read + fp32 write + 16-bit copy + row sums = 10 bytes per output element), run as TWO independent
// 256-thread workgroups per CU.  The 256^2 kernel above has one workgroup per CU whose eight waves reach the epilogue
// together: its 640 KB per tile leave the CU while the matrix pipe idles (out_proj: k-loop 0.12 ms + epilogue 0.15 ms).
// Two co-resident workgroups fall out of phase by themselves - a workgroup stuck in its stores leaves the SIMDs' MFMA
// issue slots to the other one - so one tile's epilogue traffic drains under the other tile's k-loop.
//   * 4 waves as 4(M) x 1(N): a wave owns 32 rows x all 256 columns (2 x 16 accumulators = 128 VGPRs).  Its A rows are
//     nobody else's, so the A operand never enters the LDS: each lane fetches its MFMA fragment (row fr, 16 bytes of k)
//     straight from global memory, one stage ahead, into registers.  Only W (shared by the four waves) is staged by
//     LDS-DMA: 32 KiB per 64-deep stage, 2 stages = 64 KiB per workgroup, which is what lets two workgroups share a CU
//     (the 256^2 kernel's ring is 128 KiB).  LDS-DMA issue per MFMA is the same as in the 256^2 kernel (8 pieces per wave
//     and stage), LDS reads are 0.5 instead of 0.375 ds_read_b128 per MFMA.
//   * a wave holds whole 256-column row segments, so the LayerNorm row sums of the folding producers need no exchange
//     between waves: in-lane sum, two permlane swaps, one float2 per row and 256-column tile (the rowsum_reduced layout).
//   * natural column order (lane (fr, fg) holds columns 16 jj + 4 fg + r): residual loads and fp32 stores move 64
//     contiguous bytes per row and instruction; the 16-bit copy is staged through a wave-private 2 KiB LDS block so that it
//     leaves as whole 128-byte row segments; the bias sits in LDS (once per launch).
//   * every wait is vmcnt(0): inside a workgroup the epilogue's stores and the next tile's residual loads drain before
//     the next k-loop starts (loads and stores retire out of order with each other, a counted wait would not be safe
//     here) - the overlap comes from the other workgroup, not from within.
template <class P>
__global__ __launch_bounds__(256, 2)
void gemm_pair_kernel(const GemmParams p) {
  constexpr int BM = 128, BN = 256, NW = 4;
  constexpr int WSTAGE = BN * BK * 2;                       // 32 KiB
  constexpr int PPW = BN / 8 / NW;                          // 8 LDS-DMA pieces (1 KiB = 8 W rows) per wave and stage
  constexpr int XS_PITCH = 128, XS_WAVE = 16 * XS_PITCH;    // 16-bit copy staging: 16 rows x 128 B per wave
  constexpr int XS_OFF = 2 * WSTAGE, BIAS_OFF = XS_OFF + NW * XS_WAVE;
  __shared__ __attribute__((aligned(16))) char smem[BIAS_OFF + 4096];   // 76 KiB: two workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // bias of all N <= 1024 columns, once
  {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && tid * 4 < p.N) b4 = *reinterpret_cast<const float4*>(p.bias + tid * 4);
    *reinterpret_cast<float4*>(smem + BIAS_OFF + tid * 16) = b4;
  }
  __syncthreads();

  // ---- this workgroup's tiles: XCD-contiguous range, strided by the workgroups of the XCD (as the 256^2 kernel)
  const int nwg = p.n_tiles, nb = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = nb >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int x_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int x_count = xcd < r ? q + 1 : q;
  const int my_tiles = slot < x_count ? (x_count - slot + per_xcd - 1) / per_xcd : 0;
  if (my_tiles == 0) return;
  const int nk = p.K / BK;
  const int G = my_tiles * nk;
  auto tile_coords = [&](int j, int& m0, int& n0) {
    const int wg = x_first + slot + j * per_xcd;
    const int per_group = p.sm * p.tiles_n;
    const int g = wg / per_group, first_m = g * p.sm;
    const int sm = min(p.sm, p.tiles_m - first_m);
    const int w = wg - g * per_group;
    const int chunk = w / (sm * p.sn), rr = w - chunk * (sm * p.sn);
    m0 = (first_m + rr % sm) * BM;
    n0 = (chunk * p.sn + rr / sm) * BN;
  };

  // 32-bit element offsets from p.W / p.A (host guarantees they fit).  A wave's W pieces i = 0..7 are the rows
  // (wave + 4 i) * 8 + (lane >> 3): 32 rows apart, so they share the swizzle term ((row >> 1) & 7) and ONE lane-dependent
  // offset serves all eight (piece i adds the uniform 32 i ldw)
  unsigned wsrc, asrc[2];
  auto set_src = [&](int m0, int n0) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // recomputed per tile instead of hoisted (registers), as in the 256^2 kernel
    {
      const int row = wave * 8 + (ln >> 3);
      const int chunk = (ln & 7) ^ ((row >> 1) & 7);
      wsrc = (unsigned)(n0 + row) * (unsigned)p.ldw + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int gm = m0 + wave * 32 + i * 16 + (ln & 15);
      gm = gm < p.M ? gm : p.M - 1;
      asrc[i] = (unsigned)gm * (unsigned)p.lda + (ln >> 4) * 8;
    }
  };
  // A fragments [k half][row group]: two register sets, stage g computes from set g & 1 while the loads of stage g+1 land
  // in the other one (the stage loop is unrolled by two; K % 128 == 0 is the launcher's condition)
  s16x8_t aA[2][2], aB[2][2];
  auto issue = [&](int g, int kt, s16x8_t (&an)[2][2]) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(p.W + (size_t)(unsigned)(i * 32 * (int)p.ldw + kt * BK) + (size_t)wsrc),
                                       LDS_PTR(void, smem + (g & 1) * WSTAGE + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        an[ks][i] = *reinterpret_cast<const s16x8_t*>(p.A + (size_t)(asrc[i] + (unsigned)(kt * BK + ks * 32)));
  };
  // W fragment jj, k half ks: rows 16 jj + fr of the stage, 16-byte chunk (4 ks + fg) ^ ((row >> 1) & 7)
  const int w_row = fr * 128;
  const int w_k0 = (fg ^ (fr >> 1)) << 4, w_k1 = ((4 + fg) ^ (fr >> 1)) << 4;

  f32x4_t acc[2][16];
  auto load_resid = [&](int i, int c, int mm0, int nn0) {   // rows mm0 + wave*32 + i*16 + fr, columns nn0 + 64 c + 16 q + 4 fg ..
    if (p.resid) {
      int m = mm0 + wave * 32 + i * 16 + fr;
      m = m < p.M ? m : p.M - 1;
      const float* rp = p.resid + (long)m * p.ldr + nn0 + 64 * c + 4 * fg;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) acc[i][4 * c + qq] = *reinterpret_cast<const f32x4_t*>(rp + 16 * qq);
    } else {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) acc[i][4 * c + qq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  };

  int m0, n0, m0n = 0, n0n = 0;
  tile_coords(0, m0, n0);
  set_src(m0, n0);
  int role = 0; (void)role;
#ifdef GAVA_ENABLE_ABLATE   // experiment builds: GAVA_PAIR_MODE bits, GAVA_PAIR_DELAY in 10 ns ticks (launch_pair)
  {
    // which of the CU's two workgroups is this?  bit 1: by the wave slot the hardware gave wave 0 (HW_ID[3:0]), else by
    // the position in the XCD's dispatch order (second half = second workgroup of a CU if the dispatcher goes breadth-first)
    const unsigned hw = __builtin_amdgcn_s_getreg(63492);   // hwreg(HW_REG_HW_ID, 0, 32)
    const bool second = (p.ablate & 2) ? (hw & 15u) != 0 : slot >= per_xcd / 2;
    if ((p.ablate & 1) && second) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)p.pair_delay) __builtin_amdgcn_s_sleep(32);
    }
    if (((p.ablate & 4) && !second) || ((p.ablate & 32) && second)) __builtin_amdgcn_s_setprio(2);
    // 64: split roles - a CU's first workgroup runs only k-loops, its second only epilogues (can the two phases overlap at
    // all?); 128 / 256 with it: the epilogue / the k-loop group leaves at once (each role's time alone)
    if (p.ablate & 64) {
      if (((p.ablate & 128) && second) || ((p.ablate & 256) && !second)) return;
      role = second ? 16 : 8;
    }
  }
#endif
  issue(0, 0, aA);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) load_resid(i, c, m0, n0);

  // one 64-deep stage: everything this wave has in flight lands (its W pieces and A fragments of this stage, after an
  // epilogue also that tile's stores and this tile's residual loads), the barrier makes the other waves' pieces visible
  // and frees the other ring slot, the next stage's loads go out, 64 MFMAs
  auto stage = [&](int j, int kt, s16x8_t (&ac)[2][2], s16x8_t (&an)[2][2]) {
    const int g = j * nk + kt;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only (expcnt 7, lgkmcnt 15: untouched)
    __builtin_amdgcn_s_barrier();
    if (g + 1 < G) {
      if (kt + 1 < nk) {
        issue(g + 1, kt + 1, an);
      } else {
        tile_coords(j + 1, m0n, n0n);
        set_src(m0n, n0n);
        issue(g + 1, 0, an);
      }
    }
    const char* cur = smem + (g & 1) * WSTAGE + w_row;
    // 8 steps of (4 W fragments x 2 row groups); the fragments of step s+1 are read while step s multiplies
    s16x8_t wf[2][4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) wf[0][qq] = *reinterpret_cast<const s16x8_t*>(cur + qq * 2048 + w_k0);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int ks = st >> 2, grp = st & 3;
      if (st < 7) {
        const int ks1 = (st + 1) >> 2, grp1 = (st + 1) & 3;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          wf[(st + 1) & 1][qq] = *reinterpret_cast<const s16x8_t*>(cur + (4 * grp1 + qq) * 2048 + (ks1 ? w_k1 : w_k0));
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][4 * grp + qq] = P::mfma(wf[st & 1][qq], ac[ks][i], acc[i][4 * grp + qq]);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int st = 0; st < 7; ++st) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  };

  for (int j = 0; j < my_tiles; ++j) {
    for (int kt = 0; kt < nk; kt += 2) {
      if ((ABLATE | role) & 16) break;   // experiment builds: epilogue only
      stage(j, kt, aA, aB);
      stage(j, kt + 1, aB, aA);
    }
    if ((ABLATE | role) & 8) continue;   // experiment builds: k-loop only

    // ---- epilogue of tile j: lane holds out[m][n0 + 16 jj + 4 fg + r], m = m0 + wave*32 + i*16 + fr
    // every lane-dependent address is derived here from an opaque copy of the lane id: hoisted out of the tile loop these
    // terms stay live across the k-loop, spill, and every spill reload is a vmcnt wait (as in the 256^2 kernel)
    const bool has_next = j + 1 < my_tiles;
    int le = lane;
    asm volatile("" : "+v"(le));
    const int er = le & 15, eg = le >> 4;
    char* xs = smem + XS_OFF + wave * XS_WAVE;
    const int xs_w = er * XS_PITCH + (eg & 1) * 8, xs_sw = (er >> 1) & 7, xs_hi = eg >> 1;
    const int xr_row = le >> 3, xr_chunk = le & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wave * 32 + i * 16 + er;
      const bool valid = m < p.M;
      float* orow = reinterpret_cast<float*>(p.out) + (long)m * p.ldo + n0 + 4 * eg;
      float ps1 = 0.f, ps2 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4* bp = reinterpret_cast<const float4*>(smem + BIAS_OFF + (n0 + 64 * c + 4 * eg) * 4);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float4 b = bp[4 * qq];
          const f32x4_t a = acc[i][4 * c + qq];
          const float v0 = a[0] + b.x, v1 = a[1] + b.y, v2 = a[2] + b.z, v3 = a[3] + b.w;
          if (valid) *reinterpret_cast<float4*>(orow + 64 * c + 16 * qq) = make_float4(v0, v1, v2, v3);
          if (p.x16) {
            *reinterpret_cast<uint2*>(xs + xs_w + (((2 * qq + xs_hi) ^ xs_sw) << 4)) = pack4<P>(v0, v1, v2, v3);
            ps1 += (v0 + v1) + (v2 + v3);
            ps2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
          }
        }
        if (p.x16) {
          // lane l: row (l >> 3) (+8 in the second pass), 16-byte chunk (l & 7) of the 128-byte row segment
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row = xr_row + 8 * h;
            const uint4 d = *reinterpret_cast<const uint4*>(xs + row * XS_PITCH + ((xr_chunk ^ ((row >> 1) & 7)) << 4));
            const int mm = m0 + wave * 32 + i * 16 + row;
            if (mm < p.M) *reinterpret_cast<uint4*>(p.x16 + (long)mm * p.ldx16 + n0 + 64 * c + xr_chunk * 8) = d;
          }
        }
#ifdef GAVA_ENABLE_ABLATE   // experiment builds: a throttled epilogue (GAVA_PAIR_MODE bit 512, GAVA_PAIR_SLEEP x 64 cycles per chunk)
        if (p.ablate & 512) for (int z = 0; z < p.pair_sleep; ++z) __builtin_amdgcn_s_sleep(1);
#endif
        // next tile: its residual rows go straight into the accumulators just freed
        if (has_next) {
          if (p.resid) {
            int mn = m0n + wave * 32 + i * 16 + er;
            mn = mn < p.M ? mn : p.M - 1;
            const float* rp = p.resid + (long)mn * p.ldr + n0n + 64 * c + 4 * eg;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) acc[i][4 * c + qq] = *reinterpret_cast<const f32x4_t*>(rp + 16 * qq);
          } else {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) acc[i][4 * c + qq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      if (p.x16) {
        // the 4 lanes that share a row (fg = 0..3) cover its 256 columns: one (sum, sum^2) per row and 256-column tile
        ps1 = sum_across_lane_groups(ps1); ps2 = sum_across_lane_groups(ps2);
        if (eg == 0 && valid) p.rowsum[(long)m * 4 + n0 / 256] = make_float2(ps1, ps2);
      }
    }
    m0 = m0n; n0 = n0n;
  }
}

template <class P>
int launch_pair(GemmParams gp, hipStream_t s) {
  gp.tiles_m = (gp.M + 127) / 128;
  gp.tiles_n = gp.N / 256;
  gp.n_tiles = gp.tiles_m * gp.tiles_n;
  // super-tile: ~64 concurrent tiles per XCD; the sn W panels (all of them for N <= 1024) and sm A panels share its L2
  gp.sn = gp.tiles_n < 4 ? gp.tiles_n : 4;
  gp.sm = 64 / gp.sn;
  if (gp.sm > gp.tiles_m) gp.sm = gp.tiles_m;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GAVA_ELAUNCH;
    n_cu = prop.multiProcessorCount / 8 * 8;
    if (n_cu <= 0) n_cu = 8;
  }
  static const int forced = getenv("GAVA_CU_RESERVE") ? atoi(getenv("GAVA_CU_RESERVE")) : -1;
  const int reserve = forced >= 0 ? forced : (gp.cu_reserve > 0 ? gp.cu_reserve : 0);
  const int avail = 2 * (n_cu - reserve > 8 ? (n_cu - reserve) / 8 * 8 : 8);   // two workgroups per CU
  const int blocks = gp.n_tiles < avail ? (gp.n_tiles + 7) / 8 * 8 : avail;
#ifdef GAVA_ENABLE_ABLATE
  static const int pmode = getenv("GAVA_PAIR_MODE") ? atoi(getenv("GAVA_PAIR_MODE")) : 0;
  static const int pdelay = getenv("GAVA_PAIR_DELAY") ? atoi(getenv("GAVA_PAIR_DELAY")) : 0;
  static const int psleep = getenv("GAVA_PAIR_SLEEP") ? atoi(getenv("GAVA_PAIR_SLEEP")) : 0;
  gp.ablate = pmode; gp.pair_delay = pdelay; gp.pair_sleep = psleep;
#endif
  hipLaunchKernelGGL((gemm_pair_kernel<P>), dim3(blocks), dim3(256), 0, s, gp);
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}


template <class P>
int launch_prec(const GemmParams& gp, int epi, hipStream_t s) {
  static const int variant = getenv("GAVA_GEMM_VARIANT") ? atoi(getenv("GAVA_GEMM_VARIANT")) : 0;
  if (gp.r16) {   // the 16-bit residual pair: the persistent 256^2 kernel's ping-pong loop only; never fall back silently
    const bool fits = (unsigned long long)(gp.M + 256) * gp.lda < (1ull << 31) && (unsigned long long)gp.N * gp.ldw < (1ull << 31);
    if ((gp.kernel != GAVA_KERNEL_AUTO && gp.kernel != GAVA_KERNEL_PP) || gp.N % 256 || !fits || gp.frames || gp.clips) return GAVA_EINVAL;
    return launch_256<P, 3>(gp, epi, s);
  }
  if (gp.w_lo == 2 || gp.out8 || gp.x8) {   // only the persistent 256^2 kernel implements these; never fall back silently
    const bool fits = (unsigned long long)(gp.M + 256) * gp.lda < (1ull << 31) && (unsigned long long)gp.N * gp.ldw < (1ull << 31);
    if (gp.kernel == GAVA_KERNEL_PAIR || gp.N % 256 || !fits || gp.frames || gp.clips) return GAVA_EINVAL;
    return launch_256<P, 3>(gp, epi, s);
  }
  // fp32-output GEMMs with the heavy epilogue (residual stream, folding producers): the two-workgroups-per-CU kernel can
  // be named explicitly (gava_gemm_args.kernel)
  {
    const bool fits = (unsigned long long)(gp.M + 256) * gp.lda < (1ull << 31) && (unsigned long long)gp.N * gp.ldw < (1ull << 31);
    const long tiles128 = (long)((gp.M + 127) / 128) * (gp.N / 256);
    const bool can_pair = !gp.w_lo && epi == GAVA_EPI_F32 && gp.N % 256 == 0 && gp.N <= 1024 && gp.K % 128 == 0 && fits && !gp.frames &&
                          !gp.clips && (gp.rowsum_reduced || !gp.x16);
    if (gp.kernel == GAVA_KERNEL_PAIR) return can_pair ? launch_pair<P>(gp, s) : GAVA_EINVAL;
    if (gp.kernel == GAVA_KERNEL_256 || gp.kernel == GAVA_KERNEL_PP) {
      const bool can_256 = gp.N % 256 == 0 && fits && !gp.frames && !gp.clips;
      return can_256 ? launch_256<P, 3>(gp, epi, s) : GAVA_EINVAL;
    }
    if (gp.kernel != GAVA_KERNEL_AUTO) return GAVA_EINVAL;
    // Measured in round 3 (profiles/r03_pair_kernel.txt): the pair kernel does not win - out_proj 0.299 vs 0.276 ms, fc2 0.69 vs
    // 0.575 ms on the 256^2 kernel; the epilogue of one workgroup does not drain under the k-loop of the other.  It is never taken
    // automatically and has no environment switch: only gava_gemm_args.kernel = GAVA_KERNEL_PAIR names it (tests, A/B timing).
    (void)tiles128;
  }
  if (gp.rowsum_reduced || gp.fpart) {   // only the persistent kernels implement these; never fall back silently
    const bool fits = (unsigned long long)(gp.M + 256) * gp.lda < (1ull << 31) && (unsigned long long)gp.N * gp.ldw < (1ull << 31);
    return (gp.N % 256 == 0 && fits && !gp.frames && !gp.clips) ? launch_256<P, 3>(gp, epi, s) : GAVA_EINVAL;
  }
  // small-M problems (prompt path, text tower at few classes): 128x128 tiles fill more CUs;
  // the im2col-free patch loader lives in the templated kernel
  // im2col-free patch embedding: 128 x 256 tiles (8 waves) build every A tile for 3 instead of 6 N-tiles at D = 768
  if ((gp.frames || gp.clips) && gp.N % 256 == 0 && variant != 1 && variant != 10) return launch_tile<P, 128, 256, 2>(gp, epi, s);
  // small-M problems are a latency chain (the prompt path: 8 dependent launches of <= 36 workgroups each, beside the streaming QKV GEMM):
  // a 4-deep operand ring (three k-tiles in flight; 128 KiB of LDS, one workgroup per CU - they have few) hides the latencies they see
  // there.  Round 4, same box: stand-alone they take 11-12 us each whatever the depth, the c2 forward 21.31 -> 21.15 ms
  // (profiles/r04_pingpong.txt).  GAVA_SMALL_NST=2|3|4: A/B.
  static const int small_nst = getenv("GAVA_SMALL_NST") ? atoi(getenv("GAVA_SMALL_NST")) : 4;
  if (!gp.frames && !gp.clips && gp.M <= 2048 && variant != 1 && variant != 10 && small_nst == 4) return launch_tile<P, 128, 128, 4>(gp, epi, s);
  if (!gp.frames && !gp.clips && gp.M <= 2048 && variant != 1 && variant != 10 && small_nst == 3) return launch_tile<P, 128, 128, 3>(gp, epi, s);
  if (gp.frames || gp.clips || gp.M <= 2048 || variant == 1 || variant == 10) return launch_tile<P, 128, 128, 2>(gp, epi, s);
  if (variant == 2) return launch_tile<P, 256, 128, 3>(gp, epi, s);
  if (variant == 12 && gp.N % 256 == 0) return launch_tile<P, 128, 256, 2>(gp, epi, s);
  if (variant == 13 && gp.N % 256 == 0) return launch_tile<P, 128, 256, 3>(gp, epi, s);
  // measured at M = 100864 (c2): the persistent 256^2 kernel wins for N >= 1536 (qkv 0.42 vs 0.58 ms,
  // fc1 0.55 vs 0.78 ms) and, since the static wave priority, for the deep-K N = 768 GEMM (fc2 0.59 vs
  // 0.62 ms); the shallow one (out, K = 768: 0.28 vs 0.26 ms) stays on the 128^2 kernel, whose many small
  // workgroups spread the fp32 residual traffic better over its short k-loop
  const bool fits32 = (unsigned long long)(gp.M + 256) * gp.lda < (1ull << 31) && (unsigned long long)gp.N * gp.ldw < (1ull << 31);
  const long tiles256 = (long)((gp.M + 255) / 256) * (gp.N / 256);
  // ... and, since the fp32-output kernels use the natural column order (64 contiguous bytes per row and instruction in
  // the residual loads and the stores), also for the shallow fp32 GEMM: out 0.251 vs 0.268 ms on the 128^2 kernel
  if (gp.N % 256 == 0 && fits32 &&
      (gp.N >= 1536 || (gp.K >= 2048 && tiles256 >= 512) || (epi == GAVA_EPI_F32 && tiles256 >= 512) || variant == 3 ||
       (epi == GAVA_EPI_F32_PATCH && tiles256 >= 512 && patch_on_256())))
    return launch_256<P, 3>(gp, epi, s);
  return launch_tile<P, 128, 128, 2>(gp, epi, s);
}

}  // namespace

namespace gava {
// the conditions of launch_prec / launch_256 for the HL instantiations; GAVA_PAIR_STREAM=0: never (A/B against the fp32 stream)
bool gemm_takes_pair(int M, int N, int K, long lda, long ldw) {
  const char* e = getenv("GAVA_PAIR_STREAM");      // read per call: the tests switch it inside one process
  const bool on = !(e && e[0] == '0');
  const bool fits = (unsigned long long)(M + 256) * lda < (1ull << 31) && (unsigned long long)N * ldw < (1ull << 31);
  return on && M > 0 && N % 256 == 0 && N <= 1024 && K % 128 == 0 && K >= 256 && fits;
}
}  // namespace gava

extern "C" int gava_gemm_aligned_walk(int M, int N, int cu_reserve) {
  if (M <= 0 || N <= 0 || N % 256) return 0;
  const int n_cu = persistent_cus();
  if (n_cu < 0) return 0;
  static const int forced = getenv("GAVA_CU_RESERVE") ? atoi(getenv("GAVA_CU_RESERVE")) : -1;
  const int reserve = forced >= 0 ? forced : (cu_reserve > 0 ? cu_reserve : 0);
  const int avail = n_cu - reserve > 8 ? (n_cu - reserve) / 8 * 8 : 8;
  const int tiles_m = (M + 255) / 256, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  const int blocks = n_tiles < avail ? (n_tiles + 7) / 8 * 8 : avail;
  return aligned_walk_sm(tiles_m, tiles_n, tiles_n < 4 ? tiles_n : 4, blocks, avail) ? 1 : 0;
}

extern "C" int gava_gemm(const gava_gemm_args* a, gava_stream_t stream) {
  const bool patch_u8 = a && a->epilogue == GAVA_EPI_F32_PATCH && !a->A && !a->frames && a->clips;
  const bool patch_direct = a && a->epilogue == GAVA_EPI_F32_PATCH && !a->A && (a->frames || patch_u8);
  const bool hl = a && a->resid16 != nullptr;
  if (!a || (!a->A && !patch_direct) || !a->W || (!a->out && !hl)) return GAVA_EINVAL;
  if (hl && (a->epilogue != GAVA_EPI_F32 || a->out || a->resid || !a->resid_lo || !a->x16_out || !a->xlo_out || !a->rowsum_out ||
             a->ldr % 8 || a->ldr < a->N || a->ld_x16 < a->N || a->w_lo == 2 || a->out8 || a->x8_out || a->K % 128 ||
             (((uintptr_t)a->resid16 | (uintptr_t)a->resid_lo | (uintptr_t)a->xlo_out) & 15))) return GAVA_EINVAL;
  if (!hl && (a->resid_lo || a->xlo_out)) return GAVA_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return GAVA_EINVAL;
  if (a->N % 128 || a->K % BK) return GAVA_EINVAL;
  if (a->w_lo < 0 || a->w_lo > 2) return GAVA_EINVAL;
  if (a->ldw % 8 || a->ldw < (a->w_lo == 1 ? 2 : 1) * (int64_t)a->K) return GAVA_EINVAL;
  if (a->w_lo && (a->kernel == GAVA_KERNEL_PAIR || a->epilogue == GAVA_EPI_H16_QGELU_BWD)) return GAVA_EINVAL;
  if (a->w_lo == 2 && (!a->A8 || !a->W8 || a->K % 128 || a->lda8 != 2 * a->lda || a->ldw8 != 2 * a->ldw ||
                       (((uintptr_t)a->A8 | (uintptr_t)a->W8) & 15) || a->w8_exp < -100 || a->w8_exp > 100)) return GAVA_EINVAL;
  if ((a->w_lo == 2 || a->out8 || a->x8_out) && a->prec != GAVA_PREC_F16) return GAVA_EINVAL;
  if (a->out8 && ((a->epilogue != GAVA_EPI_H16 && a->epilogue != GAVA_EPI_H16_QGELU) || a->split_out || a->ldo8 % 16 ||
                  a->ldo8 < a->N || ((uintptr_t)a->out8 & 15))) return GAVA_EINVAL;
  if (a->x8_out && (!a->x16_out || a->ld_x8 % 8 || a->ld_x8 < a->N || ((uintptr_t)a->x8_out & 7))) return GAVA_EINVAL;
  if (!patch_direct && (a->lda % 8 || a->lda < a->K)) return GAVA_EINVAL;
  if (patch_direct && (a->patch <= 0 || a->frame_size % a->patch || 3 * a->patch * a->patch > a->K ||
                       (a->frame_size / a->patch) * (a->frame_size / a->patch) != a->n_patches ||
                       (!patch_u8 && (((uintptr_t)a->frames & 15) || ((a->patch & 7) == 0 && (a->frame_size & 3))))))
    return GAVA_EINVAL;
  if (patch_u8 && !a->clip_lut) return GAVA_EINVAL;
  if (((uintptr_t)a->W | (uintptr_t)a->out) & 15) return GAVA_EINVAL;
  if (a->A && ((uintptr_t)a->A & 15)) return GAVA_EINVAL;
  if (a->bias && ((uintptr_t)a->bias & 15)) return GAVA_EINVAL;
  if (!hl && (a->ldo % 4 || a->ldo < (a->split_out ? 3 : 1) * (int64_t)a->N)) return GAVA_EINVAL;
  if (a->split_out && a->epilogue != GAVA_EPI_H16 && a->epilogue != GAVA_EPI_H16_QGELU) return GAVA_EINVAL;
  if (a->epilogue == GAVA_EPI_H16_QGELU_BWD && (!a->aux || ((uintptr_t)a->aux & 15) || a->ldo % 8 ||
                                               (a->aux_prec != GAVA_PREC_F16 && a->aux_prec != GAVA_PREC_BF16))) return GAVA_EINVAL;
  if (a->x16_out && (a->epilogue != GAVA_EPI_F32 || !a->rowsum_out || a->ld_x16 % 8 || ((uintptr_t)a->x16_out & 15) || ((uintptr_t)a->rowsum_out & 7))) return GAVA_EINVAL;
  if (a->fold_stats && ((a->epilogue != GAVA_EPI_H16 && a->epilogue != GAVA_EPI_H16_QGELU) || a->bias || !a->fold_s || !a->fold_t ||
                        (((uintptr_t)a->fold_stats | (uintptr_t)a->fold_s | (uintptr_t)a->fold_t) & 15))) return GAVA_EINVAL;
  if (a->fold_partials && (a->fold_stats || (a->epilogue != GAVA_EPI_H16 && a->epilogue != GAVA_EPI_H16_QGELU) || a->bias || !a->fold_s ||
                           !a->fold_t || a->split_out || a->K > 1024 || a->K < 4 * BK || a->N % 256 ||
                           (((uintptr_t)a->fold_partials | (uintptr_t)a->fold_s | (uintptr_t)a->fold_t) & 15))) return GAVA_EINVAL;
  if (a->rowsum_reduced && (!a->x16_out || a->N % 256 || a->N > 1024)) return GAVA_EINVAL;
  if (a->aux_out && (a->epilogue != GAVA_EPI_H16_QGELU || a->split_out || ((uintptr_t)a->aux_out & 15) || a->ldo % 8)) return GAVA_EINVAL;
  if (a->epilogue == GAVA_EPI_F32 && a->resid && (a->ldr % 4 || ((uintptr_t)a->resid & 15))) return GAVA_EINVAL;
  if (a->epilogue == GAVA_EPI_F32_PATCH &&
      (!a->pos || !a->time || a->n_patches <= 0 || a->T <= 0 || a->M % a->n_patches)) return GAVA_EINVAL;
  GemmParams gp;
  gp.A = (const unsigned short*)a->A; gp.lda = a->lda;
  gp.W = (const unsigned short*)a->W; gp.ldw = a->ldw;
  gp.bias = a->bias; gp.out = a->out; gp.ldo = a->ldo;
  gp.resid = a->resid; gp.ldr = a->ldr;
  gp.aux = (const unsigned short*)a->aux; gp.aux_f16 = a->aux_prec == GAVA_PREC_F16; gp.aux_out = (unsigned short*)a->aux_out;
  gp.x16 = (unsigned short*)a->x16_out; gp.ldx16 = a->ld_x16; gp.rowsum = (float2*)a->rowsum_out;
  gp.fstats = (const float2*)a->fold_stats; gp.fs = a->fold_s; gp.ft = a->fold_t;
  gp.rowsum_reduced = a->rowsum_reduced; gp.fpart = (const float2*)a->fold_partials;
  gp.fold_slots = (a->K + 255) / 256; gp.fold_inv_d = 1.0f / (float)a->K;
  if (a->fold_stats || a->fold_partials) gp.bias = a->fold_t;   // t_n takes the bias registers of the epilogue
  gp.M = a->M; gp.N = a->N; gp.K = a->K;
  gp.scale_cols = a->scale_cols; gp.scale = a->scale;
  gp.pos = a->pos; gp.time = a->time; gp.n_patches = a->n_patches; gp.T = a->T;
  gp.frames = (patch_direct && !patch_u8) ? a->frames : nullptr; gp.fsize = a->frame_size; gp.patch = a->patch;
  gp.clips = patch_u8 ? a->clips : nullptr; gp.clut = a->clip_lut;
  gp.split_out = a->split_out;
  gp.cu_reserve = a->cu_reserve;
  gp.kernel = a->kernel;
  gp.pair_delay = 0; gp.pair_sleep = 0; gp.stagger_mode = 0; gp.operand_l2 = 0;
  gp.w_lo = a->w_lo; gp.nka = a->K / BK;
  gp.A8 = (const unsigned char*)a->A8; gp.W8 = (const unsigned char*)a->W8;
  gp.nk8 = a->w_lo == 2 ? a->K / 128 : 0;
  gp.scale8 = 0x01010101u * (unsigned)(127 - a->w8_exp);
  gp.out8 = (unsigned char*)a->out8; gp.ldo8 = a->ldo8; gp.x8 = (unsigned char*)a->x8_out; gp.ldx8 = a->ld_x8;
  gp.r16 = (const unsigned short*)a->resid16; gp.rlo = (const unsigned short*)a->resid_lo; gp.xlo = (unsigned short*)a->xlo_out;
#ifdef GAVA_ENABLE_ABLATE   // timing-probe builds only (tools/ab_build.sh NAME -DGAVA_ENABLE_ABLATE): results are WRONG by design
  static const int ablate = getenv("GAVA_GEMM_ABLATE") ? atoi(getenv("GAVA_GEMM_ABLATE")) : 0;
  gp.ablate = ablate;
#else
  gp.ablate = 0;
#endif
#ifdef GAVA_EXP_OPERAND_L2
  static const int opl2_ = getenv("GAVA_OPERAND_L2") ? atoi(getenv("GAVA_OPERAND_L2")) : 0;
  gp.operand_l2 = opl2_;
#endif
#ifdef GAVA_EXP_STAGGER
  static const int pdelay_ = getenv("GAVA_PAIR_DELAY") ? atoi(getenv("GAVA_PAIR_DELAY")) : 0;
  static const int smode_ = getenv("GAVA_STAGGER_MODE") ? atoi(getenv("GAVA_STAGGER_MODE")) : 0;
  gp.pair_delay = pdelay_; gp.stagger_mode = smode_;
#endif
  gp.dbg = gava::debug_buffer();
  hipStream_t s = (hipStream_t)stream;
  if (a->prec == GAVA_PREC_F16) return launch_prec<PrecF16>(gp, a->epilogue, s);
  if (a->prec == GAVA_PREC_BF16) return launch_prec<PrecBF16>(gp, a->epilogue, s);
  return GAVA_EINVAL;
}
