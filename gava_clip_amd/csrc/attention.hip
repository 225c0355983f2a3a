// attention.hip — fused softmax(Q K^T) V, head dim 64, 16-bit MFMA operands, fp32 softmax.
//
// gfx950 design: a problem is one (frame, head).  Its whole key set is short (<= 320 keys: 214 for ViT-B/16 with T=8), so
// K and V of one head are staged ONCE in LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass;
// 128-byte rows whose 16-byte chunks are XOR-swizzled by row & 6 on the source address - found by search to be
// conflict-free for both the ds_read_b128 K-fragment reads and the ds_read_b64_tr_b16 transposed V reads) and softmax is
// single pass: no online rescaling.  Two kernels: attention_kernel, one 4-wave workgroup per problem (two per CU), and
// attention_persist_kernel for the vision blocks at 209..224 keys, one 8-wave workgroup per CU walking the problems with
// K/V double-buffered and the next problem staged by the wave that has no query tiles (see there).
// Round-2 measurements behind the present form (tools/attn_stamps.py, tools/archive/r2_attn_ab.sh; ViT-B/16 T=8 B=64, 0.200 ms
// before): loads alone 0.072 ms, compute alone 0.158 ms, so the load phase was only partly hidden by the CU's second
// workgroup -> persistent + prefetch; per-tile mask branches cost 750 of a pair's 5900 cycles -> branch-free masks on the
// tiles that can need them; 56 LDS-DMA issues per problem take ~3000 cycles to be accepted -> one loader wave; the kernel
// now moves its 650 MB at 4.0 TB/s (0.162 ms).  Tried and dropped: K first / V later staging with a second barrier (slower),
// a two-block online-softmax order that interleaves MFMAs with the exp/max VALU work inside a wave (instruction order
// verified in the ISA; 5 % slower than the plain phase order), 8 waves per workgroup for the 320-key class.
// Per 16-query tile a wave computes S^T = K Q^T with the KEY on the MFMA row, so every lane
// holds, for its own query (lane&15), 4 consecutive keys per 16-key tile.  That accumulator
// layout is already the B-operand layout of the second product O^T = V^T P^T (k-slot order
// permuted identically on both operands), so P never leaves registers; V^T fragments come from
// the hardware-transposing LDS read.  O^T leaves each lane with 4 consecutive head-dim columns
// of its own query: 8-byte stores, row sum in-lane.
// Vision "side" keys (global/local prompts, summary token) are gathered from a separate small
// K/V matrix while staging, so prompt tokens are never materialised per frame.
#include <cstdlib>
#include "common.h"
#include "internal.h"

namespace {

constexpr int LDS_ROW = 128;  // bytes per K/V row in LDS (64 x 2 B, unpadded: 16-byte chunk c of row r sits at chunk c ^ (r & 6))

// LDS-DMA by inline asm.  Through the builtin the compiler knows that a VMEM load writes LDS and guards every later LDS read
// with s_waitcnt vmcnt(0) (it cannot tell buffers apart, nor count past the DMA loads): in the persistent kernel that
// would wait for the NEXT problem's K/V at the first K fragment of this one.  Issued from asm the transfer is invisible
// to it; every consumer below sits behind an explicit s_waitcnt vmcnt + s_barrier.  (The compiler's own vmcnt waits for
// the loads it does track stay correct: they can only over-wait, VMEM returns in order.)  M0 = LDS base of the 1 KiB
// piece; nothing else in these kernels uses M0.
__device__ __forceinline__ unsigned lds_addr(const char* ptr) { return (unsigned)(size_t)LDS_PTR(char, ptr); }
__device__ __forceinline__ void dma16(const unsigned short* src, unsigned lds_piece) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_piece), "v"(src) : "memory");
}

struct AttnParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; long ld;
  const unsigned short* sk; const unsigned short* sv; long lds;
  unsigned short* out; long ldo;
  int batch, heads, n_q, n_kmain, n_g, T, has_summary, n_keys, causal, split;
  int qbr; long ldq;
  unsigned long long* dbg;
};

constexpr float LOG2E = 1.4426950408889634f;
#ifndef GAVA_ATTN_ONESUM   // row sums of P by one more MFMA per 32 keys (an all-ones "V" row) instead of one v_add per score
#define GAVA_ATTN_ONESUM 1
#endif

// Stage K and V of problem (n, h) by LDS-DMA: lane l of a 1 KiB piece lands at row (l >> 3), physical chunk (l & 7), so it
// fetches the logical chunk (l & 7) ^ (row & 6).  Rows beyond n_keys fetch row 0: their scores are masked to -inf, P = 0,
// and 0 * V must stay finite.  No wait here: the caller owns the s_waitcnt vmcnt and the barrier.
template <int KP, int NTH>
__device__ __forceinline__ void stage_kv(const AttnParams& p, int n, int h, char* Ks, char* Vs, int tid, int wave) {
  constexpr int NIT = (KP * 8 + NTH - 1) / NTH;      // staging tasks (16 B of K and of V) per thread
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * NTH;
    if (id < KP * 8) {                                                 // wave-uniform: KP * 8 is a multiple of 64
      const int row = id >> 3, chunk = (id & 7) ^ (row & 6);
      const int rowc = row < p.n_keys ? row : 0;
      const int sidx = rowc - p.n_kmain;                               // >= 0: side row
      const long sr = sidx < p.n_g ? sidx
                    : sidx < p.n_g + p.T ? p.n_g + (long)(n / p.T) * p.T + (sidx - p.n_g)
                                         : (long)p.n_g + p.batch + n;
      const bool is_main = rowc < p.n_kmain;
      const long off = (is_main ? ((long)n * p.n_kmain + rowc) * p.ld : sr * p.lds) + h * 64 + chunk * 8;
      const unsigned short* kb = is_main ? p.k : p.sk;
      const unsigned short* vb = is_main ? p.v : p.sv;
      const int dst = (wave * 64 + it * NTH) * 16;
      dma16(kb + off, lds_addr(Ks) + dst);
      dma16(vb + off, lds_addr(Vs) + dst);
    }
  }
}

// One wave stages a whole problem (the persistent kernel's loader wave): 1 KiB pieces of 8 rows each.  Pieces that lie
// entirely in the main rows advance two pointers by 8 rows; the few pieces with prompt/summary/padding rows take the
// general row -> source map of stage_kv.
template <int KP>
__device__ __forceinline__ void stage_kv_one_wave(const AttnParams& p, int n, int h, char* Ks, char* Vs, int lane) {
  const int r8 = lane >> 3, chunk = (lane & 7) ^ (r8 & 6);           // row & 6 == r8 & 6: pieces start at multiples of 8
  const int n_fast = p.n_kmain >> 3;
  const long o_main = ((long)n * p.n_kmain + r8) * p.ld + h * 64 + chunk * 8;
  const unsigned short* kp = p.k + o_main;
  const unsigned short* vp = p.v + o_main;
  const unsigned kdst = lds_addr(Ks), vdst = lds_addr(Vs);
  int j = 0;
  for (; j < n_fast; ++j) {
    dma16(kp, kdst + j * 1024);
    dma16(vp, vdst + j * 1024);
    kp += 8 * p.ld; vp += 8 * p.ld;
  }
  for (; j < KP / 8; ++j) {
    const int row = j * 8 + r8;
    const int rowc = row < p.n_keys ? row : 0;
    const int sidx = rowc - p.n_kmain;                               // >= 0: side row
    const long sr = sidx < p.n_g ? sidx
                  : sidx < p.n_g + p.T ? p.n_g + (long)(n / p.T) * p.T + (sidx - p.n_g)
                                       : (long)p.n_g + p.batch + n;
    const bool is_main = rowc < p.n_kmain;
    const long off = (is_main ? ((long)n * p.n_kmain + rowc) * p.ld : sr * p.lds) + h * 64 + chunk * 8;
    dma16((is_main ? p.k : p.sk) + off, kdst + j * 1024);
    dma16((is_main ? p.v : p.sv) + off, vdst + j * 1024);
  }
}

// Per-lane LDS addresses of the fragment reads (byte addresses; add the buffer base).
struct FragAddr {
  unsigned k0, k1;   // K fragments: row (tile * 16 + fr), logical chunks fg and fg + 4
  unsigned v[4];     // transposed V reads: lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3 of group dt
};
__device__ __forceinline__ FragAddr frag_addr(int fr, int fg) {
  FragAddr a;
  const int ksw = fr & 6;
  a.k0 = fr * LDS_ROW + ((fg ^ ksw) << 4);
  a.k1 = fr * LDS_ROW + (((fg + 4) ^ ksw) << 4);
  // row 4 fg + (fr >> 2) (+16, + 32 per chunk: row & 6 unchanged), bytes (fr & 3) * 8 of the 32-byte group dt:
  // logical chunk 2 dt + ((fr >> 1) & 1) -> physical chunk 2 (dt ^ gv) + ((fr >> 1) & 1), gv = (row & 6) >> 1
  const int gv = ((4 * fg + (fr >> 2)) & 6) >> 1;
  const int tr_off = (4 * fg + (fr >> 2)) * LDS_ROW + (fr & 3) * 8;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) a.v[dt] = tr_off + ((dt ^ gv) << 5);
  return a;
}

// Two query tiles (16 queries each) against all keys in LDS: every K fragment and every transposed V fragment read from
// LDS feeds two MFMAs instead of one (half the LDS traffic per query).  Returns the unnormalised O^T accumulators and the
// reciprocal row sums; the caller stores (pair_store) - the persistent kernel waits for its prefetch in between.
// FULL: leading key tiles the dispatcher guarantees to hold only valid keys (no mask code for them at all); PCH: 32-key
// chunks of V in flight per batch (registers).
// DUAL = false: the second tile of the pair does not exist (an odd number of query tiles: ViT-L/14's 257 queries are 16 full
// tiles + 1 query) - every MFMA, exp and pack of the b half is left out; same registers, a little more than half the work.
template <class P, int NKT, int FULL, int PCH = (NKT > 14 ? 2 : 4), bool DUAL = true>
__device__ __forceinline__ void pair_compute(const AttnParams& p, const char* ks, const char* vs, const FragAddr& fa, int fg,
                                             s16x8_t q0, s16x8_t q1, s16x8_t qb0, s16x8_t qb1,
                                             f32x4_t (&oa)[4], f32x4_t (&ob)[4], float& inva, float& invb,
                                             unsigned long long* stamps = nullptr) {
#ifdef GAVA_ATTN_STAMPS   // diagnostics build only (tools/attn_stamps.py): phase boundaries of the first pair of a wave
#define ATTN_STAMP(i, dep) do { if (stamps) { asm volatile("" :: "v"(dep)); stamps[i] = clock64(); } } while (0)
#else
#define ATTN_STAMP(i, dep) do { } while (0)
#endif
  ATTN_STAMP(0, q0);
  f32x4_t sa[NKT], sb[NKT];
  constexpr int QCH = NKT % 7 == 0 ? 7 : (NKT % 5 == 0 ? 5 : (NKT % 4 == 0 ? 4 : (NKT % 3 == 0 ? 3 : 2)));
  static_assert(NKT % QCH == 0, "key tiles per batch");
  const char* ka0 = ks + fa.k0;
  const char* ka1 = ks + fa.k1;
  // S^T = K Q^T.  All K fragments of a batch are requested from LDS before the first MFMA.
#pragma unroll
  for (int c0 = 0; c0 < NKT; c0 += QCH) {
    s16x8_t kf[QCH][2];
#pragma unroll
    for (int t = 0; t < QCH; ++t) {
      kf[t][0] = *reinterpret_cast<const s16x8_t*>(ka0 + (c0 + t) * 16 * LDS_ROW);
      kf[t][1] = *reinterpret_cast<const s16x8_t*>(ka1 + (c0 + t) * 16 * LDS_ROW);
    }
#pragma unroll
    for (int t = 0; t < QCH; ++t) {
      f32x4_t a = (f32x4_t){0.f, 0.f, 0.f, 0.f}, b = a;
      a = P::mfma(kf[t][0], q0, a);
      if (DUAL) b = P::mfma(kf[t][0], qb0, b);
      a = P::mfma(kf[t][1], q1, a);
      if (DUAL) b = P::mfma(kf[t][1], qb1, b);
      sa[c0 + t] = a;
      sb[c0 + t] = b;
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * QCH, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, (DUAL ? 4 : 2) * QCH, 0);
  }
  ATTN_STAMP(1, sb[NKT - 1]);
  // Invalid keys (rows past n_keys, staged as copies of row 0) get -inf: unconditionally on the tiles that may hold one -
  // a wave-uniform branch per tile costs more than the eight selects (its hoisted masks live in spilled SGPRs).
  float mxa = -INFINITY, mxb = -INFINITY;
  int lane_key = 4 * fg;
  asm volatile("" : "+v"(lane_key));   // keeps the compares inside the pair loop: hoisted, the 4 masks per tile spill
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt >= FULL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = kt * 16 + lane_key + r < p.n_keys;
        sa[kt][r] = ok ? sa[kt][r] : -INFINITY;
        if (DUAL) sb[kt][r] = ok ? sb[kt][r] : -INFINITY;
      }
    }
    mxa = fmaxf(fmaxf(mxa, sa[kt][0]), fmaxf(sa[kt][1], fmaxf(sa[kt][2], sa[kt][3])));
    if (DUAL) mxb = fmaxf(fmaxf(mxb, sb[kt][0]), fmaxf(sb[kt][1], fmaxf(sb[kt][2], sb[kt][3])));
  }
  mxa = max_across_lane_groups(mxa);
  if (DUAL) mxb = max_across_lane_groups(mxb);
  const float mna = -mxa * LOG2E, mnb = -mxb * LOG2E;
  ATTN_STAMP(2, mnb);
  float suma = 0.f, sumb = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ea = __builtin_amdgcn_exp2f(fmaf(sa[kt][r], LOG2E, mna));
      sa[kt][r] = ea;
      float eb = 0.f;
      if (DUAL) { eb = __builtin_amdgcn_exp2f(fmaf(sb[kt][r], LOG2E, mnb)); sb[kt][r] = eb; }
      if (!GAVA_ATTN_ONESUM) { suma += ea; sumb += eb; }
    }

  ATTN_STAMP(3, sb[NKT - 1]);
  // O^T = V^T P^T: every transposed V read of a batch is in flight before its MFMAs.
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { oa[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; ob[dt] = oa[dt]; }
  // The softmax denominators ride on the matrix core: an all-ones A operand makes every row of the product the sum over
  // all keys of the ROUNDED probabilities - the very numbers the numerator multiplies - with no cross-lane reduction.
  f32x4_t osa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, osb = osa;
  const unsigned short one = P::cvt(1.0f);
  const s16x8_t ones = {(short)one, (short)one, (short)one, (short)one, (short)one, (short)one, (short)one, (short)one};
  constexpr int NC2 = NKT / 2;                       // 32-key chunks
#pragma unroll
  for (int b0 = 0; b0 < NC2; b0 += PCH) {
    const int nb = NC2 - b0 < PCH ? NC2 - b0 : PCH;  // compile-time after unrolling
    s16x4_t t0[PCH][4], t1[PCH][4];
#pragma unroll
    for (int c = 0; c < PCH; ++c) {
      if (c < nb) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const char* vb = vs + fa.v[dt] + (b0 + c) * 32 * LDS_ROW;
          t0[c][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, vb));
          t1[c][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, vb + 16 * LDS_ROW));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < PCH; ++c) {
      if (c < nb) {
        const int cc = b0 + c;
        const uint2 la = pack4<P>(sa[2 * cc][0], sa[2 * cc][1], sa[2 * cc][2], sa[2 * cc][3]);
        const uint2 ha = pack4<P>(sa[2 * cc + 1][0], sa[2 * cc + 1][1], sa[2 * cc + 1][2], sa[2 * cc + 1][3]);
        const s16x8_t pfa = __builtin_bit_cast(s16x8_t, make_uint4(la.x, la.y, ha.x, ha.y));
        s16x8_t pfb = pfa;
        if (DUAL) {
          const uint2 lb = pack4<P>(sb[2 * cc][0], sb[2 * cc][1], sb[2 * cc][2], sb[2 * cc][3]);
          const uint2 hb = pack4<P>(sb[2 * cc + 1][0], sb[2 * cc + 1][1], sb[2 * cc + 1][2], sb[2 * cc + 1][3]);
          pfb = __builtin_bit_cast(s16x8_t, make_uint4(lb.x, lb.y, hb.x, hb.y));
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x8_t vf = __builtin_shufflevector(t0[c][dt], t1[c][dt], 0, 1, 2, 3, 4, 5, 6, 7);
          oa[dt] = P::mfma(vf, pfa, oa[dt]);
          if (DUAL) ob[dt] = P::mfma(vf, pfb, ob[dt]);
        }
        if (GAVA_ATTN_ONESUM) {
          osa = P::mfma(ones, pfa, osa);
          if (DUAL) osb = P::mfma(ones, pfb, osb);
        }
      }
    }
  }
  if (GAVA_ATTN_ONESUM) {
    suma = osa[0]; sumb = osb[0];
  } else {
    suma = sum_across_lane_groups(suma);
    sumb = sum_across_lane_groups(sumb);
  }
  inva = __builtin_amdgcn_rcpf(suma); invb = DUAL ? __builtin_amdgcn_rcpf(sumb) : 0.f;
  ATTN_STAMP(4, ob[3]);
#undef ATTN_STAMP
}

// O^T leaves each lane with 4 consecutive head-dim columns of its own query: 8-byte stores.
template <class P>
__device__ __forceinline__ void pair_store_at(unsigned short* opa, bool valid_a, unsigned short* opb, bool valid_b,
                                              const f32x4_t (&oa)[4], const f32x4_t (&ob)[4], float inva, float invb) {
  if (valid_a) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<uint2*>(opa + dt * 16) = pack4<P>(oa[dt][0] * inva, oa[dt][1] * inva, oa[dt][2] * inva, oa[dt][3] * inva);
  }
  if (valid_b) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<uint2*>(opb + dt * 16) = pack4<P>(ob[dt][0] * invb, ob[dt][1] * invb, ob[dt][2] * invb, ob[dt][3] * invb);
  }
}
template <class P>
__device__ __forceinline__ void pair_store(const AttnParams& p, int n, int h, int pr, int fr, int fg,
                                           const f32x4_t (&oa)[4], const f32x4_t (&ob)[4], float inva, float invb) {
  const int qia = pr * 32 + fr, qib = qia + 16;
  pair_store_at<P>(p.out + ((long)n * p.n_q + qia) * p.ldo + h * 64 + 4 * fg, qia < p.n_q,
                   p.out + ((long)n * p.n_q + qib) * p.ldo + h * 64 + 4 * fg, qib < p.n_q, oa, ob, inva, invb);
}

__device__ __forceinline__ const unsigned short* q_row_ptr(const AttnParams& p, int n, int h, int qt, int fr, int fg) {
  const int qi = qt * 16 + fr;
  const int qrow = qi < p.n_q ? qi : p.n_q - 1;   // clamped: a tile beyond n_q computes garbage, stores nothing
  return p.q + ((long)n * p.qbr + qrow) * p.ldq + h * 64 + 8 * fg;
}

// One workgroup (4 waves) per (problem, head), two workgroups per CU: 2 x 56 KiB of LDS at 224 keys, 2 x 80 KiB (all of
// it) for the 320-key class (ViT-L/14 with T = 32).
template <class P, int NKT, bool CAUSAL, bool PAIR = false, int FULL = 0>
__global__ __launch_bounds__(256, 2) void attention_kernel(const AttnParams p) {
  constexpr int NWV = 4;
  constexpr int KP = NKT * 16;
  __shared__ __attribute__((aligned(16))) char Ks[KP * LDS_ROW];
  __shared__ __attribute__((aligned(16))) char Vs[KP * LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x / p.heads, h = blockIdx.x - n * p.heads;
  const int fr = lane & 15, fg = lane >> 4;
  const int n_qt = (p.n_q + 15) >> 4;
  const unsigned long long t_start = p.dbg ? clock64() : 0;
  unsigned long long t_loads = 0, t_staged = 0;

  // ---- Q fragments of this wave's first query tile(s): in flight while K/V are staged
  auto q_ptr = [&](int qt) { return q_row_ptr(p, n, h, qt, fr, fg); };
  s16x8_t q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0;
  s16x8_t qb0 = q0, qb1 = q0;   // PAIR: second query tile of the wave's pair
  if constexpr (PAIR) {
    if (2 * wave < n_qt) {
      const unsigned short* qp = q_ptr(2 * wave);
      q0 = *reinterpret_cast<const s16x8_t*>(qp);
      q1 = *reinterpret_cast<const s16x8_t*>(qp + 32);
      const unsigned short* qq = q_ptr(2 * wave + 1);
      qb0 = *reinterpret_cast<const s16x8_t*>(qq);
      qb1 = *reinterpret_cast<const s16x8_t*>(qq + 32);
    }
  } else if (wave < n_qt) {
    const unsigned short* qp = q_ptr(wave);
    q0 = *reinterpret_cast<const s16x8_t*>(qp);
    q1 = *reinterpret_cast<const s16x8_t*>(qp + 32);
  }

  stage_kv<KP, NWV * 64>(p, n, h, Ks, Vs, tid, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.dbg) t_loads = clock64();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (p.dbg) t_staged = clock64();

  const FragAddr fa = frag_addr(fr, fg);
  const char* ks = Ks;
  const char* vs = Vs;

  if constexpr (PAIR) {
    // Tiles 2*pr and 2*pr+1; pairs are dealt round-robin to the 4 waves.
    const int n_pairs = (n_qt + 1) >> 1;
    for (int pr = wave; pr < n_pairs; pr += NWV) {
      s16x8_t na0 = q0, na1 = q1, nb0 = qb0, nb1 = qb1;
      if (pr + NWV < n_pairs) {
        const unsigned short* qp = q_ptr(2 * (pr + NWV));
        na0 = *reinterpret_cast<const s16x8_t*>(qp);
        na1 = *reinterpret_cast<const s16x8_t*>(qp + 32);
        const unsigned short* qq = q_ptr(2 * (pr + NWV) + 1);
        nb0 = *reinterpret_cast<const s16x8_t*>(qq);
        nb1 = *reinterpret_cast<const s16x8_t*>(qq + 32);
      }
      f32x4_t oa[4], ob[4];
      float inva, invb;
#ifdef GAVA_ATTN_STAMPS
      unsigned long long st[5] = {0, 0, 0, 0, 0};
      pair_compute<P, NKT, FULL>(p, ks, vs, fa, fg, q0, q1, qb0, qb1, oa, ob, inva, invb, p.dbg && pr == wave ? st : nullptr);
      if (p.dbg && pr == wave && lane == 0 && blockIdx.x < 4096) {
        unsigned long long* d = p.dbg + 4096 * 16 + (size_t)(blockIdx.x * 4 + wave) * 4;
        for (int i = 0; i < 4; ++i) d[i] = st[i + 1] - st[i];
      }
#elif defined(GAVA_ATTN_ODD_TILE)
      // experiment build (tools/ab_build.sh odd -DGAVA_ATTN_ODD_TILE; profiles/r03_attention_c5.txt): an odd last tile
      // (ViT-L/14: 17 tiles) runs the one-tile form of the same code.  5 % SLOWER at c5 (256 VGPRs + scratch with the second
      // instantiation inlined), so the product build keeps the half-empty pair.
      if (2 * pr + 1 < n_qt) pair_compute<P, NKT, FULL>(p, ks, vs, fa, fg, q0, q1, qb0, qb1, oa, ob, inva, invb);
      else pair_compute<P, NKT, FULL, (NKT > 14 ? 2 : 4), false>(p, ks, vs, fa, fg, q0, q1, qb0, qb1, oa, ob, inva, invb);
#else
      pair_compute<P, NKT, FULL>(p, ks, vs, fa, fg, q0, q1, qb0, qb1, oa, ob, inva, invb);
#endif
      pair_store<P>(p, n, h, pr, fr, fg, oa, ob, inva, invb);
      q0 = na0; q1 = na1; qb0 = nb0; qb1 = nb1;
    }
  } else
  for (int qt = wave; qt < n_qt; qt += NWV) {
    const int qi = qt * 16 + fr;
    // prefetch the next tile's Q while this one computes
    s16x8_t nq0 = q0, nq1 = q1;
    if (qt + NWV < n_qt) {
      const unsigned short* qp = q_ptr(qt + NWV);
      nq0 = *reinterpret_cast<const s16x8_t*>(qp);
      nq1 = *reinterpret_cast<const s16x8_t*>(qp + 32);
    }

    // S^T = K Q^T.  All K fragments of a chunk are requested from LDS before the first MFMA (the
    // compiler otherwise emits read -> wait -> 2 MFMA per tile and exposes the LDS latency 14 times).
    f32x4_t s[NKT];
    constexpr int QCH = NKT <= 14 ? NKT : NKT / 2;
    const char* ka0 = ks + fa.k0;
    const char* ka1 = ks + fa.k1;
#pragma unroll
    for (int c0 = 0; c0 < NKT; c0 += QCH) {
      s16x8_t kf[QCH][2];
#pragma unroll
      for (int t = 0; t < QCH; ++t) {
        kf[t][0] = *reinterpret_cast<const s16x8_t*>(ka0 + (c0 + t) * 16 * LDS_ROW);
        kf[t][1] = *reinterpret_cast<const s16x8_t*>(ka1 + (c0 + t) * 16 * LDS_ROW);
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
    // mask only the key tiles that can hold invalid keys (wave-uniform test), then row max
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const bool partial = (kt * 16 + 16 > p.n_keys) || (CAUSAL && kt * 16 + 15 > qt * 16);
      if (partial) {
        asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch: full tiles skip the masks
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * fg + r;
          const bool ok = key < p.n_keys && (!CAUSAL || key <= qi);
          s[kt][r] = ok ? s[kt][r] : -INFINITY;
        }
      }
      mx = fmaxf(fmaxf(mx, s[kt][0]), fmaxf(s[kt][1], fmaxf(s[kt][2], s[kt][3])));
    }
    mx = max_across_lane_groups(mx);
    const float mneg = -mx * LOG2E;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], LOG2E, mneg));   // exp(s - mx)
        s[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);

    // O^T = V^T P^T.  Same idea: every transposed V read of a chunk is in flight before its MFMAs.
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    constexpr int NC2 = NKT / 2;                       // 32-key chunks
    constexpr int PCH = NC2 <= 7 ? NC2 : NC2 / 2;      // chunks per batch
#pragma unroll
    for (int b0 = 0; b0 < NC2; b0 += PCH) {
      s16x4_t t0[PCH][4], t1[PCH][4];
#pragma unroll
      for (int c = 0; c < PCH; ++c) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const char* vb = vs + fa.v[dt] + (b0 + c) * 32 * LDS_ROW;
          t0[c][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, vb));
          t1[c][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, vb + 16 * LDS_ROW));
        }
      }
#pragma unroll
      for (int c = 0; c < PCH; ++c) {
        const int cc = b0 + c;
        const uint2 lo = pack4<P>(s[2 * cc][0], s[2 * cc][1], s[2 * cc][2], s[2 * cc][3]);
        const uint2 hi = pack4<P>(s[2 * cc + 1][0], s[2 * cc + 1][1], s[2 * cc + 1][2], s[2 * cc + 1][3]);
        const s16x8_t pf = __builtin_bit_cast(s16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x8_t vf = __builtin_shufflevector(t0[c][dt], t1[c][dt], 0, 1, 2, 3, 4, 5, 6, 7);
          o[dt] = P::mfma(vf, pf, o[dt]);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8 * PCH, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * PCH, 0);
    }
    if (qi < p.n_q) {
      unsigned short* op = p.out + ((long)n * p.n_q + qi) * p.ldo + h * 64 + 4 * fg;
      const int Dm = p.heads * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if (p.split) {
          uint2 hi, lo;
          split4<P>(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv, hi, lo);
          *reinterpret_cast<uint2*>(op + dt * 16) = hi;
          *reinterpret_cast<uint2*>(op + dt * 16 + Dm) = lo;
          *reinterpret_cast<uint2*>(op + dt * 16 + 2 * Dm) = hi;
        } else {
          *reinterpret_cast<uint2*>(op + dt * 16) =
              pack4<P>(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
        }
      }
    }
    q0 = nq0; q1 = nq1;
  }
  if (p.dbg && lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 4 + wave) * 4;
    d[0] = t_loads - t_start; d[1] = t_staged - t_loads; d[2] = clock64() - t_staged; d[3] = (n_qt - wave + NWV - 1) / NWV;
  }
}

#ifndef GAVA_PERSIST_PCH
#define GAVA_PERSIST_PCH 2
#endif
// Persistent form for the vision blocks: one workgroup of 8 waves per CU walks the (frame, head) problems with a stride of
// the grid, K/V double-buffered in LDS (2 x 2 x KP rows): the LDS-DMA loads and the Q fragments of the NEXT problem are
// issued before the current one is computed, so the load phase (a quarter of a workgroup's life in the kernel above, and
// only partly covered by the second workgroup of the CU) disappears behind the MFMA/softmax work.  One query-tile pair
// per wave (n_q <= 256).  One barrier per problem: a wave arrives after its own share of the next problem's loads has
// landed (vmcnt(0) BEFORE its output stores, which then drain under the next problem) and after its last LDS read of the
// current buffer, which is all the next iteration needs.
template <class P, int NKT, int FULL>
__global__ __launch_bounds__(512, 1) void attention_persist_kernel(const AttnParams p, const int n_prob) {
  constexpr int NWV = 8;
  constexpr int KP = NKT * 16;
  constexpr int BUF = KP * LDS_ROW;
  __shared__ __attribute__((aligned(16))) char Ks[2 * BUF];
  __shared__ __attribute__((aligned(16))) char Vs[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n_qt = (p.n_q + 15) >> 4;
  const bool has_pair = 2 * wave < n_qt;
  const FragAddr fa = frag_addr(fr, fg);
  const char* ks = Ks;
  const char* vs = Vs;

  int prob = blockIdx.x;
  if (prob >= n_prob) return;
  int n = prob / p.heads, h = prob - n * p.heads;
  // problem-independent parts of the Q / O addresses this thread forms, once
  const int qia = wave * 32 + fr, qib = qia + 16;
  const int qoa = (qia < p.n_q ? qia : p.n_q - 1) * (int)p.ldq + 8 * fg;   // a tile beyond n_q computes garbage, stores nothing
  const int qob = (qib < p.n_q ? qib : p.n_q - 1) * (int)p.ldq + 8 * fg;
  const int ooa = qia * (int)p.ldo + 4 * fg, oob = qib * (int)p.ldo + 4 * fg;
  const bool valid_a = qia < p.n_q, valid_b = qib < p.n_q;
  auto q_base = [&](int n_, int h_) { return p.q + ((long)n_ * p.qbr * p.ldq + h_ * 64); };
  // The last wave never has a query-tile pair here (n_q <= n_keys <= 224: at most 7 pairs) and alone issues the next
  // problem's K/V: 56 LDS-DMA instructions take ~3000 cycles to get accepted by the memory pipeline (measured), which the
  // computing waves would otherwise all spend at the top of every problem before their first MFMA.
  auto stage = [&](int n_, int h_, char* kd, char* vd) {
    if (wave == NWV - 1) stage_kv_one_wave<KP>(p, n_, h_, kd, vd, lane);
  };
  s16x8_t q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0, qb0 = q0, qb1 = q0;
  stage(n, h, Ks, Vs);
  if (has_pair) {
    const unsigned short* qp = q_base(n, h);
    q0 = *reinterpret_cast<const s16x8_t*>(qp + qoa);
    q1 = *reinterpret_cast<const s16x8_t*>(qp + qoa + 32);
    qb0 = *reinterpret_cast<const s16x8_t*>(qp + qob);
    qb1 = *reinterpret_cast<const s16x8_t*>(qp + qob + 32);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(qb0), "+v"(qb1) : : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#ifdef GAVA_ATTN_STAMPS   // per-wave sums over the problems: issue | compute | wait for the prefetch | stores | barrier
  unsigned long long acc[5] = {0, 0, 0, 0, 0}, tp = clock64();
  const unsigned long long w0 = wall_clock64(), c0 = tp;
#define PSTAMP(i) do { const unsigned long long t = clock64(); acc[i] += t - tp; tp = t; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
  int buf = 0;
  for (;;) {
    const int nxt = prob + gridDim.x;
    const bool more = nxt < n_prob;                     // workgroup-uniform
    const int nn = nxt / p.heads, nh = nxt - nn * p.heads;
    s16x8_t na0 = q0, na1 = q1, nb0 = qb0, nb1 = qb1;
    if (more) {
      stage(nn, nh, Ks + (buf ^ 1) * BUF, Vs + (buf ^ 1) * BUF);
      if (has_pair) {
        const unsigned short* qp = q_base(nn, nh);
        na0 = *reinterpret_cast<const s16x8_t*>(qp + qoa);
        na1 = *reinterpret_cast<const s16x8_t*>(qp + qoa + 32);
        nb0 = *reinterpret_cast<const s16x8_t*>(qp + qob);
        nb1 = *reinterpret_cast<const s16x8_t*>(qp + qob + 32);
      }
    }
    PSTAMP(0);
    if (has_pair) {
      f32x4_t oa[4], ob[4];
      float inva, invb;
      pair_compute<P, NKT, FULL, GAVA_PERSIST_PCH>(p, ks + buf * BUF, vs + buf * BUF, fa, fg, q0, q1, qb0, qb1, oa, ob, inva, invb);
#ifdef GAVA_ATTN_STAMPS
      asm volatile("" :: "v"(ob[3]), "v"(inva), "v"(invb));
#endif
      PSTAMP(1);
      // the next problem's K/V pieces and Q fragments of this wave have landed; nothing of this problem is stored yet
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(na0), "+v"(na1), "+v"(nb0), "+v"(nb1) : : "memory");
      PSTAMP(2);
      unsigned short* ob_ = p.out + ((long)n * p.n_q * p.ldo + h * 64);
      pair_store_at<P>(ob_ + ooa, valid_a, ob_ + oob, valid_b, oa, ob, inva, invb);
      PSTAMP(3);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PSTAMP(4);
    if (!more) break;
    q0 = na0; q1 = na1; qb0 = nb0; qb1 = nb1;
    prob = nxt; n = nn; h = nh; buf ^= 1;
  }
#ifdef GAVA_ATTN_STAMPS
  if (p.dbg && lane == 0) {
    unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 5; ++i) d[i] = acc[i];
    d[5] = clock64() - c0; d[6] = wall_clock64() - w0; d[7] = (n_prob - blockIdx.x + gridDim.x - 1) / gridDim.x;
  }
#endif
#undef PSTAMP
}

template <class P>
int launch_attn(const AttnParams& p, hipStream_t s) {
  const int n_prob = p.batch * p.heads;
  dim3 grid(n_prob), blk(256);
  const int tiles = (p.n_keys + 15) / 16;
  // two query tiles per wave for the big non-causal problems (vision blocks); GAVA_ATTN_PAIR=0 turns it off (A/B)
  static const bool pair_ok = !(getenv("GAVA_ATTN_PAIR") && getenv("GAVA_ATTN_PAIR")[0] == '0');
  const bool pair = pair_ok && !p.causal && !p.split && p.n_q >= 64;
  // persistent double-buffered form: 209..224 keys (14 key tiles, only the last can be partial) with at most 8 query-tile
  // pairs, and enough problems to keep every CU busy for several rounds; GAVA_ATTN_PERSIST=0 turns it off (A/B)
  static const bool persist_ok = !(getenv("GAVA_ATTN_PERSIST") && getenv("GAVA_ATTN_PERSIST")[0] == '0');
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GAVA_ELAUNCH;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 8;
  }
  #ifdef GAVA_ATTN_STAMPS
  const bool dbg_ok = true;
#else
  const bool dbg_ok = !p.dbg;
#endif
  if (persist_ok && pair && tiles == 14 && p.n_q <= 14 * 16 && dbg_ok && n_prob >= 4 * n_cu) {
    hipLaunchKernelGGL((attention_persist_kernel<P, 14, 13>), dim3(n_cu), dim3(512), 0, s, p, n_prob);
    GAVA_CHECK_LAUNCH();
    return GAVA_OK;
  }
  // (N, LOW): LOW = tile count of the class below, so tiles 0 .. LOW-1 are full; tiles == N: only the last can be partial
#define GAVA_ATTN(N, LOW)                                                                         \
  do {                                                                                            \
    if (p.causal) hipLaunchKernelGGL((attention_kernel<P, N, true, false>), grid, blk, 0, s, p);  \
    else if (pair && N >= 14 && tiles == N)                                                       \
      hipLaunchKernelGGL((attention_kernel<P, (N >= 14 ? N : 14), false, true, (N >= 14 ? N : 14) - 1>), grid, blk, 0, s, p); \
    else if (pair && N >= 14)                                                                     \
      hipLaunchKernelGGL((attention_kernel<P, (N >= 14 ? N : 14), false, true, (N >= 14 ? LOW : 6)>), grid, blk, 0, s, p); \
    else hipLaunchKernelGGL((attention_kernel<P, N, false, false>), grid, blk, 0, s, p);          \
  } while (0)
  if (tiles <= 2) GAVA_ATTN(2, 0);
  else if (tiles <= 6) GAVA_ATTN(6, 2);
  else if (tiles <= 14) GAVA_ATTN(14, 6);
  else if (tiles <= 20) GAVA_ATTN(20, 14);
  else return GAVA_EINVAL;
#undef GAVA_ATTN
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

// Exact-fp32 attention for short sequences (gava_attention_f32): one workgroup per (sequence, head), thread t owns query t.
// K and V of the head sit in LDS as fp32; every thread walks the keys in the same order, so each LDS read is a broadcast.
// Two passes over the keys (row max, then exp / sum / P.V): no rescaling, scores recomputed (64 FMAs) instead of kept.
template <class P>
__global__ __launch_bounds__(128) void attention_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, long ld, unsigned short* out, long ldo,
                                                            int heads, int L, int causal, int split, float scale) {
  extern __shared__ __attribute__((aligned(16))) float kv[];   // [2][L][64]
  float* Ks = kv;
  float* Vs = kv + (size_t)L * 64;
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const int t = threadIdx.x;
  for (int i = t; i < L * 16; i += blockDim.x) {
    const int row = i >> 4, c = (i & 15) * 4;
    const long off = ((long)n * L + row) * ld + h * 64 + c;
    *reinterpret_cast<float4*>(Ks + row * 64 + c) = *reinterpret_cast<const float4*>(k + off);
    *reinterpret_cast<float4*>(Vs + row * 64 + c) = *reinterpret_cast<const float4*>(v + off);
  }
  float qr[64];
  const int tq = t < L ? t : L - 1;
#pragma unroll
  for (int c = 0; c < 64; c += 4) {
    const float4 x = *reinterpret_cast<const float4*>(q + ((long)n * L + tq) * ld + h * 64 + c);
    qr[c] = x.x * scale; qr[c + 1] = x.y * scale; qr[c + 2] = x.z * scale; qr[c + 3] = x.w * scale;
  }
  __syncthreads();
  const int lim = causal ? tq : L - 1;                                     // last key this query sees
  const int kend = causal ? min(L, ((t >> 6) + 1) * 64) : L;              // wave-uniform loop bound
  auto dot = [&](int key) {
    const float4* kr = reinterpret_cast<const float4*>(Ks + key * 64);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 x = kr[c];
      s0 = fmaf(qr[4 * c], x.x, s0); s1 = fmaf(qr[4 * c + 1], x.y, s1);
      s2 = fmaf(qr[4 * c + 2], x.z, s2); s3 = fmaf(qr[4 * c + 3], x.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
  };
  float mx = -INFINITY;
  for (int key = 0; key < kend; ++key) {
    const float sc = dot(key);
    mx = key <= lim ? fmaxf(mx, sc) : mx;
  }
  float o[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) o[c] = 0.f;
  float sum = 0.f;
  for (int key = 0; key < kend; ++key) {
    const float sc = dot(key);
    const float pr = key <= lim ? __builtin_amdgcn_exp2f((sc - mx) * LOG2E) : 0.f;
    sum += pr;
    const float4* vr = reinterpret_cast<const float4*>(Vs + key * 64);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 x = vr[c];
      o[4 * c] = fmaf(pr, x.x, o[4 * c]); o[4 * c + 1] = fmaf(pr, x.y, o[4 * c + 1]);
      o[4 * c + 2] = fmaf(pr, x.z, o[4 * c + 2]); o[4 * c + 3] = fmaf(pr, x.w, o[4 * c + 3]);
    }
  }
  if (t >= L) return;
  const float inv = 1.0f / sum;
  unsigned short* op = out + ((long)n * L + t) * ldo + h * 64;
  const int Dm = heads * 64;
#pragma unroll
  for (int c = 0; c < 64; c += 4) {
    if (split) {
      uint2 hi, lo;
      split4<P>(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv, hi, lo);
      *reinterpret_cast<uint2*>(op + c) = hi;
      *reinterpret_cast<uint2*>(op + c + Dm) = lo;
      *reinterpret_cast<uint2*>(op + c + 2 * Dm) = hi;
    } else {
      *reinterpret_cast<uint2*>(op + c) = pack4<P>(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
    }
  }
}

}  // namespace

extern "C" int gava_attention_f32(const gava_attention_f32_args* a, gava_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return GAVA_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->L <= 0 || a->L > 128) return GAVA_EINVAL;
  if (a->ld % 4 || a->ld_out % 4 || a->ld < (int64_t)a->heads * 64) return GAVA_EINVAL;
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15 || ((uintptr_t)a->out & 7)) return GAVA_EINVAL;
  if (a->ld_out < (a->split_out ? 3 : 1) * (int64_t)a->heads * 64) return GAVA_EINVAL;
  const size_t lds = (size_t)2 * a->L * 64 * sizeof(float);
  const dim3 grid(a->batch * a->heads), blk(a->L <= 64 ? 64 : 128);
  hipStream_t s = (hipStream_t)stream;
  if (a->prec == GAVA_PREC_F16)
    hipLaunchKernelGGL((attention_f32_kernel<PrecF16>), grid, blk, lds, s, a->q, a->k, a->v, (long)a->ld, (unsigned short*)a->out,
                       (long)a->ld_out, a->heads, a->L, a->causal, a->split_out, a->scale);
  else if (a->prec == GAVA_PREC_BF16)
    hipLaunchKernelGGL((attention_f32_kernel<PrecBF16>), grid, blk, lds, s, a->q, a->k, a->v, (long)a->ld, (unsigned short*)a->out,
                       (long)a->ld_out, a->heads, a->L, a->causal, a->split_out, a->scale);
  else return GAVA_EINVAL;
  GAVA_CHECK_LAUNCH();
  return GAVA_OK;
}

extern "C" int gava_attention(const gava_attention_args* a, gava_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return GAVA_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_kmain <= 0) return GAVA_EINVAL;
  if (a->ld_qkv % 8 || a->ld_out % 4) return GAVA_EINVAL;
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15) return GAVA_EINVAL;
  if ((uintptr_t)a->out & 7) return GAVA_EINVAL;
  int n_side = 0;
  if (a->n_g || a->T || a->has_summary) {
    if (!a->side_k || !a->side_v || a->ld_side % 8 || a->n_g < 0 || a->T <= 0) return GAVA_EINVAL;
    if (((uintptr_t)a->side_k | (uintptr_t)a->side_v) & 15) return GAVA_EINVAL;
    if (a->batch % a->T) return GAVA_EINVAL;
    n_side = a->n_g + a->T + (a->has_summary ? 1 : 0);
  }
  AttnParams p;
  p.q = (const unsigned short*)a->q; p.k = (const unsigned short*)a->k; p.v = (const unsigned short*)a->v;
  p.ld = a->ld_qkv;
  p.sk = (const unsigned short*)a->side_k; p.sv = (const unsigned short*)a->side_v; p.lds = a->ld_side;
  p.out = (unsigned short*)a->out; p.ldo = a->ld_out;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_kmain = a->n_kmain;
  p.n_g = a->n_g; p.T = n_side ? a->T : 1; p.has_summary = a->has_summary;
  p.n_keys = a->n_kmain + n_side; p.causal = a->causal; p.split = a->split_out;
  if (a->split_out && a->ld_out < 3 * (int64_t)a->heads * 64) return GAVA_EINVAL;
  p.dbg = gava::debug_buffer();
  p.qbr = a->q_batch_rows > 0 ? a->q_batch_rows : a->n_q;
  p.ldq = a->ld_q > 0 ? a->ld_q : a->ld_qkv;
  if (p.ldq % 8 || p.qbr < a->n_q) return GAVA_EINVAL;
  if (p.n_keys > 320) return GAVA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->prec == GAVA_PREC_F16) return launch_attn<PrecF16>(p, s);
  if (a->prec == GAVA_PREC_BF16) return launch_attn<PrecBF16>(p, s);
  return GAVA_EINVAL;
}
