// forward.hip — host-side drivers that chain the kernels into the vision / text forward passes.
// Pure launch code: no allocation, no synchronisation, capturable into a hipGraph.
#include "common.h"
#include "internal.h"
#include <stdlib.h>
#include <mutex>

namespace {

// Per-device launch context.  The library allocates nothing per call; the only objects it ever creates are, once per
// device and on first use, a second stream for the prompt ("side") path of every block with its fork/join events (a
// capturing caller records a proper fork-join graph; GAVA_SIDE_STREAM=0 keeps everything on the caller's stream) and the
// event pairs of the fc1 probe.  They are keyed by the CURRENT device (the caller's stream must belong to it, as for any
// HIP launch), and a per-device mutex serialises the drivers' enqueue sequences: two host threads may call into the
// library concurrently, on the same or on different devices.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork[64], join[64];
  bool ok = false, tried = false;
  bool get() {
    if (tried) return ok;
    tried = true;
    const char* e = getenv("GAVA_SIDE_STREAM");
    if (e && e[0] == '0') return false;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return false;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) return false;
    for (int i = 0; i < 64; ++i)
      if (hipEventCreateWithFlags(&fork[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess) return false;
    ok = true;
    return true;
  }
};

// fc1 probe (gava_probe_fc1_enable / _read): event pairs around the roofline kernel inside the forward
struct Fc1Probe {
  hipEvent_t ev[64][2];
  bool made = false;
  int on = 0;     // 0 = off, else which kernel of a block is bracketed (GAVA_PROBE_* in gava_hip.h)
  int n = 0;
  unsigned long long valid = 0;   // block slots whose pair was recorded by the last forward
  bool ensure() {
    if (made) return true;
    for (int i = 0; i < 64; ++i)
      for (int k = 0; k < 2; ++k)
        if (hipEventCreate(&ev[i][k]) != hipSuccess) return false;
    made = true;
    return true;
  }
};

constexpr int MAX_DEVICES = 64;
struct DeviceCtx {
  std::mutex mu;
  SideStream side;
  Fc1Probe probe;
};
DeviceCtx g_dev[MAX_DEVICES];
DeviceCtx& device_ctx() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return g_dev[d % MAX_DEVICES];
}

// CUs the persistent QKV GEMM leaves to the prompt-path kernels on the side stream.  Measured (tools/side_probe.py): with
// 8 or 16 CUs left out a 128-workgroup kernel on the second stream still waits for the GEMM's first workgroups to retire
// (0.23 ms instead of 0.01 ms) - workgroups are dealt round-robin to the XCDs' shader engines and the dispatch blocks on
// the first engine without a free CU; with 32 (one CU per engine) it runs at its stand-alone speed and the GEMM is 4 % slower.
// When LN1 runs before the QKV GEMM most of the prompt path finishes beside LN1 and only its tail waits: 8 is then the
// better trade (23.85 vs 24.0 ms per forward); with LN1 folded away the whole path sits beside the GEMM: 32 (23.4 vs 23.75).
int side_cus(bool ln1_folded) {
  static const int forced = getenv("GAVA_SIDE_CUS") ? atoi(getenv("GAVA_SIDE_CUS")) : -1;
  return forced >= 0 ? forced : (ln1_folded ? 32 : 8);
}

struct Carver {
  char* base; size_t off, cap;
  Carver(void* b, size_t c) : base((char*)b), off(0), cap(c) {}
  void* take(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    return base ? (void*)(base + a) : nullptr;
  }
};

#define TRY(x)              \
  do {                      \
    const int _e = (x);     \
    if (_e != GAVA_OK) return _e; \
  } while (0)

int ln(const float* in, long in_stride, const int32_t* idx, const float* g, const float* b, void* o16, long o16s,
       float* o32, long o32s, int rows, int D, int prec, gava_stream_t s, int split = 0, const float* g2 = nullptr,
       const float* b2 = nullptr, void* ohi = nullptr, void* olo = nullptr, long ohls = 0) {
  gava_layernorm_args a{};
  a.gamma2 = g2; a.beta2 = b2;
  a.out_hi = ohi; a.out_lo = olo; a.out_hl_stride = ohls;
  a.in = in; a.in_stride = in_stride; a.in_row_index = idx; a.gamma = g; a.beta = b;
  a.out16 = o16; a.out16_stride = o16s; a.out32 = o32; a.out32_stride = o32s;
  a.rows = rows; a.D = D; a.prec = prec; a.split_out = split;
  return gava_layernorm(&a, s);
}

// LayerNorm folding (gava_gemm_args): producer outputs and/or consumer inputs of one GEMM call
struct Fold {
  void* x16 = nullptr; long ld_x16 = 0; float* rowsum = nullptr; int reduced = 0;         // producer
  const float* stats = nullptr; const float* s = nullptr; const float* t = nullptr;       // consumer
  const float* partials = nullptr;                                                        // consumer, no row_stats launch
  // producer with the residual stream as a 16-bit pair (gava_gemm_args.resid16): pair in (pitch = the call's ldr), lo out beside x16
  const void* r16 = nullptr; const void* rlo = nullptr; void* xlo = nullptr;
};

// the 8-bit side of a GEMM in the w_lo = 2 mode (gava_gemm_args): its own lo product (W8 != NULL) and / or the bf8 copies it
// leaves for the next one
struct Lo8 {
  const void* A8 = nullptr; const void* W8 = nullptr; int exp = 0;    // lda8 = 2 lda, ldw8 = 2 ldw (the kernel's contract)
  void* out8 = nullptr; long ldo8 = 0; void* x8 = nullptr; long ldx8 = 0;
};

// w_lo (gava_gemm_args.w_lo = 1): W is packed [W_hi | W_lo]; `ldw` is still given as the row pitch of a plain weight (K columns)
int gemm_x(int w_lo, const void* A, long lda, const void* W, long ldw, const float* bias, void* out, long ldo, int M, int N, int K,
           int epi, int prec, gava_stream_t s, const float* resid = nullptr, long ldr = 0, int scale_cols = 0,
           float scale = 1.f, int split_out = 0, void* aux_out = nullptr, const Fold* fold = nullptr, int cu_reserve = 0,
           const Lo8* l8 = nullptr) {
  gava_gemm_args a{};
  a.w_lo = w_lo ? 1 : 0;
  if (w_lo) ldw *= 2;
  if (l8) {
    if (l8->W8) {   // the 16-bit loop reads the W_hi half of the [W_hi | W_lo] rows, the 8-bit loop W8
      a.w_lo = 2; a.A8 = l8->A8; a.lda8 = 2 * lda; a.W8 = l8->W8; a.ldw8 = 2 * ldw; a.w8_exp = l8->exp;
    }
    a.out8 = l8->out8; a.ldo8 = l8->ldo8; a.x8_out = l8->x8; a.ld_x8 = l8->ldx8;
  }
  a.cu_reserve = cu_reserve;
  if (fold) {
    a.x16_out = fold->x16; a.ld_x16 = fold->ld_x16; a.rowsum_out = fold->rowsum; a.rowsum_reduced = fold->reduced;
    a.fold_stats = fold->stats; a.fold_s = fold->s; a.fold_t = fold->t; a.fold_partials = fold->partials;
    a.resid16 = fold->r16; a.resid_lo = fold->rlo; a.xlo_out = fold->xlo;
  }
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo;
  a.resid = resid; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.epilogue = epi; a.prec = prec;
  a.scale_cols = scale_cols; a.scale = scale; a.split_out = split_out; a.aux_out = aux_out;
  return gava_gemm(&a, s);
}
int gemm(const void* A, long lda, const void* W, long ldw, const float* bias, void* out, long ldo, int M, int N, int K,
         int epi, int prec, gava_stream_t s, const float* resid = nullptr, long ldr = 0, int scale_cols = 0,
         float scale = 1.f, int split_out = 0, void* aux_out = nullptr, const Fold* fold = nullptr, int cu_reserve = 0) {
  return gemm_x(0, A, lda, W, ldw, bias, out, ldo, M, N, K, epi, prec, s, resid, ldr, scale_cols, scale, split_out, aux_out, fold, cu_reserve);
}

struct VisionWs {
  float* X; void* Xn; void* QKV; void* MIX; void* HID;
  void* Xlo;                                     // the lo half of the residual stream's 16-bit pair (hi = Xn), see `pair` in the driver
  void* CLS16; float* CP; void* CPn; void* SQKV; void* SMIX; float* SUMM; void* SIDEn; void* SIDEKV;
  void* CLSPOST; float* PROJ;
  void* XNC; void* QC; void* MIXC; void* HIDC;   // last block: CLS rows only
  float* RSUM; float* STATS;                     // LayerNorm folding: row-sum partials [R][D/64][2], (mean, rstd) [R up to 256][2]
  void* Xn8; void* HID8;                         // w_lo = 2: bf8 copies of Xn / HID, rows 2D / 2F bytes apart
  size_t total;
};

int patch_k(const gava_vision_model* m) { return (3 * m->P * m->P + 63) / 64 * 64; }

// The A operand of the patch-embedding GEMM.
//   fp32 clips (the reference's input, the benchmark's), small batches: IM2COL-FREE - the GEMM builds its A tiles in the k-loop straight
//   from the NCTHW frames (two coalesced float4 loads per 8 k-values, converted and written to the LDS stage: gemm_kernel's
//   patch_load / patch_write); no patch matrix exists, one launch.  Big batches (enough 256 x 256 tiles for the persistent kernel,
//   round 4): two passes - the GEMM then runs the ping-pong loop of the persistent kernel on the 16-bit patch matrix, which the
//   in-loop loader (a VALU-built LDS tile per stage) cannot: c2 forward 20.06 -> 19.91 ms same box (profiles/r04_pingpong.txt).
//   decoded uint8 videos (forward_frames): two passes - gava_patchify writes the 16-bit patch matrix once (0.46 ms: every
//   element is a bilinear sample, 4 byte loads + the lerp), the GEMM stages it by LDS-DMA (0.26 ms).  Built inside the k-loop
//   the same samples are recomputed for each of the three 256-column tiles and cannot hide behind 0.23 GF of MFMA work per
//   frame: 2.33 ms.  GAVA_PATCH_DIRECT=0 / 1 forces the two-pass / the in-loop form for either source (A/B, op tests).
int patch_operand(const gava_vision_model* m, const float* x, void* scratch, int Kp, int prec, gava_stream_t stream,
                  gava_gemm_args* a) {
  static const int forced = getenv("GAVA_PATCH_DIRECT") ? atoi(getenv("GAVA_PATCH_DIRECT")) : -1;
  // (P % 8 != 0 - ViT-L/14 - has no float4 form of the in-loop loader: eight scalar loads per chunk; it stays two-pass)
  // the in-loop loader reads the frames with float4 loads: a clip tensor whose storage is not 16-byte aligned (a view at an odd
  // offset) or whose rows are not a multiple of 4 floats takes the two-pass form, which has no such requirement
  const bool aligned = m->clips != nullptr || ((((uintptr_t)x) & 15) == 0 && m->size % 4 == 0);
  const long gp_ = m->size / m->P, rows = (long)m->B * m->T_in * gp_ * gp_;
  const bool big = m->D % 256 == 0 && (rows + 255) / 256 * (m->D / 256) >= 512;      // gemm.hip's condition for the persistent kernel
  const bool direct = forced >= 0 ? (forced != 0 && aligned) : (m->clips == nullptr && m->P % 8 == 0 && aligned && !big);
  a->lda = Kp; a->frame_size = m->size; a->patch = m->P;
  if (direct) {
    a->A = nullptr; a->frames = m->clips ? nullptr : x; a->clips = m->clips; a->clip_lut = m->clip_lut;
    return GAVA_OK;
  }
  gava_patchify_args pa{};
  pa.x = m->clips ? nullptr : x; pa.clips = m->clips; pa.clip_lut = m->clip_lut;
  pa.B = m->B; pa.T = m->T_in; pa.size = m->size; pa.patch = m->P; pa.prec = prec;
  pa.out = scratch; pa.ldo = Kp;
  a->A = scratch;
  return gava_patchify(&pa, stream);
}

VisionWs carve_vision(const gava_vision_model* m, void* ws, size_t cap) {
  const long g = m->size / m->P, n = g * g, BT = (long)m->B * m->T_in, R = BT * (n + 1);
  const long D = m->D, F = m->F, E = m->E, SR = m->G + 2 * BT;
  Carver c(ws, cap);
  VisionWs w;
  w.X = (float*)c.take(R * D * 4);
  w.Xn = c.take(R * D * 2);
  w.Xlo = c.take(R * D * 2);
  w.QKV = c.take(R * 3 * D * 2);
  w.MIX = c.take(R * D * 2);
  {  // the fc1 output; before the blocks it parks the 16-bit patch matrix (patch_operand)
    const size_t hid = (size_t)R * F * 2, pm = (size_t)BT * n * patch_k(m) * 2;
    w.HID = c.take(hid > pm ? hid : pm);
  }
  w.CLS16 = c.take(BT * D * 2);
  w.CP = (float*)c.take(BT * D * 4);
  w.CPn = c.take(BT * D * 2);
  w.SQKV = c.take(BT * 3 * D * 2);
  w.SMIX = c.take(BT * D * 2);
  w.SUMM = (float*)c.take(BT * D * 4);
  w.SIDEn = c.take(SR * D * 2);
  w.SIDEKV = c.take(SR * 2 * D * 2);
  w.CLSPOST = c.take(BT * 3 * D * 2);
  w.PROJ = (float*)c.take(BT * E * 4);
  w.XNC = c.take(BT * 3 * D * 2);     // x3: the last block's CLS rows may run in split precision ([hi | lo | hi] rows)
  w.QC = c.take(BT * D * 2);
  w.MIXC = c.take(BT * 3 * D * 2);
  w.HIDC = c.take(BT * 3 * F * 2);
  {
    const long per_slot = R * (D / 64) * 8, reduced = (R + 255) / 256 * 256 * 4 * 8 + 1024;   // [R][D/64] or [R up to 256][4] float2
    w.RSUM = (float*)c.take(per_slot > reduced ? per_slot : reduced);
  }
  w.STATS = (float*)c.take((R + 255) / 256 * 256 * 8);
  w.Xn8 = m->w_lo == 2 ? c.take((size_t)R * 2 * D) : nullptr;
  w.HID8 = m->w_lo == 2 ? c.take((size_t)R * 2 * F) : nullptr;
  w.total = (c.off + 255) & ~(size_t)255;
  return w;
}

int check_vision(const gava_vision_model* m) {
  if (!m || !m->layer) return GAVA_EINVAL;
  if (m->B <= 0 || m->T_in <= 0 || m->T_model <= 0 || m->layers <= 0) return GAVA_EINVAL;
  if (m->size % m->P || m->D != m->H * 64 || m->D % 128 || m->F % 128 || m->E % 128) return GAVA_EINVAL;
  if (m->D > 1024) return GAVA_EINVAL;
  if (((long)m->B * m->T_in) % m->T_model) return GAVA_EINVAL;  // reference: view(B,T,C) fails (utils:160-163)
  const long g = m->size / m->P;
  if (g * g + 1 + m->G + m->T_model + 1 > 320) return GAVA_EINVAL;
  return GAVA_OK;
}

// does the inference driver keep the residual stream as a 16-bit pair for this model and batch? (see `pair` in the driver)
bool pair_stream(const gava_vision_model* m) {
  const long g = m->size / m->P, R = (long)m->B * m->T_in * (g * g + 1);
  const int D = m->D, F = m->F, WL = m->w_lo ? 2 : 1;
  bool pair = m->w_lo != 2 && m->layers >= 2 && R >= 8192 && D % 256 == 0 && D >= 256 &&
              !(getenv("GAVA_FUSED_STATS") && getenv("GAVA_FUSED_STATS")[0] == '0') && !getenv("GAVA_NO_PREFUSE") && !getenv("GAVA_NO_LASTFOLD") &&
              gava::gemm_takes_pair((int)R, D, D, D, (long)WL * D) && gava::gemm_takes_pair((int)R, D, F, F, (long)WL * F);
  for (int i = 0; pair && i + 1 < m->layers; ++i) pair = m->layer[i].w_fc1_fold && m->layer[i + 1].w_qkv_fold;
  return pair;
}

}  // namespace

extern "C" int gava_vision_pair_stream(const gava_vision_model* m) {
  if (check_vision(m) != GAVA_OK) return 0;
  return pair_stream(m) ? 1 : 0;
}

// The ABI version is a hash of include/gava_hip.h, baked in by gava_clip_amd/build.py (-DGAVA_ABI_HASH): a library built from
// another header than the one its caller mirrors is refused at load time (gava_clip_amd/hip.py load()).
#ifndef GAVA_ABI_HASH
#error "build with -DGAVA_ABI_HASH=<first 31 bits of sha256(include/gava_hip.h)> (python -m gava_clip_amd.build does)"
#endif
extern "C" int gava_abi_version(void) { return (int)(GAVA_ABI_HASH); }

extern "C" int gava_struct_sizes(size_t* out, int cap) {
  const size_t v[] = {sizeof(gava_gemm_args), sizeof(gava_layernorm_args), sizeof(gava_attention_args), sizeof(gava_attention_f32_args),
                      sizeof(gava_clip_desc), sizeof(gava_vision_layer), sizeof(gava_vision_layer8), sizeof(gava_vision_model),
                      sizeof(gava_text_layer), sizeof(gava_text_model), sizeof(gava_layernorm_bwd_args), sizeof(gava_attention_bwd_args),
                      sizeof(gava_vision_saved), sizeof(gava_preprocess_args), sizeof(gava_patchify_args)};
  const int n = (int)(sizeof(v) / sizeof(v[0]));
  for (int i = 0; out && i < n && i < cap; ++i) out[i] = v[i];
  return n;
}

extern "C" int gava_probe_fc1_enable(int on) {
  Fc1Probe& g_probe = device_ctx().probe;
  if (on < 0 || on > GAVA_PROBE_ATTN) return GAVA_EINVAL;
  g_probe.on = on; if (!on) g_probe.n = 0; return GAVA_OK;
}
extern "C" int gava_probe_fc1_read(float* ms, int cap) {
  Fc1Probe& g_probe = device_ctx().probe;
  const int n = g_probe.n;
  if (n <= 0 || !ms) return 0;
  if (hipEventSynchronize(g_probe.ev[n - 1][1]) != hipSuccess) return 0;
  for (int i = 0; i < n && i < cap; ++i) {
    ms[i] = -1.0f;    // a block the probed kernel did not run in
    if (((g_probe.valid >> i) & 1) && hipEventElapsedTime(&ms[i], g_probe.ev[i][0], g_probe.ev[i][1]) != hipSuccess) return 0;
  }
  return n;
}

extern "C" size_t gava_vision_workspace_bytes(const gava_vision_model* m) {
  if (check_vision(m) != GAVA_OK) return 0;
  return carve_vision(m, nullptr, 0).total;
}

extern "C" int gava_vision_forward(const gava_vision_model* m, const float* x, float* cls_x, float* summary,
                                   float* debug_cls, void* workspace, size_t workspace_bytes, gava_stream_t stream) {
  return gava_vision_forward_train(m, x, cls_x, summary, debug_cls, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int gava_vision_forward_train(const gava_vision_model* m, const float* x, float* cls_x, float* summary,
                                         float* debug_cls, float* saved_x, void* workspace, size_t workspace_bytes,
                                         gava_stream_t stream) {
  TRY(check_vision(m));
  if ((!x && !m->clips) || !cls_x || !summary || !workspace) return GAVA_EINVAL;
  const VisionWs w = carve_vision(m, workspace, workspace_bytes);
  if (w.total > workspace_bytes) return GAVA_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  DeviceCtx& dc = device_ctx();
  std::lock_guard<std::mutex> lock(dc.mu);
  SideStream& g_side = dc.side;
  Fc1Probe& g_probe = dc.probe;
  const int g = m->size / m->P, n = g * g, BT = m->B * m->T_in, R = BT * (n + 1);
  const int D = m->D, F = m->F, E = m->E, G = m->G, Tm = m->T_model, pr = m->prec;
  const int Kp = patch_k(m), SR = G + 2 * BT;
  const long fs = (long)(n + 1) * D;  // frame stride in X
  // weight-lo pass (gava_vision_model.w_lo): every weight below except the *_split ones is [W_hi | W_lo], 2x the columns
  const int wl = m->w_lo ? 1 : 0, WL = wl ? 2 : 1;
  if (wl && saved_x) return GAVA_EINVAL;
  // w_lo = 2: the LayerNorm-folded qkv / fc1 GEMMs and fc2 run their lo product at 8 bits (shapes the persistent kernel takes)
  const bool l8 = m->w_lo == 2 && m->layer8 && pr == GAVA_PREC_F16 && D % 256 == 0 && F % 256 == 0;
  const char* h16 = nullptr; (void)h16;

  // ---- embedding (VitaCLIP_vision_encoder.py:105-113)
  {
    // im2col-free for fp32 clips, patch matrix + LDS-DMA GEMM for decoded uint8 videos: see patch_operand
    gava_gemm_args a{};
    TRY(patch_operand(m, x, w.HID, Kp, pr, stream, &a));
    a.W = m->w_patch; a.ldw = WL * Kp; a.bias = m->b_patch; a.w_lo = wl;
    a.out = w.X; a.ldo = D; a.M = BT * n; a.N = D; a.K = Kp; a.epilogue = GAVA_EPI_F32_PATCH; a.prec = pr;
    a.pos = m->pos_embed; a.time = m->time_embed; a.n_patches = n; a.T = m->T_in;
    TRY(gava_gemm(&a, stream));
  }
  TRY(gava::cls_embed(w.X, m->cls_token, m->pos_embed, m->time_embed, BT, m->T_in, D, fs, s));
  // training: slot 0 = the embedding before ln_pre, slot 1+i = the input of block i, slot layers+1 = the final stream
  auto keep = [&](int slot) -> int {
    if (!saved_x) return GAVA_OK;
    return hipMemcpyAsync(saved_x + (size_t)slot * R * D, w.X, (size_t)R * D * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess
               ? GAVA_OK : GAVA_ELAUNCH;
  };
  TRY(keep(0));
  // ln_pre, and in the same row pass norm1 of block 0 (both read the same rows: one read of the embedding instead of two)
  static const bool no_prefuse = getenv("GAVA_NO_PREFUSE") != nullptr;     // A/B switches (tools/ab_env.py)
  static const bool no_lastfold = getenv("GAVA_NO_LASTFOLD") != nullptr;
  // timing diagnostics (results are WRONG): how much of a kernel's duration the forward gives back when it is removed.
  // Compiled in only by experiment builds (tools/ab_build.sh NAME -DGAVA_ENABLE_ABLATE), never into the product library.
#ifdef GAVA_ENABLE_ABLATE
  static const bool skip_attn = getenv("GAVA_DIAG_SKIP_ATTN") != nullptr, skip_stats = getenv("GAVA_DIAG_SKIP_STATS") != nullptr;
#else
  constexpr bool skip_attn = false, skip_stats = false;
#endif
  const bool pre_fused = m->layers >= 1 && !no_prefuse;
  // The residual stream as a 16-bit pair (inference, big batches, every block's LayerNorms folded): between ln_pre and the last
  // block the stream exists only as hi = h16(x) - which IS the operand of the GEMM consuming the fold (Xn) - and lo = fp16(x - hi)
  // (Xlo); out_proj / fc2 read and write the pair (8 bytes per element through their epilogue instead of 10).  Block 0's operand
  // is the NORMALISED stream (norm1 is fused into ln_pre's pass), so its hi half parks in HID until fc1 overwrites that; the last
  // block works on fp32 CLS rows again (join_rows).  |x - hi - lo| <= 2^-22 |x|.  GAVA_PAIR_STREAM=0: the fp32 stream (A/B).
  const bool pair = !saved_x && pre_fused && !no_lastfold && pair_stream(m);
  if (pre_fused) TRY(ln(w.X, D, nullptr, m->lnpre_g, m->lnpre_b, w.Xn, D, pair ? nullptr : w.X, D, R, D, pr, stream, 0, m->layer[0].ln1_g, m->layer[0].ln1_b,
                        pair ? w.HID : nullptr, pair ? w.Xlo : nullptr, D));
  else TRY(ln(w.X, D, nullptr, m->lnpre_g, m->lnpre_b, nullptr, 0, w.X, D, R, D, pr, stream));

  // ---- blocks (VitaCLIP_vision_encoder.py:115-121, VitaCLIP_vision_encoder_utils.py:155-203)
  bool folded_in = false;   // Xn / STATS already hold this block's un-normalised input and its row statistics
  const int probe = (g_probe.on && m->layers <= 64 && g_probe.ensure()) ? g_probe.on : 0;
  if (probe) { g_probe.n = 0; g_probe.valid = 0; }
  // event pair k (0 = before, 1 = after) of block i around the kernel the probe names
  auto mark = [&](int which, int i, int k) -> int {
    if (probe != which) return GAVA_OK;
    if (hipEventRecord(g_probe.ev[i][k], s) != hipSuccess) return GAVA_ELAUNCH;
    if (k == 1) { g_probe.n = i + 1; g_probe.valid |= 1ull << i; }
    return GAVA_OK;
  };
  auto not_last_blk = [&](int i) { return i + 1 < m->layers; };
  for (int i = 0; i < m->layers; ++i) {
    const gava_vision_layer& L = m->layer[i];
    TRY(keep(1 + i));
    const unsigned short* wqkv = (const unsigned short*)L.w_qkv;
    // prompt ("side") path: cls_proj, summary token, local prompts -> K/V-only rows.  The main path does not
    // need it before attention, so it runs on the side stream next to LN1 + the QKV GEMM.
    const bool two = g_side.get() && m->layers <= 64;
    gava_stream_t ss = two ? (gava_stream_t)g_side.s : stream;
    if (two) {
      if (hipEventRecord(g_side.fork[i], s) != hipSuccess || hipStreamWaitEvent(g_side.s, g_side.fork[i], 0) != hipSuccess)
        return GAVA_ELAUNCH;
    }
    // the CLS rows as cls_proj's operand: h16 of the stream - with the pair, its hi half where it lies
    const void* hi_in = i == 0 ? w.HID : w.Xn;
    if (pair) {
      TRY(gemm_x(wl, hi_in, fs, L.w_cls, D, L.b_cls, w.CP, D, BT, D, D, GAVA_EPI_F32, pr, ss));
    } else {
      TRY(ln(w.X, fs, nullptr, nullptr, nullptr, w.CLS16, D, nullptr, 0, BT, D, pr, ss));
      TRY(gemm_x(wl, w.CLS16, D, L.w_cls, D, L.b_cls, w.CP, D, BT, D, D, GAVA_EPI_F32, pr, ss));
    }
    TRY(ln(w.CP, D, nullptr, L.sln_g, L.sln_b, w.CPn, D, nullptr, 0, BT, D, pr, ss));
    TRY(gemm_x(wl, w.CPn, D, L.w_sqkv, D, L.b_sqkv, w.SQKV, 3 * D, BT, 3 * D, D, GAVA_EPI_H16, pr, ss, nullptr, 0, D, 0.125f));
    {
      gava_attention_args a{};
      const unsigned short* q = (const unsigned short*)w.SQKV;
      a.q = q; a.k = q + D; a.v = q + 2 * D; a.ld_qkv = 3 * D; a.out = w.SMIX; a.ld_out = D;
      a.batch = BT / Tm; a.heads = m->H; a.n_q = Tm; a.n_kmain = Tm; a.prec = pr;
      TRY(gava_attention(&a, ss));
    }
    TRY(gemm_x(wl, w.SMIX, D, L.w_sout, D, L.b_sout, w.SUMM, D, BT, D, D, GAVA_EPI_F32, pr, ss, w.CP, D));
    TRY(gava::side_ln(L.global_prompts, L.local_prompts, w.CP, w.SUMM, L.ln1_g, L.ln1_b, w.SIDEn, G, Tm, BT, D, pr, (hipStream_t)ss));
    TRY(gemm_x(wl, w.SIDEn, D, wqkv + (long)D * WL * D, D, L.b_qkv + D, w.SIDEKV, 2 * D, SR, 2 * D, D, GAVA_EPI_H16, pr, ss));
    if (two && hipEventRecord(g_side.join[i], g_side.s) != hipSuccess) return GAVA_ELAUNCH;
    const int resv = two ? side_cus(folded_in) : 0;   // the persistent QKV GEMM leaves side_cus() CUs to the side kernels
    // main path.  LayerNorm folding (inference only, when the model carries the folded weights): norm2 of every block
    // but the last and norm1 of blocks 1..layers-2 are not launched; the producing GEMM (out_proj / fc2 of the block
    // before) leaves a 16-bit copy of x in Xn plus row-sum partials, gava_row_stats makes (mean, rstd) of them and the
    // consuming GEMM (fc1 / qkv) applies the normalisation in its epilogue.
    const bool not_last = i + 1 < m->layers;
    const bool fold2 = !saved_x && not_last && L.w_fc1_fold && D % 64 == 0;
    const bool fold1 = folded_in;                                                    // set by the previous block's fc2
    // (the last block consumes the fold too: its K/V GEMM over all rows; its CLS queries get a LayerNorm of their own)
    const bool fold1_next = !saved_x && i + (no_lastfold ? 2 : 1) < m->layers && m->layer[i + 1].w_qkv_fold && D % 64 == 0;
    // big batches: the producers pre-reduce their row sums per 256-column tile and the consumers turn them into
    // (mean, rstd) themselves - no gava_row_stats launch between producer and consumer (each cost ~33 us of forward time
    // for a 9 us kernel: two kernel boundaries behind a persistent GEMM).  GAVA_FUSED_STATS=0: the launch stays (A/B).
    static const bool fused_env = !(getenv("GAVA_FUSED_STATS") && getenv("GAVA_FUSED_STATS")[0] == '0');
    const bool fused = fused_env && R >= 8192 && D % 256 == 0 && D <= 1024 && D >= 256;
    Fold produce; produce.x16 = w.Xn; produce.ld_x16 = D; produce.rowsum = w.RSUM; produce.reduced = fused ? 1 : 0;
    if (pair) { produce.rlo = w.Xlo; produce.xlo = w.Xlo; }
    auto consume = [&](const float* s_, const float* t_) {
      Fold c; c.s = s_; c.t = t_;
      if (fused) c.partials = w.RSUM; else c.stats = w.STATS;
      return c;
    };
    if (!fold1 && !(i == 0 && pre_fused)) TRY(ln(w.X, D, nullptr, L.ln1_g, L.ln1_b, w.Xn, D, nullptr, 0, R, D, pr, stream));
    const unsigned short* sk = (const unsigned short*)w.SIDEKV;
    const gava_vision_layer8* L8 = l8 ? &m->layer8[i] : nullptr;
    if (not_last) {
      if (fold1) TRY(mark(GAVA_PROBE_QKV, i, 0));     // the folded form only: the instantiation bench.py's table names
      if (fold1) {
        Fold c = consume(L8 ? L8->qkv_fold_s8 : L.qkv_fold_s, L.qkv_fold_t);
        Lo8 lo; if (L8) { lo.A8 = w.Xn8; lo.W8 = L8->w_qkv_fold8; lo.exp = L8->qkv_fold_exp; }
        TRY(gemm_x(wl, w.Xn, D, L.w_qkv_fold, D, nullptr, w.QKV, 3 * D, R, 3 * D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f, 0, nullptr, &c, resv, L8 ? &lo : nullptr));
      } else
      TRY(gemm_x(wl, w.Xn, D, L.w_qkv, D, L.b_qkv, w.QKV, 3 * D, R, 3 * D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f, 0, nullptr, nullptr, resv));
      if (fold1) TRY(mark(GAVA_PROBE_QKV, i, 1));
      if (two && hipStreamWaitEvent(s, g_side.join[i], 0) != hipSuccess) return GAVA_ELAUNCH;
      {
        gava_attention_args a{};
        const unsigned short* q = (const unsigned short*)w.QKV;
        a.q = q; a.k = q + D; a.v = q + 2 * D; a.ld_qkv = 3 * D;
        a.side_k = sk; a.side_v = sk + D; a.ld_side = 2 * D;
        a.out = w.MIX; a.ld_out = D;
        a.batch = BT; a.heads = m->H; a.n_q = n + 1; a.n_kmain = n + 1;
        a.n_g = G; a.T = Tm; a.has_summary = 1; a.prec = pr;
        TRY(mark(GAVA_PROBE_ATTN, i, 0));
        if (!skip_attn) TRY(gava_attention(&a, stream));
        TRY(mark(GAVA_PROBE_ATTN, i, 1));
      }
      TRY(mark(GAVA_PROBE_OUT, i, 0));
      if (fold2) {
        Lo8 po; if (L8) { po.x8 = w.Xn8; po.ldx8 = 2 * D; }      // out_proj: 16-bit lo product, bf8 copy of x16 for fc1
        if (pair) {
          produce.r16 = hi_in;
          TRY(gemm_x(wl, w.MIX, D, L.w_out, D, L.b_out, nullptr, 0, R, D, D, GAVA_EPI_F32, pr, stream, nullptr, D, 0, 1.f, 0, nullptr, &produce));
        } else
        TRY(gemm_x(wl, w.MIX, D, L.w_out, D, L.b_out, w.X, D, R, D, D, GAVA_EPI_F32, pr, stream, w.X, D, 0, 1.f, 0, nullptr, &produce, 0, L8 ? &po : nullptr));
        TRY(mark(GAVA_PROBE_OUT, i, 1));
        if (!skip_stats && !fused) TRY(gava_row_stats(w.RSUM, D / 64, D, R, w.STATS, stream));
        Fold c = consume(L8 ? L8->fc1_fold_s8 : L.fc1_fold_s, L.fc1_fold_t);
        Lo8 lo; if (L8) { lo.A8 = w.Xn8; lo.W8 = L8->w_fc1_fold8; lo.exp = L8->fc1_fold_exp; lo.out8 = w.HID8; lo.ldo8 = 2 * F; }
        TRY(mark(GAVA_PROBE_FC1, i, 0));
        TRY(gemm_x(wl, w.Xn, D, L.w_fc1_fold, D, nullptr, w.HID, F, R, F, D, GAVA_EPI_H16_QGELU, pr, stream, nullptr, 0, 0, 1.f, 0, nullptr, &c, 0, L8 ? &lo : nullptr));
        TRY(mark(GAVA_PROBE_FC1, i, 1));
      } else {
        TRY(gemm_x(wl, w.MIX, D, L.w_out, D, L.b_out, w.X, D, R, D, D, GAVA_EPI_F32, pr, stream, w.X, D));
        TRY(mark(GAVA_PROBE_OUT, i, 1));
        TRY(ln(w.X, D, nullptr, L.ln2_g, L.ln2_b, w.Xn, D, nullptr, 0, R, D, pr, stream));
        TRY(mark(GAVA_PROBE_FC1, i, 0));
        TRY(gemm_x(wl, w.Xn, D, L.w_fc1, D, L.b_fc1, w.HID, F, R, F, D, GAVA_EPI_H16_QGELU, pr, stream));
        TRY(mark(GAVA_PROBE_FC1, i, 1));
      }
      TRY(mark(GAVA_PROBE_FC2, i, 0));
      // fc2: 8-bit lo product when fc1 left the bf8 copy of its output (HID8); its own x16 gets a bf8 copy for the next qkv
      Lo8 f2; const bool f2_8 = L8 && fold2;
      if (f2_8) { f2.A8 = w.HID8; f2.W8 = L8->w_fc28; f2.exp = L8->fc2_exp; }
      if (fold1_next) {
        if (L8) { f2.x8 = w.Xn8; f2.ldx8 = 2 * D; }
        if (pair) {
          produce.r16 = w.Xn;
          TRY(gemm_x(wl, w.HID, F, L.w_fc2, F, L.b_fc2, nullptr, 0, R, D, F, GAVA_EPI_F32, pr, stream, nullptr, D, 0, 1.f, 0, nullptr, &produce));
        } else
        TRY(gemm_x(wl, w.HID, F, L.w_fc2, F, L.b_fc2, w.X, D, R, D, F, GAVA_EPI_F32, pr, stream, w.X, D, 0, 1.f, 0, nullptr, &produce, 0, L8 ? &f2 : nullptr));
        TRY(mark(GAVA_PROBE_FC2, i, 1));
        if (!skip_stats && !fused) TRY(gava_row_stats(w.RSUM, D / 64, D, R, w.STATS, stream));
      } else {
        TRY(gemm_x(wl, w.HID, F, L.w_fc2, F, L.b_fc2, w.X, D, R, D, F, GAVA_EPI_F32, pr, stream, w.X, D, 0, 1.f, 0, nullptr, nullptr, 0, f2_8 ? &f2 : nullptr));
        TRY(mark(GAVA_PROBE_FC2, i, 1));
      }
      folded_in = fold1_next;
    } else {
      // Last block: only the CLS row of each frame reaches the outputs (VitaCLIP_vision_encoder.py:126
      // takes x[:,0]; the summary token comes from the prompt path above).  Keys/values are still
      // needed for every token, queries / out_proj / MLP only for the B*T CLS rows: same results,
      // 1/197 of the row work.
      if (pair) TRY(gava::join_rows(w.Xn, w.Xlo, fs, w.X, fs, BT, D, pr, s));      // those rows as fp32 again
      if (fold1) {   // norm1 folded into the K/V GEMM: Xn holds the 16-bit copy of the un-normalised stream
        Fold c = consume((L8 ? L8->qkv_fold_s8 : L.qkv_fold_s) + D, L.qkv_fold_t + D);
        Lo8 lo; if (L8) { lo.A8 = w.Xn8; lo.W8 = (const char*)L8->w_qkv_fold8 + (size_t)D * 4 * D; lo.exp = L8->qkv_fold_exp; }
        TRY(gemm_x(wl, w.Xn, D, (const unsigned short*)L.w_qkv_fold + (long)D * WL * D, D, nullptr, (unsigned short*)w.QKV + D, 3 * D, R, 2 * D, D,
                 GAVA_EPI_H16, pr, stream, nullptr, 0, 0, 1.f, 0, nullptr, &c, resv, L8 ? &lo : nullptr));
      } else {
        TRY(gemm_x(wl, w.Xn, D, wqkv + (long)D * WL * D, D, L.b_qkv + D, (unsigned short*)w.QKV + D, 3 * D, R, 2 * D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, 0, 1.f, 0, nullptr, nullptr, resv));
      }
      // split precision for these B*T rows when the model carries the split-packed weights (gava_vision_layer)
      const int sp = (!saved_x && L.w_q_split && L.w_out_split && L.w_fc1_split && L.w_fc2_split) ? 1 : 0;
      const int S = sp ? 3 : 1;
      if (sp) {
        TRY(ln(w.X, fs, nullptr, L.ln1_g, L.ln1_b, w.XNC, 3 * D, nullptr, 0, BT, D, pr, stream, 1));
        TRY(gemm(w.XNC, 3 * D, L.w_q_split, 3 * D, L.b_qkv, w.QC, D, BT, D, 3 * D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f));
      } else if (fold1) {
        TRY(ln(w.X, fs, nullptr, L.ln1_g, L.ln1_b, w.XNC, D, nullptr, 0, BT, D, pr, stream));     // the B*T CLS rows only
        TRY(gemm_x(wl, w.XNC, D, L.w_qkv, D, L.b_qkv, w.QC, D, BT, D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f));
      } else {
        TRY(gemm_x(wl, w.Xn, fs, L.w_qkv, D, L.b_qkv, w.QC, D, BT, D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f));
      }
      if (two && hipStreamWaitEvent(s, g_side.join[i], 0) != hipSuccess) return GAVA_ELAUNCH;
      {
        gava_attention_args a{};
        const unsigned short* q = (const unsigned short*)w.QKV;
        a.q = w.QC; a.ld_q = D; a.q_batch_rows = 1;
        a.k = q + D; a.v = q + 2 * D; a.ld_qkv = 3 * D;
        a.side_k = sk; a.side_v = sk + D; a.ld_side = 2 * D;
        a.out = w.MIXC; a.ld_out = S * D; a.split_out = sp;
        a.batch = BT; a.heads = m->H; a.n_q = 1; a.n_kmain = n + 1;
        a.n_g = G; a.T = Tm; a.has_summary = 1; a.prec = pr;
        TRY(gava_attention(&a, stream));
      }
      TRY(gemm_x(sp ? 0 : wl, w.MIXC, S * D, sp ? L.w_out_split : L.w_out, S * D, L.b_out, w.X, fs, BT, D, S * D, GAVA_EPI_F32, pr, stream, w.X, fs));
      TRY(ln(w.X, fs, nullptr, L.ln2_g, L.ln2_b, w.XNC, S * D, nullptr, 0, BT, D, pr, stream, sp));
      TRY(gemm_x(sp ? 0 : wl, w.XNC, S * D, sp ? L.w_fc1_split : L.w_fc1, S * D, L.b_fc1, w.HIDC, S * F, BT, F, S * D, GAVA_EPI_H16_QGELU, pr, stream, nullptr, 0, 0, 1.f, sp));
      TRY(gemm_x(sp ? 0 : wl, w.HIDC, S * F, sp ? L.w_fc2_split : L.w_fc2, S * F, L.b_fc2, w.X, fs, BT, D, S * F, GAVA_EPI_F32, pr, stream, w.X, fs));
    }
    if (debug_cls) {
      if (pair && not_last_blk(i)) TRY(gava::join_rows(w.Xn, w.Xlo, fs, debug_cls + (long)i * BT * D, D, BT, D, pr, s));
      else TRY(gava::copy_rows(w.X, fs, debug_cls + (long)i * BT * D, BT, D, s));
    }
  }

  TRY(keep(1 + m->layers));
  // ---- head (VitaCLIP_vision_encoder.py:126-130)
  // split precision (3 MFMA passes): M = BT rows only, and its rounding lands directly on the output
  TRY(ln(w.X, fs, nullptr, m->lnpost_g, m->lnpost_b, w.CLSPOST, 3 * D, nullptr, 0, BT, D, pr, stream, 1));
  TRY(gemm(w.CLSPOST, 3 * D, m->w_proj, 3 * D, nullptr, w.PROJ, E, BT, E, 3 * D, GAVA_EPI_F32, pr, stream));
  TRY(gava::mean_rows(w.PROJ, cls_x, m->B, m->T_in, E, s));
  TRY(gava::mean_rows(w.SUMM, summary, BT / Tm, Tm, D, s));
  return GAVA_OK;
}

// Training forward that KEEPS the activations the backward needs instead of leaving them to be recomputed
// (288 GB of HBM: ~21 GB at c2).  Nothing is copied: the residual stream hops from buffer to buffer
// (x[i] -> x1[i] -> x[i+1], the GEMMs' residual input and output being different pointers), the QKV GEMM and the prompt
// path write straight into the per-block slots, and fc1 stores its pre-activation beside the QuickGELU output.
// With last_q / last_x1 / last_pre the last block runs on the CLS rows only, like the inference driver.
extern "C" int gava_vision_forward_keep(const gava_vision_model* m, const float* x, float* cls_x, float* summary,
                                        const gava_vision_saved* sv, void* workspace, size_t workspace_bytes,
                                        gava_stream_t stream) {
  TRY(check_vision(m));
  if (m->w_lo) return GAVA_EINVAL;     // the weight-lo pass is an inference mode (gava_hip.h)
  if ((!x && !m->clips) || !cls_x || !summary || !workspace || !sv || !sv->e0 || !sv->x || !sv->x1 || !sv->qkv || !sv->pre || !sv->sidekv)
    return GAVA_EINVAL;
  const VisionWs w = carve_vision(m, workspace, workspace_bytes);
  if (w.total > workspace_bytes) return GAVA_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  DeviceCtx& dc = device_ctx();
  std::lock_guard<std::mutex> lock(dc.mu);
  SideStream& g_side = dc.side;
  const int g = m->size / m->P, n = g * g, BT = m->B * m->T_in, R = BT * (n + 1);
  const int D = m->D, F = m->F, E = m->E, G = m->G, Tm = m->T_model, pr = m->prec;
  const int Kp = patch_k(m), SR = G + 2 * BT;
  const long fs = (long)(n + 1) * D;
  const size_t RD = (size_t)R * D;
  {
    gava_gemm_args a{};
    TRY(patch_operand(m, x, w.HID, Kp, pr, stream, &a));
    a.W = m->w_patch; a.ldw = Kp; a.bias = m->b_patch;
    a.out = sv->e0; a.ldo = D; a.M = BT * n; a.N = D; a.K = Kp; a.epilogue = GAVA_EPI_F32_PATCH; a.prec = pr;
    a.pos = m->pos_embed; a.time = m->time_embed; a.n_patches = n; a.T = m->T_in;
    TRY(gava_gemm(&a, stream));
  }
  TRY(gava::cls_embed(sv->e0, m->cls_token, m->pos_embed, m->time_embed, BT, m->T_in, D, fs, s));
  // LayerNorm folding as in the inference driver (norm1 / norm2 of the vision blocks are frozen, VitaCLIP_model.py:230-239, so
  // the folded weight copies stay valid while training; the backward recomputes the statistics from the kept fp32 stream and
  // never needs the normalised activations).  GAVA_TRAIN_FOLD=0: separate LayerNorm launches (A/B).
  static const bool fold_env = !(getenv("GAVA_TRAIN_FOLD") && getenv("GAVA_TRAIN_FOLD")[0] == '0');
  static const bool fused_env = !(getenv("GAVA_FUSED_STATS") && getenv("GAVA_FUSED_STATS")[0] == '0');
  const bool can_fold = fold_env && D % 64 == 0 && m->layers >= 1 && m->layer[0].w_fc1_fold && m->layer[0].w_qkv_fold;
  const bool fused = fused_env && R >= 8192 && D % 256 == 0 && D <= 1024 && D >= 256;
  Fold produce; produce.x16 = w.Xn; produce.ld_x16 = D; produce.rowsum = w.RSUM; produce.reduced = fused ? 1 : 0;
  auto consume = [&](const float* s_, const float* t_) {
    Fold c; c.s = s_; c.t = t_;
    if (fused) c.partials = w.RSUM; else c.stats = w.STATS;
    return c;
  };
  // ln_pre, and with folding norm1 of block 0 in the same row pass
  if (can_fold) TRY(ln(sv->e0, D, nullptr, m->lnpre_g, m->lnpre_b, w.Xn, D, sv->x, D, R, D, pr, stream, 0, m->layer[0].ln1_g, m->layer[0].ln1_b));
  else TRY(ln(sv->e0, D, nullptr, m->lnpre_g, m->lnpre_b, nullptr, 0, sv->x, D, R, D, pr, stream));
  bool folded_in = false;   // Xn / RSUM already hold this block's un-normalised input and its row-sum partials
  for (int i = 0; i < m->layers; ++i) {
    const gava_vision_layer& L = m->layer[i];
    const unsigned short* wqkv = (const unsigned short*)L.w_qkv;
    float* Xin = sv->x + (size_t)i * RD;
    float* X1 = sv->x1 + (size_t)i * RD;
    float* Xout = sv->x + (size_t)(i + 1) * RD;
    unsigned short* QKV = (unsigned short*)sv->qkv + (size_t)i * R * 3 * D;
    unsigned short* PRE = (unsigned short*)sv->pre + (size_t)i * R * F;
    unsigned short* SKV = (unsigned short*)sv->sidekv + (size_t)i * SR * 2 * D;
    const bool two = g_side.get() && m->layers <= 64;
    gava_stream_t ss = two ? (gava_stream_t)g_side.s : stream;
    if (two) {
      if (hipEventRecord(g_side.fork[i], s) != hipSuccess || hipStreamWaitEvent(g_side.s, g_side.fork[i], 0) != hipSuccess)
        return GAVA_ELAUNCH;
    }
    TRY(ln(Xin, fs, nullptr, nullptr, nullptr, w.CLS16, D, nullptr, 0, BT, D, pr, ss));
    TRY(gemm(w.CLS16, D, L.w_cls, D, L.b_cls, w.CP, D, BT, D, D, GAVA_EPI_F32, pr, ss));
    TRY(ln(w.CP, D, nullptr, L.sln_g, L.sln_b, w.CPn, D, nullptr, 0, BT, D, pr, ss));
    TRY(gemm(w.CPn, D, L.w_sqkv, D, L.b_sqkv, w.SQKV, 3 * D, BT, 3 * D, D, GAVA_EPI_H16, pr, ss, nullptr, 0, D, 0.125f));
    {
      gava_attention_args a{};
      const unsigned short* q = (const unsigned short*)w.SQKV;
      a.q = q; a.k = q + D; a.v = q + 2 * D; a.ld_qkv = 3 * D; a.out = w.SMIX; a.ld_out = D;
      a.batch = BT / Tm; a.heads = m->H; a.n_q = Tm; a.n_kmain = Tm; a.prec = pr;
      TRY(gava_attention(&a, ss));
    }
    TRY(gemm(w.SMIX, D, L.w_sout, D, L.b_sout, w.SUMM, D, BT, D, D, GAVA_EPI_F32, pr, ss, w.CP, D));
    TRY(gava::side_ln(L.global_prompts, L.local_prompts, w.CP, w.SUMM, L.ln1_g, L.ln1_b, w.SIDEn, G, Tm, BT, D, pr, (hipStream_t)ss));
    TRY(gemm(w.SIDEn, D, wqkv + (long)D * D, D, L.b_qkv + D, SKV, 2 * D, SR, 2 * D, D, GAVA_EPI_H16, pr, ss));
    if (two && hipEventRecord(g_side.join[i], g_side.s) != hipSuccess) return GAVA_ELAUNCH;
    const bool last_cls = i + 1 == m->layers && sv->last_q && sv->last_x1 && sv->last_pre;
    const bool fold1 = folded_in;                                               // set by the previous block's fc2
    const bool fold2 = can_fold && !last_cls && L.w_fc1_fold;                   // out_proj produces, fc1 consumes
    const bool fold1_next = can_fold && i + 1 < m->layers && m->layer[i + 1].w_qkv_fold;   // fc2 produces for the next qkv / K,V GEMM
    const int resv = two ? side_cus(fold1) : 0;
    if (!fold1 && !(i == 0 && can_fold)) TRY(ln(Xin, D, nullptr, L.ln1_g, L.ln1_b, w.Xn, D, nullptr, 0, R, D, pr, stream));
    if (last_cls) {
      // Last block, as in the inference driver: keys/values for every row, queries / out_proj / MLP for the B*T CLS
      // rows only (VitaCLIP_vision_encoder.py:126 reads x[:,0]).  Kept: K/V in the QKV slot, the CLS queries, the CLS
      // rows of the stream after attention, the CLS pre-activations.
      unsigned short* QC = (unsigned short*)sv->last_q;
      unsigned short* PREC = (unsigned short*)sv->last_pre;
      if (fold1) {   // norm1 folded into the K/V GEMM; the B*T CLS queries get a LayerNorm of their own
        Fold c = consume(L.qkv_fold_s + D, L.qkv_fold_t + D);
        TRY(gemm(w.Xn, D, (const unsigned short*)L.w_qkv_fold + (long)D * D, D, nullptr, QKV + D, 3 * D, R, 2 * D, D, GAVA_EPI_H16, pr, stream,
                 nullptr, 0, 0, 1.f, 0, nullptr, &c, resv));
        TRY(ln(Xin, fs, nullptr, L.ln1_g, L.ln1_b, w.XNC, D, nullptr, 0, BT, D, pr, stream));
        TRY(gemm(w.XNC, D, L.w_qkv, D, L.b_qkv, QC, D, BT, D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f));
      } else {
        TRY(gemm(w.Xn, D, wqkv + (long)D * D, D, L.b_qkv + D, QKV + D, 3 * D, R, 2 * D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, 0, 1.f, 0, nullptr, nullptr, resv));
        TRY(gemm(w.Xn, fs, L.w_qkv, D, L.b_qkv, QC, D, BT, D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f));
      }
      if (two && hipStreamWaitEvent(s, g_side.join[i], 0) != hipSuccess) return GAVA_ELAUNCH;
      {
        gava_attention_args a{};
        a.q = QC; a.ld_q = D; a.q_batch_rows = 1;
        a.k = QKV + D; a.v = QKV + 2 * D; a.ld_qkv = 3 * D;
        a.side_k = SKV; a.side_v = SKV + D; a.ld_side = 2 * D;
        a.out = w.MIXC; a.ld_out = D;
        a.batch = BT; a.heads = m->H; a.n_q = 1; a.n_kmain = n + 1;
        a.n_g = G; a.T = Tm; a.has_summary = 1; a.prec = pr;
        TRY(gava_attention(&a, stream));
      }
      TRY(gemm(w.MIXC, D, L.w_out, D, L.b_out, sv->last_x1, D, BT, D, D, GAVA_EPI_F32, pr, stream, Xin, fs));
      TRY(ln(sv->last_x1, D, nullptr, L.ln2_g, L.ln2_b, w.XNC, D, nullptr, 0, BT, D, pr, stream));
      TRY(gemm(w.XNC, D, L.w_fc1, D, L.b_fc1, w.HIDC, F, BT, F, D, GAVA_EPI_H16_QGELU, pr, stream, nullptr, 0, 0, 1.f, 0, PREC));
      TRY(gemm(w.HIDC, F, L.w_fc2, F, L.b_fc2, Xout, fs, BT, D, F, GAVA_EPI_F32, pr, stream, sv->last_x1, D));
      continue;
    }
    if (fold1) {
      Fold c = consume(L.qkv_fold_s, L.qkv_fold_t);
      TRY(gemm(w.Xn, D, L.w_qkv_fold, D, nullptr, QKV, 3 * D, R, 3 * D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f, 0, nullptr, &c, resv));
    } else {
      TRY(gemm(w.Xn, D, L.w_qkv, D, L.b_qkv, QKV, 3 * D, R, 3 * D, D, GAVA_EPI_H16, pr, stream, nullptr, 0, D, 0.125f, 0, nullptr, nullptr, resv));
    }
    if (two && hipStreamWaitEvent(s, g_side.join[i], 0) != hipSuccess) return GAVA_ELAUNCH;
    {
      gava_attention_args a{};
      a.q = QKV; a.k = QKV + D; a.v = QKV + 2 * D; a.ld_qkv = 3 * D;
      a.side_k = SKV; a.side_v = SKV + D; a.ld_side = 2 * D;
      a.out = w.MIX; a.ld_out = D;
      a.batch = BT; a.heads = m->H; a.n_q = n + 1; a.n_kmain = n + 1;
      a.n_g = G; a.T = Tm; a.has_summary = 1; a.prec = pr;
      TRY(gava_attention(&a, stream));
    }
    if (fold2) {
      TRY(gemm(w.MIX, D, L.w_out, D, L.b_out, X1, D, R, D, D, GAVA_EPI_F32, pr, stream, Xin, D, 0, 1.f, 0, nullptr, &produce));
      if (!fused) TRY(gava_row_stats(w.RSUM, D / 64, D, R, w.STATS, stream));
      Fold c = consume(L.fc1_fold_s, L.fc1_fold_t);
      TRY(gemm(w.Xn, D, L.w_fc1_fold, D, nullptr, w.HID, F, R, F, D, GAVA_EPI_H16_QGELU, pr, stream, nullptr, 0, 0, 1.f, 0, PRE, &c));
    } else {
      TRY(gemm(w.MIX, D, L.w_out, D, L.b_out, X1, D, R, D, D, GAVA_EPI_F32, pr, stream, Xin, D));
      TRY(ln(X1, D, nullptr, L.ln2_g, L.ln2_b, w.Xn, D, nullptr, 0, R, D, pr, stream));
      TRY(gemm(w.Xn, D, L.w_fc1, D, L.b_fc1, w.HID, F, R, F, D, GAVA_EPI_H16_QGELU, pr, stream, nullptr, 0, 0, 1.f, 0, PRE));
    }
    if (fold1_next) {
      TRY(gemm(w.HID, F, L.w_fc2, F, L.b_fc2, Xout, D, R, D, F, GAVA_EPI_F32, pr, stream, X1, D, 0, 1.f, 0, nullptr, &produce));
      if (!fused) TRY(gava_row_stats(w.RSUM, D / 64, D, R, w.STATS, stream));
    } else {
      TRY(gemm(w.HID, F, L.w_fc2, F, L.b_fc2, Xout, D, R, D, F, GAVA_EPI_F32, pr, stream, X1, D));
    }
    folded_in = fold1_next;
  }
  const float* Xf = sv->x + (size_t)m->layers * RD;
  TRY(ln(Xf, fs, nullptr, m->lnpost_g, m->lnpost_b, w.CLSPOST, 3 * D, nullptr, 0, BT, D, pr, stream, 1));
  TRY(gemm(w.CLSPOST, 3 * D, m->w_proj, 3 * D, nullptr, w.PROJ, E, BT, E, 3 * D, GAVA_EPI_F32, pr, stream));
  TRY(gava::mean_rows(w.PROJ, cls_x, m->B, m->T_in, E, s));
  TRY(gava::mean_rows(w.SUMM, summary, BT / Tm, Tm, D, s));
  return GAVA_OK;
}

// ---------------------------------------------------------------------------------------------
namespace {
struct TextWs { float* X; void* Xn; void* QKV; void* MIX; void* HID; void* EOT16; float* QKV32; size_t total; };
// fp32 q/k/v + fp32 softmax core (gava_text_model.attn_f32): inference with split-precision GEMMs and a short sequence
bool text_attn_f32(const gava_text_model* m) { return m->attn_f32 && m->split && m->L <= 128; }

TextWs carve_text(const gava_text_model* m, void* ws, size_t cap) {
  const long R = (long)m->n_prompts * m->L, W = m->W, S = m->split ? 3 : 1;
  Carver c(ws, cap);
  TextWs w;
  w.X = (float*)c.take(R * W * 4);
  w.Xn = c.take(R * S * W * 2);
  w.QKV = c.take(R * 3 * W * 2);
  w.MIX = c.take(R * S * W * 2);
  w.HID = c.take(R * S * 4 * W * 2);
  w.EOT16 = c.take((long)m->n_prompts * S * W * 2);
  w.QKV32 = text_attn_f32(m) ? (float*)c.take(R * 3 * W * 4) : nullptr;
  w.total = (c.off + 255) & ~(size_t)255;
  return w;
}

int check_text(const gava_text_model* m) {
  if (!m || !m->layer) return GAVA_EINVAL;
  if (m->n_prompts <= 0 || m->L <= 0 || m->L > 320 || m->layers <= 0) return GAVA_EINVAL;
  if (m->W != m->H * 64 || m->W % 128 || m->E % 128 || m->W > 1024) return GAVA_EINVAL;
  // L == n_ctx + 1 is a valid trimmed length: knowledge-aware prompts without descriptions put the EOT look-up inside
  // the context slots (text_encoder.py:169,298), and text_embed_kernel handles a prompt that ends with its context
  if (m->n_ctx < 0 || m->n_ctx + 1 > m->L) return GAVA_EINVAL;
  return GAVA_OK;
}
}  // namespace

extern "C" size_t gava_text_workspace_bytes(const gava_text_model* m) {
  if (check_text(m) != GAVA_OK) return 0;
  return carve_text(m, nullptr, 0).total;
}

extern "C" int gava_text_forward(const gava_text_model* m, const int32_t* tokens, const float* ctx,
                                 const int32_t* eot_index, float* out, void* workspace, size_t workspace_bytes,
                                 gava_stream_t stream) {
  return gava_text_forward_train(m, tokens, ctx, eot_index, out, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int gava_text_forward_train(const gava_text_model* m, const int32_t* tokens, const float* ctx,
                                       const int32_t* eot_index, float* out, float* saved_x, void* workspace,
                                       size_t workspace_bytes, gava_stream_t stream) {
  TRY(check_text(m));
  if (!ctx || !eot_index || !out || !workspace) return GAVA_EINVAL;   // tokens == NULL: direct mode (gava_hip.h)
  const TextWs w = carve_text(m, workspace, workspace_bytes);
  if (w.total > workspace_bytes) return GAVA_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int R = m->n_prompts * m->L, W = m->W, pr = m->prec;
  const int sp = m->split ? 1 : 0, S = sp ? 3 : 1;  // split precision: A rows are [hi|lo|hi], K' = 3K
  const bool f32_core = !saved_x && text_attn_f32(m);   // inference only: the backward recomputes blocks with the 16-bit core
  TRY(gava::text_embed(m->token_embedding, m->positional_embedding, ctx, tokens, w.X, m->n_prompts, m->L, W, m->n_ctx, s));
  auto keep = [&](int i) -> int {   // training: the input of block i (i == layers: the final stream)
    if (!saved_x) return GAVA_OK;
    return hipMemcpyAsync(saved_x + (size_t)i * R * W, w.X, (size_t)R * W * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess
               ? GAVA_OK : GAVA_ELAUNCH;
  };
  for (int i = 0; i < m->layers; ++i) {
    const gava_text_layer& L = m->layer[i];
    TRY(keep(i));
    TRY(ln(w.X, W, nullptr, L.ln1_g, L.ln1_b, w.Xn, S * W, nullptr, 0, R, W, pr, stream, sp));
    if (f32_core) {
      // q, k, v straight out of the split-precision GEMM in fp32, softmax core in fp32: nothing of the attention branch is
      // rounded to 16 bits before the [hi | lo | hi] operand of the out-projection
      TRY(gemm(w.Xn, S * W, L.w_qkv, S * W, L.b_qkv, w.QKV32, 3 * W, R, 3 * W, S * W, GAVA_EPI_F32, pr, stream));
      gava_attention_f32_args a{};
      a.q = w.QKV32; a.k = w.QKV32 + W; a.v = w.QKV32 + 2 * W; a.ld = 3 * W; a.out = w.MIX; a.ld_out = S * W;
      a.batch = m->n_prompts; a.heads = m->H; a.L = m->L; a.causal = 1; a.prec = pr; a.split_out = sp; a.scale = 0.125f;
      TRY(gava_attention_f32(&a, stream));
    } else {
      TRY(gemm(w.Xn, S * W, L.w_qkv, S * W, L.b_qkv, w.QKV, 3 * W, R, 3 * W, S * W, GAVA_EPI_H16, pr, stream, nullptr, 0, W, 0.125f));
      gava_attention_args a{};
      const unsigned short* q = (const unsigned short*)w.QKV;
      a.q = q; a.k = q + W; a.v = q + 2 * W; a.ld_qkv = 3 * W; a.out = w.MIX; a.ld_out = S * W;
      a.batch = m->n_prompts; a.heads = m->H; a.n_q = m->L; a.n_kmain = m->L; a.causal = 1; a.prec = pr;
      a.split_out = sp;
      TRY(gava_attention(&a, stream));
    }
    TRY(gemm(w.MIX, S * W, L.w_out, S * W, L.b_out, w.X, W, R, W, S * W, GAVA_EPI_F32, pr, stream, w.X, W));
    TRY(ln(w.X, W, nullptr, L.ln2_g, L.ln2_b, w.Xn, S * W, nullptr, 0, R, W, pr, stream, sp));
    TRY(gemm(w.Xn, S * W, L.w_fc, S * W, L.b_fc, w.HID, S * 4 * W, R, 4 * W, S * W, GAVA_EPI_H16_QGELU, pr, stream, nullptr, 0, 0, 1.f, sp));
    TRY(gemm(w.HID, S * 4 * W, L.w_proj, S * 4 * W, L.b_proj, w.X, W, R, W, S * 4 * W, GAVA_EPI_F32, pr, stream, w.X, W));
  }
  TRY(keep(m->layers));
  // ln_final on the EOT rows only, then text_projection (VitaCLIP_text_encoder.py:164-169)
  TRY(ln(w.X, W, eot_index, m->lnf_g, m->lnf_b, w.EOT16, S * W, nullptr, 0, m->n_prompts, W, pr, stream, sp));
  TRY(gemm(w.EOT16, S * W, m->w_tproj, S * W, nullptr, out, m->E, m->n_prompts, m->E, S * W, GAVA_EPI_F32, pr, stream));
  return GAVA_OK;
}
