/* gava_hip.h — C ABI of libgava_hip.so: the MI355X (gfx950) implementation of the GaVA-CLIP
 * video-frame forward path.
 *
 * The reference (lisqzqng/GaVA-CLIP) is pure PyTorch and has no FFI or operator registry; its
 * boundary for this path is the Python class VitaCLIP(nn.Module)
 * (training/VitaCLIP_model.py:22-401).  Each entry point below replaces the stock PyTorch ops
 * of one stretch of that forward; the citation says which.  The host-side mirror
 * (gava_clip_amd/model.py) binds them with ctypes, see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; nothing is allocated, freed or
 *     synchronised inside a call; all work is enqueued on `stream` (a hipStream_t).
 *   - return value: 0 on success, negative GAVA_E* on a rejected call (nothing launched).
 *   - "h16" = 16-bit MFMA operand storage, fp16 or bf16 as selected by `prec`
 *     (GAVA_PREC_*).  Accumulators, LayerNorm, softmax, residual stream and the
 *     similarity head are always fp32.
 *   - weights are [out_features][in_features] row-major (nn.Linear layout).
 */
#ifndef GAVA_HIP_H
#define GAVA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAVA_OK 0
#define GAVA_EINVAL (-1)   /* shape/alignment the kernels do not support */
#define GAVA_EWORKSPACE (-2) /* workspace too small */
#define GAVA_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after a launch */

#define GAVA_PREC_F16 0
#define GAVA_PREC_BF16 1

/* GEMM epilogues */
#define GAVA_EPI_H16 0        /* out16 = (acc + bias) * (col < scale_cols ? scale : 1)        */
#define GAVA_EPI_H16_QGELU 1  /* out16 = quickgelu(acc + bias), x*sigmoid(1.702x)              */
#define GAVA_EPI_F32 2        /* out32 = acc + bias (+ resid32, may alias out32)               */
#define GAVA_EPI_F32_PATCH 3  /* patch-embed: row remap + pos/time embedding (see below)       */
#define GAVA_EPI_H16_QGELU_BWD 4 /* backward: out16 = acc * quickgelu'(aux16), aux = pre-activation [M][ldo] */

typedef void* gava_stream_t;

int gava_abi_version(void);
/* Debug hook: device buffer (u64[8] per wave) that instrumented kernels fill with per-segment cycle
 * sums; NULL (default) disables every stamp.  Not used by the product path. */
int gava_debug_set_buffer(void* dev_u64);
/* Measurement hook (bench.py `roofline`): while enabled, gava_vision_forward brackets ONE kernel of every full-width block
 * with a pair of HIP events on the stream it launches on: on = GAVA_PROBE_FC1 (the fc1 GEMM), _OUT (out_proj), _FC2,
 * _QKV (the LayerNorm-folded qkv GEMM: blocks 1 .. layers-2) or _ATTN; 0 = off.  gava_probe_fc1_read waits for the last
 * pair and returns the number of block slots, writing up to `cap` elapsed times in ms (-1 in the slots of blocks the probed
 * kernel did not run in: _QKV leaves slot 0 unused).  Off by default; never enabled by the model code. */
#define GAVA_PROBE_FC1 1
#define GAVA_PROBE_OUT 2
#define GAVA_PROBE_FC2 3
#define GAVA_PROBE_QKV 4
#define GAVA_PROBE_ATTN 5
int gava_probe_fc1_enable(int on);
int gava_probe_fc1_read(float* ms, int cap);

/* C[M,N] = A[M,K] · W[N,K]^T with a fused epilogue.  Replaces nn.Linear / the conv-as-GEMM of
 * ImagePatchEmbed2D (VitaCLIP_vision_encoder_utils.py:66,79,110-115,215-219;
 * VitaCLIP_text_encoder.py:73-77 and the projections inside nn.MultiheadAttention :71).
 * Requirements: N % 128 == 0, K % 64 == 0, lda/ldw multiples of 8 elements, 16-byte aligned
 * pointers.  M is arbitrary. */
/* One decoded video for the fused input path: the patch-embedding GEMM builds its A tiles straight from
 * the uint8 RGB frames [n_frames][height][width][3] - temporal crop, (u8/255 - mean)/std, bilinear short-side resize and
 * centre crop of video_dataset/dataset.py:117-139 are evaluated per loaded pixel, with the arithmetic of
 * gava_preprocess_clip (bit-identical values), and the fp32 clip is never written to HBM.  gava_clip_geometry() fills the
 * derived fields from the video's size.  Device array, one entry per clip of the batch. */
typedef struct {
  const uint8_t* frames; int n_frames, height, width;
  int t_st, rate;                   /* frame(t) = min(t_st + t*rate, n_frames-1)                    */
  int h_st, w_st;                   /* crop offset inside the resized frame                          */
  float scale_h, scale_w;           /* source / resized extent (align_corners = False)               */
} gava_clip_desc;
/* host helper: the integer / float geometry of dataset.py:124-129,163-186 for one video; 0 on success */
int gava_clip_geometry(gava_clip_desc* d, int T, int rate, int size, int first_temporal_view, int first_spatial_view);

typedef struct {
  const void* A; int64_t lda;       /* h16 [M][lda]                                    */
  const void* W; int64_t ldw;       /* h16 [N][ldw]                                    */
  const float* bias;                /* [N] or NULL                                     */
  void* out; int64_t ldo;           /* h16 or fp32 [rows][ldo]                         */
  const float* resid; int64_t ldr;  /* EPI_F32: optional fp32 residual, may alias out  */
  int M, N, K;
  int epilogue;                     /* GAVA_EPI_*                                      */
  int prec;                         /* GAVA_PREC_*                                     */
  int scale_cols; float scale;      /* EPI_H16: columns [0,scale_cols) are multiplied  */
  /* EPI_F32_PATCH: GEMM row m = frame*n_patches + p is written to output row
   * frame*(n_patches+1) + 1 + p with + pos[(1+p)][:] + time[(frame % T)][:]
   * (VitaCLIP_vision_encoder.py:108-111,86-100). */
  const float* pos; const float* time; int n_patches; int T;
  /* EPI_F32_PATCH with A == NULL: im2col-free patch embedding.  The A tile is built on the fly from the
   * frames `frames` = fp32 (B,3,T,size,size): row m = frame*n_patches + p, column k = (c,ky,kx) reads
   * frames[b][c][t][py*P+ky][px*P+kx] (frame = b*T+t); loaded coalesced, converted to h16 and written to
   * the swizzled LDS tile, never materialised in HBM.  K = 3*P*P rounded up to 64 (W zero-padded). */
  const float* frames; int frame_size; int patch;
  /* EPI_H16 / EPI_H16_QGELU: split-precision output.  Row layout becomes [hi(N) | lo(N) | hi(N)]
   * (ldo >= 3N) with lo = h16(v - hi): the A operand of a following GEMM whose weight is packed
   * [W_hi | W_hi | W_lo] (K' = 3K), i.e. A_hi W_hi + A_lo W_hi + A_hi W_lo in one pass. */
  int split_out;
  const void* aux;                  /* EPI_H16_QGELU_BWD: h16 pre-activations, rows as out (ld = ldo) */
  int aux_prec;                     /* ... stored as GAVA_PREC_F16 / _BF16 (may differ from `prec`: activations kept
                                     * from an fp16 forward, gradients in bf16) */
  void* aux_out;                    /* EPI_H16_QGELU (training forward): also store the pre-activation acc + bias as
                                     * h16 [M][ldo] here; NULL = off */
  /* LayerNorm folded into the NEXT GEMM ("LayerNorm folding").  Producer side, EPI_F32: besides out32 also
   * store x16 = h16(out) [M][ld_x16] and, per row and 64-column group, the partial sums (sum x, sum x^2) as
   * float2 rowsum[M][N/64].  gava_row_stats turns them into (mean, rstd).  Consumer side, EPI_H16 / EPI_H16_QGELU with
   * A = x16 and W = h16(gamma * W): out = rstd_m * (acc - mean_m * fold_s[n]) + fold_t[n], fold_s[n] = sum_k W[n][k],
   * fold_t[n] = sum_k beta[k] * W0[n][k] + bias[n] (bias must then be NULL). */
  void* x16_out; int64_t ld_x16; float* rowsum_out;
  const float* fold_stats; const float* fold_s; const float* fold_t;
  /* persistent (one workgroup per CU) kernels only: leave this many CUs out of the grid, so that small kernels the
   * caller runs on another stream beside this GEMM get CUs of their own (0 = use every CU).  Per call: the library keeps
   * no launch state between calls. */
  int cu_reserve;
  /* LayerNorm folding without the gava_row_stats launch in between (big-M GEMMs: the persistent kernel only; N % 256 == 0,
   * N <= 1024; anything else is rejected, never silently routed to another path).  Producer: rowsum_reduced != 0 makes
   * rowsum_out float2 [ceil(M/256)*256][4] - slot n/256 of row m holds (sum x, sum x^2) over columns [256*(n/256), +256),
   * partial sums added in a fixed order.  Consumer: fold_partials = that array (instead of fold_stats); the
   * kernel derives (mean, rstd) of its rows itself (eps 1e-5, variance = E[x^2] - mean^2, fixed summation order). */
  int rowsum_reduced;
  const float* fold_partials;
  /* EPI_F32_PATCH with A == NULL and frames == NULL: the uint8 source (see gava_clip_desc): clips = device array
   * [M / (n_patches * T)], frame_size = the crop size, clip_lut = device fp32 [3][256]: lut[c][v] = (v/255 - mean[c]) / std[c]
   * as the caller's fp32 arithmetic gives it (the reference's own torch expression: the normalisation is then exact). */
  const gava_clip_desc* clips; const float* clip_lut;
  /* Kernel selection.  0 = automatic (by shape and epilogue, the only value the forward drivers pass).  Tests and A/B
   * timing may name a kernel: GAVA_KERNEL_256 = the persistent 256 x 256 kernel (one 512-thread workgroup per CU),
   * GAVA_KERNEL_PAIR = the persistent 128 x 256 kernel run as two 256-thread workgroups per CU (EPI_F32 only: the
   * residual-stream GEMMs out_proj / fc2 and their folding-producer form; N % 256 == 0, N <= 1024, K % 128 == 0).  A named
   * kernel that does not take the shape is rejected (GAVA_EINVAL), never replaced. */
  int kernel;
  /* Weight-lo pass (DESIGN.md "Numerics", round 4).  With 16-bit operands the rounding of the frozen WEIGHTS, not of the
   * activations, sets the logits error (tools/error_budget.py), so the product can be completed by A . W_lo^T with
   * W_lo = W - h16(W) while A stays a single 16-bit operand:
   *   w_lo = 1: W rows are [W_hi (K) | W_lo (K)] h16, ldw >= 2K; the k-loop runs 2K deep and re-reads A for the second half
   *             (nothing is stored twice).  K stays the number of columns of A.  Every kernel but GAVA_KERNEL_PAIR.
   *   w_lo = 2: the lo product at 8 bits on the block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4: twice the 16-bit
   *             rate): after the K-deep 16-bit loop, K more columns with A8 = bf8 (e5m2) copy of A, bytes [M][lda8], and
   *             W8 = e4m3 rows [N][ldw8] holding 2^w8_exp * W_lo (the MFMA's E8M0 scale operand takes the 2^-w8_exp back, so
   *             both products share the accumulators).  W stays [N][K] = W_hi.  K % 128 == 0; the 8-bit rows keep the BYTE
   *             pitch of their 16-bit twins, lda8 == 2 lda and ldw8 == 2 ldw (the kernel then addresses both phases with one
   *             set of offsets; half of each 8-bit row's pitch is unused); the persistent 256 x 256 kernel only (rejected
   *             elsewhere, never replaced).
   * Producers of a w_lo = 2 operand: out8 (EPI_H16 / EPI_H16_QGELU) = bf8 copy of the 16-bit output rows, bytes [M][ldo8];
   * x8_out (EPI_F32 with x16_out) = bf8 copy of x16_out, bytes [M][ld_x8].  bf8 = the fp16 value rounded to 2 mantissa
   * bits (fp16 operands only). */
  int w_lo;
  const void* A8; int64_t lda8; const void* W8; int64_t ldw8; int w8_exp;
  void* out8; int64_t ldo8; void* x8_out; int64_t ld_x8;
  /* The residual stream as a 16-bit pair (round 4; inference driver of the vision tower, DESIGN.md "Residual stream").  EPI_F32
   * with resid16 != NULL: the residual is read as resid16 + resid_lo (both [M][ldr]: hi in the operand type, lo in fp16 whatever
   * the operand type) and the result x = resid + A W^T + bias leaves as x16_out = h16(x) and xlo_out = fp16(x - h16(x)) (both
   * [M][ld_x16]) together with the producer's row sums (rowsum_out, computed from the fp32 x): 8 bytes per element through the
   * epilogue instead of 10 (fp32 in, fp32 out, 16-bit copy), and the hi half IS the operand of the GEMM that consumes the
   * LayerNorm fold.  |x - hi - lo| <= 2^-22 |x| (fp16 operands; 2^-20 with bf16).  out and resid must be NULL, x16_out /
   * xlo_out may alias resid16 / resid_lo (every element is read and written by the same lane).  The persistent 256 x 256
   * kernel with the ping-pong loop only (N % 256 == 0, K % 128 == 0, K >= 256, w_lo <= 1); rejected elsewhere. */
  const void* resid16; const void* resid_lo; void* xlo_out;
} gava_gemm_args;
#define GAVA_KERNEL_AUTO 0
#define GAVA_KERNEL_256 3
#define GAVA_KERNEL_PAIR 4
#define GAVA_KERNEL_PP 5    /* round 4: the persistent 256 x 256 kernel with the "ping-pong" k-loop (four 16-MFMA phases per k-tile,
                               the two wave groups one barrier apart, half-tiles staged six ahead): EPI_H16 / EPI_H16_QGELU
                               with the LayerNorm fold, plain EPI_H16, EPI_F32 with a residual (and its producer extras);
                               N % 256 == 0, K % 128 == 0, K >= 256 */
int gava_gemm(const gava_gemm_args* a, gava_stream_t stream);
/* Measurement helper (bench.py names the kernel instantiation a GEMM runs as): 1 when an EPI_F32 GEMM of M x N on the
 * persistent 256 x 256 kernel of the CURRENT device takes the aligned tile walk (the <..., ALIGN = true> instantiation:
 * super-tile blocks dealt to the XCDs), 0 when it takes the plain walk.  The same decision gava_gemm makes. */
int gava_gemm_aligned_walk(int M, int N, int cu_reserve);

/* Row LayerNorm (eps 1e-5, affine) fp32 -> h16 and/or fp32; one wave per row.  Replaces
 * LayerNorm (VitaCLIP_vision_encoder_utils.py:22-28, VitaCLIP_text_encoder.py:19-25).
 * gamma == NULL means "no normalisation": the row is only converted (used to feed cls_proj,
 * utils:164-165).  in_row_index (optional, int32[rows]) gathers input rows. D % 4 == 0, D <= 1024. */
typedef struct {
  const float* in; int64_t in_stride; const int32_t* in_row_index;
  const float* gamma; const float* beta;
  void* out16; int64_t out16_stride;     /* may be NULL */
  float* out32; int64_t out32_stride;    /* may be NULL; may alias in */
  int rows, D, prec;
  int split_out;                         /* out16 row = [hi(D) | lo(D) | hi(D)], see gava_gemm_args */
  /* Optional second LayerNorm applied to the first one's fp32 result in the same pass (both NULL = off): out32 then
   * receives LN(in; gamma, beta) and out16 receives LN(LN(in; gamma, beta); gamma2, beta2).  ln_pre followed by norm1 of
   * block 0 (VitaCLIP_vision_encoder.py:113, VitaCLIP_vision_encoder_utils.py:190): one read of the embedding instead
   * of two row passes. */
  const float* gamma2; const float* beta2;
  /* Optional 16-bit pair of what out32 receives (see gava_gemm_args.resid16): out_hi = h16(y), out_lo = fp16(y - h16(y)),
   * rows out_hl_stride apart, y = the FIRST LayerNorm's result (or the converted row when gamma == NULL).  Both or neither;
   * out32 may then be NULL. */
  void* out_hi; void* out_lo; int64_t out_hl_stride;
} gava_layernorm_args;
int gava_layernorm(const gava_layernorm_args* a, gava_stream_t stream);

/* Fused softmax(QK^T)V for head dim 64, whole key row resident in LDS (single-pass softmax).
 * Replaces Attention.forward's einsum/softmax/einsum (VitaCLIP_vision_encoder_utils.py:71-77)
 * and the scaled-dot-product inside nn.MultiheadAttention (VitaCLIP_text_encoder.py:81-83).
 * Q must already carry the 1/sqrt(64) factor.  Problem (n, h): queries are rows
 * n*n_q .. n*n_q+n_q-1 of q; keys/values are n_kmain rows n*n_kmain.. of k/v followed by
 * "side" rows taken from side_k/side_v (vision prompt tokens, see gava_vision_forward):
 *   n_g rows [0,n_g)                            shared by every problem   (global prompts)
 *   T   rows n_g + (n / T)*T + [0,T)            shared by the T frames of a clip (local prompts)
 *   1   row  n_g + batch + n                    per problem               (summary token)
 * With n_side == 0 there are no side rows.  causal: key j is visible to query i iff j <= i.
 * Total keys <= 320. */
typedef struct {
  const void* q; const void* k; const void* v; int64_t ld_qkv;  /* h16, element strides */
  const void* side_k; const void* side_v; int64_t ld_side;
  void* out; int64_t ld_out;                                     /* h16 [batch*n_q][ld_out] */
  int batch, heads, n_q, n_kmain;
  int n_g, T, has_summary;                                       /* side-row structure      */
  int causal, prec;
  int split_out;   /* out row = [hi(heads*64) | lo | hi], see gava_gemm_args */
  /* optional query addressing (0 = defaults): queries of problem n are rows n*q_batch_rows + [0,n_q)
   * of q with row stride ld_q.  Defaults: q_batch_rows = n_q, ld_q = ld_qkv.  Lets the last vision
   * block run only its CLS queries (n_q = 1, q_batch_rows = tokens per frame). */
  int q_batch_rows; int64_t ld_q;
} gava_attention_args;
int gava_attention(const gava_attention_args* a, gava_stream_t stream);

/* Exact-fp32 softmax attention for SHORT sequences (the text tower in inference: 77 padded tokens, 15-20 after the rows behind
 * the EOT are trimmed): q, k, v fp32 rows [batch*L][ld] (head h = columns [64h, 64h+64)), scores = scale * q.k (+ causal
 * mask), softmax and P.V in fp32 on the vector ALU - no 16-bit rounding of Q, K, V or P anywhere.  out: h16 rows
 * [batch*L][ld_out]; split_out != 0 writes [hi | lo | hi] copies heads*64 columns apart (A operand of a split-precision
 * GEMM, see gava_gemm_args.split_out).  Replaces the scaled-dot-product core of nn.MultiheadAttention
 * (VitaCLIP_text_encoder.py:71,83) where 16-bit MFMA operands would set the error of the text features.  L <= 128. */
typedef struct {
  const float* q; const float* k; const float* v; int64_t ld;
  void* out; int64_t ld_out;
  int batch, heads, L, causal, prec, split_out;
  float scale;
} gava_attention_f32_args;
int gava_attention_f32(const gava_attention_f32_args* a, gava_stream_t stream);

/* ---- fused drivers ---------------------------------------------------------------------- */

typedef struct {                 /* one TransformerEncoderLayer, utils:93-203 */
  const void* w_qkv; const float* b_qkv;   /* [3D][D]: q_proj;k_proj;v_proj stacked          */
  const void* w_out; const float* b_out;   /* [D][D]                                         */
  const void* w_fc1; const float* b_fc1;   /* [F][D]                                         */
  const void* w_fc2; const float* b_fc2;   /* [D][F]                                         */
  const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
  const void* w_cls; const float* b_cls;   /* cls_proj [D][D]                                */
  const float* sln_g; const float* sln_b;  /* summary_ln                                     */
  const void* w_sqkv; const float* b_sqkv; /* summary_attn_layer q;k;v stacked [3D][D]       */
  const void* w_sout; const float* b_sout;
  const float* local_prompts;              /* [T][D] fp32                                    */
  const float* global_prompts;             /* [G][D] fp32 (row i of visual.global_prompts)   */
  /* Optional (all NULL = off): LayerNorm folded into the consumer GEMM, inference driver only (see gava_gemm_args).
   * w_*_fold = h16(gamma * W) [N][D]; *_fold_s[n] = sum_k w_fold[n][k] (of the ROUNDED weight);
   * *_fold_t[n] = sum_k beta[k] * W[n][k] + bias[n] in fp32.  qkv: norm1 of this block, fc1: norm2. */
  const void* w_qkv_fold; const float* qkv_fold_s; const float* qkv_fold_t;
  const void* w_fc1_fold; const float* fc1_fold_s; const float* fc1_fold_t;
  /* Optional (all NULL = off), read for the LAST block of the inference driver only: its B*T CLS rows - the only rows
   * of that block that reach the outputs (VitaCLIP_vision_encoder.py:126) - run q_proj, out_proj, fc1 and fc2 in split
   * precision (weights packed [W_hi | W_hi | W_lo], see gava_gemm_args.split_out): 1/197 of the block's rows, and their
   * rounding lands on the video feature directly.  w_q_split [D][3D], w_out_split [D][3D], w_fc1_split [F][3D],
   * w_fc2_split [D][3F]. */
  const void* w_q_split; const void* w_out_split; const void* w_fc1_split; const void* w_fc2_split;
} gava_vision_layer;

typedef struct {
  int B, T_in, T_model;          /* clips, frames per clip in x, num_frames of the model      */
  int size, P, D, H, layers, F, E, G;
  int prec;
  const void* w_patch; const float* b_patch;   /* [D][Kp] h16, Kp = 3*P*P rounded up to 64   */
  const float* cls_token; const float* pos_embed; const float* time_embed; /* time: [T_in][D] */
  const float* lnpre_g; const float* lnpre_b; const float* lnpost_g; const float* lnpost_b;
  const void* w_proj;                          /* visual.proj transposed, split-packed: [E][3D] =
                                                  [W_hi | W_hi | W_lo] h16 (see gava_gemm_args)  */
  const gava_vision_layer* layer;              /* host array [layers]                         */
  /* optional uint8 input (gava_gemm_args.clips): when clips != NULL the drivers' `x` argument is ignored (may be NULL) */
  const gava_clip_desc* clips; const float* clip_lut;   /* fp32 [3][256], see gava_gemm_args */
  /* Weight-lo pass (gava_gemm_args.w_lo), inference drivers only (gava_vision_forward; the training drivers reject it).
   * w_lo = 1: w_patch and every 16-bit weight of gava_vision_layer except the *_split ones are packed [W_hi | W_lo] with
   * twice the columns ([3D][2D], [F][2D], [D][2F], ...; the folded copies too, their fold_s = row sums of hi + lo).
   * w_lo = 2 (fp16 operands): packed as 1, and the LayerNorm-folded qkv / fc1 GEMMs and fc2 over all B*T*(n+1) rows run their
   * lo product at 8 bits from the layer's *8 weights (gava_vision_layer8), their A operands leaving the producers (out_proj,
   * fc2, fc1) with a bf8 copy; every other GEMM (and every shape the persistent kernel does not take) as in w_lo = 1. */
  int w_lo;
  const struct gava_vision_layer8* layer8;     /* host array [layers], w_lo = 2 only */
} gava_vision_model;

/* e4m3 rows of 2^exp * W_lo for the 8-bit lo product (gava_gemm_args.W8 / w8_exp) of the GEMMs that run it - the
 * LayerNorm-folded qkv and fc1 (W_lo of the folded weight h16(gamma * W)) and fc2: [3D] / [F] rows of D bytes, [D] rows of F
 * bytes, each row 4x its length apart (ldw8 = 2 ldw, and the 16-bit weight rows are [W_hi | W_lo], gava_vision_model.w_lo).
 * *_fold_s8[n] = sum_k (W_hi[n][k] + 2^-exp W8[n][k]), the row sums of the weight these GEMMs multiply by.
 * out_proj keeps the 16-bit lo product (its A operand, the attention output, has no 8-bit copy). */
typedef struct gava_vision_layer8 {
  const void* w_qkv_fold8; const void* w_fc1_fold8; const void* w_fc28;
  const float* qkv_fold_s8; const float* fc1_fold_s8;
  int qkv_fold_exp, fc1_fold_exp, fc2_exp;
} gava_vision_layer8;

/* CLIPVisionEncoder.forward (VitaCLIP_vision_encoder.py:102-132).
 * x: fp32 (B,3,T_in,size,size) contiguous.  cls_x: fp32 [B][E] (un-normalised, :126-128),
 * summary: fp32 [B][D] (:129-130).  debug_cls (optional): fp32 [layers][B*T_in][D] receives the
 * CLS row of every frame after each block. */
size_t gava_vision_workspace_bytes(const gava_vision_model* m);
/* Measurement / test helper: 1 when gava_vision_forward keeps the residual stream of this model and batch as a 16-bit pair
 * between ln_pre and the last block (gava_gemm_args.resid16: big batches with every block's LayerNorms folded), else 0.  The
 * decision the driver makes (environment switch GAVA_PAIR_STREAM=0: never). */
int gava_vision_pair_stream(const gava_vision_model* m);
int gava_vision_forward(const gava_vision_model* m, const float* x, float* cls_x, float* summary,
                        float* debug_cls, void* workspace, size_t workspace_bytes,
                        gava_stream_t stream);

typedef struct {                 /* ResidualAttentionBlock, VitaCLIP_text_encoder.py:67-88 */
  const void* w_qkv; const float* b_qkv;   /* attn.in_proj_weight [3W][W]                    */
  const void* w_out; const float* b_out;
  const void* w_fc; const float* b_fc;     /* mlp.c_fc [4W][W]                               */
  const void* w_proj; const float* b_proj; /* mlp.c_proj [W][4W]                             */
  const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
} gava_text_layer;

typedef struct {
  int n_prompts, L, W, H, layers, E, n_ctx, prec;
  int split;                               /* 1: split-precision GEMMs; every weight below is then
                                              packed [W_hi | W_hi | W_lo] with 3x the columns    */
  int attn_f32;                            /* with split, inference, L <= 128: q/k/v leave their GEMM in fp32 and the
                                              attention core runs in fp32 (gava_attention_f32)  */
  const float* token_embedding;            /* [vocab][W] fp32                                */
  const float* positional_embedding;       /* [L][W]                                         */
  const float* lnf_g; const float* lnf_b;
  const void* w_tproj;                     /* text_projection transposed [E][W] h16          */
  const gava_text_layer* layer;            /* host array [layers]                            */
} gava_text_model;

/* TextPromptLearner.forward + CLIPTextEncoder.forward for all prompts in one batch
 * (VitaCLIP_text_encoder.py:310-332,154-171; the per-class Python loop of
 * VitaCLIP_model.py:282-285).  tokens: int32 [n_prompts][L]; ctx: fp32 [n_prompts][n_ctx][W];
 * eot_index: int32 [n_prompts], FLAT row n*L + column of the token vocab-1 (text_encoder.py:169).
 * out: fp32 [n_prompts][E].
 * tokens == NULL: direct mode = CLIPTextEncoder.forward(prompts, tokenized_prompts) called on ready-made prompt
 * embeddings (evaluation/zero_shot.py:75-76, utils/prepare_embedding.py): ctx is then fp32 [n_prompts][L][W], the
 * whole `prompts` argument; n_ctx is ignored. */
size_t gava_text_workspace_bytes(const gava_text_model* m);
int gava_text_forward(const gava_text_model* m, const int32_t* tokens, const float* ctx,
                      const int32_t* eot_index, float* out, void* workspace,
                      size_t workspace_bytes, gava_stream_t stream);

/* Similarity head (VitaCLIP_model.py:255,287-293): L2-normalise video [B][E] and text
 * [C*n_kv][E] rows (no epsilon), logits[b][c] = exp(logit_scale) * mean_k <v_b, t_{c,k}>
 * (+ logit_bias if not NULL), computed as one exact-fp32 GEMM against the class means of the unit
 * prompt features (the mean commutes with the dot product); text_features[c] = normalise(mean_k
 * t_{c,k}) [C][E].  n_kv = prompts per class (1 for plain prompts, the number of knowledge versions
 * for KAPT).  All fp32.  video_norm receives the normalised video features. */
int gava_similarity_head(const float* video, const float* text, const float* logit_scale,
                         const float* logit_bias, int B, int C, int n_kv, int E, float* logits,
                         float* text_features, float* video_norm, gava_stream_t stream);

/* ---- Backward of the trainable subset (training/train.py:441-490) --------------------------
 * Every transformer weight of the reference is frozen (VitaCLIP_model.py:230-239): the gradient only has to flow
 * THROUGH the GEMMs to the prompt parameters.  dgrad is gava_gemm on a transposed weight copy
 * (dX = dY . W == gemm(A = dY, W = W^T stored [in][out])); the three kernels below are the rest of the chain.
 * Gradient operands are 16-bit (bf16 recommended: fp32 range), accumulation and LayerNorm arithmetic fp32. */

/* LayerNorm backward (torch.nn.functional.layer_norm, eps 1e-5): dx = rstd*(g - mean(g) - xhat*mean(g*xhat)),
 * g = dy*gamma.  x rows may be gathered (x_row_index) and dx rows scattered (dx_row_index); accumulate != 0 adds
 * into dx (the residual branch).  dgamma/dbeta (both or neither) are accumulated with atomics.  dx16 (optional,
 * not with dx_row_index): an h16 copy of the final dx rows, i.e. the A operand of the next dgrad GEMM. */
typedef struct {
  const float* x; int64_t x_stride; const int32_t* x_row_index;
  const float* gamma;
  const float* dy; int64_t dy_stride;
  float* dx; int64_t dx_stride; const int32_t* dx_row_index;
  float* dgamma; float* dbeta;
  int rows, D, accumulate;
  void* dx16; int64_t dx16_stride; int prec;
} gava_layernorm_bwd_args;
int gava_layernorm_backward(const gava_layernorm_bwd_args* a, gava_stream_t stream);

/* QuickGELU backward: dpre = dh * s*(1 + 1.702*pre*(1-s)), s = sigmoid(1.702*pre); h16 in/out, n % 4 == 0. */
int gava_qgelu_backward(const void* pre, const void* dh, void* dpre, size_t n, int prec, gava_stream_t stream);

/* Softmax-attention backward (MFMA kernels).  q (already scaled by 1/sqrt(dh)), k, v, dout: h16 rows [batch*n][ld],
 * head h at columns [64h, 64h+64); writes dq (times q_scale, the factor folded into q), dk, dv as h16 rows
 * [batch*n][ld_dqkv].  n (+ prompt rows) <= 320.
 *   - plain or causal sequences without prompt rows (side_k == NULL): the text tower (nn.MultiheadAttention at
 *     text_encoder.py:71,83) and the T-token summary attention (vision_encoder_utils.py:169-170);
 *   - vision blocks (vision_encoder_utils.py:190-191): keys = the n rows of the frame + the gathered prompt rows
 *     [n_g global | T local rows of the clip | the frame's summary row] of side_k/side_v (same layout as
 *     gava_attention).  Prompt rows are shared between frames, so their gradients come out as
 *     per-frame partials dside_k / dside_v fp32 [batch][n_g + T + 1][ld_dside] (plain stores; row order global,
 *     local, summary) which the caller sums over the frames sharing a row.  n_q != 0: only the first n_q rows of
 *     each frame are queries.
 * `workspace`: gava_attention_backward_workspace_bytes(batch, heads, n_q) bytes of scratch (per-query softmax
 * statistics handed from the dQ kernel to the dK/dV kernel). */
typedef struct {
  const void* q; const void* k; const void* v; int64_t ld_qkv;
  const void* dout; int64_t ld_dout;
  void* dq; void* dk; void* dv; int64_t ld_dqkv;
  int batch, heads, n, causal, prec;
  float q_scale;
  const void* side_k; const void* side_v; int64_t ld_side;
  float* dside_k; float* dside_v; int64_t ld_dside;
  int n_g, T, has_summary, n_q;
  void* workspace;
  int act_prec_set, act_prec;   /* vision path only: q/k/v/side_* are stored as act_prec (fp16 activations kept from the
                                 * forward) while dout / dq / dk / dv use prec (bf16); 0 = same as prec */
  int q_batch_rows; int64_t ld_q, ld_dq; /* vision path, != 0: q, dout and dq are separate buffers holding q_batch_rows rows
                                 * per frame (strides ld_q, ld_dout, ld_dq), as gava_attention's q_batch_rows: the CLS-only
                                 * last block */
} gava_attention_bwd_args;
int gava_attention_backward(const gava_attention_bwd_args* a, gava_stream_t stream);
size_t gava_attention_backward_workspace_bytes(int batch, int heads, int n_q);

/* gava_text_forward that also keeps what the backward recomputes from: saved_x fp32 [layers+1][n_prompts*L][W] =
 * the input of every residual block and the final stream (before ln_final). */
int gava_text_forward_train(const gava_text_model* m, const int32_t* tokens, const float* ctx,
                            const int32_t* eot_index, float* out, float* saved_x, void* workspace,
                            size_t workspace_bytes, gava_stream_t stream);

/* (mean, rstd) per row from the producers' partial sums: rowsum float2 [rows][slots] -> stats float2 [rows], D columns,
 * eps 1e-5, variance = E[x^2] - mean^2 in fp32 with a fixed summation order (deterministic). */
int gava_row_stats(const float* rowsum, int slots, int D, int rows, float* stats, gava_stream_t stream);

/* gava_vision_forward that also keeps what the backward recomputes from: saved_x fp32 [layers+2][B*T_in*(n+1)][D] =
 * the embedding before ln_pre, the input of every block, and the final residual stream. */
int gava_vision_forward_train(const gava_vision_model* m, const float* x, float* cls_x, float* summary,
                              float* debug_cls, float* saved_x, void* workspace, size_t workspace_bytes,
                              gava_stream_t stream);

/* Training forward that keeps the backward's activations instead of leaving them to be recomputed.  With
 * R = B*T_in*(n+1), SR = G + 2*B*T_in:  e0 fp32 [R][D] (embedding before ln_pre);  x fp32 [layers+1][R][D] (block inputs,
 * last slot = final stream);  x1 fp32 [layers][R][D] (stream after the attention branch);  qkv h16 [layers][R][3D];
 * pre h16 [layers][R][F] (fc1 output before QuickGELU);  sidekv h16 [layers][SR][2D] (prompt-row keys/values).
 * Nothing is copied: the residual stream hops x[i] -> x1[i] -> x[i+1]. */
typedef struct {
  float* e0; float* x; float* x1; void* qkv; void* pre; void* sidekv;
  /* optional (all three or none): run the LAST block on the CLS rows only, as gava_vision_forward does, and keep its
   * CLS-row activations here: last_q h16 [B*T_in][D] (scaled queries), last_x1 fp32 [B*T_in][D], last_pre h16
   * [B*T_in][F].  The block's K/V then sit in columns D..3D of its qkv slot; x1 / pre of that block are unused and only
   * the CLS rows of the final stream are written. */
  void* last_q; float* last_x1; void* last_pre;
} gava_vision_saved;
int gava_vision_forward_keep(const gava_vision_model* m, const float* x, float* cls_x, float* summary,
                             const gava_vision_saved* saved, void* workspace, size_t workspace_bytes,
                             gava_stream_t stream);

/* Clip preprocessing of the evaluation data path (video_dataset/dataset.py:117-139, the
 * num_spatial_views = num_temporal_views = 1 case that every eval script uses): from the decoded RGB frames of
 * one video, uint8 [n_frames][height][width][3], to the model input slot fp32 [3][T][size][size]:
 *   temporal crop   frame(t) = min(st + t*rate, n_frames-1), st = max(n_frames - ((T-1)*rate+1), 0) / 2
 *                   (:163-177, a short video repeats its last frame);
 *   value           (u8/255 - mean[c]) / std[c]                                       (:119,121);
 *   resize          short side -> size, bilinear, align_corners=False, no antialias   (:124-133), i.e.
 *                   src = max((dst+0.5)*in/out - 0.5, 0), taps floor(src) and min(+1, in-1);
 *   crop            centre size x size window                                          (:181-186).
 * out_stride_c / out_stride_t are element strides of the output (frames are contiguous size*size planes), so
 * the kernel can write straight into clip b of a [B][3][T][size][size] batch.  HBM-bound byte work. */
typedef struct {
  const uint8_t* frames; int n_frames, height, width;
  float mean[3], std[3];
  int T, rate, size;
  float* out; int64_t out_stride_c, out_stride_t;
  /* the dataset returns only the FIRST of its num_spatial_views x num_temporal_views crops (`frames = frames[0]`,
   * :136-139): with more than one temporal view that is the crop starting at frame 0 (:171-172), with three spatial
   * views the top / left one (:188-199).  first_temporal_view / first_spatial_view != 0 select those; 0 = centred. */
  int first_temporal_view, first_spatial_view;
  /* optional device table fp32 [3][256] with (v/255 - mean[c]) / std[c] for every byte value (see gava_gemm_args.clip_lut):
   * when given, the kernel looks the normalised value up instead of dividing - the same bits as the fused uint8 patch loader */
  const float* lut;
} gava_preprocess_args;
int gava_preprocess_clip(const gava_preprocess_args* a, gava_stream_t stream);

/* The unfold half of ImagePatchEmbed2D (VitaCLIP_vision_encoder_utils.py:31-53: Conv2d(3, D, P, stride P) == patches x
 * W^T) as one HBM-bound pass: the 16-bit patch matrix the patch-embedding GEMM (gava_gemm, EPI_F32_PATCH with A given)
 * then reads by LDS-DMA like any other operand.
 *   out[(b*T + t) * n_patches + py*g + px][c*P*P + ky*P + kx] = h16( in(b, c, t, py*P + ky, px*P + kx) ),  g = size / P,
 * columns [3*P*P, ldo) are zeroed (K is padded to a multiple of 64 for ViT-L/14).  The input is either
 *   x      fp32 [B][3][T][size][size], the tensor VitaCLIP.forward takes (VitaCLIP_model.py:241), or
 *   clips  decoded uint8 videos: device array of B descriptors (gava_clip_desc) + clip_lut fp32 [3][256], evaluated with
 *          the arithmetic of gava_preprocess_clip (video_dataset/dataset.py:117-139) - the same bits as preprocessing to
 *          fp32 first, a quarter of the HBM traffic and no fp32 clip in memory.
 * Rounding to 16 bits is the one the GEMM's own fp32 loader applies (round to nearest even), so the product is unchanged. */
typedef struct {
  const float* x;
  const gava_clip_desc* clips; const float* clip_lut;
  int B, T, size, patch, prec;
  void* out; int64_t ldo;             /* ldo >= 3*patch*patch, multiple of 8; out 16-byte aligned */
} gava_patchify_args;
int gava_patchify(const gava_patchify_args* a, gava_stream_t stream);

/* fp32 -> h16 conversion of a contiguous array (weight packing at load time). */
int gava_convert_h16(const float* in, void* out, size_t n, int prec, gava_stream_t stream);

/* sizeof of every ABI struct as the library was compiled, in the order gemm_args, layernorm_args, attention_args,
 * attention_f32_args, clip_desc, vision_layer, vision_layer8, vision_model, text_layer, text_model, layernorm_bwd_args,
 * attention_bwd_args, vision_saved, preprocess_args, patchify_args.  Writes min(cap, 15) entries, returns 15.  A binding
 * compares them with its own mirrors at load time (gava_clip_amd/hip.py does): gava_abi_version ties the library to the
 * header, this ties the header to the mirrors. */
int gava_struct_sizes(size_t* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
