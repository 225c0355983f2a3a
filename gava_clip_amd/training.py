"""Backward of the trainable subset through the HIP kernels (first stage: the text side).

The reference trains `prompt_learner.ctx` (CoOp context vectors), the vision prompts and `logit_scale` with every
transformer weight frozen (VitaCLIP_model.py:230-239; training/train.py:441-490 calls `loss.backward()`).  The
gradient therefore only has to flow THROUGH the frozen GEMMs:

    d logits -> d text features -> text_projection^T -> ln_final' (EOT rows) -> 12 x residual block' -> d ctx

This module implements that chain for the text tower as a `torch.autograd.Function` around the C ABI:
forward = `gava_text_forward_train` (keeps the fp32 input of every block), backward = per block, in reverse:
recompute the block's activations from its saved input (LayerNorm, QKV GEMM, attention, out-proj GEMM, LayerNorm,
fc GEMM - the forward kernels), then dgrad GEMMs on transposed weight copies + `gava_qgelu_backward`,
`gava_layernorm_backward`, `gava_attention_backward`.  Gradient operands are bf16 (fp32 exponent range, so no
loss scaling inside the library; the reference's fp16 autocast needs its GradScaler), accumulation fp32.

The second half of this module does the same for the vision tower (prompt parameters, summary path, time_embed).
"""
import ctypes as C

import torch

from . import hip

BWD = hip.PREC_BF16


def _bf16(t):
    return hip.convert_h16(t.detach().float().contiguous(), BWD)


def pack_text_backward(model):
    """bf16 copies of the frozen text weights in both orientations (forward for the recompute, transposed for dgrad)."""
    t = model.textual
    layers = []
    for blk in t.transformer.resblocks:
        f32 = lambda p: p.detach().float().contiguous()
        layers.append(dict(
            w_qkv=_bf16(blk.attn.in_proj_weight), w_qkv_t=_bf16(blk.attn.in_proj_weight.detach().t()),
            b_qkv=f32(blk.attn.in_proj_bias),
            w_out=_bf16(blk.attn.out_proj.weight), w_out_t=_bf16(blk.attn.out_proj.weight.detach().t()),
            b_out=f32(blk.attn.out_proj.bias),
            w_fc=_bf16(blk.mlp.c_fc.weight), w_fc_t=_bf16(blk.mlp.c_fc.weight.detach().t()), b_fc=f32(blk.mlp.c_fc.bias),
            w_proj_t=_bf16(blk.mlp.c_proj.weight.detach().t()),
            ln1_g=f32(blk.ln_1.weight), ln1_b=f32(blk.ln_1.bias), ln2_g=f32(blk.ln_2.weight), ln2_b=f32(blk.ln_2.bias)))
    return dict(layers=layers, lnf_g=t.ln_final.weight.detach().float().contiguous(),
                # out = x @ text_projection (W,E): dx = dout @ text_projection^T = gemm(A = dout, W = text_projection)
                w_tproj=_bf16(t.text_projection))


def text_forward_train(model, ctx_param):
    """-> (text features (C,E) fp32, saved block inputs fp32 [layers+1, C*L, W])."""
    lib = hip.load()
    pk, sh = model._pack(), model._shape
    tok = pk["tokens"]
    n = tok.shape[0]
    m = hip.TextModel()
    m.n_prompts, m.L, m.W, m.H, m.layers = n, model.text_rows_per_prompt, sh["W"], sh["TH"], sh["TL"]
    m.E, m.n_ctx, m.prec = sh["E"], sh["n_ctx"], model.prec
    m.split = int(model.text_split_precision)
    for k, val in pk["txt"].items():
        setattr(m, k, val)
    m.layer = C.cast(pk["txt_layers"], C.POINTER(hip.TextLayer))
    nbytes = lib.gava_text_workspace_bytes(C.byref(m))
    if nbytes == 0:
        raise hip.GavaError(f"unsupported text shape: {sh}")
    ws = model._workspace("text", nbytes, tok.device)
    ctx = ctx_param.detach().float().contiguous()
    out = torch.empty(n, sh["E"], dtype=torch.float32, device=tok.device)
    saved = torch.empty(sh["TL"] + 1, n * model.text_rows_per_prompt, sh["W"], dtype=torch.float32, device=tok.device)
    hip.check(lib.gava_text_forward_train(C.byref(m), hip.ptr(tok), hip.ptr(ctx), hip.ptr(pk["eot"]), hip.ptr(out),
                                          hip.ptr(saved), hip.ptr(ws), ws.numel(), hip.stream_ptr()),
              "gava_text_forward_train")
    return out, saved


def text_backward(model, saved, dtext):
    """d(text features) (C,E) -> d ctx (C, n_ctx, W), through the frozen text tower."""
    sh = model._shape
    pk = model._pack()
    bw = model._pack_text_backward()
    n, L, W, H, E, n_ctx = pk["tokens"].shape[0], model.text_rows_per_prompt, sh["W"], sh["TH"], sh["E"], sh["n_ctx"]
    R = n * L
    dev = dtext.device
    bf = torch.bfloat16
    new = lambda *s, dtype=bf: torch.empty(*s, dtype=dtype, device=dev)
    eot = pk["eot"]
    dX = torch.zeros(R, W, dtype=torch.float32, device=dev)
    # text_projection^T and ln_final' on the EOT rows (VitaCLIP_text_encoder.py:164-170)
    dtext16 = hip.convert_h16(dtext.float().contiguous(), BWD)
    dEOT = new(n, W, dtype=torch.float32)
    hip.gemm(dtext16, bw["w_tproj"], None, dEOT, epilogue=hip.EPI_F32, prec=BWD)
    hip.layernorm_backward(saved[sh["TL"]], bw["lnf_g"], dEOT, dX, x_row_index=eot, dx_row_index=eot, rows=n)
    xn, qkv, mix, pre = new(R, W), new(R, 3 * W), new(R, W), new(R, 4 * W)
    X1, dx16, dhid, dmix, dqkv = new(R, W, dtype=torch.float32), new(R, W), new(R, 4 * W), new(R, W), new(R, 3 * W)
    dxn = new(R, W, dtype=torch.float32)
    hip.check(hip.load().gava_convert_h16(hip.ptr(dX), hip.ptr(dx16), dX.numel(), BWD, hip.stream_ptr()), "convert")
    for i in reversed(range(sh["TL"])):
        P, X0 = bw["layers"][i], saved[i]
        # ---- recompute the block from its input (forward kernels, bf16 operands)
        hip.layernorm(X0, P["ln1_g"], P["ln1_b"], out16=xn, prec=BWD)
        hip.gemm(xn, P["w_qkv"], P["b_qkv"], qkv, epilogue=hip.EPI_H16, prec=BWD, scale_cols=W, scale=0.125)
        hip.attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], mix, batch=n, heads=H, n_q=L, n_kmain=L, prec=BWD, causal=True)
        hip.gemm(mix, P["w_out"], P["b_out"], X1, epilogue=hip.EPI_F32, prec=BWD, resid=X0)
        hip.layernorm(X1, P["ln2_g"], P["ln2_b"], out16=xn, prec=BWD)
        hip.gemm(xn, P["w_fc"], P["b_fc"], pre, epilogue=hip.EPI_H16, prec=BWD)
        # ---- MLP branch: x2 = x1 + c_proj(gelu(c_fc(ln_2 x1)))            (VitaCLIP_text_encoder.py:73-77,86)
        #      (dx16 = bf16 copy of dX, written by the LayerNorm' that produced dX)
        hip.gemm(dx16, P["w_proj_t"], None, dhid, epilogue=hip.EPI_H16_QGELU_BWD, prec=BWD, aux=pre)   # c_proj^T, gelu' fused
        hip.gemm(dhid, P["w_fc_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)
        hip.layernorm_backward(X1, P["ln2_g"], dxn, dX, accumulate=True, dx16=dx16)
        # ---- attention branch: x1 = x0 + out_proj(attn(in_proj(ln_1 x0)))     (VitaCLIP_text_encoder.py:81-85)
        hip.gemm(dx16, P["w_out_t"], None, dmix, epilogue=hip.EPI_H16, prec=BWD)
        hip.attention_backward(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dmix, dqkv[:, :W], dqkv[:, W:2 * W], dqkv[:, 2 * W:],
                               batch=n, heads=H, n=L, prec=BWD, causal=True, q_scale=0.125)
        hip.gemm(dqkv, P["w_qkv_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)
        hip.layernorm_backward(X0, P["ln1_g"], dxn, dX, accumulate=True, dx16=dx16)
    # x0 = [SOS | ctx[c] | suffix] + positional_embedding  (VitaCLIP_text_encoder.py:323-332,157): ctx rows 1..n_ctx
    return dX.view(n, L, W)[:, 1:1 + n_ctx].clone()


class TextTowerFn(torch.autograd.Function):
    """text features = f(ctx) with the HIP text tower in both directions."""

    @staticmethod
    def forward(fctx, model, ctx_param):
        out, saved = text_forward_train(model, ctx_param)
        fctx.model = model
        fctx.save_for_backward(saved)
        return out

    @staticmethod
    def backward(fctx, dtext):
        (saved,) = fctx.saved_tensors
        dctx = text_backward(fctx.model, saved, dtext)
        return None, dctx.to(dtext.dtype)


# =================================================================================================
# Vision tower (second stage): gradients of the Vita-CLIP prompt parameters through the frozen ViT
# =================================================================================================
# Trainable on the vision side (VitaCLIP_model.py:230-234: names containing summary / local / global / time_embed):
#   visual.global_prompts (layers,G,D), blocks.i.local_prompts (1,T,D), blocks.i.summary_ln.{weight,bias},
#   blocks.i.summary_attn_layer.{q,k,v,out}_proj.{weight,bias}, visual.time_embed (T,D).
# Chain (VitaCLIP_vision_encoder.py:102-132, VitaCLIP_vision_encoder_utils.py:155-203), in reverse:
#   d cls_x -> mean over T -> proj^T -> ln_post' (CLS rows) -> 12 x block' -> ln_pre' -> sum over tokens = d time_embed
#   block': MLP' and attention' on all B*T*197 rows (dgrad GEMMs + qgelu' + LayerNorm' + gava_attention_backward);
#   the attention' also yields the gradient of the shared prompt K/V rows -> K/V projection^T -> norm1' ->
#   d global_prompts, d local_prompts, and through the summary path (T-token attention, summary_ln, cls_proj) both the
#   parameter gradients of summary_ln / summary_attn_layer (wgrad = gava_gemm on transposed operands) and a
#   contribution to the CLS rows of dX.
# Activations are recomputed per block from its saved fp32 input (the forward kernels, bf16 operands).

def _vision_trainables(model):
    """Ordered (name, parameter) list of the vision-side parameters the reference leaves trainable."""
    return [(n, p) for n, p in model.visual.named_parameters()
            if ("summary" in n or "local" in n or "global" in n or "time_embed" in n)]


def pack_vision_backward(model):
    v = model.visual
    f32 = lambda p: p.detach().float().contiguous()
    layers = []
    for blk in v.blocks:
        a, s = blk.attn, blk.summary_attn_layer
        wqkv = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).detach()
        wsqkv = torch.cat([s.q_proj.weight, s.k_proj.weight, s.v_proj.weight], 0).detach()
        D = wqkv.shape[1]
        layers.append(dict(
            w_qkv=_bf16(wqkv), w_qkv_t=_bf16(wqkv.t()), b_qkv=f32(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)),
            w_kv=_bf16(wqkv[D:]), w_kv_t=_bf16(wqkv[D:].t()),
            w_out=_bf16(a.out_proj.weight), w_out_t=_bf16(a.out_proj.weight.detach().t()), b_out=f32(a.out_proj.bias),
            w_fc1=_bf16(blk.mlp.fc1.weight), w_fc1_t=_bf16(blk.mlp.fc1.weight.detach().t()), b_fc1=f32(blk.mlp.fc1.bias),
            w_fc2_t=_bf16(blk.mlp.fc2.weight.detach().t()),
            w_cls=_bf16(blk.cls_proj.weight), w_cls_t=_bf16(blk.cls_proj.weight.detach().t()), b_cls=f32(blk.cls_proj.bias),
            w_sqkv=_bf16(wsqkv), w_sqkv_t=_bf16(wsqkv.t()),
            w_sout=_bf16(s.out_proj.weight), w_sout_t=_bf16(s.out_proj.weight.detach().t())))
    return dict(layers=layers, proj=_bf16(v.proj),     # cls_x = ln_post(x) @ proj (D,E): dx = d @ proj^T = gemm(d, W=proj)
                summary_ver=model._summary_weight_versions())


def refresh_vision_backward(model, bw):
    """The summary-attention projections are the only trainable weights with bf16 copies: re-convert just those (in place)
    after an optimizer step."""
    cur = model._summary_weight_versions()
    if cur == bw["summary_ver"]:
        return
    for i, blk in enumerate(model.visual.blocks):
        if cur[i] != bw["summary_ver"][i]:
            s = blk.summary_attn_layer
            wsqkv = torch.cat([s.q_proj.weight, s.k_proj.weight, s.v_proj.weight], 0).detach()
            P = bw["layers"][i]
            P["w_sqkv"].copy_(_bf16(wsqkv)); P["w_sqkv_t"].copy_(_bf16(wsqkv.t()))
            P["w_sout"].copy_(_bf16(s.out_proj.weight)); P["w_sout_t"].copy_(_bf16(s.out_proj.weight.detach().t()))
    bw["summary_ver"] = cur


def _pad_k(t, mult=64):
    """[N][K] -> K padded with zero columns to a multiple of `mult` (GEMM reduction-dim granularity)."""
    k = t.shape[1]
    kp = (k + mult - 1) // mult * mult
    return t.contiguous() if kp == k else torch.nn.functional.pad(t, (0, kp - k)).contiguous()


def _wgrad(dy16, x16):
    """dW[o][i] = sum_m dy[m][o] * x[m][i] for an nn.Linear weight [out][in]: gava_gemm on the transposed operands."""
    A, W = _pad_k(dy16.t()), _pad_k(x16.t())
    out = torch.empty(A.shape[0], W.shape[0], dtype=torch.float32, device=dy16.device)
    hip.gemm(A, W, None, out, epilogue=hip.EPI_F32, prec=BWD)
    return out


def kept_bytes(model, B, T):
    sh = model._shape
    n1 = (sh["size"] // sh["P"]) ** 2 + 1
    R, D, F, NL = B * T * n1, sh["D"], sh["F"], sh["layers"]
    return R * D * 4 * (2 * NL + 1) + NL * R * 3 * D * 2 + (NL - 1) * R * F * 2 + NL * (sh["G"] + 2 * B * T) * 2 * D * 2


def alloc_kept(model, B, T, device):
    """Per-block activation buffers for gava_vision_forward_keep (include/gava_hip.h, gava_vision_saved)."""
    sh = model._shape
    n1 = (sh["size"] // sh["P"]) ** 2 + 1
    R, D, F, NL, SR = B * T * n1, sh["D"], sh["F"], sh["layers"], sh["G"] + 2 * B * T
    h16 = hip.h16_dtype(model.prec)
    e = lambda *s, dtype=torch.float32: torch.empty(*s, dtype=dtype, device=device)
    BT = B * T
    return dict(e0=e(R, D), x=e(NL + 1, R, D), x1=e(max(NL - 1, 1), R, D), qkv=e(NL, R, 3 * D, dtype=h16),
                pre=e(max(NL - 1, 1), R, F, dtype=h16), sidekv=e(NL, SR, 2 * D, dtype=h16),
                # the last block runs on the CLS rows only (as in inference): its CLS queries / stream / pre-activations
                last_q=e(BT, D, dtype=h16), last_x1=e(BT, D), last_pre=e(BT, F, dtype=h16))


def vision_backward(model, saved, dcls_x, B, T, dsummary=None, kept=None):
    """d cls_x (B,E) [+ d summary (B,D), the auxiliary NTE head's input] -> {parameter name: gradient} for the trainable
    vision parameters.  `kept` (alloc_kept, filled by the forward) replaces the per-block recomputation of the main rows:
    activations are then in the forward's operand type (fp16 by default) while gradients stay bf16 - the attention
    backward converts K/V/Q as it stages them, the QuickGELU' epilogue decodes the pre-activation by its own flag."""
    sh = model._shape
    bw = model._pack_vision_backward()
    D, H, F, E, G, NL = sh["D"], sh["H"], sh["F"], sh["E"], sh["G"], sh["layers"]
    n1 = (sh["size"] // sh["P"]) ** 2 + 1
    # T_in frames per input clip; the blocks regroup the B*T_in frames by the MODEL's num_frames regardless
    # (vision_encoder_utils.py:160-162): inside the block loop (B, T) are (groups, num_frames), the head and the temporal
    # embedding keep the input's (B_in, T_in)
    B_in, T_in = B, T
    T = model.num_frames
    B = B_in * T_in // T
    BT, R, SR = B * T, B * T * n1, G + 2 * B * T
    dev = dcls_x.device
    bf = torch.bfloat16
    new = lambda *s, dtype=bf: torch.empty(*s, dtype=dtype, device=dev)
    conv = lambda src, dst: hip.check(hip.load().gava_convert_h16(hip.ptr(src), hip.ptr(dst), src.numel(), BWD, hip.stream_ptr()), "convert")
    v = model.visual
    grads = {}
    cls_idx = (torch.arange(BT, device=dev, dtype=torch.int32) * n1).contiguous()

    # ---- head: cls_x = mean_t(ln_post(x_cls) @ proj)                          (VitaCLIP_vision_encoder.py:126-128)
    dproj = (dcls_x.float() / T_in).unsqueeze(1).expand(B_in, T_in, E).reshape(BT, E).contiguous()
    dclspost = new(BT, D, dtype=torch.float32)
    hip.gemm(hip.convert_h16(dproj, BWD), bw["proj"], None, dclspost, epilogue=hip.EPI_F32, prec=BWD)
    dX = torch.zeros(R, D, dtype=torch.float32, device=dev)
    x_final = kept["x"][NL] if kept is not None else saved[NL + 1]
    hip.layernorm_backward(x_final, v.ln_post.weight.detach().float().contiguous(), dclspost, dX,
                           x_row_index=cls_idx, dx_row_index=cls_idx, rows=BT)

    if kept is None:
        xn, qkv_buf, mix, pre_buf, X1_buf = new(R, D), new(R, 3 * D), new(R, D), new(R, F), new(R, D, dtype=torch.float32)
    dx16, dhid, dmix, dqkv = new(R, D), new(R, F), new(R, D), new(R, 3 * D)
    dxn = new(R, D, dtype=torch.float32)
    conv(dX, dx16)
    dgp = torch.zeros_like(v.global_prompts, dtype=torch.float32)
    for i in reversed(range(NL)):
        P, blk = bw["layers"][i], v.blocks[i]
        X0 = kept["x"][i] if kept is not None else saved[1 + i]
        ACT = model.prec if kept is not None else BWD          # storage type of the activations used below
        f32 = lambda p: p.detach().float().contiguous()
        ln1_g, ln1_b, ln2_g, ln2_b = f32(blk.norm1.weight), f32(blk.norm1.bias), f32(blk.norm2.weight), f32(blk.norm2.bias)
        sln_g, sln_b = f32(blk.summary_ln.weight), f32(blk.summary_ln.bias)
        s = blk.summary_attn_layer
        b_sqkv = f32(torch.cat([s.q_proj.bias, s.k_proj.bias, s.v_proj.bias], 0))
        # ---- recompute: prompt ("side") path                                 (vision_encoder_utils.py:164-190)
        cls16 = hip.convert_h16(X0.view(BT, n1, D)[:, 0].contiguous(), BWD)
        CP = new(BT, D, dtype=torch.float32)
        hip.gemm(cls16, P["w_cls"], P["b_cls"], CP, epilogue=hip.EPI_F32, prec=BWD)
        CPn = new(BT, D)
        hip.layernorm(CP, sln_g, sln_b, out16=CPn, prec=BWD)
        SQKV = new(BT, 3 * D)
        hip.gemm(CPn, P["w_sqkv"], b_sqkv, SQKV, epilogue=hip.EPI_H16, prec=BWD, scale_cols=D, scale=0.125)
        SMIX = new(BT, D)
        hip.attention(SQKV[:, :D], SQKV[:, D:2 * D], SQKV[:, 2 * D:], SMIX, batch=BT // T, heads=H, n_q=T, n_kmain=T, prec=BWD)
        SUMM = new(BT, D, dtype=torch.float32)
        hip.gemm(SMIX, P["w_sout"], f32(s.out_proj.bias), SUMM, epilogue=hip.EPI_F32, prec=BWD, resid=CP)
        lp = blk.local_prompts.detach().float()[0]                                     # (T, D)
        SIDE = torch.cat([v.global_prompts.detach().float()[i], (CP.view(B, T, D) + lp).view(BT, D), SUMM], 0).contiguous()
        cls_only = kept is not None and i == NL - 1 and "last_q" in kept
        if cls_only:
            SIDEKV, qkv = kept["sidekv"][i], kept["qkv"][i]
        elif kept is not None:
            SIDEKV, qkv, X1, pre = kept["sidekv"][i], kept["qkv"][i], kept["x1"][i], kept["pre"][i]
        else:
            SIDEn = new(SR, D)
            hip.layernorm(SIDE, ln1_g, ln1_b, out16=SIDEn, prec=BWD)
            SIDEKV = new(SR, 2 * D)
            hip.gemm(SIDEn, P["w_kv"], P["b_qkv"][D:], SIDEKV, epilogue=hip.EPI_H16, prec=BWD)
            # ---- recompute: main rows
            qkv, X1, pre = qkv_buf, X1_buf, pre_buf
            hip.layernorm(X0, ln1_g, ln1_b, out16=xn, prec=BWD)
            hip.gemm(xn, P["w_qkv"], P["b_qkv"], qkv, epilogue=hip.EPI_H16, prec=BWD, scale_cols=D, scale=0.125)
            hip.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], mix, batch=BT, heads=H, n_q=n1, n_kmain=n1, prec=BWD,
                          side_k=SIDEKV[:, :D], side_v=SIDEKV[:, D:], n_g=G, T=T, has_summary=True)
            hip.gemm(mix, P["w_out"], P["b_out"], X1, epilogue=hip.EPI_F32, prec=BWD, resid=X0)
            hip.layernorm(X1, ln2_g, ln2_b, out16=xn, prec=BWD)
            hip.gemm(xn, P["w_fc1"], P["b_fc1"], pre, epilogue=hip.EPI_H16, prec=BWD)
        part = new(BT, G + T + 1, 2 * D, dtype=torch.float32)     # per-frame partials of the shared prompt rows
        dside = part.view(BT * (G + T + 1), 2 * D)
        if cls_only:
            # last block: only the CLS rows carry a gradient (VitaCLIP_vision_encoder.py:126) - MLP', out_proj' and the
            # query side of attention' on B*T rows; keys / values (and through them every row of the block input) in full
            dXc = dX.view(BT, n1, D)[:, 0].contiguous()
            dxc16 = hip.convert_h16(dXc, BWD)
            dhid_c = new(BT, F)
            hip.gemm(dxc16, P["w_fc2_t"], None, dhid_c, epilogue=hip.EPI_H16_QGELU_BWD, prec=BWD, aux=kept["last_pre"], aux_prec=ACT)
            dxn_c = new(BT, D, dtype=torch.float32)
            hip.gemm(dhid_c, P["w_fc1_t"], None, dxn_c, epilogue=hip.EPI_F32, prec=BWD)
            hip.layernorm_backward(kept["last_x1"], ln2_g, dxn_c, dXc, accumulate=True, dx16=dxc16)
            dmix_c, dq_c = new(BT, D), new(BT, D)
            hip.gemm(dxc16, P["w_out_t"], None, dmix_c, epilogue=hip.EPI_H16, prec=BWD)
            hip.attention_backward(kept["last_q"], qkv[:, D:2 * D], qkv[:, 2 * D:], dmix_c, dq_c, dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                                   batch=BT, heads=H, n=n1, prec=BWD, q_scale=0.125,
                                   side_k=SIDEKV[:, :D], side_v=SIDEKV[:, D:], dside_k=dside[:, :D], dside_v=dside[:, D:],
                                   n_g=G, T=T, has_summary=True, act_prec=ACT, n_q=1, q_batch_rows=1)
            hip.gemm(dqkv[:, D:], P["w_kv_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)          # [dK dV] . [Wk; Wv]
            dxn_cls = dxn.view(BT, n1 * D)[:, :D]                                                     # CLS rows, stride n1*D
            hip.gemm(dq_c, P["w_qkv_t"][:, :D], None, dxn_cls, epilogue=hip.EPI_F32, prec=BWD, resid=dxn_cls)   # + dQ . Wq
            dX.view(BT, n1, D)[:, 0] = dXc
        else:
            # ---- MLP'                                                       (vision_encoder_utils.py:109-115,199)
            #      (dx16 = bf16 copy of dX, written by the LayerNorm' that produced dX)
            hip.gemm(dx16, P["w_fc2_t"], None, dhid, epilogue=hip.EPI_H16_QGELU_BWD, prec=BWD, aux=pre, aux_prec=ACT)   # fc2^T, gelu' fused
            hip.gemm(dhid, P["w_fc1_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)
            hip.layernorm_backward(X1, ln2_g, dxn, dX, accumulate=True, dx16=dx16)
            # ---- attention'                                                 (vision_encoder_utils.py:61-81,190-191)
            hip.gemm(dx16, P["w_out_t"], None, dmix, epilogue=hip.EPI_H16, prec=BWD)
            hip.attention_backward(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], dmix, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                                   batch=BT, heads=H, n=n1, prec=BWD, q_scale=0.125,
                                   side_k=SIDEKV[:, :D], side_v=SIDEKV[:, D:], dside_k=dside[:, :D], dside_v=dside[:, D:],
                                   n_g=G, T=T, has_summary=True, act_prec=ACT)
            hip.gemm(dqkv, P["w_qkv_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)
        # (norm1' of the main rows is applied at the end of the block, after the prompt path has added its share to
        #  the CLS rows of dX: it also writes the bf16 copy of the finished dX for the next block)
        # ---- prompt rows': sum the partials over the frames that share a row (global: all; local: the T frames of the
        #      clip; summary: its own frame), then K/V projection^T, norm1', split into global / local / summary
        pv = part.view(B, T, G + T + 1, 2 * D)
        dsidekv = torch.cat([pv[:, :, :G].sum(dim=(0, 1)), pv[:, :, G:G + T].sum(dim=1).reshape(BT, 2 * D),
                             pv[:, :, G + T].reshape(BT, 2 * D)], 0).contiguous()
        dSIDEn = new(SR, D, dtype=torch.float32)
        hip.gemm(hip.convert_h16(dsidekv, BWD), P["w_kv_t"], None, dSIDEn, epilogue=hip.EPI_F32, prec=BWD)
        dSIDE = new(SR, D, dtype=torch.float32)
        hip.layernorm_backward(SIDE, ln1_g, dSIDEn, dSIDE)
        dgp[i] = dSIDE[:G]
        dlocal, dSUMM = dSIDE[G:G + BT], dSIDE[G + BT:]
        if dsummary is not None and i == NL - 1:
            # summary = mean over T of the last block's summary tokens (VitaCLIP_vision_encoder.py:129-130)
            dSUMM = dSUMM + (dsummary.float() / T).repeat_interleave(T, dim=0)
        grads[f"blocks.{i}.local_prompts"] = dlocal.view(B, T, D).sum(0).unsqueeze(0)
        dCP = (dlocal + dSUMM).contiguous()                      # local = lp + CP;  SUMM = CP + out_proj(...)
        # ---- summary attention' (T tokens per clip) with parameter gradients
        dSUMM16 = hip.convert_h16(dSUMM.contiguous(), BWD)
        dSMIX = new(BT, D)
        hip.gemm(dSUMM16, P["w_sout_t"], None, dSMIX, epilogue=hip.EPI_H16, prec=BWD)
        grads[f"blocks.{i}.summary_attn_layer.out_proj.weight"] = _wgrad(dSUMM16, SMIX)
        grads[f"blocks.{i}.summary_attn_layer.out_proj.bias"] = dSUMM.sum(0)
        dSQKV = new(BT, 3 * D)
        hip.attention_backward(SQKV[:, :D], SQKV[:, D:2 * D], SQKV[:, 2 * D:], dSMIX, dSQKV[:, :D], dSQKV[:, D:2 * D], dSQKV[:, 2 * D:],
                               batch=BT // T, heads=H, n=T, prec=BWD, q_scale=0.125)
        dWs = _wgrad(dSQKV, CPn)
        dbs = dSQKV.float().sum(0)
        for k, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            grads[f"blocks.{i}.summary_attn_layer.{nm}.weight"] = dWs[k * D:(k + 1) * D]
            grads[f"blocks.{i}.summary_attn_layer.{nm}.bias"] = dbs[k * D:(k + 1) * D]
        dCPn = new(BT, D, dtype=torch.float32)
        hip.gemm(dSQKV, P["w_sqkv_t"], None, dCPn, epilogue=hip.EPI_F32, prec=BWD)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        hip.layernorm_backward(CP, sln_g, dCPn, dCP, accumulate=True, dgamma=dg, dbeta=db)
        grads[f"blocks.{i}.summary_ln.weight"], grads[f"blocks.{i}.summary_ln.bias"] = dg, db
        # ---- cls_proj' (frozen weight): back onto the CLS rows of the block input
        dCLS = new(BT, D, dtype=torch.float32)
        hip.gemm(hip.convert_h16(dCP, BWD), P["w_cls_t"], None, dCLS, epilogue=hip.EPI_F32, prec=BWD)
        dX.view(BT, n1, D)[:, 0] += dCLS
        hip.layernorm_backward(X0, ln1_g, dxn, dX, accumulate=True, dx16=dx16)
    grads["global_prompts"] = dgp
    # ---- ln_pre' and the temporal embedding (VitaCLIP_vision_encoder.py:86-100,108-113): time_embed[t] is added to
    #      every token of frame t
    hip.layernorm_backward(kept["e0"] if kept is not None else saved[0], v.ln_pre.weight.detach().float().contiguous(), dX, dX)
    dte = dX.view(B_in, T_in, n1, D).sum(dim=(0, 2))
    if T_in != T:   # nearest-resized time_embed (VitaCLIP_vision_encoder.py:91-95): row t of the resized table is row floor(t*T/T_in)
        src = (torch.arange(T_in, device=dev) * T) // T_in
        dte = torch.zeros(T, D, dtype=dte.dtype, device=dev).index_add_(0, src, dte)
    grads["time_embed"] = dte
    return grads


class VisionTowerFn(torch.autograd.Function):
    """cls_x = f(x; prompt parameters) with the HIP vision tower in both directions.  `params` are the trainable vision
    parameters in `_vision_trainables` order (they enter only so that autograd routes their gradients)."""

    @staticmethod
    def forward(fctx, model, x, *params):
        sh = model._shape
        B, _, T = x.shape[:3]
        n1 = (sh["size"] // sh["P"]) ** 2 + 1
        fctx.model, fctx.BT = model, (B, T)
        if kept_bytes(model, B, T) <= model.keep_activation_bytes:
            fctx.kept = alloc_kept(model, B, T, x.device)
            cls_x, summary = model.encode_video(x, kept=fctx.kept)
            fctx.save_for_backward()
        else:
            fctx.kept = None
            saved = torch.empty(sh["layers"] + 2, B * T * n1, sh["D"], dtype=torch.float32, device=x.device)
            cls_x, summary = model.encode_video(x, saved=saved)
            fctx.save_for_backward(saved)
        return cls_x, summary

    @staticmethod
    def backward(fctx, dcls_x, dsummary):
        model, kept = fctx.model, fctx.kept
        saved = fctx.saved_tensors[0] if kept is None else None
        dev = (kept["x"] if kept is not None else saved).device
        if dcls_x is None:
            dcls_x = torch.zeros(fctx.BT[0], model._shape["E"], device=dev)
        g = vision_backward(model, saved, dcls_x.contiguous(), *fctx.BT, dsummary=dsummary, kept=kept)
        fctx.kept = None      # release the activation buffers with the graph
        out = []
        for name, p in _vision_trainables(model):
            gi = g.get(name)
            out.append(gi.reshape(p.shape).to(p.dtype) if (gi is not None and p.requires_grad) else None)
        return (None, None, *out)
