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

The vision tower's backward (attention over 214 keys with shared prompt rows, MFMA work) is the next stage; until
then video features enter the loss as constants, which is exactly CoOp text-prompt tuning on frozen video features.
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
    m.n_prompts, m.L, m.W, m.H, m.layers = n, sh["L"], sh["W"], sh["TH"], sh["TL"]
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
    saved = torch.empty(sh["TL"] + 1, n * sh["L"], sh["W"], dtype=torch.float32, device=tok.device)
    hip.check(lib.gava_text_forward_train(C.byref(m), hip.ptr(tok), hip.ptr(ctx), hip.ptr(pk["eot"]), hip.ptr(out),
                                          hip.ptr(saved), hip.ptr(ws), ws.numel(), hip.stream_ptr()),
              "gava_text_forward_train")
    return out, saved


def text_backward(model, saved, dtext):
    """d(text features) (C,E) -> d ctx (C, n_ctx, W), through the frozen text tower."""
    sh = model._shape
    pk = model._pack()
    bw = model._pack_text_backward()
    n, L, W, H, E, n_ctx = pk["tokens"].shape[0], sh["L"], sh["W"], sh["TH"], sh["E"], sh["n_ctx"]
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
        hip.load().gava_convert_h16(hip.ptr(dX), hip.ptr(dx16), dX.numel(), BWD, hip.stream_ptr())
        hip.gemm(dx16, P["w_proj_t"], None, dhid, epilogue=hip.EPI_H16, prec=BWD)
        hip.qgelu_backward(pre, dhid, dhid, BWD)
        hip.gemm(dhid, P["w_fc_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)
        hip.layernorm_backward(X1, P["ln2_g"], dxn, dX, accumulate=True)
        # ---- attention branch: x1 = x0 + out_proj(attn(in_proj(ln_1 x0)))     (VitaCLIP_text_encoder.py:81-85)
        hip.load().gava_convert_h16(hip.ptr(dX), hip.ptr(dx16), dX.numel(), BWD, hip.stream_ptr())
        hip.gemm(dx16, P["w_out_t"], None, dmix, epilogue=hip.EPI_H16, prec=BWD)
        hip.attention_backward(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dmix, dqkv[:, :W], dqkv[:, W:2 * W], dqkv[:, 2 * W:],
                               batch=n, heads=H, n=L, prec=BWD, causal=True, q_scale=0.125)
        hip.gemm(dqkv, P["w_qkv_t"], None, dxn, epilogue=hip.EPI_F32, prec=BWD)
        hip.layernorm_backward(X0, P["ln1_g"], dxn, dX, accumulate=True)
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
