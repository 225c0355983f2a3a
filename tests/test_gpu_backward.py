"""GPU: backward kernels and the text-side training path against torch autograd / the oracle.

Tolerances: LayerNorm backward is fp32 arithmetic -> 1e-5 of the gradient's range.  QuickGELU / attention
backward take bf16 operands (8-bit mantissa): inputs are rounded to bf16 first, so what is left is the output
rounding, 2^-8 relative.  The end-to-end gradient of the context vectors goes through 12 blocks with bf16
operands on our side and fp32 in the oracle: norm-wise 3e-2."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from gava_clip_amd import VitaCLIP, hip, synth  # noqa: E402
from gava_clip_amd.config import TINY, VIT_B16_T8  # noqa: E402
from oracle.vita_oracle import Oracle  # noqa: E402  (checker only)
from helpers import CLASSES_3, model_kwargs, synth_torch_state  # noqa: E402

BF = hip.PREC_BF16


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("rows,D", [(9, 128), (231, 512), (40, 768), (5, 1024)])
def test_layernorm_backward_matches_autograd(rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.3).requires_grad_()
    gamma = (torch.rand(D, generator=g) + 0.5).requires_grad_()
    beta = torch.randn(D, generator=g).requires_grad_()
    dy = torch.randn(rows, D, generator=g)
    torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5).backward(dy)
    xd, dyd = x.detach().cuda(), dy.cuda()
    base = torch.randn(rows, D, generator=g).cuda()
    dx = base.clone()
    dgm, dbt = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    hip.layernorm_backward(xd, gamma.detach().cuda(), dyd, dx, accumulate=True, dgamma=dgm, dbeta=dbt)
    assert (dx.cpu() - base.cpu() - x.grad).abs().max() <= 1e-5 * x.grad.abs().max() + 1e-6
    assert (dgm.cpu() - gamma.grad).abs().max() <= 1e-4 * gamma.grad.abs().max()
    assert (dbt.cpu() - beta.grad).abs().max() <= 1e-4 * beta.grad.abs().max()


def test_layernorm_backward_gather_scatter_rows():
    g = torch.Generator().manual_seed(3)
    X = torch.randn(50, 256, generator=g)
    idx = torch.tensor([7, 30, 49], dtype=torch.int32)
    gamma = torch.rand(256, generator=g) + 0.5
    dy = torch.randn(3, 256, generator=g)
    xs = X[idx.long()].clone().requires_grad_()
    torch.nn.functional.layer_norm(xs, (256,), gamma, torch.zeros(256), 1e-5).backward(dy)
    dX = torch.zeros(50, 256, device="cuda")
    hip.layernorm_backward(X.cuda(), gamma.cuda(), dy.cuda(), dX, x_row_index=idx.cuda(), dx_row_index=idx.cuda(), rows=3)
    ref = torch.zeros(50, 256)
    ref[idx.long()] = xs.grad
    assert (dX.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_qgelu_backward_matches_autograd():
    g = torch.Generator().manual_seed(5)
    pre = (torch.randn(231, 2048, generator=g) * 2).bfloat16()
    dh = torch.randn(231, 2048, generator=g).bfloat16()
    x = pre.float().requires_grad_()
    (x * torch.sigmoid(1.702 * x)).backward(dh.float())
    out = torch.empty_like(pre, device="cuda")
    hip.qgelu_backward(pre.cuda(), dh.cuda(), out, BF)
    assert (out.float().cpu() - x.grad).abs().max() <= 2 ** -8 * x.grad.abs().max() * 1.01


@pytest.mark.parametrize("batch,heads,n,causal", [(3, 8, 77, True), (2, 2, 77, False), (4, 3, 8, True), (1, 1, 88, True), (5, 2, 1, True)])
def test_attention_backward_small_matches_autograd(batch, heads, n, causal):
    g = torch.Generator().manual_seed(n * 7 + heads)
    W = heads * 64
    qkv = torch.randn(batch * n, 3 * W, generator=g).bfloat16()
    do = torch.randn(batch * n, W, generator=g).bfloat16()
    q, k, v = [t.float().view(batch, n, heads, 64).transpose(1, 2).requires_grad_() for t in qkv.split(W, dim=1)]
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((n, n), float("-inf")).triu_(1)
    o = s.softmax(-1) @ v
    o.backward(do.float().view(batch, n, heads, 64).transpose(1, 2))
    # the kernel takes q already scaled (as the forward's QKV GEMM writes it) and folds the scale back into dq
    qkv_s = qkv.float().clone()
    qkv_s[:, :W] *= 0.125          # exact in bf16 (power of two)
    qd = qkv_s.bfloat16().cuda()
    dqkv = torch.empty(batch * n, 3 * W, dtype=torch.bfloat16, device="cuda")
    hip.attention_backward(qd[:, :W], qd[:, W:2 * W], qd[:, 2 * W:], do.cuda(), dqkv[:, :W], dqkv[:, W:2 * W], dqkv[:, 2 * W:],
                           batch=batch, heads=heads, n=n, prec=BF, causal=causal, q_scale=0.125)
    for name, grad, sl in (("dq", q.grad, slice(0, W)), ("dk", k.grad, slice(W, 2 * W)), ("dv", v.grad, slice(2 * W, 3 * W))):
        ref = grad.transpose(1, 2).reshape(batch * n, W)
        got = dqkv[:, sl].float().cpu()
        assert (got - ref).abs().max() <= 2 ** -8 * ref.abs().max() * 1.05 + 1e-6, name


def _oracle_text_grads(cfg, sd, tokens, x, wlog):
    p = {k: v.clone().float() for k, v in sd.items()}
    p["prompt_learner.ctx"].requires_grad_()
    p["logit_scale"].requires_grad_()
    o = Oracle(cfg, p, tokens)
    with torch.no_grad():
        vf, _ = o.vision(x.float())
        vf = vf / vf.norm(dim=-1, keepdim=True)
    tf = o.text(o.prompts())
    tf = tf / tf.norm(dim=-1, keepdim=True)
    logits = p["logit_scale"].exp() * vf @ tf.t()
    (logits * wlog).sum().backward()
    return logits.detach(), p["prompt_learner.ctx"].grad, p["logit_scale"].grad


@pytest.mark.parametrize("cfg,B", [(TINY, 2), (VIT_B16_T8, 1)])
def test_text_prompt_gradients_match_oracle_autograd(cfg, B):
    """loss.backward() through VitaCLIP.forward: d ctx and d logit_scale vs the fp32 oracle under torch autograd."""
    sd = synth_torch_state(cfg, 3)
    m = VitaCLIP(**model_kwargs(cfg, CLASSES_3))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size))
    wlog = torch.randn(B, 3, generator=torch.Generator().manual_seed(11))
    tokens = torch.cat(m.tokenized_prompts).cpu()
    ref_logits, ref_dctx, ref_dls = _oracle_text_grads(cfg, sd, tokens, x, wlog)
    logits, lmt, lvm = m(x.cuda())
    assert logits.requires_grad and lmt is None and lvm is None
    (logits * wlog.cuda()).sum().backward()
    assert (logits.detach().cpu() - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    g = m.prompt_learner.ctx.grad
    assert g is not None and g.shape == ref_dctx.shape and bool(torch.isfinite(g).all())
    assert rel(g.cpu(), ref_dctx) <= 3e-2, rel(g.cpu(), ref_dctx)
    assert abs(float(m.logit_scale.grad) - float(ref_dls)) <= 2e-3 * abs(float(ref_dls)) + 1e-6
    # frozen parameters stay without gradient, as in the reference (VitaCLIP_model.py:230-239)
    assert m.textual.ln_final.weight.grad is None
    # an optimizer step on ctx must not trigger a re-pack of the frozen weights and must change the output
    key = m._pack_key()
    with torch.no_grad():
        m.prompt_learner.ctx -= 0.1 * g
    assert m._pack_key() == key
    with torch.no_grad():
        l2 = m(x.cuda())[0]
    assert not torch.equal(l2, logits.detach())


@pytest.mark.parametrize("BT,T,heads,n,G,n_q", [(16, 8, 12, 197, 8, 0), (8, 4, 2, 17, 4, 0), (4, 2, 3, 33, 0, 0), (8, 8, 2, 50, 8, 1),
                                                 (32, 32, 2, 257, 8, 0),     # ViT-L/14, T = 32: 298 keys, 17 query tiles
                                                 (16, 16, 2, 197, 8, 0)])    # T = 16: 222 keys
def test_attention_backward_vision_with_prompt_rows_matches_autograd(BT, T, heads, n, G, n_q):
    """Vision block attention (vision_encoder_utils.py:176-191): keys = the frame's n rows + G global prompt rows +
    the T local-prompt rows of its clip + its summary row.  dq/dk/dv of the frame rows and the ACCUMULATED gradients of
    the shared prompt rows against torch autograd on the same bf16-rounded operands."""
    g = torch.Generator().manual_seed(BT * 31 + n)
    D = heads * 64
    qkv = torch.randn(BT * n, 3 * D, generator=g).bfloat16()
    side = torch.randn(G + 2 * BT, 2 * D, generator=g).bfloat16()
    nq = n_q or n
    do = torch.randn(BT * n, D, generator=g).bfloat16()
    q32 = qkv[:, :D].float().requires_grad_()
    k32 = qkv[:, D:2 * D].float().requires_grad_()
    v32 = qkv[:, 2 * D:].float().requires_grad_()
    s32 = side.float().requires_grad_()
    outs = []
    for f in range(BT):
        rows = slice(f * n, (f + 1) * n)
        clip = f // T
        srow = torch.cat([torch.arange(G), G + clip * T + torch.arange(T), torch.tensor([G + BT + f])])
        K = torch.cat([k32[rows], s32[srow, :D]]).view(-1, heads, 64).transpose(0, 1)
        V = torch.cat([v32[rows], s32[srow, D:]]).view(-1, heads, 64).transpose(0, 1)
        Q = q32[rows][:nq].view(nq, heads, 64).transpose(0, 1) * 0.125
        o = (Q @ K.transpose(-1, -2)).softmax(-1) @ V
        outs.append(o.transpose(0, 1).reshape(nq, D))
    dout = do.float().view(BT, n, D)[:, :nq]
    (torch.stack(outs) * dout).sum().backward()
    qs = qkv.float().clone(); qs[:, :D] *= 0.125
    qd, sd_ = qs.bfloat16().cuda(), side.cuda()
    act = None
    if BT == 16 and T == 8:
        # activations kept from an fp16 forward, gradients in bf16: the values are bf16-exact, so storing them as fp16 is
        # lossless here and the result must be the same as in the all-bf16 case
        qd, sd_, act = qd.half(), sd_.half(), hip.PREC_F16
    dqkv = torch.zeros(BT * n, 3 * D, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(BT, G + T + 1, 2 * D, dtype=torch.float32, device="cuda")
    dside = part.view(-1, 2 * D)
    q_arg, do_arg, dq_arg, qbr = qd[:, :D], do.cuda(), dqkv[:, :D], 0
    if n_q == 1:
        # the CLS-only last block: queries, dout and dq live in their own [BT][D] buffers (one row per frame)
        q_arg = qd.view(BT, n, 3 * D)[:, 0, :D].contiguous()
        do_arg = do.cuda().view(BT, n, D)[:, 0].contiguous()
        dq_sep = torch.zeros(BT, D, dtype=torch.bfloat16, device="cuda")
        dq_arg, qbr = dq_sep, 1
    hip.attention_backward(q_arg, qd[:, D:2 * D], qd[:, 2 * D:], do_arg, dq_arg, dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                           batch=BT, heads=heads, n=n, prec=BF, q_scale=0.125, q_batch_rows=qbr,
                           side_k=sd_[:, :D], side_v=sd_[:, D:], dside_k=dside[:, :D], dside_v=dside[:, D:],
                           n_g=G, T=T, has_summary=True, n_q=n_q, act_prec=act)
    tol = 3 * 2 ** -8     # P and dS are rounded to bf16 between the two MFMA products (as P is in the forward kernel)
    ref_q = q32.grad.view(BT, n, D)[:, :nq]
    got_q = dq_sep.float().cpu().view(BT, 1, D) if n_q == 1 else dqkv[:, :D].float().cpu().view(BT, n, D)[:, :nq]
    assert (got_q - ref_q).abs().max() <= tol * ref_q.abs().max() + 1e-6
    for name, got, ref in (("dk", dqkv[:, D:2 * D].float().cpu(), k32.grad), ("dv", dqkv[:, 2 * D:].float().cpu(), v32.grad)):
        assert (got - ref).abs().max() <= tol * ref.abs().max() + 1e-6, name
    # prompt rows: per-frame fp32 partials, summed here over the frames that share a row
    pv = part.cpu().view(BT // T, T, G + T + 1, 2 * D)
    total = torch.cat([pv[:, :, :G].sum(dim=(0, 1)), pv[:, :, G:G + T].sum(dim=1).reshape(BT, 2 * D), pv[:, :, G + T].reshape(BT, 2 * D)])
    assert (total - s32.grad).abs().max() <= tol * s32.grad.abs().max(), "dside"


def _oracle_all_grads(cfg, sd, tokens, x, wlog):
    """fp32 oracle under torch autograd: gradients of every parameter the reference trains (VitaCLIP_model.py:230-239)."""
    p = {k: v.clone().float() for k, v in sd.items()}
    names = [k for k in p if k == "logit_scale" or k == "prompt_learner.ctx" or
             (k.startswith("visual.") and any(t in k for t in ("summary", "local", "global", "time_embed")))]
    for k in names:
        p[k].requires_grad_()
    o = Oracle(cfg, p, tokens)
    vf, _ = o.vision(x.float())
    vf = vf / vf.norm(dim=-1, keepdim=True)
    tf = o.text(o.prompts())
    tf = tf / tf.norm(dim=-1, keepdim=True)
    logits = p["logit_scale"].exp() * vf @ tf.t()
    (logits * wlog).sum().backward()
    return logits.detach(), {k: p[k].grad for k in names}


@pytest.mark.parametrize("cfg,B", [(TINY, 2), (VIT_B16_T8, 1)])
def test_all_trainable_gradients_match_oracle_autograd(cfg, B):
    """loss.backward() through the whole drop-in model in train mode (training/train.py:441-490): every parameter the
    reference trains gets the oracle's gradient (norm-wise, bf16 backward operands vs fp32 autograd), frozen ones none."""
    sd = synth_torch_state(cfg, 3)
    m = VitaCLIP(**model_kwargs(cfg, CLASSES_3))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size))
    wlog = torch.randn(B, 3, generator=torch.Generator().manual_seed(5))
    tokens = torch.cat(m.tokenized_prompts).cpu()
    ref_logits, ref = _oracle_all_grads(cfg, sd, tokens, x, wlog)
    logits = m(x.cuda())[0]
    (logits * wlog.cuda()).sum().backward()
    assert (logits.detach().cpu() - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    got = {n: p.grad for n, p in m.named_parameters()}
    worst = {}
    for name, g_ref in ref.items():
        g = got[name]
        assert g is not None, name
        assert g.shape == g_ref.shape and bool(torch.isfinite(g).all()), name
        if name.endswith("k_proj.bias"):
            # softmax is invariant to a constant added to every key: this gradient is exactly zero in exact arithmetic
            # (the oracle returns ~1e-9 noise), so it is measured against the scale of its sibling, d q_proj.bias
            scale = ref[name.replace("k_proj", "q_proj")].norm()
            worst[name] = float(g.cpu().norm() / scale) * 0.1
            assert float(g_ref.norm()) <= 1e-4 * float(scale)
            continue
        worst[name] = rel(g.cpu(), g_ref)
    bad = {k: v for k, v in worst.items() if v > 4e-2}
    assert not bad, bad
    for n, p in m.named_parameters():
        if n not in ref:
            assert p.grad is None, n


# ---- against the REFERENCE's own gradients (tools/gen_golden.py runs training/VitaCLIP_model.py under torch autograd
#      in the build container and commits them as fixtures) -----------------------------------------------------------

def _grad_sample(g, n=2048):
    flat = g.reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n]


def _check_against_reference_grads(m, gold, sampled=False):
    got = {n: p.grad for n, p in m.named_parameters()}
    names = [k.split(".", 1)[1] for k in gold.files if k.startswith("gradsub." if sampled else "grad.")]
    assert len(names) > 20
    worst = {}
    for name in names:
        g = got[name]
        assert g is not None, name
        if sampled:
            ref_n, ref_s = float(gold["gradnorm." + name]), torch.from_numpy(gold["gradsub." + name])
            gs = _grad_sample(g.float().cpu())
            scale = ref_n if ref_n > 0 else 1.0
            if name.endswith("k_proj.bias"):      # exactly zero in exact arithmetic (see the oracle test above)
                q_n = float(gold["gradnorm." + name.replace("k_proj", "q_proj")])
                assert float(g.norm()) <= 0.2 * q_n and ref_n <= 1e-4 * q_n
                continue
            worst[name] = max(abs(float(g.float().norm()) - ref_n) / scale,
                              float((gs - ref_s).norm() / (ref_s.norm() + 1e-30)))
        else:
            ref = torch.from_numpy(gold["grad." + name])
            assert g.shape == ref.shape, name
            if name.endswith("k_proj.bias") and "summary" in name:
                q_n = torch.from_numpy(gold["grad." + name.replace("k_proj", "q_proj")]).norm()
                assert float(g.norm()) <= 0.2 * float(q_n) and float(ref.norm()) <= 1e-4 * float(q_n)
                continue
            worst[name] = rel(g.cpu(), ref)
    bad = {k: v for k, v in worst.items() if v > 4e-2}
    assert not bad, bad
    for n, p in m.named_parameters():
        if n not in names and not n.endswith("k_proj.bias"):
            assert p.grad is None, n
    return worst


@pytest.mark.parametrize("keep", [True, False])
def test_gradients_match_reference_tiny(golden_dir, keep):
    """keep=True: the forward keeps the per-block activations (gava_vision_forward_keep, fp16 activations + bf16
    gradients in the backward); keep=False: the backward recomputes every block from its saved input."""
    import numpy as np, os
    gold = np.load(os.path.join(golden_dir, "tiny_grads.npz"))
    m = VitaCLIP(**model_kwargs(TINY, CLASSES_3))
    m.load_state_dict(synth_torch_state(TINY, 3), strict=True)
    m = m.cuda().train()
    if not keep:
        m.keep_activation_bytes = 0
    x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size)).cuda()
    logits, lmt, lvm = m(x)
    assert lmt is None and lvm is None
    assert (logits.detach().cpu() - torch.from_numpy(gold["logits"])).abs().max() <= 1e-3 * np.abs(gold["logits"]).max()
    (logits * torch.from_numpy(gold["w_logits"]).cuda()).sum().backward()
    _check_against_reference_grads(m, gold)


def test_gradients_match_reference_with_auxiliary_heads(golden_dir):
    """add_nte + use_support_memory (VitaCLIP_model.py:311-398): the NTE head hangs off `summary`, the memory head off
    `text_features`; their outputs and every gradient (including sum_proj / tf_project / memory_project, handled by
    torch, and the prompt parameters they reach THROUGH the HIP backward) against the reference."""
    import numpy as np, os
    gold = np.load(os.path.join(golden_dir, "tiny_aux_grads.npz"))
    m = VitaCLIP(**model_kwargs(TINY, CLASSES_3), add_nte=True, use_support_memory=True, detach_features=False, num_classes=3)
    sd = synth_torch_state(TINY, 3)
    sd.update({k: torch.from_numpy(v) for k, v in synth.synth_aux_state(TINY, 3).items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size)).cuda()
    nte, mem = synth.synth_aux_inputs(2, TINY.embed_dim)
    logits, lmt, lvm = m(x, memory=torch.from_numpy(mem).cuda(), video_nte=torch.from_numpy(nte).cuda())
    for got, key in ((logits, "logits"), (lvm, "logits_vm"), (lmt, "logits_mt")):
        ref = torch.from_numpy(gold[key])
        assert (got.detach().cpu() - ref).abs().max() <= 2e-3 * ref.abs().max(), key
    loss = (logits * torch.from_numpy(gold["w_logits"]).cuda()).sum() + (lvm * torch.from_numpy(gold["w_vm"]).cuda()).sum() \
        + (lmt * torch.from_numpy(gold["w_mt"]).cuda()).sum()
    loss.backward()
    _check_against_reference_grads(m, gold)


def test_gradients_match_reference_vit_b16(golden_dir):
    import numpy as np, os
    gold = np.load(os.path.join(golden_dir, "b16_grads.npz"))
    m = VitaCLIP(**model_kwargs(VIT_B16_T8, CLASSES_3))
    m.load_state_dict(synth_torch_state(VIT_B16_T8, 3), strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_clip(1, VIT_B16_T8.num_frames, VIT_B16_T8.input_size)).cuda()
    logits = m(x)[0]
    assert (logits.detach().cpu() - torch.from_numpy(gold["logits"])).abs().max() <= 1e-3 * np.abs(gold["logits"]).max()
    (logits * torch.from_numpy(gold["w_logits"]).cuda()).sum().backward()
    _check_against_reference_grads(m, gold, sampled=True)


def test_kapt_forward_and_gradients_match_reference(golden_dir, tmp_path, monkeypatch):
    """Knowledge-aware prompts (training/kapt_head.py; `cntn_split_uni_disc`, 3 knowledge versions -> 3 prompts per
    class) on synthetic knowledge files: eval logits (mean over the class's prompts), class text features,
    per-description logits, and every train-mode gradient including the per-class context MLPs - against the fixture
    the reference produced on the same files."""
    import numpy as np, os
    gold = np.load(os.path.join(golden_dir, "tiny_kapt.npz"))
    synth.synth_knowledge_files(str(tmp_path), "updrs", 3, ["v1", "v2", "v3"])
    monkeypatch.chdir(tmp_path)
    m = VitaCLIP(**{**model_kwargs(TINY, CLASSES_3), "text_prompt_init": "cntn_split_uni_disc", "knowledge_version": ["v1", "v2", "v3"]})
    sd = synth_torch_state(TINY, 3)
    sd.update({k: torch.from_numpy(v) for k, v in synth.synth_kapt_state(TINY, 3).items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size)).cuda()
    with torch.no_grad():
        logits = m(x)[0]
        tfeat = m.text_features.clone()
        desc = m(x, desc_wise=True)[0]
    for got, key in ((logits, "logits"), (tfeat, "text_features"), (torch.stack(desc), "desc_logits")):
        ref = torch.from_numpy(gold[key])
        assert got.shape == ref.shape, key
        assert (got.cpu() - ref).abs().max() <= 1e-3 * ref.abs().max(), key
    m.train()
    lg = m(x)[0]
    (lg * torch.from_numpy(gold["w_logits"]).cuda()).sum().backward()
    worst = _check_against_reference_grads(m, gold)
    assert any("context_prompt_learner.projector" in k for k in worst)


def test_optimizer_steps_refresh_only_trainable_copies_and_match_a_fresh_model():
    """Two SGD steps on the drop-in model == the same two steps where the model is rebuilt from its state_dict before
    each forward (i.e. every 16-bit weight copy re-made from scratch): the in-place refresh of the summary-attention
    copies (the only trainable weights that are packed) is complete, and frozen weights are not re-converted."""
    sd = synth_torch_state(TINY, 3)
    x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size)).cuda()
    y = torch.tensor([0, 2], device="cuda")

    def make(state):
        m = VitaCLIP(**model_kwargs(TINY, CLASSES_3))
        m.load_state_dict(state, strict=True)
        return m.cuda().train()

    m = make(sd)
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.5)
    ref_state = {k: v.clone() for k, v in sd.items()}
    for step in range(2):
        key_before = m._pack_key()
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x)[0], y)
        loss.backward()
        opt.step()
        assert m._pack_key() == key_before            # frozen copies are never invalidated by the step
        fresh = make(ref_state)
        fopt = torch.optim.SGD([p for p in fresh.parameters() if p.requires_grad], lr=0.5)
        floss = torch.nn.functional.cross_entropy(fresh(x)[0], y)
        floss.backward()
        fopt.step()
        assert abs(float(loss.detach()) - float(floss.detach())) <= 1e-5 * abs(float(floss.detach())), step
        ref_state = {k: v.detach().cpu().clone() for k, v in fresh.state_dict().items()}
        # not bit-for-bit: d gamma / d beta of summary_ln are accumulated with fp32 atomics (order varies run to run)
        for (n, p), (_, q) in zip(m.named_parameters(), fresh.named_parameters()):
            assert torch.allclose(p, q, rtol=1e-4, atol=1e-6), (step, n)


def test_seventy_frame_clips_forward_and_backward_match_oracle():
    """train_scripts/updrs_3cls_train_tulip.sh trains with --num_frames 70: 70 local-prompt rows per frame, a 70-token
    summary attention (small backward kernel at n = 70), 96-key tiles in the main attention backward."""
    import dataclasses
    cfg = dataclasses.replace(TINY, num_frames=70)
    sd = synth_torch_state(cfg, 3)
    m = VitaCLIP(**model_kwargs(cfg, CLASSES_3))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_clip(1, 70, cfg.input_size))
    wlog = torch.randn(1, 3, generator=torch.Generator().manual_seed(9))
    ref_logits, ref = _oracle_all_grads(cfg, sd, torch.cat(m.tokenized_prompts).cpu(), x, wlog)
    logits = m(x.cuda())[0]
    assert (logits.detach().cpu() - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    (logits * wlog.cuda()).sum().backward()
    got = {n: p.grad for n, p in m.named_parameters()}
    bad = {}
    for name, g_ref in ref.items():
        if name.endswith("k_proj.bias"):
            continue
        r = rel(got[name].cpu(), g_ref)
        if r > 4e-2:
            bad[name] = r
    assert not bad, bad


def test_full_size_backward_is_additive_over_the_batch():
    """BASELINE config c2's training shape (64 clips, ViT-B/16): with a loss that is a sum over clips, the gradient of
    the 64-clip batch equals the sum of the gradients of its two 32-clip halves (clips are independent, every trainable
    parameter is shared) - a size-independent check of the full-size backward path (persistent dgrad GEMMs, MFMA
    attention backward on 512 frames, kept activations).  bf16 gradient operands: norm-wise 2e-2."""
    m = VitaCLIP(**model_kwargs(VIT_B16_T8, CLASSES_3))
    m.load_state_dict(synth_torch_state(VIT_B16_T8, 3), strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_clip(64, VIT_B16_T8.num_frames, VIT_B16_T8.input_size, seed=21)).cuda()
    w = torch.randn(64, 3, generator=torch.Generator().manual_seed(4)).cuda()
    names = [n for n, p in m.named_parameters() if p.requires_grad]

    def grads(xs, ws):
        for p in m.parameters():
            p.grad = None
        (m(xs)[0] * ws).sum().backward()
        return {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}

    full = grads(x, w)
    a, b = grads(x[:32], w[:32]), grads(x[32:], w[32:])
    assert set(full) == set(names) == set(a) == set(b)
    worst = ("", 0.0)
    for n in names:
        s = a[n] + b[n]
        assert torch.isfinite(full[n]).all()
        if float(s.norm()) < 1e-12:      # the exactly-zero key-bias gradient of the summary attention
            assert float(full[n].norm()) < 1e-6
            continue
        e = rel(full[n], s)
        if e > worst[1]:
            worst = (n, e)
    print(f"\n[c2 backward additivity] worst parameter {worst[0]}: {worst[1]:.3e}")
    assert worst[1] < 2e-2, worst


@pytest.mark.parametrize("keep", [True, False])
def test_training_with_more_frames_than_num_frames_matches_oracle_autograd(keep):
    """T_in = 2 * num_frames under autograd: frames are regrouped by the model's num_frames (utils:160-162) and the
    temporal embedding is nearest-resized (vision_encoder.py:91-95), so d time_embed gathers two input frames per row."""
    cfg, B = TINY, 2
    sd = synth_torch_state(cfg, 3)
    m = VitaCLIP(**model_kwargs(cfg, CLASSES_3))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    if not keep:
        m.keep_activation_bytes = 0
    x = torch.from_numpy(synth.synth_clip(B, 2 * cfg.num_frames, cfg.input_size, seed=13))
    wlog = torch.randn(B, 3, generator=torch.Generator().manual_seed(6))
    ref_logits, ref = _oracle_all_grads(cfg, sd, torch.cat(m.tokenized_prompts).cpu(), x, wlog)
    logits = m(x.cuda())[0]
    (logits * wlog.cuda()).sum().backward()
    assert (logits.detach().cpu() - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    got = {n: p.grad for n, p in m.named_parameters()}
    bad = {}
    for name, g_ref in ref.items():
        if name.endswith("k_proj.bias"):
            continue
        e = rel(got[name].cpu(), g_ref)
        if e > 4e-2:
            bad[name] = e
    assert not bad, bad
    assert got["visual.time_embed"].shape == (cfg.num_frames, cfg.feature_dim)


def test_kapt_descriptor_mode_forward_and_gradients_match_reference(golden_dir, tmp_path, monkeypatch):
    """KAPT with use_descriptor=True: a ragged number of prompts per class (2, 3, 1 descriptors).  Eval logits (mean over
    each class's own prompts), class features, the per-descriptor logits and every train-mode gradient against the
    fixture the reference produced on the same synthetic descriptor files."""
    import numpy as np, os
    gold = np.load(os.path.join(golden_dir, "tiny_kapt_desc.npz"))
    synth.synth_descriptor_files(str(tmp_path), "updrs", (2, 3, 1))
    monkeypatch.chdir(tmp_path)
    m = VitaCLIP(**{**model_kwargs(TINY, CLASSES_3), "text_prompt_init": "cntn_split_uni_disc", "knowledge_version": ["v1", "v2", "v3"],
                    "use_descriptor": True})
    sd = synth_torch_state(TINY, 3)
    sd.update({k: torch.from_numpy(v) for k, v in synth.synth_kapt_state(TINY, 3).items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size)).cuda()
    with torch.no_grad():
        logits = m(x)[0]
        tfeat = m.text_features.clone()
        desc = m(x, desc_wise=True)[0]
    assert [tuple(d.shape) for d in desc] == [(2, 2), (2, 3), (2, 1)]
    for got, key in ((logits, "logits"), (tfeat, "text_features"), (torch.cat(desc, 1), "desc_logits")):
        ref = torch.from_numpy(gold[key])
        assert got.shape == ref.shape, key
        assert (got.cpu() - ref).abs().max() <= 1e-3 * ref.abs().max(), key
    m.train()
    lg = m(x)[0]
    (lg * torch.from_numpy(gold["w_logits"]).cuda()).sum().backward()
    worst = _check_against_reference_grads(m, gold)
    assert any("context_prompt_learner.projector" in k for k in worst)
