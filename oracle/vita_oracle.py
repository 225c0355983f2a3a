"""ORACLE — CPU fp32 restatement of the GaVA-CLIP forward path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (gava_clip_amd) never does and fails loudly without its HIP
library.

Every function restates, in plain torch-CPU float32 tensor algebra, what the reference
computes, citing the file:line it follows (all under /root/reference/training/).  It is a
floating-point path, so the oracle is torch fp32 rather than C/numpy.  Parity of this
oracle with the reference itself is PINNED: tools/gen_golden.py imports the reference in
the build container, runs it on the synthetic weights/inputs of gava_clip_amd/synth.py and
commits its outputs under tests/golden/; tests/test_oracle_golden.py checks this file
against them (the reference has no tests or golden vectors of its own).

``operand_dtype`` emulates a 16-bit MFMA path on the CPU: every GEMM/attention operand is
rounded to that dtype, products accumulate in fp32, everything else stays fp32.  It is
used to size the numerical budget of the HIP kernels, not for parity.
"""
import math
import torch
import torch.nn.functional as F


class Oracle:
    def __init__(self, cfg, params, tokenized_prompts, operand_dtype=None):
        """params: dict name -> float32 torch tensor (reference state_dict keys).
        tokenized_prompts: (C, 77) int tensor, one prompt per class (n_kv = 1, the plain
        CoOp path of every BASELINE config)."""
        self.cfg = cfg
        self.p = params
        self.tok = torch.as_tensor(tokenized_prompts).long()
        self.od = operand_dtype
        self.trace = None

    # -- primitive ops -----------------------------------------------------------------
    def _r(self, t):
        return t if self.od is None else t.to(self.od).to(torch.float32)

    def linear(self, x, w, b=None):
        y = self._r(x) @ self._r(w).t()
        return y if b is None else y + b

    @staticmethod
    def layer_norm(x, w, b):
        # VitaCLIP_vision_encoder_utils.py:22-28 / VitaCLIP_text_encoder.py:19-25, eps 1e-5
        return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5)

    @staticmethod
    def quick_gelu(x):
        # VitaCLIP_vision_encoder_utils.py:18-20
        return x * torch.sigmoid(1.702 * x)

    def _rec(self, name, t):
        if self.trace is not None:
            self.trace[name] = t.detach().clone()

    # -- vision ------------------------------------------------------------------------
    def attention(self, pre, x, heads):
        """VitaCLIP_vision_encoder_utils.py:61-81 with q=k=v=x (N,L,D)."""
        p = self.p
        N, L, D = x.shape
        q = self.linear(x, p[pre + "q_proj.weight"], p[pre + "q_proj.bias"])
        k = self.linear(x, p[pre + "k_proj.weight"], p[pre + "k_proj.bias"])
        v = self.linear(x, p[pre + "v_proj.weight"], p[pre + "v_proj.bias"])
        dh = D // heads
        q = q.view(N, L, heads, dh)
        k = k.view(N, L, heads, dh)
        v = v.view(N, L, heads, dh)
        aff = torch.einsum("nqhc,nkhc->nqkh", self._r(q / (dh ** 0.5)), self._r(k))
        aff = aff.softmax(dim=-2)
        mix = torch.einsum("nqlh,nlhc->nqhc", self._r(aff), self._r(v))
        return self.linear(mix.flatten(-2), p[pre + "out_proj.weight"], p[pre + "out_proj.bias"])

    def block(self, i, x, B, T):
        """VitaCLIP_vision_encoder_utils.py:155-203 (summary token + local prompts on).
        x: (B*T, 1+G+n, D).  Frames are regrouped with the model's num_frames, exactly as
        the reference does (utils:160-162)."""
        p, cfg = self.p, self.cfg
        pre = f"visual.blocks.{i}."
        BT, N, C = x.shape
        Tm = cfg.num_frames
        Bm = BT // Tm
        cls_proj = self.linear(x[:, 0, :].view(Bm, Tm, C), p[pre + "cls_proj.weight"], p[pre + "cls_proj.bias"])
        s_norm = self.layer_norm(cls_proj, p[pre + "summary_ln.weight"], p[pre + "summary_ln.bias"])
        summ = cls_proj + self.attention(pre + "summary_attn_layer.", s_norm, cfg.num_heads)
        x = torch.cat([x, summ.reshape(BT, 1, C)], dim=1)                       # utils:171-172
        lp = p[pre + "local_prompts"].expand(Bm, -1, -1) + cls_proj             # utils:176-184
        lp = lp.repeat_interleave(repeats=Tm, dim=0)                            # utils:187
        x = torch.cat((x[:, :1, :], lp, x[:, 1:, :]), dim=1)                    # utils:188
        xn = self.layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
        x = x + self.attention(pre + "attn.", xn, cfg.num_heads)                # utils:190-191
        x = x[:, :-1, :]                                                        # utils:194-195
        x = torch.cat((x[:, :1, :], x[:, lp.shape[1] + 1:, :]), dim=1)          # utils:196-197
        h = self.layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
        h = self.quick_gelu(self.linear(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
        x = x + self.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])  # utils:199
        return x, summ

    def patch_embed(self, x):
        """VitaCLIP_vision_encoder_utils.py:205-220: Conv2d(3,D,k=P,s=P)+bias as a GEMM
        over (c,ky,kx)."""
        p, cfg = self.p, self.cfg
        P, g = cfg.patch_size, cfg.grid
        BT = x.shape[0]
        cols = x.view(BT, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(BT, g * g, 3 * P * P)
        w = p["visual.patch_embed.proj.weight"].reshape(cfg.feature_dim, -1)
        return self.linear(cols, w, p["visual.patch_embed.proj.bias"])

    def time_embed(self, T):
        """VitaCLIP_vision_encoder.py:86-100: nearest resize when T != len(time_embed)."""
        te = self.p["visual.time_embed"]
        if T != te.shape[0]:
            te = F.interpolate(te.unsqueeze(0).transpose(1, 2), size=(T), mode="nearest")
            te = te.transpose(1, 2).squeeze(0)
        return te

    def vision(self, x):
        """VitaCLIP_vision_encoder.py:102-132 -> (cls_x (B,E), summary (B,D))."""
        p, cfg = self.p, self.cfg
        B, C, T, H, W = x.shape
        G = cfg.num_global_prompts
        x = x.permute(0, 2, 1, 3, 4).flatten(0, 1)                              # :105
        x = self.patch_embed(x)                                                 # :107
        x = torch.cat([p["visual.cls_token"].view(1, 1, -1).repeat(x.size(0), 1, 1), x], dim=1)
        x = x + p["visual.pos_embed"]                                           # :110
        te = self.time_embed(T)
        x = (x.view(B, T, x.shape[1], -1) + te.view(1, T, 1, -1)).view(B * T, x.shape[1], -1)  # :111
        self._rec("embed", x)
        x = self.layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"])  # :113
        self._rec("ln_pre", x)
        summ = None
        for i in range(cfg.num_layers):                                         # :115-121
            gp = p["visual.global_prompts"][i].expand(B * T, -1, -1)
            x = torch.cat((x[:, :1, :], gp, x[:, 1:, :]), dim=1)
            x, summ = self.block(i, x, B, T)
            x = torch.cat((x[:, :1, :], x[:, G + 1:, :]), dim=1)
            self._rec(f"block{i}", x)
            self._rec(f"summ{i}", summ)
        cls_x = self.layer_norm(x[:, 0, :], p["visual.ln_post.weight"], p["visual.ln_post.bias"])
        cls_x = self.linear(cls_x, p["visual.proj"].t())                        # :126-127
        cls_x = cls_x.view(B, T, -1).mean(dim=1)                                # :128
        summary = summ.mean(dim=1)                                              # :129-130
        return cls_x, summary

    # -- text --------------------------------------------------------------------------
    def prompts(self):
        """VitaCLIP_text_encoder.py:296-332 with CSC ctx: [SOS | ctx[c] | suffix] -> (C,77,W)."""
        p, cfg = self.p, self.cfg
        emb = p["textual.token_embedding.weight"][self.tok]                     # :284
        n = cfg.text_num_prompts
        return torch.cat([emb[:, :1, :], p["prompt_learner.ctx"], emb[:, 1 + n:, :]], dim=1)

    def text_block(self, i, x):
        """VitaCLIP_text_encoder.py:67-88; nn.MultiheadAttention restated:
        packed in_proj split in thirds, heads = contiguous channel groups, scores
        (q/sqrt(dh))k^T + causal -inf mask, softmax over keys, out_proj.  x: (n,L,W)."""
        p, cfg = self.p, self.cfg
        pre = f"textual.transformer.resblocks.{i}."
        n, L, W = x.shape
        H = cfg.text_heads
        dh = W // H
        h = self.layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"])
        qkv = self.linear(h, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"])
        q, k, v = qkv.split(W, dim=-1)
        q = q.view(n, L, H, dh).transpose(1, 2)
        k = k.view(n, L, H, dh).transpose(1, 2)
        v = v.view(n, L, H, dh).transpose(1, 2)
        s = self._r(q / math.sqrt(dh)) @ self._r(k).transpose(-1, -2)
        mask = torch.full((L, L), float("-inf")).triu_(1)                       # :146-152
        a = (s + mask).softmax(dim=-1)
        o = (self._r(a) @ self._r(v)).transpose(1, 2).reshape(n, L, W)
        x = x + self.linear(o, p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"])
        h = self.layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
        h = self.quick_gelu(self.linear(h, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"]))
        return x + self.linear(h, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"])

    def text(self, prompts):
        """VitaCLIP_text_encoder.py:154-171 -> (n,E)."""
        p, cfg = self.p, self.cfg
        x = prompts + p["textual.positional_embedding"]
        for i in range(cfg.text_layers):
            x = self.text_block(i, x)
        x = self.layer_norm(x, p["textual.ln_final.weight"], p["textual.ln_final.bias"])
        eot = (self.tok == cfg.text_vocab_size - 1).nonzero()[:, 1]            # :169
        x = x[torch.arange(x.shape[0]), eot]
        return self.linear(x, p["textual.text_projection"].t())

    # -- model -------------------------------------------------------------------------
    def forward(self, x, trace=False):
        """VitaCLIP_model.py:241-309,401 (prompt-learning branch, n_kv=1) ->
        dict(logits (B,C), video_features (B,E) normalised, text_features (C,E) normalised,
        summary (B,D))."""
        self.trace = {} if trace else None
        with torch.no_grad():
            logit_scale = self.p["logit_scale"].exp()                          # :248
            vf, summary = self.vision(x.float())
            vf = vf / vf.norm(dim=-1, keepdim=True)                            # :255
            tf = self.text(self.prompts())                                     # :282-285 batched
            tf = tf / tf.norm(dim=-1, keepdim=True)                            # :287
            logits = logit_scale * vf @ tf.t()                                 # :288-289 (n_kv=1)
            tfeat = tf / tf.norm(dim=-1, keepdim=True)                         # :291
        out = dict(logits=logits, video_features=vf, text_features=tfeat, summary=summary)
        if trace:
            out["trace"] = self.trace
        return out
