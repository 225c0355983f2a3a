"""Algorithmic FLOPs of the forward path: 2 x MACs of the dense ops the REFERENCE executes
(BASELINE.md §3).  This is the figure roofline fractions are quoted against; work
the HIP path skips (prompt-token rows whose outputs the reference discards) is not subtracted."""
from .config import VitaConfig


def vision_layer_macs(cfg: VitaConfig, T=None) -> float:
    T = cfg.num_frames if T is None else T
    D, F = cfg.feature_dim, cfg.mlp_dim
    L_mlp = 1 + cfg.num_global_prompts + cfg.num_patches
    L_att = L_mlp + 1 + T
    return (D * D + 4 * D * D + 2 * T * D        # cls_proj, summary q/k/v/o, summary attention
            + 3 * L_att * D * D + 2 * L_att * L_att * D + L_att * D * D
            + 2 * L_mlp * D * F)


def vision_flops_per_frame(cfg: VitaConfig, T=None) -> float:
    patch = 2.0 * cfg.num_patches * 3 * cfg.patch_size ** 2 * cfg.feature_dim
    proj = 2.0 * cfg.feature_dim * cfg.embed_dim
    return patch + 2.0 * cfg.num_layers * vision_layer_macs(cfg, T) + proj


def vision_flops_per_clip(cfg: VitaConfig, T=None) -> float:
    T = cfg.num_frames if T is None else T
    return T * vision_flops_per_frame(cfg, T)


def text_flops_per_prompt(cfg: VitaConfig) -> float:
    W, L = cfg.text_width, cfg.text_context_length
    return 2.0 * (cfg.text_layers * (L * 12 * W * W + 2 * L * L * W) + W * cfg.embed_dim)


def forward_flops(cfg: VitaConfig, B: int, n_cls: int, T=None) -> float:
    return B * vision_flops_per_clip(cfg, T) + n_cls * text_flops_per_prompt(cfg) + 2.0 * B * n_cls * cfg.embed_dim


def executed_flops(cfg: VitaConfig, B: int, n_cls: int, T=None, text_rows=None) -> float:
    """2 x MACs of the dense work the HIP path EXECUTES for one forward ("dead work"): q / out_proj /
    MLP only for the cls + patch rows (the reference also pushes the G global-prompt rows through them and discards the
    result), K/V of the prompt rows once per block (G global rows) or once per frame (local, summary) instead of once
    per frame and key, the last block on the CLS rows only, and the text tower on the first `text_rows` positions of
    each prompt (rows behind the last EOT are never read).  A split-precision GEMM (text tower, final projection) is
    counted once, not with its three MFMA passes.  SURVEY.md section 8d: skipped work is not credited - this is the
    figure `mfma_frac_executed` uses, next to the reference's `forward_flops`."""
    T = cfg.num_frames if T is None else T
    D, F, G, n1 = cfg.feature_dim, cfg.mlp_dim, cfg.num_global_prompts, cfg.tokens_main
    BT = B * T
    L_att = cfg.attn_keys(T)
    side = D * D + 4 * D * D + 2 * T * D                       # cls_proj, summary q/k/v/o, T x T summary attention: per frame
    side_kv = (G + 2 * BT) * 2 * D * D                         # K/V projection of the prompt rows: per block
    full = BT * (3 * n1 * D * D + 2 * n1 * L_att * D + n1 * D * D + 2 * n1 * D * F + side) + side_kv
    last = BT * (2 * n1 * D * D + D * D + 2 * L_att * D + D * D + 2 * D * F + side) + side_kv
    patch = BT * cfg.num_patches * 3 * cfg.patch_size ** 2 * D
    vision = 2.0 * (patch + (cfg.num_layers - 1) * full + last + BT * D * cfg.embed_dim)
    W, L = cfg.text_width, (cfg.text_context_length if text_rows is None else text_rows)
    text = 2.0 * n_cls * (cfg.text_layers * (L * 12 * W * W + 2 * L * L * W) + W * cfg.embed_dim)
    return vision + text + 2.0 * B * n_cls * cfg.embed_dim
