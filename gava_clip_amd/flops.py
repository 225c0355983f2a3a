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
