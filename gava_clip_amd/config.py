"""Shape configuration of the VitaCLIP forward path.

Mirrors the constructor keywords of the reference model
(/root/reference/training/VitaCLIP_model.py:24-74) that decide tensor shapes.
Only the prompt-enabled configuration is valid in the reference (SURVEY.md §7,
hard part 5: the non-global-prompt branch is broken upstream), so summary token,
local prompts and global prompts are always on here.
"""
from dataclasses import dataclass, asdict
from collections import OrderedDict


@dataclass(frozen=True)
class VitaConfig:
    input_size: int = 224
    num_frames: int = 8
    feature_dim: int = 768          # D
    patch_size: int = 16            # P
    num_heads: int = 12             # H (head dim must be 64)
    num_layers: int = 12
    mlp_factor: float = 4.0
    embed_dim: int = 512            # E
    num_global_prompts: int = 8     # G
    text_context_length: int = 77
    text_vocab_size: int = 49408
    text_width: int = 512           # W
    text_heads: int = 8
    text_layers: int = 12
    text_num_prompts: int = 8       # n_ctx

    @property
    def grid(self) -> int:
        return self.input_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def mlp_dim(self) -> int:
        return int(round(self.mlp_factor * self.feature_dim))

    @property
    def tokens_main(self) -> int:
        """cls + patches: the rows of a frame that survive every block."""
        return 1 + self.num_patches

    def attn_keys(self, T=None) -> int:
        """L_att of the reference = 1 + T + G + patches + 1 (utils:171-188)."""
        T = self.num_frames if T is None else T
        return self.tokens_main + self.num_global_prompts + T + 1

    def as_dict(self):
        return asdict(self)


# named configurations from BASELINE.json / SURVEY.md §8
VIT_B16_T8 = VitaConfig()
VIT_B16_T16 = VitaConfig(num_frames=16)
VIT_L14_T32 = VitaConfig(num_frames=32, feature_dim=1024, patch_size=14, num_heads=16,
                         num_layers=24, embed_dim=768, text_width=768, text_heads=12)
# tiny configuration for exhaustive intermediate-tensor fixtures (head dim stays 64)
TINY = VitaConfig(input_size=64, num_frames=4, feature_dim=128, patch_size=16, num_heads=2,
                  num_layers=2, embed_dim=128, num_global_prompts=4, text_width=128,
                  text_heads=2, text_layers=2, text_num_prompts=4)


def param_shapes(cfg: VitaConfig, n_cls: int) -> "OrderedDict[str, tuple]":
    """state_dict keys and shapes of the reference model, in the
    order ``VitaCLIP(...).state_dict()`` yields them (checked by strict=True load in
    tools/gen_golden.py)."""
    D, E, W = cfg.feature_dim, cfg.embed_dim, cfg.text_width
    P, T, G = cfg.patch_size, cfg.num_frames, cfg.num_global_prompts
    F = cfg.mlp_dim
    s = OrderedDict()
    s["logit_scale"] = ()
    s["visual.cls_token"] = (D,)
    s["visual.pos_embed"] = (cfg.num_patches + 1, D)
    s["visual.time_embed"] = (T, D)
    s["visual.proj"] = (D, E)
    s["visual.global_prompts"] = (cfg.num_layers, G, D)
    s["visual.patch_embed.proj.weight"] = (D, 3, P, P)
    s["visual.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.num_layers):
        p = f"visual.blocks.{i}."
        s[p + "local_prompts"] = (1, T, D)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"attn.{n}.weight"] = (D, D)
            s[p + f"attn.{n}.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (F, D)
        s[p + "mlp.fc1.bias"] = (F,)
        s[p + "mlp.fc2.weight"] = (D, F)
        s[p + "mlp.fc2.bias"] = (D,)
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (D,)
            s[p + n + ".bias"] = (D,)
        s[p + "cls_proj.weight"] = (D, D)
        s[p + "cls_proj.bias"] = (D,)
        s[p + "summary_ln.weight"] = (D,)
        s[p + "summary_ln.bias"] = (D,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"summary_attn_layer.{n}.weight"] = (D, D)
            s[p + f"summary_attn_layer.{n}.bias"] = (D,)
    for n in ("ln_pre", "ln_post"):
        s[f"visual.{n}.weight"] = (D,)
        s[f"visual.{n}.bias"] = (D,)
    s["textual.positional_embedding"] = (cfg.text_context_length, W)
    s["textual.text_projection"] = (W, E)
    for i in range(cfg.text_layers):
        p = f"textual.transformer.resblocks.{i}."
        s[p + "attn.in_proj_weight"] = (3 * W, W)
        s[p + "attn.in_proj_bias"] = (3 * W,)
        s[p + "attn.out_proj.weight"] = (W, W)
        s[p + "attn.out_proj.bias"] = (W,)
        s[p + "ln_1.weight"] = (W,)
        s[p + "ln_1.bias"] = (W,)
        s[p + "mlp.c_fc.weight"] = (4 * W, W)
        s[p + "mlp.c_fc.bias"] = (4 * W,)
        s[p + "mlp.c_proj.weight"] = (W, 4 * W)
        s[p + "mlp.c_proj.bias"] = (W,)
        s[p + "ln_2.weight"] = (W,)
        s[p + "ln_2.bias"] = (W,)
    s["textual.token_embedding.weight"] = (cfg.text_vocab_size, W)
    s["textual.ln_final.weight"] = (W,)
    s["textual.ln_final.bias"] = (W,)
    s["prompt_learner.ctx"] = (n_cls, cfg.text_num_prompts, W)
    return s
