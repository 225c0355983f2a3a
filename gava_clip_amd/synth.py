"""Platform-exact synthetic weights and inputs.

The reference ships no checkpoints usable offline and its 185 M parameters cannot be
committed as fixtures (hard part 6), so both the golden-vector generator
(tools/gen_golden.py, runs the imported reference) and the GPU parity tests rebuild the
same tensors from (name, shape, seed) with integer hashing only: splitmix64 counters ->
24-bit uniforms -> IEEE float64 arithmetic -> float32.  No libm transcendental is
involved, so the bits are identical on any machine.

Per-tensor scales follow the reference's initialisers where it has them
(VitaCLIP_vision_encoder.py:62-84, VitaCLIP_vision_encoder_utils.py:54-57,144-152,
VitaCLIP_text_encoder.py:238) and the CLIP convention where the reference leaves
``torch.empty`` (VitaCLIP_text_encoder.py:141,143).  Biases and LayerNorm affines are
deliberately non-trivial so that a dropped bias/affine shows up in parity.
"""
import math
import numpy as np

from .config import VitaConfig, param_shapes

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(key: int, n: int, stream: int = 0, chunk: int = 1 << 24) -> np.ndarray:
    """n float64 values in (0,1), 24-bit resolution, from counter-mode splitmix64."""
    out = np.empty(n, dtype=np.float64)
    base = np.uint64((key + stream * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ctr = np.arange(s, e, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = _splitmix64(base + ctr * _GOLD)
        out[s:e] = ((z >> np.uint64(40)).astype(np.float64) + 0.5) * (1.0 / 16777216.0)
    return out


def uniform_pm1(name: str, shape, seed: int = 0) -> np.ndarray:
    """float64 uniform in (-1, 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = _fnv1a64(name) ^ ((seed * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF)
    return (2.0 * _uniform01(key, n) - 1.0).reshape(shape)


def normalish(name: str, shape, seed: int = 0) -> np.ndarray:
    """float64, zero mean, unit variance, bell-shaped (Irwin-Hall of 4 uniforms)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = _fnv1a64(name) ^ ((seed * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF)
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += _uniform01(key, n, stream=k + 1)
    return ((acc - 2.0) * math.sqrt(3.0)).reshape(shape)


def _rule(name: str, shape, cfg: VitaConfig):
    """-> (kind, scale, offset) with kind in {'u','n','const'}."""
    D, W, P = cfg.feature_dim, cfg.text_width, cfg.patch_size
    if name == "logit_scale":
        return "const", 0.0, math.log(1.0 / 0.07)
    leaf = name.rsplit(".", 1)[-1]
    is_ln = any(t in name for t in ("ln_pre", "ln_post", "norm1", "norm2", "summary_ln",
                                    "ln_1", "ln_2", "ln_final"))
    if is_ln:
        return ("u", 0.1, 1.0) if leaf == "weight" else ("u", 0.05, 0.0)
    if leaf in ("bias", "in_proj_bias"):
        return "u", 0.02, 0.0
    if name in ("visual.cls_token", "visual.pos_embed", "visual.time_embed"):
        return "n", 0.02, 0.0
    if name == "visual.proj":
        return "n", D ** -0.5, 0.0
    if name == "visual.global_prompts" or leaf == "local_prompts":
        return "u", math.sqrt(6.0 / (3 * P * P + D)), 0.0
    if name == "visual.patch_embed.proj.weight":
        return "u", 1.0 / math.sqrt(3 * P * P), 0.0
    if name.startswith("visual.blocks.") and leaf == "weight":
        fan_out, fan_in = shape
        if "cls_proj" in name:
            return "u", 1.0 / math.sqrt(fan_in), 0.0
        return "u", math.sqrt(6.0 / (fan_in + fan_out)), 0.0
    if name == "textual.positional_embedding":
        return "n", 0.01, 0.0
    if name == "textual.text_projection":
        return "n", W ** -0.5, 0.0
    if name == "textual.token_embedding.weight":
        return "n", 0.02, 0.0
    if name.startswith("textual.transformer.resblocks."):
        L = cfg.text_layers
        if leaf == "in_proj_weight":
            return "n", W ** -0.5, 0.0
        if "out_proj" in name or "c_proj" in name:
            return "n", (W ** -0.5) * ((2 * L) ** -0.5), 0.0
        if "c_fc" in name:
            return "n", (2 * W) ** -0.5, 0.0
    if name == "prompt_learner.ctx":
        return "n", 0.02, 0.0
    # auxiliary heads (VitaCLIP_model.py:148-199): sum_proj, tf_project.*, memory_project.*.*; their logit scales are
    # kept small so that the log-softmax outputs of the synthetic case are not saturated
    if name in ("logit_scale_vm", "logit_scale_mt"):
        return "const", 0.0, 4.0
    if name.startswith(("sum_proj.", "tf_project.", "memory_project.")) and leaf == "weight":
        return "u", 1.0 / math.sqrt(shape[1]), 0.0
    if "context_prompt_learner.projector." in name and leaf == "weight":   # zero-initialised upstream; non-zero here
        return "u", 0.5 / math.sqrt(shape[1]), 0.0
    raise KeyError(f"no synth rule for {name}")


def synth_param(name: str, shape, cfg: VitaConfig, seed: int = 0) -> np.ndarray:
    kind, scale, offset = _rule(name, shape, cfg)
    if kind == "const":
        return np.full(shape, offset, dtype=np.float32)
    base = uniform_pm1(name, shape, seed) if kind == "u" else normalish(name, shape, seed)
    return (base * scale + offset).astype(np.float32)


def synth_state_dict(cfg: VitaConfig, n_cls: int, seed: int = 0):
    """OrderedDict name -> float32 ndarray for every key of the reference state_dict."""
    from collections import OrderedDict
    out = OrderedDict()
    for name, shape in param_shapes(cfg, n_cls).items():
        out[name] = synth_param(name, shape, cfg, seed)
    return out


def synth_aux_state(cfg: VitaConfig, n_cls: int, seed: int = 0):
    """Parameters of the auxiliary heads (add_nte + use_support_memory), reference key order."""
    from collections import OrderedDict
    D, E = cfg.feature_dim, cfg.embed_dim
    shapes = OrderedDict()
    shapes["logit_scale_vm"] = ()
    shapes["logit_scale_mt"] = ()
    shapes["sum_proj.weight"], shapes["sum_proj.bias"] = (E, D), (E,)
    mlp = [("0.weight", (E // 4, E)), ("0.bias", (E // 4,)), ("2.weight", (E // 8, E // 4)), ("2.bias", (E // 8,))]
    for k, sh in mlp:
        shapes["tf_project." + k] = sh
    for c in range(n_cls):
        for k, sh in mlp:
            shapes[f"memory_project.{c}.{k}"] = sh
    return OrderedDict((k, synth_param(k, sh, cfg, seed)) for k, sh in shapes.items())


def synth_aux_inputs(B: int, E: int, n_mem: int = 5, seed: int = 1234):
    """(video_nte (B, 70, E), memory (B, n_mem, E)) for the auxiliary heads."""
    return (normalish("input.video_nte", (B, 70, E), seed).astype(np.float32),
            normalish("input.memory", (B, n_mem, E), seed).astype(np.float32))


KAPT_DESCRIPTIONS = (   # synthetic stand-ins for ./data/ke_<type>/simQdesc_<kv>.txt (one line per class and version)
    "a person walking with {} gait impairment", "video of a patient whose walk shows {} signs",
    "{} severity of slowness and reduced arm swing", "clinical gait recording rated {}", "footage of {} shuffling steps")


def synth_knowledge_files(root: str, cls_type: str, n_cls: int, versions, seed: int = 0, inp_dim: int = 768):
    """Write synthetic KEPLER knowledge files where training/kapt_head.py:60,94-112 looks for them:
    <root>/data/ke_<type>/EntityEmb_<kv>.npy (n_cls, 768) and simQdesc_<kv>.txt (n_cls lines).  The real files are
    not part of the reference repository; the same synthetic ones feed the reference (tools/gen_golden.py) and the tests."""
    import os
    d = os.path.join(root, "data", "ke_" + cls_type.lower().split("_")[0])
    os.makedirs(d, exist_ok=True)
    levels = ("no", "slight", "mild", "moderate", "severe", "very severe")
    for i, kv in enumerate(versions):
        np.save(os.path.join(d, f"EntityEmb_{kv}.npy"), normalish(f"kapt.entity.{kv}", (n_cls, inp_dim), seed).astype(np.float32))
        with open(os.path.join(d, f"simQdesc_{kv}.txt"), "w") as f:
            for c in range(n_cls):
                f.write(KAPT_DESCRIPTIONS[i % len(KAPT_DESCRIPTIONS)].format(levels[c % len(levels)]) + "\n")
    return d


KAPT_DESCRIPTORS = ("arms swing {} while walking", "{} shuffling of the feet", "the trunk is {} bent forward",
                    "turning takes {} many steps", "{} freezing at the start")


def synth_descriptor_files(root: str, cls_type: str, counts, seed: int = 0, inp_dim: int = 768):
    """Synthetic per-class descriptor files for KAPT's use_descriptor mode (kapt_head.py:65-88): all.npy (n_cls, 768),
    descriptor_<c>.txt (counts[c] lines) and descriptor_<c>.npy (counts[c], 768).  A ragged number of prompts per class."""
    import os
    d = os.path.join(root, "data", "ke_" + cls_type.lower().split("_")[0])
    os.makedirs(d, exist_ok=True)
    levels = ("no", "slight", "mild", "moderate", "severe", "very severe")
    np.save(os.path.join(d, "all.npy"), normalish("kapt.entity.all", (len(counts), inp_dim), seed).astype(np.float32))
    for c, k in enumerate(counts):
        np.save(os.path.join(d, f"descriptor_{c}.npy"), normalish(f"kapt.descriptor.{c}", (k, inp_dim), seed).astype(np.float32))
        with open(os.path.join(d, f"descriptor_{c}.txt"), "w") as f:
            for j in range(k):
                f.write(KAPT_DESCRIPTORS[(c + j) % len(KAPT_DESCRIPTORS)].format(levels[c % len(levels)]) + "\n")
    return d


def synth_kapt_state(cfg: VitaConfig, n_cls: int, seed: int = 0, inp_dim: int = 768):
    from collections import OrderedDict
    W = cfg.text_width
    out = OrderedDict()
    for c in range(n_cls):
        for k, sh in (("0.weight", (W // 4, inp_dim)), ("2.weight", (W, W // 4))):
            name = f"prompt_learner.context_prompt_learner.projector.{c}.{k}"
            out[name] = synth_param(name, sh, cfg, seed)
    return out


def synth_clip(B: int, T: int, size: int, seed: int = 1234) -> np.ndarray:
    """(B,3,T,size,size) float32, ~N(0,1): the post-normalisation frame statistics of
    /root/reference/video_dataset/dataset.py:117-139."""
    return normalish("input.clip", (B, 3, T, size, size), seed).astype(np.float32)
