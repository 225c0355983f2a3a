"""Drop-in ``VitaCLIP(nn.Module)`` whose forward runs on libgava_hip.so (MI355X / gfx950).

Mirrors the reference boundary /root/reference/training/VitaCLIP_model.py:22-401: same
constructor keywords (:24-74), same ``forward(x, memory=None, video_nte=None, desc_wise=False)``
-> ``(logits, logits_mt, logits_vm)`` (:241-244,401), same ``state_dict`` keys,
same public attributes (``visual``, ``textual``, ``prompt_learner``, ``tokenized_prompts``,
``logit_scale``, ``text_features``), same freezing of parameters (:222-239).

The sub-modules below are parameter containers with the reference's names and initialisers; they
compute nothing in PyTorch.  All arithmetic of the hot path happens in the HIP kernels through
the C ABI (gava_clip_amd/hip.py); PyTorch only owns the device memory.  There is no CPU or eager
fallback: a non-device input or a missing library raises.
"""
import ctypes as C
import math
import os
import weakref
from collections import OrderedDict
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .tokenizer import tokenize, read_class_names, prompt_texts

NUM_COMB = 70  # /root/reference/video_dataset/dataset.py:19


# ---------------------------------------------------------------------------------------------
# parameter containers (names/initialisers: VitaCLIP_vision_encoder*.py, VitaCLIP_text_encoder.py)
# ---------------------------------------------------------------------------------------------
class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - never called
        raise RuntimeError("gava_clip_amd sub-modules hold parameters only; call VitaCLIP.forward")


class _Encoder(_Params):
    """The two L1 encoders of the reference are callable on their own (`videoEncoder(data)`, evaluation/iwa.py:212,230;
    `text_model(token_features, tokenized_prompts)`, evaluation/zero_shot.py:75-76, utils/prepare_embedding.py): forward()
    hands the call to the HIP host that packs this encoder's weights - the VitaCLIP it belongs to, or a private one when
    the encoder was constructed stand-alone."""

    def _host(self):
        ref = self.__dict__.get("_host_ref")
        host = ref() if ref is not None else None
        if host is not None and getattr(host, "visual", None) is not self and getattr(host, "textual", None) is not self:
            host = None          # a deep copy still pointing at the model it was copied from
        if host is None:
            host = self.__dict__.get("_own_host")
            if host is None:
                host = _StandaloneHost(self)
                self.__dict__["_own_host"] = host
        return host


class Attention(_Params):
    """q/k/v/out projections (VitaCLIP_vision_encoder_utils.py:31-57)."""

    def __init__(self, dim):
        super().__init__()
        self.q_proj = nn.Linear(dim, dim)
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)
        for m in (self.q_proj, self.k_proj, self.v_proj, self.out_proj):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0.)


class LayerNorm(nn.LayerNorm):
    pass


class TransformerEncoderLayer(_Params):
    """VitaCLIP_vision_encoder_utils.py:83-152 (summary token and local prompts always on)."""

    def __init__(self, dim, num_heads, mlp_factor, num_frames, patch_size):
        super().__init__()
        self.attn = Attention(dim)
        mlp_dim = round(mlp_factor * dim)
        self.mlp = nn.Sequential(OrderedDict([("fc1", nn.Linear(dim, mlp_dim)), ("act", nn.Identity()),
                                              ("dropout", nn.Identity()), ("fc2", nn.Linear(mlp_dim, dim))]))
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.cls_proj = nn.Linear(dim, dim)
        self.num_frames = num_frames
        self.summary_ln = LayerNorm(dim)
        self.summary_attn_layer = Attention(dim)
        self.local_prompts = nn.Parameter(torch.zeros(1, num_frames, dim))
        val = math.sqrt(6. / float(3 * patch_size[0] * patch_size[1] + dim))
        nn.init.uniform_(self.local_prompts.data, -val, val)
        for m in (self.mlp.fc1, self.mlp.fc2):
            nn.init.xavier_uniform_(m.weight)
            nn.init.normal_(m.bias, std=1e-6)


class _PatchEmbed(_Params):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class CLIPVisionEncoder(_Encoder):
    """VitaCLIP_vision_encoder.py:19-84 (same constructor keywords); forward(x) -> (cls_x (B,E), summary (B,D)), :102-132."""

    def __init__(self, input_size=(224, 224), num_frames=8, feature_dim=768, patch_size=(16, 16), num_heads=12,
                 num_layers=12, mlp_factor=4.0, act=None, embed_dim=512, use_summary_token=False, use_local_prompts=False,
                 use_global_prompts=False, num_global_prompts=8):
        super().__init__()
        if not (use_summary_token and use_local_prompts and use_global_prompts):
            raise NotImplementedError("only use_summary_token=use_local_prompts=use_global_prompts=True is a working "
                                      "configuration of the reference (VitaCLIP_vision_encoder.py:123-124,129)")
        if isinstance(input_size, int):
            input_size = (input_size, input_size)
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size)
        self.feature_dim = feature_dim
        self.num_frames = num_frames
        self._hip_shape = dict(size=input_size[0], P=patch_size[0], D=feature_dim, H=num_heads, layers=num_layers,
                               F=round(mlp_factor * feature_dim), E=embed_dim, G=num_global_prompts)
        self.patch_embed = _PatchEmbed(patch_size[0], feature_dim)
        self.num_patches = int(np.prod([x // y for x, y in zip(input_size, patch_size)])) + 1
        self.cls_token = nn.Parameter(torch.zeros([feature_dim]))
        self.pos_embed = nn.Parameter(torch.zeros([self.num_patches, feature_dim]))
        self.time_embed = nn.Parameter(torch.zeros([num_frames, feature_dim]))
        self.blocks = nn.ModuleList([TransformerEncoderLayer(feature_dim, num_heads, mlp_factor, num_frames, patch_size)
                                     for _ in range(num_layers)])
        self.ln_pre = LayerNorm(feature_dim)
        self.ln_post = LayerNorm(feature_dim)
        self.proj = nn.Parameter(feature_dim ** -0.5 * torch.randn(feature_dim, embed_dim))
        self.use_global_prompts = True
        self.num_global_prompts = num_global_prompts
        self.global_prompts = nn.Parameter(torch.zeros(num_layers, num_global_prompts, feature_dim))
        val = math.sqrt(6. / float(3 * patch_size[0] * patch_size[1] + feature_dim))
        nn.init.uniform_(self.global_prompts.data, -val, val)
        nn.init.normal_(self.cls_token, std=0.02)
        nn.init.normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.time_embed, std=0.02)

    def forward(self, x):
        return self._host().encode_video(x)


class _MHAParams(_Params):
    """Parameter layout of nn.MultiheadAttention (packed in_proj), VitaCLIP_text_encoder.py:71."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)


class ResidualAttentionBlock(_Params):
    def __init__(self, width):
        super().__init__()
        self.attn = _MHAParams(width)
        self.ln_1 = LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, width * 4)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(width * 4, width))]))
        self.ln_2 = LayerNorm(width)


class Transformer(_Params):
    def __init__(self, width, layers):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width) for _ in range(layers)])


class CLIPTextEncoder(_Encoder):
    """VitaCLIP_text_encoder.py:120-143.  positional_embedding / text_projection are
    ``torch.empty`` upstream (uninitialised without a checkpoint); here they get the CLIP
    initialisers so that a random-weight model is finite."""

    def __init__(self, embed_dim=512, context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                 transformer_layers=12):
        super().__init__()
        self._hip_shape = dict(W=transformer_width, TH=transformer_heads, TL=transformer_layers, L=context_length,
                               E=embed_dim, n_ctx=0)
        self.context_length = context_length
        self.transformer = Transformer(transformer_width, transformer_layers)
        self.heads = transformer_heads
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=transformer_width ** -0.5)

    def forward(self, prompts, tokenized_prompts):
        """(n, L, W) prompt embeddings + (n, L) token ids -> (n, E) (VitaCLIP_text_encoder.py:154-171)."""
        return self._host().encode_prompt_embeddings(prompts, tokenized_prompts)


class ContextualPromptLearner(nn.Module):
    """Knowledge-aware prompts, training/kapt_head.py:24-214, in the configuration that works upstream and that the
    reference's training scripts use (train_scripts/updrs_3cls_train_tulip.sh:31-36): `cntn_split_uni[_disc]`, one
    bias-free two-layer MLP per class (768 -> W/4 -> W, zero-initialised, kapt_head.py:127-133,161) applied to the
    class's KEPLER entity embeddings, one prompt per knowledge version (n_kv).  Reads the same files as upstream,
    relative to the working directory: ./data/ke_<type>/EntityEmb_<kv>.npy, simQdesc_<kv>.txt (kapt_head.py:60,94-112).
    This small module (n_cls * n_kv rows) is evaluated by torch; its output joins the context vectors that enter the
    HIP text tower, and its gradients come back through TextTowerFn."""

    def __init__(self, use_cntn, cntn_split, uni_mlp, use_disc, emb_dim, out_dim, inp_dim=768, n_cls=4, n_tokens=16,
                 cls_type="updrs", knowledge_version=("v0",), use_descriptor=False, token_wise_mlp=False):
        super().__init__()
        if token_wise_mlp:   # upstream: asserts against class_wise_mlp=True (kapt_head.py:63) and reads an unbound idc (:201)
            raise NotImplementedError("KAPT: token_wise_mlp=True does not run in the reference")
        if not (use_cntn and cntn_split and uni_mlp):
            # upstream: without `split` ke_v0_path is read before assignment (kapt_head.py:91), without `uni` a Python
            # list is .expand()-ed (:187-191), without `cntn` a 2-D ctx is concatenated with 3-D prefixes (text_encoder.py:325)
            raise NotImplementedError("KAPT: only text_prompt_init with cntn+split+uni (optionally disc) runs in the reference")
        assert len(knowledge_version) > 0 or use_descriptor, "No knowledge is specified."
        self.type = cls_type.lower().split("_")[0]
        self.n_cls, self.n_tokens = n_cls, n_tokens
        self.use_descriptor = use_descriptor
        self.updrs_ke_dir = f"./data/ke_{self.type}"
        assert os.path.isdir(self.updrs_ke_dir), f"{self.updrs_ke_dir} (KEPLER knowledge files) not found"
        embeds = torch.empty(n_cls, 0, inp_dim)
        cls_disc = [[] for _ in range(n_cls)]
        if use_descriptor:
            # per-class descriptors instead of one description per knowledge version (kapt_head.py:65-88): class c has as
            # many prompts as descriptor_<c>.txt has lines, each with its own entity embedding (descriptor_<c>.npy)
            np.load(os.path.join(self.updrs_ke_dir, "all.npy"), allow_pickle=False)     # read upstream as well (:68)
            embeds = []
            for idc in range(n_cls):
                with open(os.path.join(self.updrs_ke_dir, f"descriptor_{idc}.txt")) as f:
                    cls_disc[idc] = [ln.strip() for ln in f]
                ent = np.load(os.path.join(self.updrs_ke_dir, f"descriptor_{idc}.npy"), allow_pickle=False)
                assert ent.shape[0] == len(cls_disc[idc])
                embeds.append(torch.from_numpy(ent).float())
            knowledge_version = ()
        for kv in knowledge_version:
            ent = np.load(os.path.join(self.updrs_ke_dir, f"EntityEmb_{kv}.npy"), allow_pickle=False)[:n_cls]
            embeds = torch.cat([embeds, torch.from_numpy(ent).float().unsqueeze(1)], dim=1)
            if use_disc:
                with open(os.path.join(self.updrs_ke_dir, f"simQdesc_{kv}.txt")) as f:
                    lines = [ln.strip() for ln in f]
                for idc in range(n_cls):
                    cls_disc[idc].append(lines[idc])
            else:
                for idc in range(n_cls):
                    cls_disc[idc].append("")
        self.projector = nn.ModuleList([nn.Sequential(nn.Linear(inp_dim, emb_dim, bias=False), nn.ReLU(inplace=True),
                                                      nn.Linear(emb_dim, out_dim, bias=False)) for _ in range(n_cls)])
        for m in self.projector.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.weight)
        self.cntn_embeds = list(embeds)          # plain attribute like upstream (:166-167): n_cls x (n_kv, inp_dim)
        self.cls_disc = cls_disc

    def forward(self, ctx_prompt):
        """(n_cls, N, W) context parameters -> list of n_cls tensors (n_kv, N, W) (kapt_head.py:177-214)."""
        prompts = []
        for idc in range(self.n_cls):
            e = self.cntn_embeds[idc] = self.cntn_embeds[idc].to(ctx_prompt.device)
            emb = self.projector[idc](e).unsqueeze(1).expand(-1, self.n_tokens, -1)
            prompts.append(ctx_prompt[idc].unsqueeze(0) + emb)
        return prompts


class TextPromptLearner(_Params):
    """VitaCLIP_text_encoder.py:174-332: class-specific context vectors, plain ("X X ... name.") or knowledge-aware."""

    def __init__(self, classnames, text_model, num_prompts, prompts_init="", CSC=False, ctx_pos="end", cls_type="updrs",
                 knowledge_version=("v0",), use_descriptor=False, token_wise_mlp=False):
        super().__init__()
        ctx_init = prompts_init.lower()
        assert ctx_init == "" or set(ctx_init.split("_")).issubset({"split", "uni", "cntn", "disc"}), "Invalid prompt initialization"
        if ctx_pos != "end":
            raise NotImplementedError(f"Unsupported class token position: {ctx_pos}")
        n_cls, n_ctx = len(classnames), num_prompts
        ctx_dim = text_model.ln_final.weight.shape[0]
        self.knowledge_aware_prompt = ctx_init != ""
        names = [name.replace("_", " ") for name in classnames]
        if self.knowledge_aware_prompt:
            flags = set(ctx_init.split("_"))
            self.context_prompt_learner = ContextualPromptLearner(
                use_cntn="cntn" in flags, cntn_split="split" in flags, uni_mlp="uni" in flags, use_disc="disc" in flags,
                emb_dim=ctx_dim // 4, out_dim=ctx_dim, n_cls=n_cls, n_tokens=n_ctx, cls_type=cls_type,
                knowledge_version=list(knowledge_version), use_descriptor=use_descriptor, token_wise_mlp=token_wise_mlp)
            self.ctx = nn.Parameter(torch.zeros(n_cls, n_ctx, ctx_dim))                      # :218-220
            if use_descriptor:                                                               # :253-255
                texts = [[d + " " + names[c] for d in self.context_prompt_learner.cls_disc[c]] for c in range(n_cls)]
            else:
                texts = [[self.context_prompt_learner.cls_disc[c][k] + " " + names[c] for k in range(len(knowledge_version))]
                         for c in range(n_cls)]                                              # :256-258
        else:
            if not CSC:
                # upstream: a generic (n_ctx, dim) ctx is unsqueezed to (n_ctx,1,dim) and cannot be
                # concatenated (text_encoder.py:317,325-332) -> only CSC=True works there either.
                raise NotImplementedError("text_prompt_CSC=False is broken in the reference (SURVEY.md §7.5); use CSC=True")
            ctx = torch.empty(n_cls, n_ctx, ctx_dim)
            nn.init.normal_(ctx, std=0.02)
            self.ctx = nn.Parameter(ctx)
            texts = [[t] for t in prompt_texts(classnames, n_ctx)]
        # list of (n_kv, 77) int tensors, one per class, like upstream (:266-270)
        self.tokenized_prompts = [torch.from_numpy(np.concatenate([tokenize(t, text_model.context_length) for t in ts]))
                                  for ts in texts]
        assert max(int((tp == 49407).nonzero()[:, -1].max()) for tp in self.tokenized_prompts) <= 77, \
            "The tokenized prompt is too long"
        self.n_cls, self.n_ctx = n_cls, n_ctx
        # prompts per class: uniform (n_kv knowledge versions) or ragged (use_descriptor); n_kv is None when ragged
        self.kv_counts = [int(tp.shape[0]) for tp in self.tokenized_prompts]
        self.n_kv = self.kv_counts[0] if len(set(self.kv_counts)) == 1 else None
        self.class_token_position = ctx_pos

    def embedding_token_ids(self):
        """(n_cls*n_kv, L) token ids in the order their EMBEDDINGS sit in the prompt: [SOS | n_ctx context slots | rest].
        Plain prompts carry "X" placeholders at the context slots, so this is the tokenised text itself (:300);
        knowledge-aware prompts have no placeholders - upstream inserts the context after SOS and drops the last n_ctx
        positions (`embedding[:, 1:-n_ctx]`, :298), i.e. every later token moves n_ctx places to the right.  The EOT
        look-up keeps using the un-shifted ids (:169 with tokenized_prompts) - reproduced as is."""
        tok = torch.cat(self.tokenized_prompts)
        if not self.knowledge_aware_prompt:
            return tok
        eff = tok.clone()
        eff[:, 1 + self.n_ctx:] = tok[:, 1:tok.shape[1] - self.n_ctx]
        return eff

    def full_context(self):
        """(n_cls*n_kv, n_ctx, W) context rows of every prompt (text_encoder.py:310-317)."""
        if self.knowledge_aware_prompt:
            return torch.cat(self.context_prompt_learner(self.ctx), dim=0)
        return self.ctx


class _HipHost:
    """Weight packing and tower launches on libgava_hip.so, shared by VitaCLIP and by stand-alone encoders.  Expects
    `visual` and / or `textual` attributes (parameter containers), `num_frames`, and the nn.Module parameter iterators."""

    def _hip_init(self, shape, operand_dtype=None):
        operand_dtype = operand_dtype or os.environ.get("GAVA_PREC", "fp16")
        self.prec, self.w_lo = self._parse_operand_dtype(operand_dtype)
        # text tower GEMMs in split precision (hi+lo operands, 3 MFMA passes): the text side is <1 % of
        # the work at the headline configs but dominates the logits error at plain 16-bit operands
        self.text_split_precision = os.environ.get("GAVA_TEXT_SPLIT", "1") != "0"
        # with it (inference): q/k/v leave their GEMM in fp32 and the softmax core runs in fp32 - the text features then carry
        # no 16-bit rounding at all (GAVA_TEXT_ATTN_F32=0: the MFMA attention on 16-bit q/k/v, as in rounds 1-2)
        self.text_attention_fp32 = os.environ.get("GAVA_TEXT_ATTN_F32", "1") != "0"
        # eval-time text-feature cache: in eval mode the text tower is input
        # independent; opt-in because a benchmark must not skip work.  Invalidated by any parameter update.
        self.cache_text_features = False
        self.text_on_side_stream = os.environ.get("GAVA_TEXT_STREAM", "1") != "0"
        # inference: LayerNorm folded into the qkv / fc1 GEMMs (two row passes per block less); GAVA_LN_FOLD=0 turns it off
        self.fold_layernorm = os.environ.get("GAVA_LN_FOLD", "1") != "0"
        self.trim_text_rows = os.environ.get("GAVA_TEXT_TRIM", "1") != "0"   # skip the rows behind the last EOT (see _pack)
        # inference, opt-in: the last block's B*T CLS rows (the only ones that reach the outputs) in split precision.
        # Measured at c1 (tools/archive/r2_diag.py): video-feature rms error 2.06e-5 with, 2.10e-5 without - the error of the
        # features is made in the fp16 K/V and in the eleven blocks before, not here - so it is off by default.
        self.split_last_block = os.environ.get("GAVA_LAST_SPLIT", "0") != "0"
        self.text_rows_per_prompt = shape.get("L", 77)
        # training: keep the backward's activations (~21 GB at B = 64, T = 8) instead of recomputing them per block, as
        # long as they fit this budget; beyond it the backward recomputes from the block inputs only
        self.keep_activation_bytes = int(float(os.environ.get("GAVA_KEEP_ACT_GB", "96")) * 2 ** 30)
        self._text_stream = None
        self._text_cache = None
        self.gather_across_ranks = True     # RCCL all-gather of clip embeddings when world_size > 1
        self.shard_text_across_ranks = True  # eval, world_size > 1: each rank encodes a slice of the prompts (+ all-gather)
        self.debug_taps = False             # keep per-layer CLS rows of the last forward
        self._shape = dict(shape)
        self._packed = None
        self._packed_key = None
        self._ws = {}
        self.last = {}

    # ---- weight packing -----------------------------------------------------------------------
    @staticmethod
    def _parse_operand_dtype(name):
        """"fp16" | "bf16", optionally "+wlo" / "+wlo8": the weight-lo pass of the vision tower in inference (include/gava_hip.h,
        gava_gemm_args.w_lo; DESIGN.md "Numerics") - the 16-bit rounding of the frozen weights is what misses north_star's 1e-3
        on some models, so every vision GEMM also multiplies by W - h16(W): "+wlo" as a second 16-bit k-loop over the
        re-read activations, "+wlo8" at 8 bits (block-scaled MFMA, twice the rate) for the four full-width GEMMs of a block."""
        base, _, mode = name.partition("+")
        if mode not in ("", "wlo", "wlo8") or base not in hip.PREC_NAMES:
            raise ValueError(f"operand_dtype must be fp16 | bf16 [+wlo | +wlo8], got {name!r}")
        if mode == "wlo8" and hip.PREC_NAMES[base] != hip.PREC_F16:
            raise ValueError("the 8-bit weight-lo pass takes its bf8 activations from fp16 operands: use fp16+wlo8")
        return hip.PREC_NAMES[base], {"": 0, "wlo": 1, "wlo8": 2}[mode]

    def set_operand_dtype(self, name: str):
        self.prec, self.w_lo = self._parse_operand_dtype(name)
        self._packed = None

    _PASS_THROUGH = ("prompt_learner.", "logit_scale", "global_prompts", "local_prompts", "token_embedding",
                     "pos_embed", "time_embed", "positional_embedding", "cls_token", "sum_proj", "tf_project",
                     "memory_project")

    def _pack_key(self):
        """Changes when a packed 16-bit copy goes stale.  fp32 pass-through parameters (prompts, embeddings, LN
        affines, biases: the structs hold pointers into their own storage) only count by address, so an optimizer
        step on the prompt parameters does not re-convert 180 M frozen weights."""
        ver, addr = 0, 0
        for name, p in self.named_parameters():
            addr ^= p.data_ptr()
            if p.dim() >= 2 and not p.requires_grad and not any(k in name for k in self._PASS_THROUGH):
                ver += p._version
        if self.fold_layernorm and hasattr(self, "visual"):   # norm1 / norm2 affines and the qkv / fc1 biases are baked into the folded copies
            for blk in self.visual.blocks:
                for q in (blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias, blk.attn.q_proj.bias,
                          blk.attn.k_proj.bias, blk.attn.v_proj.bias, blk.mlp.fc1.bias):
                    ver += q._version
        ps = next(self.parameters())
        return (self.prec, self.w_lo, self.text_split_precision, self.trim_text_rows, self.fold_layernorm, self.split_last_block,
                ps.device, addr, ver)

    def _summary_weight_versions(self):
        """Versions of the only TRAINABLE weights that have 16-bit copies (summary_attn_layer projections): an optimizer
        step refreshes just those copies in place (same device pointers) instead of re-converting every frozen weight."""
        if not hasattr(self, "visual"):
            return []
        return [tuple(w._version for w in (b.summary_attn_layer.q_proj.weight, b.summary_attn_layer.k_proj.weight,
                                           b.summary_attn_layer.v_proj.weight, b.summary_attn_layer.out_proj.weight,
                                           b.summary_attn_layer.q_proj.bias, b.summary_attn_layer.k_proj.bias,
                                           b.summary_attn_layer.v_proj.bias))
                for b in self.visual.blocks]

    def _refresh_summary_weights(self, packed):
        cur = self._summary_weight_versions()
        if cur == packed["summary_ver"]:
            return
        for i, blk in enumerate(self.visual.blocks):
            if cur[i] != packed["summary_ver"][i]:
                s_ = blk.summary_attn_layer
                packed["w_sqkv"][i].copy_(self._h16(torch.cat([s_.q_proj.weight, s_.k_proj.weight, s_.v_proj.weight], 0)))
                packed["w_sout"][i].copy_(self._h16(s_.out_proj.weight))
                packed["b_sqkv"][i].copy_(torch.cat([s_.q_proj.bias, s_.k_proj.bias, s_.v_proj.bias], 0).detach().float())
                if "w_sqkv_wlo" in packed:
                    packed["w_sqkv_wlo"][i].copy_(self._hl16(torch.cat([s_.q_proj.weight, s_.k_proj.weight, s_.v_proj.weight], 0)))
                    packed["w_sout_wlo"][i].copy_(self._hl16(s_.out_proj.weight))
                    packed["b_sqkv_wlo"][i].copy_(torch.cat([s_.q_proj.bias, s_.k_proj.bias, s_.v_proj.bias], 0).detach().float())
        packed["summary_ver"] = cur

    def _pack_vision_backward(self):
        from . import training
        key = self._pack_key()
        if getattr(self, "_bwd_pack_v", None) is None or self._bwd_pack_v[0] != key:
            self._bwd_pack_v = (key, training.pack_vision_backward(self))
        training.refresh_vision_backward(self, self._bwd_pack_v[1])
        return self._bwd_pack_v[1]

    def _pack_text_backward(self):
        from . import training
        key = self._pack_key()
        if getattr(self, "_bwd_pack", None) is None or self._bwd_pack[0] != key:
            self._bwd_pack = (key, training.pack_text_backward(self))
        return self._bwd_pack[1]

    def _h16(self, t):
        return hip.convert_h16(t.detach().float(), self.prec)

    def _f32(self, t):
        return t.detach().float().contiguous()

    def _pack(self):
        key = self._pack_key()
        if self._packed is not None and self._packed_key == key:
            self._refresh_summary_weights(self._packed)
            return self._packed
        sh = self._shape
        keep = []  # tensors referenced by raw pointers in the structs
        w_sqkv_t, w_sout_t, b_sqkv_t = [], [], []

        def K(t):
            keep.append(t)
            return C.c_void_p(t.data_ptr())

        packed = dict(keep=keep, summary_ver=self._summary_weight_versions())
        if hasattr(self, "visual"):
            self._pack_vision(packed, K, w_sqkv_t, w_sout_t, b_sqkv_t)
        if hasattr(self, "textual"):
            self._pack_text(packed, K)
        self._packed, self._packed_key = packed, key
        return packed

    def _hl16(self, t):
        """[N][K] fp32 -> [N][2K] h16 = [W_hi | W_lo], W_lo = h16(W - W_hi): the weight side of gava_gemm_args.w_lo = 1."""
        t = t.detach().float().contiguous()
        hi = hip.convert_h16(t, self.prec)
        return torch.cat([hi, hip.convert_h16(t - hi.float(), self.prec)], dim=1).contiguous()

    def _pack_vision(self, packed, K, w_sqkv_t, w_sout_t, b_sqkv_t):
        self._pack_vision_set(packed, K, w_sqkv_t, w_sout_t, b_sqkv_t, 0)
        if self.w_lo:
            # inference with the weight-lo pass: a second set of structs whose 16-bit weights are [W_hi | W_lo]; the training
            # drivers keep using the plain set (fp32 pass-through tensors are shared by address: K() dedups nothing, they are views)
            self._pack_vision_set(packed, K, [], [], [], self.w_lo)

    def _pack_vision_set(self, packed, K, w_sqkv_t, w_sout_t, b_sqkv_t, w_lo):
        sh, v = self._shape, self.visual
        h16 = self._hl16 if w_lo else self._h16
        Kp = (3 * sh["P"] ** 2 + 63) // 64 * 64
        wpatch = v.patch_embed.proj.weight.detach().float().reshape(sh["D"], -1)
        if Kp != wpatch.shape[1]:
            wpatch = F.pad(wpatch, (0, Kp - wpatch.shape[1]))
        vis = dict(w_patch=K(h16(wpatch)), b_patch=K(self._f32(v.patch_embed.proj.bias)),
                   cls_token=K(self._f32(v.cls_token)), pos_embed=K(self._f32(v.pos_embed)),
                   lnpre_g=K(self._f32(v.ln_pre.weight)), lnpre_b=K(self._f32(v.ln_pre.bias)),
                   lnpost_g=K(self._f32(v.ln_post.weight)), lnpost_b=K(self._f32(v.ln_post.bias)),
                   w_proj=K(hip.split_pack_weight(v.proj.detach().float().t(), self.prec)))
        layers = (hip.VisionLayer * sh["layers"])()
        layers8 = (hip.VisionLayer8 * sh["layers"])() if (w_lo == 2 and self.fold_layernorm) else None
        for i, blk in enumerate(v.blocks):
            a, s = blk.attn, blk.summary_attn_layer
            L = layers[i]
            if layers8 is not None:
                _, w8, e8, _s = hip.pack_w8(blk.mlp.fc2.weight, self.prec)
                layers8[i].w_fc28, layers8[i].fc2_exp = K(w8), e8
            L.w_qkv = K(h16(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)))
            L.b_qkv = K(self._f32(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)))
            L.w_out, L.b_out = K(h16(a.out_proj.weight)), K(self._f32(a.out_proj.bias))
            L.w_fc1, L.b_fc1 = K(h16(blk.mlp.fc1.weight)), K(self._f32(blk.mlp.fc1.bias))
            L.w_fc2, L.b_fc2 = K(h16(blk.mlp.fc2.weight)), K(self._f32(blk.mlp.fc2.bias))
            L.ln1_g, L.ln1_b = K(self._f32(blk.norm1.weight)), K(self._f32(blk.norm1.bias))
            L.ln2_g, L.ln2_b = K(self._f32(blk.norm2.weight)), K(self._f32(blk.norm2.bias))
            if self.fold_layernorm:
                # LayerNorm folded into the consumer GEMM (inference driver): W' = h16(gamma * W),
                # s = row sums of the ROUNDED W', t = W beta + b in fp32
                wq = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).detach().float()
                bq = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).detach().float()
                for tag, w0, b0, nrm in (("qkv", wq, bq, blk.norm1), ("fc1", blk.mlp.fc1.weight.detach().float(),
                                                                     blk.mlp.fc1.bias.detach().float(), blk.norm2)):
                    gam, bet = nrm.weight.detach().float(), nrm.bias.detach().float()
                    wf = h16(w0 * gam)
                    if layers8 is not None:      # the 8-bit lo operand of the folded weight (gava_vision_layer8)
                        _, w8, e8, s8 = hip.pack_w8(w0 * gam, self.prec)
                        setattr(layers8[i], f"w_{tag}_fold8", K(w8))
                        setattr(layers8[i], f"{tag}_fold_s8", K(s8))
                        setattr(layers8[i], f"{tag}_fold_exp", e8)
                    setattr(L, f"w_{tag}_fold", K(wf))
                    # ([W_hi | W_lo]: the row sum over both halves = the sum of the weight the kernel multiplies by)
                    setattr(L, f"{tag}_fold_s", K(wf.float().sum(1).contiguous()))
                    setattr(L, f"{tag}_fold_t", K((w0.double() @ bet.double() + b0.double()).float().contiguous()))
            if self.split_last_block and i == len(v.blocks) - 1:
                L.w_q_split = K(hip.split_pack_weight(a.q_proj.weight, self.prec))
                L.w_out_split = K(hip.split_pack_weight(a.out_proj.weight, self.prec))
                L.w_fc1_split = K(hip.split_pack_weight(blk.mlp.fc1.weight, self.prec))
                L.w_fc2_split = K(hip.split_pack_weight(blk.mlp.fc2.weight, self.prec))
            L.w_cls, L.b_cls = K(h16(blk.cls_proj.weight)), K(self._f32(blk.cls_proj.bias))
            L.sln_g, L.sln_b = K(self._f32(blk.summary_ln.weight)), K(self._f32(blk.summary_ln.bias))
            w_sqkv_t.append(h16(torch.cat([s.q_proj.weight, s.k_proj.weight, s.v_proj.weight], 0)))
            w_sout_t.append(h16(s.out_proj.weight))
            L.w_sqkv, L.w_sout = K(w_sqkv_t[-1]), K(w_sout_t[-1])
            b_sqkv_t.append(self._f32(torch.cat([s.q_proj.bias, s.k_proj.bias, s.v_proj.bias], 0)))
            L.b_sqkv = K(b_sqkv_t[-1])
            L.b_sout = K(self._f32(s.out_proj.bias))
            L.local_prompts = K(self._f32(blk.local_prompts[0]))
            L.global_prompts = K(self._f32(v.global_prompts[i]))
        if w_lo:
            packed.update(vis_wlo=vis, vis_layers_wlo=layers, w_sqkv_wlo=w_sqkv_t, w_sout_wlo=w_sout_t, b_sqkv_wlo=b_sqkv_t,
                          vis_layers8=layers8)
        else:
            packed.update(vis=vis, vis_layers=layers, w_sqkv=w_sqkv_t, w_sout=w_sout_t, b_sqkv=b_sqkv_t)

    def _pack_text(self, packed, K):
        sh = self._shape
        if True:
            t = self.textual
            tlayers = (hip.TextLayer * sh["TL"])()
            tw = (lambda w: hip.split_pack_weight(w, self.prec)) if self.text_split_precision else self._h16
            for i, blk in enumerate(t.transformer.resblocks):
                L = tlayers[i]
                L.w_qkv, L.b_qkv = K(tw(blk.attn.in_proj_weight)), K(self._f32(blk.attn.in_proj_bias))
                L.w_out, L.b_out = K(tw(blk.attn.out_proj.weight)), K(self._f32(blk.attn.out_proj.bias))
                L.w_fc, L.b_fc = K(tw(blk.mlp.c_fc.weight)), K(self._f32(blk.mlp.c_fc.bias))
                L.w_proj, L.b_proj = K(tw(blk.mlp.c_proj.weight)), K(self._f32(blk.mlp.c_proj.bias))
                L.ln1_g, L.ln1_b = K(self._f32(blk.ln_1.weight)), K(self._f32(blk.ln_1.bias))
                L.ln2_g, L.ln2_b = K(self._f32(blk.ln_2.weight)), K(self._f32(blk.ln_2.bias))
            packed.update(txt=dict(token_embedding=K(self._f32(t.token_embedding.weight)),
                                   positional_embedding=K(self._f32(t.positional_embedding)),
                                   lnf_g=K(self._f32(t.ln_final.weight)), lnf_b=K(self._f32(t.ln_final.bias)),
                                   w_tproj=K(tw(t.text_projection.detach().float().t().contiguous()))),
                          txt_layers=tlayers)
            if not getattr(self, "use_text_prompt_learning", False):
                return          # a bare text tower (direct calls only): no prompt learner, no token table
            dev = t.token_embedding.weight.device
            tok0 = torch.cat(self.tokenized_prompts).to(device=dev)
            eot_col = (tok0 == t.vocab_size - 1).nonzero()[:, -1]
            assert eot_col.numel() == tok0.shape[0], "every prompt must contain exactly one EOT token"
            # Causal attention: the EOT row - the only one the text features read (text_encoder.py:169) - depends on the
            # rows up to itself only, and LayerNorm / MLP are row-wise.  Rows behind the last EOT of any prompt are
            # dead work (the reference pads every prompt to 77): the tower runs on the first L_eff positions, results
            # identical.  ("X X .. name." prompts: ~15-20 of 77.)
            L_eff = min(sh["L"], max(int(eot_col.max()) + 1, 1 + sh["n_ctx"]))
            self.text_rows_per_prompt = L_eff if self.trim_text_rows else sh["L"]
            L_eff = self.text_rows_per_prompt
            eot = (torch.arange(tok0.shape[0], device=dev) * L_eff + eot_col).to(torch.int32).contiguous()
            tok = self.prompt_learner.embedding_token_ids()[:, :L_eff].to(device=dev, dtype=torch.int32).contiguous()
            packed.update(tokens=tok, eot=eot)

    def _workspace(self, tag, nbytes, device):
        ws = self._ws.get(tag)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._ws[tag] = ws
        return ws

    # ---- encoders -----------------------------------------------------------------------------
    def encode_video(self, x, saved=None, kept=None, clips=None):
        """CLIPVisionEncoder.forward on the HIP path -> (cls_x (B,E), summary (B,D)), fp32.
        saved: optional fp32 [layers+2, B*T*(n+1), D] that receives what the backward recomputes from;
        kept: optional dict of per-block activation buffers (training.alloc_kept) filled by gava_vision_forward_keep;
        clips: instead of x, the decoded uint8 videos as (descriptor tensor, B, T, normalisation table, device) - the patch embedding
        then reads the frames themselves (hip.clip_descriptors)."""
        if clips is None and not x.is_cuda:
            raise hip.GavaError("VitaCLIP (gava_clip_amd) runs on the HIP device only: move the model and the "
                                "input with .cuda(); there is no CPU fallback")
        lib = hip.load()
        pk, sh = self._pack(), self._shape
        if clips is not None:
            desc, B, T, lut, dev = clips
            x = desc            # only its device is used below
        else:
            B, Cc, T, Hh, Ww = x.shape
            assert Cc == 3 and Hh == sh["size"] and Ww == sh["size"], "input must be (B,3,T,size,size)"
            x = x.detach().float().contiguous()
        if B == 0:   # an empty batch is not an error upstream (every op of the reference accepts zero clips): empty features
            self.last["summary"] = torch.zeros(0, sh["D"], device=x.device)
            return torch.zeros(0, sh["E"], device=x.device), self.last["summary"]
        te = self.visual.time_embed.detach().float()
        if T != te.size(0):  # VitaCLIP_vision_encoder.py:91-95
            te = F.interpolate(te.unsqueeze(0).transpose(1, 2), size=(T), mode='nearest').transpose(1, 2).squeeze(0)
        te = te.contiguous()
        m = hip.VisionModel()
        m.B, m.T_in, m.T_model = B, T, self.num_frames
        for k in ("size", "P", "D", "H", "layers", "F", "E", "G"):
            setattr(m, k, sh[k])
        m.prec = self.prec
        wl = self.w_lo if (saved is None and kept is None) else 0      # the weight-lo pass is an inference mode
        for k, val in pk["vis_wlo" if wl else "vis"].items():
            setattr(m, k, val)
        m.time_embed = C.c_void_p(te.data_ptr())
        m.layer = C.cast(pk["vis_layers_wlo" if wl else "vis_layers"], C.POINTER(hip.VisionLayer))
        m.w_lo = wl
        if wl == 2 and pk.get("vis_layers8") is not None:
            m.layer8 = C.cast(pk["vis_layers8"], C.c_void_p)
        elif wl == 2:
            m.w_lo = 1            # no folded weights packed (fold_layernorm off): every lo product stays 16-bit
        if clips is not None:
            m.clips, m.clip_lut = hip.ptr(desc), hip.ptr(lut)
        nbytes = lib.gava_vision_workspace_bytes(C.byref(m))
        # (measurement / tests) does the inference driver keep the residual stream of this batch as a 16-bit pair?
        self.last["pair_stream"] = bool(saved is None and kept is None and lib.gava_vision_pair_stream(C.byref(m)))
        if nbytes == 0:
            raise hip.GavaError(f"unsupported vision shape: B={B} T={T} num_frames={self.num_frames} {sh}")
        ws = self._workspace("vision", nbytes, x.device)
        cls_x = torch.empty(B, sh["E"], dtype=torch.float32, device=x.device)
        summary = torch.empty(B * T // self.num_frames, sh["D"], dtype=torch.float32, device=x.device)
        dbg = torch.empty(sh["layers"], B * T, sh["D"], dtype=torch.float32, device=x.device) if self.debug_taps else None
        # the drivers key their side stream by the CURRENT device: make it the input's
        with torch.cuda.device(x.device):
            if kept is not None:
                sv = hip.VisionSaved(hip.ptr(kept["e0"]), hip.ptr(kept["x"]), hip.ptr(kept["x1"]), hip.ptr(kept["qkv"]),
                                     hip.ptr(kept["pre"]), hip.ptr(kept["sidekv"]), hip.ptr(kept.get("last_q")),
                                     hip.ptr(kept.get("last_x1")), hip.ptr(kept.get("last_pre")))
                hip.check(lib.gava_vision_forward_keep(C.byref(m), None if clips is not None else hip.ptr(x), hip.ptr(cls_x), hip.ptr(summary), C.byref(sv),
                                                       hip.ptr(ws), ws.numel(), hip.stream_ptr(x.device)), "gava_vision_forward_keep")
            else:
                hip.check(lib.gava_vision_forward_train(C.byref(m), None if clips is not None else hip.ptr(x), hip.ptr(cls_x), hip.ptr(summary), hip.ptr(dbg),
                                                        hip.ptr(saved), hip.ptr(ws), ws.numel(), hip.stream_ptr(x.device)),
                          "gava_vision_forward")
        self.last["cls_rows"] = dbg
        return cls_x, summary

    def _overlapping_stream(self, device):
        """A new stream that really runs BESIDE the current one.  HIP multiplexes streams onto a few hardware queues in creation
        order; a stream that shares the current stream's queue is served in order with it, and the text tower then runs serially
        in front of the vision tower (23.0 instead of 20.7 ms per c2 forward, profiles/r04_text_cost.txt) - which queue a new stream
        gets depends on how many the process has created before (data loaders, DDP).  So the candidate is tried once: a ~1 ms spin
        on the current stream, a one-element fill on the candidate; a candidate whose fill is not done long before the spin ends is
        set aside (kept alive, so that the next one gets another queue) and the next is tried, four at most."""
        if torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device=device)
        main = torch.cuda.current_stream(device)
        tried = []
        with torch.cuda.device(device):
            for _ in range(4):
                cand = torch.cuda.Stream(device=device)
                tried.append(cand)
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                with torch.cuda.stream(cand):
                    torch.zeros(1, device=device)                # first use of the stream (its queue is set up here), untimed
                torch.cuda.synchronize(device)
                e0.record(main)
                torch.cuda._sleep(2_000_000)
                e1.record(main)
                with torch.cuda.stream(cand):
                    torch.zeros(1, device=device)
                    e2.record(cand)
                torch.cuda.synchronize(device)
                if e2.elapsed_time(e1) > 0.3 * e0.elapsed_time(e1):     # the fill was done while the spin still had most of its time to go
                    break
        self._queue_sharing_streams = tried[:-1]
        self.last["text_stream_candidates"] = len(tried)
        return tried[-1]

    def _text_shard(self, n):
        """(lo, hi, rows per rank, world) when the prompts are sharded over the ranks, else None (SURVEY.md 8f row 2:
        every rank would otherwise run the whole text tower redundantly - 2.4 TF per forward at 400 classes)."""
        import torch.distributed as dist
        if not (self.shard_text_across_ranks and self.gather_across_ranks and not torch.is_grad_enabled()
                and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        world, rank = dist.get_world_size(), dist.get_rank()
        if n < 2 * world:
            return None
        per = (n + world - 1) // world
        return min(n, rank * per), min(n, (rank + 1) * per), per, world

    def encode_text(self):
        """prompt_learner() + textual(...) for all classes in one batch -> (C, E) fp32.  With several ranks (eval) each
        rank encodes a contiguous slice of the prompts and the rows are all-gathered: prompts are independent, so the
        result is the same as every rank encoding all of them."""
        pk, sh = self._pack(), self._shape
        n = pk["tokens"].shape[0]
        shard = self._text_shard(n)
        if shard is None:
            return self._encode_text_rows(0, n)
        lo, hi, per, world = shard
        part = torch.zeros(per, sh["E"], dtype=torch.float32, device=pk["tokens"].device)
        if hi > lo:
            part[:hi - lo] = self._encode_text_rows(lo, hi)
        # rank r holds prompts [r*per, (r+1)*per): the rank-major concatenation is already in prompt order, padding last
        return self._gather(part)[:n].contiguous()

    def _text_model(self, n, L, n_ctx):
        pk, sh = self._pack(), self._shape
        m = hip.TextModel()
        m.n_prompts, m.L, m.W, m.H, m.layers = n, L, sh["W"], sh["TH"], sh["TL"]
        m.E, m.n_ctx, m.prec = sh["E"], n_ctx, self.prec
        m.split = int(self.text_split_precision)
        m.attn_f32 = int(self.text_split_precision and self.text_attention_fp32)
        for k, val in pk["txt"].items():
            setattr(m, k, val)
        m.layer = C.cast(pk["txt_layers"], C.POINTER(hip.TextLayer))
        return m

    def _run_text(self, m, tok, ctx, eot, device):
        lib = hip.load()
        nbytes = lib.gava_text_workspace_bytes(C.byref(m))
        if nbytes == 0:
            raise hip.GavaError(f"unsupported text shape: n={m.n_prompts} L={m.L} {self._shape}")
        ws = self._workspace("text", nbytes, device)
        out = torch.empty(m.n_prompts, self._shape["E"], dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            hip.check(lib.gava_text_forward(C.byref(m), hip.ptr(tok), hip.ptr(ctx), hip.ptr(eot), hip.ptr(out),
                                            hip.ptr(ws), ws.numel(), hip.stream_ptr(device)), "gava_text_forward")
        return out

    def _encode_text_rows(self, lo, hi):
        """The text tower on prompts [lo, hi)."""
        pk, sh = self._pack(), self._shape
        L = self.text_rows_per_prompt
        tok = pk["tokens"][lo:hi].contiguous()
        ctx = self.prompt_learner.full_context().detach().float()[lo:hi].contiguous()
        eot = (pk["eot"][lo:hi] - lo * L).to(torch.int32).contiguous()      # flat row index n*L + column, rebased to the slice
        return self._run_text(self._text_model(tok.shape[0], L, sh["n_ctx"]), tok, ctx, eot, tok.device)

    def encode_prompt_embeddings(self, prompts, tokenized_prompts):
        """CLIPTextEncoder.forward(prompts, tokenized_prompts) (VitaCLIP_text_encoder.py:154-171) on ready-made prompt
        embeddings (n, L, W) - the call of evaluation/zero_shot.py:75-76 and utils/prepare_embedding.py - through the
        HIP text tower's direct mode (no token table, no context splice).  Inference only: training goes through
        VitaCLIP.forward, whose TextTowerFn owns the backward."""
        if not prompts.is_cuda:
            raise hip.GavaError("CLIPTextEncoder (gava_clip_amd) runs on the HIP device only; there is no CPU fallback")
        if torch.is_grad_enabled() and prompts.requires_grad:
            raise hip.GavaError("CLIPTextEncoder.forward called stand-alone has no backward: train through VitaCLIP.forward")
        self._pack()
        tok = tokenized_prompts.to(prompts.device)
        n, L, W = prompts.shape
        assert tuple(tok.shape) == (n, L) and W == self._shape["W"] and L <= self._shape["L"]
        hit = (tok == self.textual.vocab_size - 1).nonzero()
        assert hit.shape[0] == n, "every prompt must contain exactly one EOT token"      # text_encoder.py:169
        eot_col = hit[:, -1]
        L_eff = int(eot_col.max()) + 1 if self.trim_text_rows else L        # rows behind the last EOT are dead work (causal)
        x = prompts.detach().float()[:, :L_eff].contiguous()
        eot = (torch.arange(n, device=tok.device) * L_eff + eot_col).to(torch.int32).contiguous()
        return self._run_text(self._text_model(n, L_eff, 0), None, x, eot, prompts.device)


class _StandaloneHost(_HipHost):
    """Host of an encoder that was constructed on its own (evaluation/zero_shot.py:42-52 builds a bare CLIPTextEncoder,
    evaluation/iwa.py a bare CLIPVisionEncoder): packs that encoder's weights and launches its tower."""

    def __init__(self, enc):
        self._enc = weakref.ref(enc)
        self.use_text_prompt_learning = False
        if isinstance(enc, CLIPVisionEncoder):
            self.num_frames = enc.num_frames
        self._hip_init(enc._hip_shape)

    def __getattr__(self, name):      # `visual` / `textual` resolve to the encoder without keeping it alive (no cycle)
        if name in ("visual", "textual"):
            enc = self.__dict__["_enc"]()
            if enc is not None and isinstance(enc, CLIPVisionEncoder if name == "visual" else CLIPTextEncoder):
                return enc
        raise AttributeError(name)

    def named_parameters(self):
        return self._enc().named_parameters()

    def parameters(self):
        return self._enc().parameters()


# ---------------------------------------------------------------------------------------------
class VitaCLIP(nn.Module, _HipHost):

    def __init__(
        self,
        backbone_path: str = '',
        input_size: Tuple[int, int] = (224, 224),
        num_frames: int = 16,
        use_fp16: bool = False,
        cls_type: str = 'updrs',
        num_classes: int = 4,
        feature_dim: int = 768,
        patch_size: Tuple[int, int] = (16, 16),
        num_heads: int = 12,
        num_layers: int = 12,
        mlp_factor: float = 4.0,
        embed_dim: int = 512,
        use_summary_token: bool = False,
        use_local_prompts: bool = False,
        use_global_prompts: bool = False,
        num_global_prompts: int = 8,
        use_text_prompt_learning: bool = False,
        text_context_length: int = 77,
        text_vocab_size: int = 49408,
        text_transformer_width: int = 512,
        text_transformer_heads: int = 8,
        text_transformer_layers: int = 12,
        text_num_prompts: int = 8,
        text_prompt_pos: str = 'end',
        text_prompt_init: str = '',
        text_prompt_CSC: bool = False,
        text_prompt_classes_path: str = '',
        knowledge_version: List[str] = ['v0'],
        use_descriptor: bool = False,
        token_wise_mlp: bool = False,
        zeroshot_evaluation: bool = False,
        zeroshot_text_features_path: str = '',
        use_support_memory: bool = False,
        detach_features: bool = False,
        memory_batch_size: int = 64,
        add_nte: bool = False,
        use_sigmoid_loss: bool = False,
        *,
        operand_dtype: str = None,
    ):
        super().__init__()
        if not (use_summary_token and use_local_prompts and use_global_prompts):
            # upstream's non-global-prompt branch assigns the block's tuple to x and leaves
            # `summary` undefined (VitaCLIP_vision_encoder.py:123-124,129): it cannot run.
            raise NotImplementedError("only use_summary_token=use_local_prompts=use_global_prompts=True is a "
                                      "working configuration of the reference (SURVEY.md §7.5)")
        if isinstance(input_size, int):
            input_size = (input_size, input_size)
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size)
        self.fp16 = use_fp16
        self.num_frames = num_frames
        self.num_classes = num_classes
        self.text_context_length = text_context_length
        self.text_transformer_width = text_transformer_width
        self.use_summary_token = use_summary_token
        self.use_sigmoid_loss = use_sigmoid_loss
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.logit_bias = None
        self.zeroshot_evaluation = zeroshot_evaluation
        if self.zeroshot_evaluation:
            self.text_features = torch.load(zeroshot_text_features_path, map_location='cpu',
                                            weights_only=True)['text_features']
        self.visual = CLIPVisionEncoder(input_size=input_size, num_frames=num_frames, feature_dim=feature_dim,
                                        patch_size=patch_size, num_heads=num_heads, num_layers=num_layers,
                                        mlp_factor=mlp_factor, embed_dim=embed_dim, use_summary_token=True,
                                        use_local_prompts=True, use_global_prompts=True,
                                        num_global_prompts=num_global_prompts)
        self.use_text_prompt_learning = use_text_prompt_learning
        if self.use_text_prompt_learning:
            self.textual = CLIPTextEncoder(embed_dim, text_context_length, text_vocab_size, text_transformer_width,
                                           text_transformer_heads, text_transformer_layers)
        if backbone_path:
            ckpt = torch.load(backbone_path, map_location='cpu', weights_only=True)
            self.load_state_dict(ckpt, strict=False)

        self.use_support_memory = use_support_memory
        self.memoty_batch_size = memory_batch_size
        self.detach_features = detach_features
        self.add_nte = add_nte
        if self.add_nte:
            self.sum_proj = nn.Linear(feature_dim, embed_dim)
            self.logit_scale_vm = nn.Parameter(torch.ones([]) * (np.log(10.) if use_sigmoid_loss else 100.))
        if self.use_support_memory:
            def _mlp():
                return nn.Sequential(nn.Linear(embed_dim, embed_dim // 4), nn.Tanh(), nn.Linear(embed_dim // 4, embed_dim // 8))
            self.tf_project = _mlp()
            self.memory_project = nn.ModuleList([_mlp() for _ in range(num_classes)])
            if use_sigmoid_loss:
                self.logit_scale_mt = nn.Parameter(torch.ones([]) * np.log(10.))
                self.logit_bias_mt = nn.Parameter(torch.ones([]) * -10.)
            else:
                self.logit_scale_mt = nn.Parameter(torch.ones([]) * 100.)
                self.logit_bias_mt = None
        if use_sigmoid_loss:
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(np.log(10.)))
            self.logit_bias = nn.Parameter(torch.ones([]) * -10.)

        if self.use_text_prompt_learning:
            classes = read_class_names(text_prompt_classes_path)
            self.prompt_learner = TextPromptLearner(classnames=classes, text_model=self.textual,
                                                    num_prompts=text_num_prompts, prompts_init=text_prompt_init,
                                                    CSC=text_prompt_CSC, ctx_pos=text_prompt_pos, cls_type=cls_type,
                                                    knowledge_version=knowledge_version, use_descriptor=use_descriptor,
                                                    token_wise_mlp=token_wise_mlp)
            self.tokenized_prompts = self.prompt_learner.tokenized_prompts

        # freeze (VitaCLIP_model.py:222-239)
        for name, param in self.visual.named_parameters():
            if not ('summary' in name or 'local' in name or 'global' in name or 'time_embed' in name):
                param.requires_grad = False
        if hasattr(self, "textual"):
            for param in self.textual.named_parameters():
                param[1].requires_grad = False

        # ---- HIP-path state (not part of the reference surface)
        self._hip_init(dict(size=input_size[0], P=patch_size[0], D=feature_dim, H=num_heads, layers=num_layers,
                            F=round(mlp_factor * feature_dim), E=embed_dim, G=num_global_prompts,
                            W=text_transformer_width, TH=text_transformer_heads, TL=text_transformer_layers,
                            L=text_context_length, n_ctx=text_num_prompts), operand_dtype)
        self._attach_encoders()

    def _attach_encoders(self):
        """The L1 encoders stay callable on their own (evaluation/iwa.py:212, zero_shot.py:75): they run on this host."""
        self.visual.__dict__["_host_ref"] = weakref.ref(self)
        if hasattr(self, "textual"):
            self.textual.__dict__["_host_ref"] = weakref.ref(self)

    def _gather(self, feats):
        """RCCL all-gather over xGMI of the per-clip embeddings (north_star): each rank
        holds whole clips, every rank ends with the embeddings (and logits) of the global batch.
        Contract: with an initialised process group of world size N and grad disabled, forward() returns
        logits of shape (N*B, C), rank-major; `local_logits()` gives a rank its own B rows - what the reference's
        evaluate() loops index with their local labels (training/train.py:661-670).  Under autograd nothing is gathered
        (every rank keeps its own clips, as the reference's DDP does)."""
        import torch.distributed as dist
        if not (self.gather_across_ranks and dist.is_available() and dist.is_initialized()
                and dist.get_world_size() > 1):
            return feats
        feats = feats.contiguous()
        # all_gather_into_tensor needs the same number of rows on every rank: checked once per row count (a 1-int
        # all-gather), so that a ragged last batch fails with a message instead of corrupting the gathered matrix
        checked = self.__dict__.setdefault("_gather_checked", set())
        if feats.shape[0] not in checked:
            mine = torch.tensor([feats.shape[0]], dtype=torch.int64, device=feats.device if dist.get_backend() != "gloo" else "cpu")
            rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(rows, mine)
            if any(int(r) != feats.shape[0] for r in rows):
                raise hip.GavaError(f"gather_across_ranks needs the same batch size on every rank, got {[int(r) for r in rows]}; "
                                    "pad the last batch or set model.gather_across_ranks = False")
            checked.add(feats.shape[0])
        if dist.get_backend() == "gloo" and feats.is_cuda:
            # CPU rendezvous (tests on a single-GPU box): stage through the host
            out = torch.empty(dist.get_world_size() * feats.shape[0], feats.shape[1], dtype=feats.dtype)
            dist.all_gather_into_tensor(out, feats.cpu())
            return out.to(feats.device)
        out = torch.empty(dist.get_world_size() * feats.shape[0], feats.shape[1], dtype=feats.dtype, device=feats.device)
        dist.all_gather_into_tensor(out, feats)
        return out

    def local_logits(self, logits):
        """Rows of this rank in the gathered logits (identity when nothing was gathered)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return logits
        b = logits.shape[0] // dist.get_world_size()
        return logits if logits.shape[0] == self.last.get("local_batch", -1) else logits[dist.get_rank() * b:(dist.get_rank() + 1) * b]

    def _class_mean_matrix(self, device):
        """(n_prompts, n_cls) matrix with 1/count_c in the rows of class c's prompts: flat logits @ A = per-class means."""
        counts = self.prompt_learner.kv_counts
        A = torch.zeros(sum(counts), len(counts), dtype=torch.float32, device=device)
        r = 0
        for c, k in enumerate(counts):
            A[r:r + k, c] = 1.0 / k
            r += k
        return A

    def _vision_trainables(self):
        from .training import _vision_trainables
        return _vision_trainables(self)

    def _train_head(self, video, text, summary, desc_wise):
        """Similarity head under autograd (VitaCLIP_model.py:248,255,287-293,308-309): 2*B*C*E flop on (B,E)/(C,E)
        tensors, traced by torch so that d logits reaches the text tower's backward and logit_scale."""
        assert not desc_wise
        vf = video / video.norm(dim=-1, keepdim=True)
        tf = text / text.norm(dim=-1, keepdim=True)
        logits = self.logit_scale.exp() * vf @ tf.t()
        n_kv = self.prompt_learner.n_kv if self.use_text_prompt_learning else 1
        if n_kv is None:                               # ragged (use_descriptor): per-class means by an averaging matrix
            A = self._class_mean_matrix(tf.device)
            logits, tf = logits @ A, A.t() @ tf
        elif n_kv > 1:                                 # VitaCLIP_model.py:288-290
            logits = logits.view(logits.shape[0], -1, n_kv).mean(-1)
            tf = tf.view(-1, n_kv, tf.shape[-1]).mean(1)
        if self.logit_bias is not None:
            logits = logits + self.logit_bias
        self.text_features = tf / tf.norm(dim=-1, keepdim=True)
        self.last.update(video_features=vf.detach(), summary=summary.detach())
        return logits

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, memory=None, video_nte=None, desc_wise=False):
        if not x.is_cuda:
            raise hip.GavaError("VitaCLIP (gava_clip_amd) runs on the HIP device only: move the model and the "
                                "input with .cuda(); there is no CPU fallback")
        # every launch below names "the current stream": make the input's device the current one (a model on cuda:k
        # called without torch.cuda.set_device(k) would otherwise launch on another device's stream)
        with torch.cuda.device(x.device):
            return self._forward_impl(x, memory, video_nte, desc_wise)

    def forward_frames(self, videos, preprocessor, memory=None, video_nte=None, desc_wise=False):
        """forward() on DECODED videos instead of a preprocessed batch: `videos` is a list of uint8 device tensors
        [n_i, H_i, W_i, 3] (PyAV's to_rgb().to_ndarray() order), `preprocessor` a gava_clip_amd.preprocess.ClipPreprocessor
        (the evaluation branch of video_dataset/dataset.py:117-139).  The patch-embedding GEMM reads the frames themselves
        (temporal crop, normalisation, bilinear resize and centre crop per loaded pixel): same logits, bit for bit, as
        forward(preprocessor.batch(videos)), without the fp32 clip ever being written to HBM - 4x fewer input bytes.
        Inference only (the training branch of the reference's data path augments on the host)."""
        if torch.is_grad_enabled():
            raise hip.GavaError("forward_frames is the evaluation data path: call it under torch.no_grad()")
        pre = preprocessor
        assert pre.spatial_size == self._shape["size"], "the preprocessor's crop size must be the model's input size"
        dev = videos[0].device
        with torch.cuda.device(dev):
            pre.check(videos)
            desc, keep = hip.clip_descriptors(videos, T=pre.num_frames, rate=pre.sampling_rate, size=pre.spatial_size,
                                              first_temporal_view=pre.num_temporal_views > 1,
                                              first_spatial_view=pre.num_spatial_views == 3)
            clips = (desc, len(videos), pre.num_frames, pre.lut(dev), dev)
            return self._forward_impl(desc, memory, video_nte, desc_wise, clips=clips)

    def _forward_impl(self, x, memory, video_nte, desc_wise, clips=None):
        lib = hip.load()
        if clips is not None:
            B, T = clips[1], clips[2]
        else:
            B, Cc, T, Hh, Ww = x.size()
        self.last["local_batch"] = B
        sh = self._shape
        if not x.is_cuda:
            raise hip.GavaError("VitaCLIP (gava_clip_amd) runs on the HIP device only: move the model and the "
                                "input with .cuda(); there is no CPU fallback")
        text, text_stream = None, None
        self._attach_encoders()
        self._pack()   # (re)pack weights on the caller's stream, before any fork
        if self.use_text_prompt_learning:
            if desc_wise:
                assert self.training == False
            # the cached text features depend on the packed weights AND on the (pass-through) context vectors
            pl_params = list(self.prompt_learner.parameters())
            key = ((self._pack_key(), tuple(q._version for q in pl_params), tuple(q.data_ptr() for q in pl_params))
                   if (self.cache_text_features and not self.training) else None)
            train_text = torch.is_grad_enabled() and any(q.requires_grad for q in pl_params)
            if train_text:
                # differentiable text tower (gava_clip_amd/training.py): HIP kernels in both directions; the
                # knowledge-aware context MLP (if any) is torch glue in front of it
                from .training import TextTowerFn
                ctx_full = self.prompt_learner.full_context()
                if self.text_on_side_stream:
                    # own stream, like the inference path; autograd runs TextTowerFn.backward on this stream too
                    # (and inserts the cross-stream waits), so the text tower overlaps the vision tower both ways
                    main = torch.cuda.current_stream(x.device)
                    if self._text_stream is None or self._text_stream.device != x.device:
                        self._text_stream = self._overlapping_stream(x.device)
                    text_stream = self._text_stream
                    text_stream.wait_stream(main)
                    with torch.cuda.stream(text_stream):
                        text = TextTowerFn.apply(self, ctx_full)
                else:
                    text = TextTowerFn.apply(self, ctx_full)
            elif key is not None and self._text_cache is not None and self._text_cache[0] == key:
                text = self._text_cache[1]
            else:
                # The text tower does not depend on the clip: it runs on its own HIP stream beside the
                # vision tower (its few-workgroup kernels slot in between the big GEMMs) and joins at the head.
                main = torch.cuda.current_stream(x.device)
                if self.text_on_side_stream:
                    if self._text_stream is None or self._text_stream.device != x.device:
                        self._text_stream = self._overlapping_stream(x.device)
                    text_stream = self._text_stream
                    text_stream.wait_stream(main)
                    with torch.cuda.stream(text_stream):
                        text = self.encode_text()
                else:
                    text = self.encode_text()
                self._text_cache = (key, text) if key is not None else None
        else:
            text = self.text_features.to(device=x.device, dtype=torch.float32).contiguous()
        train_vision = torch.is_grad_enabled() and any(p.requires_grad for _, p in self._vision_trainables())
        if train_vision:
            # differentiable vision tower (gava_clip_amd/training.py): gradients of the prompt parameters
            from .training import VisionTowerFn
            cls_x, summary = VisionTowerFn.apply(self, x, *[p for _, p in self._vision_trainables()])
        else:
            cls_x, summary = self.encode_video(x, clips=clips)
        # The all-gather of the clip embeddings goes out on the main stream as soon as the vision tower is enqueued - BEFORE the
        # join with the text stream, so that it overlaps what is left of the text tower (it does not depend on it).
        # Under autograd every rank keeps its own clips (the reference's DDP computes the loss on local logits).
        video = cls_x if cls_x.requires_grad else self._gather(cls_x)
        if text_stream is not None:
            torch.cuda.current_stream(x.device).wait_stream(text_stream)
            text.record_stream(torch.cuda.current_stream(x.device))
        if torch.is_grad_enabled() and (text.requires_grad or video.requires_grad or self.logit_scale.requires_grad):
            # training: the 2*B*C*E-flop head is traced by torch so that d logits reaches both towers' HIP backward
            logits = self._train_head(video, text, summary, desc_wise)
        else:
            Bg, Cn = video.shape[0], text.shape[0]
            n_kv = self.prompt_learner.n_kv if self.use_text_prompt_learning else 1
            ragged = n_kv is None        # use_descriptor: the head runs per prompt, the (tiny) class means follow in torch
            if (desc_wise and self.use_text_prompt_learning) or ragged:
                n_cls, n_kv = Cn, 1      # per-description logits (VitaCLIP_model.py:265-276): every prompt is its own "class"
            else:
                n_cls = Cn // n_kv
            empty = Bg == 0          # no clips (upstream returns (0, C) logits and still refreshes text_features): one dummy row
            if empty:
                video, Bg = torch.zeros(1, sh["E"], dtype=torch.float32, device=x.device), 1
            logits = torch.empty(Bg, n_cls, dtype=torch.float32, device=x.device)
            tfeat = torch.empty(n_cls, sh["E"], dtype=torch.float32, device=x.device)
            vnorm = torch.empty(Bg, sh["E"], dtype=torch.float32, device=x.device)
            ls = self.logit_scale.detach().float().reshape(1)
            lb = self.logit_bias.detach().float().reshape(1) if self.logit_bias is not None else None
            hip.check(lib.gava_similarity_head(hip.ptr(video), hip.ptr(text), hip.ptr(ls), hip.ptr(lb), Bg, n_cls, n_kv,
                                               sh["E"], hip.ptr(logits), hip.ptr(tfeat), hip.ptr(vnorm), hip.stream_ptr(x.device)),
                      "gava_similarity_head")
            if empty:
                logits, vnorm = logits[:0], vnorm[:0]
            self.last.update(video_features=vnorm, summary=summary)
            if desc_wise and self.use_text_prompt_learning:
                self.text_features = tfeat            # upstream leaves the last class's features here; not relied upon
                logits = list(torch.split(logits, self.prompt_learner.kv_counts, dim=1))     # list of (B, n_kv_c)
            elif ragged:
                A = self._class_mean_matrix(x.device)                                        # VitaCLIP_model.py:288-291
                logits = logits @ A      # (a logit_bias, added per prompt by the head, survives the mean unchanged)
                tf = A.t() @ tfeat
                self.text_features = tf / tf.norm(dim=-1, keepdim=True)
            elif self.use_text_prompt_learning:
                self.text_features = tfeat            # VitaCLIP_model.py:293

        # auxiliary heads: inactive at every accelerated configuration; kept as PyTorch glue on the device so that
        # callers passing video_nte / memory still get the reference's outputs - and, in training, its gradients
        # (`summary` and `text_features` carry the towers' autograd nodes).
        if self.add_nte and video_nte is not None:              # VitaCLIP_model.py:311-345
            sp = self.sum_proj(summary)
            sp = sp / sp.norm(dim=-1, keepdim=True)
            with torch.no_grad():
                valid_idx = ((video_nte.sum(dim=-1).sum(dim=-1)) != 0).float()
                valid_mat = valid_idx.unsqueeze(1) * valid_idx.unsqueeze(0)
            video_nte = video_nte / video_nte.norm(dim=-1, keepdim=True)
            similarity = torch.bmm(sp.unsqueeze(0).expand(NUM_COMB, -1, -1), video_nte.permute(1, 2, 0)).mean(0)
            logits_mat = self.logit_scale_vm * (similarity * valid_mat)
            logits_vm = F.log_softmax(logits_mat, dim=-1) + F.log_softmax(logits_mat, dim=-2)
        else:
            logits_vm = None
        if self.use_support_memory and memory is not None:      # VitaCLIP_model.py:347-398
            text_features = self.text_features.detach() if self.detach_features else self.text_features
            memory = memory.mean(dim=1)
            logits_mt = torch.empty(memory.size(0), 0).to(memory.device)
            for cid, mproj in enumerate(self.memory_project):
                tf = self.tf_project(text_features[cid])
                tf = tf / tf.norm(dim=-1, keepdim=True)
                memo = mproj(memory)
                memo = memo / memo.norm(dim=-1, keepdim=True)
                logits_mt = torch.concat([logits_mt, (self.logit_scale_mt * memo @ tf.t()).unsqueeze(-1)], dim=1)
            logits_mt = F.log_softmax(logits_mt, dim=-1)
            if self.logit_bias_mt is not None:
                logits_mt += self.logit_bias_mt
        else:
            logits_mt = None
        return logits, logits_mt, logits_vm
