#!/usr/bin/env python
"""Generate tests/golden/* by running the REFERENCE implementation in the build container.

Runs only where /root/reference exists (never on the GPU box).  The reference is imported
read-only with two module shims: ``ftfy`` (identity on the ASCII class
lists) and ``video_dataset`` (only the constant NUM_COMB=70,
/root/reference/video_dataset/dataset.py:19).  Weights and inputs come from
gava_clip_amd/synth.py and are loaded with ``strict=True`` (which also pins state_dict key
parity).  Outputs are data only: logits, features, per-layer CLS rows, token ids.

    python tools/gen_golden.py            # writes tests/golden/{c1_b16,tiny}.npz, *_grads.npz, tokens_*.json
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from gava_clip_amd.config import VIT_B16_T8, TINY, param_shapes  # noqa: E402
from gava_clip_amd import synth  # noqa: E402

CLASSES = os.path.join(REPO, "gava_clip_amd", "data", "classes")


def import_reference():
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    sys.modules["ftfy"] = ftfy
    vd = types.ModuleType("video_dataset")
    vd.NUM_COMB = 70
    sys.modules["video_dataset"] = vd
    sys.path.insert(0, os.path.join(REF, "training"))
    import VitaCLIP_model  # noqa
    import VitaCLIP_text_encoder  # noqa
    return VitaCLIP_model, VitaCLIP_text_encoder


def build_reference(mod, cfg, class_file, **extra):
    kw = dict(
        backbone_path="", input_size=(cfg.input_size, cfg.input_size), num_frames=cfg.num_frames,
        feature_dim=cfg.feature_dim, patch_size=(cfg.patch_size, cfg.patch_size),
        num_heads=cfg.num_heads, num_layers=cfg.num_layers, mlp_factor=cfg.mlp_factor,
        embed_dim=cfg.embed_dim, use_summary_token=True, use_local_prompts=True,
        use_global_prompts=True, num_global_prompts=cfg.num_global_prompts,
        use_text_prompt_learning=True, text_context_length=cfg.text_context_length,
        text_vocab_size=cfg.text_vocab_size, text_transformer_width=cfg.text_width,
        text_transformer_heads=cfg.text_heads, text_transformer_layers=cfg.text_layers,
        text_num_prompts=cfg.text_num_prompts, text_prompt_pos="end", text_prompt_init="",
        text_prompt_CSC=True, text_prompt_classes_path=class_file)
    kw.update(extra)
    return mod.VitaCLIP(**kw)


def load_synth(model, cfg, n_cls, seed=0, aux=False, kapt=False):
    sd = synth.synth_state_dict(cfg, n_cls, seed)
    if aux:
        sd.update(synth.synth_aux_state(cfg, n_cls, seed))
    if kapt:
        sd.update(synth.synth_kapt_state(cfg, n_cls, seed))
    ref_keys = list(model.state_dict().keys())
    assert sorted(ref_keys) == sorted(sd.keys()), "state_dict key mismatch with reference"
    if not aux and not kapt:
        assert ref_keys == list(sd.keys()), "state_dict key order/name mismatch with reference"
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    # TextPromptLearner caches token embeddings at construction (text_encoder.py:284,296-300):
    # rebuild them from the loaded embedding table exactly as its __init__ does.
    pl = model.prompt_learner
    with torch.no_grad():
        for idc in range(pl.n_cls):
            emb = model.textual.token_embedding(pl.tokenized_prompts[idc])
            pl.token_prefix[idc] = emb[:, :1, :]
            pl.token_suffix[idc] = emb[:, 1: -pl.n_ctx, :] if pl.knowledge_aware_prompt else emb[:, 1 + pl.n_ctx:, :]
    return sd


def run_case(mod, cfg, class_file, B, name, full_trace, wseed=0, xseed=1234, compact=False):
    """One eval forward of the REFERENCE (evaluation/evaluate.py:263-283).  `compact` keeps only the outputs (logits,
    scores, features, summary): the round-3 fixtures at other seeds / shapes."""
    from gava_clip_amd.tokenizer import read_class_names
    n_cls = len(read_class_names(class_file))
    model = build_reference(mod, cfg, class_file)
    load_synth(model, cfg, n_cls, seed=wseed)
    model.eval()
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=xseed))
    blocks_out = []
    hooks = [blk.register_forward_hook(lambda m, i, o: blocks_out.append((o[0].detach(), o[1].detach())))
             for blk in model.visual.blocks]
    vis_out = []
    hooks.append(model.visual.register_forward_hook(lambda m, i, o: vis_out.append(o)))
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits, lmt, lvm = model(x)
    assert lmt is None and lvm is None
    for h in hooks:
        h.remove()
    cls_x, summary = vis_out[0]
    vf = cls_x / cls_x.norm(dim=-1, keepdim=True)
    G = cfg.num_global_prompts
    out = dict(
        logits=logits.numpy(), scores=logits.softmax(-1).numpy(), video_features=vf.numpy(),
        text_features=model.text_features.numpy(), summary=summary.numpy(),
        cls_rows=np.stack([b[0][:, 0, :].numpy() for b in blocks_out]),
        patch_row7=np.stack([b[0][:, G + 1 + 7, :].numpy() for b in blocks_out]),
        block_fro=np.array([float(torch.cat((b[0][:, :1], b[0][:, G + 1:]), 1).norm()) for b in blocks_out]),
        summ_last=blocks_out[-1][1].numpy(),
        x_checksum=np.array([float(x.double().sum()), float(x.double().abs().sum())]),
        tokens=torch.cat(model.tokenized_prompts).numpy().astype(np.int32),
        wseed=np.array(wseed), xseed=np.array(xseed),
    )
    if compact:
        for k in ("cls_rows", "patch_row7", "block_fro", "summ_last", "tokens"):
            del out[k]
    if full_trace:
        for i, b in enumerate(blocks_out):
            out[f"block{i}"] = torch.cat((b[0][:, :1], b[0][:, G + 1:]), 1).numpy()
            out[f"summ{i}"] = b[1].numpy()
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, {k: np.asarray(v).shape for k, v in out.items() if not k.startswith("block")})
    return out


def grad_sample(g, n=2048):
    """Small fixture of a large gradient: its norm and a strided sample of <= n entries (same rule in the tests)."""
    flat = g.reshape(-1)
    step = max(1, flat.numel() // n)
    return float(flat.double().norm()), flat[::step][:n].clone()


def run_grad_case(mod, cfg, class_file, B, name, aux, sampled=False):
    """The REFERENCE in train mode under torch autograd (training/train.py:441-470): gradients of every parameter it
    leaves trainable, for loss = sum(w * logits) (+ the auxiliary heads' outputs when aux).  Pins the HIP backward."""
    from gava_clip_amd.tokenizer import read_class_names
    n_cls = len(read_class_names(class_file))
    extra = dict(add_nte=True, use_support_memory=True, detach_features=False, num_classes=n_cls) if aux else {}
    model = build_reference(mod, cfg, class_file, **extra)
    load_synth(model, cfg, n_cls, aux=aux)
    model.train()
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=1234))
    g = torch.Generator().manual_seed(2024)
    w1 = torch.randn(B, n_cls, generator=g)
    kw = {}
    if aux:
        nte, mem = synth.synth_aux_inputs(B, cfg.embed_dim)
        kw = dict(video_nte=torch.from_numpy(nte), memory=torch.from_numpy(mem))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits, lmt, lvm = model(x, **kw)
    loss = (logits * w1).sum()
    out = dict(logits=logits.detach().numpy(), w_logits=w1.numpy())
    if aux:
        w2, w3 = torch.randn(B, B, generator=g), torch.randn(B, n_cls, generator=g)
        loss = loss + (lvm * w2).sum() + (lmt * w3).sum()
        out.update(logits_vm=lvm.detach().numpy(), logits_mt=lmt.detach().numpy(), w_vm=w2.numpy(), w_mt=w3.numpy())
    loss.backward()
    n = 0
    for pname, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, pname
            if sampled:
                nrm, sub = grad_sample(p.grad)
                out["gradnorm." + pname], out["gradsub." + pname] = np.array(nrm), sub.numpy()
            else:
                out["grad." + pname] = p.grad.numpy()
            n += 1
        else:
            assert p.grad is None
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, n, "gradients")


KAPT_VERSIONS = ["v1", "v2", "v3"]


KAPT_DESC_COUNTS = (2, 3, 1)


def run_kapt_case(mod, cfg, class_file, B, name, descriptors=False, init="cntn_split_uni_disc"):
    """Knowledge-aware prompts (training/kapt_head.py, text_prompt_init='cntn_split_uni_disc', the configuration of
    train_scripts/updrs_3cls_train_tulip.sh) on SYNTHETIC knowledge files (the real ./data/ke_* are not distributed):
    eval logits / text features / per-description logits, and train-mode gradients incl. the context MLPs."""
    import tempfile
    from gava_clip_amd.tokenizer import read_class_names
    n_cls = len(read_class_names(class_file))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        if descriptors:   # use_descriptor=True: a ragged number of prompts per class (kapt_head.py:65-88)
            synth.synth_descriptor_files(tmp, "updrs", KAPT_DESC_COUNTS)
        else:
            synth.synth_knowledge_files(tmp, "updrs", n_cls, KAPT_VERSIONS)
        os.chdir(tmp)
        try:
            model = build_reference(mod, cfg, class_file, text_prompt_init=init,
                                    knowledge_version=list(KAPT_VERSIONS), cls_type="updrs", use_descriptor=descriptors)
        finally:
            os.chdir(cwd)
    load_synth(model, cfg, n_cls, kapt=True)
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=1234))
    model.eval()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits, _, _ = model(x)
        tfeat = model.text_features.clone()
        desc, _, _ = model(x, desc_wise=True)
    out = dict(logits=logits.numpy(), text_features=tfeat.numpy(),
               desc_logits=(np.concatenate([d.numpy() for d in desc], 1) if descriptors else np.stack([d.numpy() for d in desc])),
               tokens=torch.cat(model.tokenized_prompts).numpy().astype(np.int32))
    model.train()
    w1 = torch.randn(B, n_cls, generator=torch.Generator().manual_seed(2024))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lg = model(x)[0]
    (lg * w1).sum().backward()
    out["w_logits"] = w1.numpy()
    n = 0
    for pname, p in model.named_parameters():
        if p.requires_grad:
            out["grad." + pname] = p.grad.numpy()
            n += 1
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, n, "gradients", out["logits"].shape, out["desc_logits"].shape)


def run_zeroshot_case(mod, cfg, class_file, B, name):
    """The zero-shot branch (VitaCLIP_model.py:97-98,295-306): use_text_prompt_learning=False, class text features read
    from a file, logits = exp(logit_scale) * normalise(video) @ normalise(text).T.  The text features are data of the
    fixture (3 x E, synthetic, deliberately NOT unit length)."""
    import tempfile
    from gava_clip_amd.tokenizer import read_class_names
    n_cls = len(read_class_names(class_file))
    tf = torch.from_numpy(synth.normalish("input.zeroshot_text_features", (n_cls, cfg.embed_dim), 0).astype(np.float32)) * 1.7
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "tf.pth")
        torch.save({"text_features": tf}, path)
        model = build_reference(mod, cfg, class_file, use_text_prompt_learning=False, zeroshot_evaluation=True,
                                zeroshot_text_features_path=path)
    sd = synth.synth_state_dict(cfg, n_cls, 0)
    keys = list(model.state_dict().keys())
    assert set(keys) <= set(sd.keys()) and not any(k.startswith(("textual.", "prompt_learner.")) for k in keys)
    model.load_state_dict({k: torch.from_numpy(sd[k]) for k in keys}, strict=True)
    model.eval()
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=1234))
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits, lmt, lvm = model(x)
        cls_x, summary = model.visual(x)           # the L1 interface callers use directly (evaluation/iwa.py:212,230)
    assert lmt is None and lvm is None
    out = dict(logits=logits.numpy(), text_features_in=tf.numpy(), state_keys=np.array(keys),
               visual_cls_x=cls_x.numpy(), visual_summary=summary.numpy())
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, out["logits"].shape)


def run_sigmoid_case(mod, cfg, class_file, B, name):
    """use_sigmoid_loss=True (VitaCLIP_model.py:193-197,308-309): logit_scale starts at log(log 10), a learnable
    logit_bias is added to the logits.  Both are set to non-default values here so that the test pins them."""
    from gava_clip_amd.tokenizer import read_class_names
    n_cls = len(read_class_names(class_file))
    model = build_reference(mod, cfg, class_file, use_sigmoid_loss=True)
    sd = synth.synth_state_dict(cfg, n_cls, 0)
    sd["logit_scale"] = np.array(0.9, dtype=np.float32)
    sd["logit_bias"] = np.array(-3.5, dtype=np.float32)
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    pl = model.prompt_learner
    with torch.no_grad():
        for idc in range(pl.n_cls):
            emb = model.textual.token_embedding(pl.tokenized_prompts[idc])
            pl.token_prefix[idc] = emb[:, :1, :]
            pl.token_suffix[idc] = emb[:, 1 + pl.n_ctx:, :]
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=1234))
    model.eval()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits = model(x)[0]
        # the text tower called directly, as evaluation/zero_shot.py:75-76 and utils/prepare_embedding.py do
        tok = torch.cat(model.tokenized_prompts)
        direct = model.textual(model.textual.token_embedding(tok), tok)
    model.train()
    w1 = torch.randn(B, n_cls, generator=torch.Generator().manual_seed(2024))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lg = model(x)[0]
    (lg * w1).sum().backward()
    out = dict(logits=logits.numpy(), w_logits=w1.numpy(), logit_scale=sd["logit_scale"], logit_bias=sd["logit_bias"],
               textual_direct=direct.numpy(), tokens=tok.numpy().astype(np.int32))
    for pname in ("logit_scale", "logit_bias", "prompt_learner.ctx", "visual.global_prompts"):
        out["grad." + pname] = dict(model.named_parameters())[pname].grad.numpy()
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, out["logits"])


def run_preprocess_cases():
    """The evaluation branch of the REFERENCE's VideoDataset.__getitem__ (video_dataset/dataset.py:78-139,163-200) run on
    synthetic decoded videos.  Its module needs PyAV and torchvision, which this container lacks: `av` is replaced by a
    stand-in whose container yields frames carrying synthetic uint8 arrays (decoding is out of scope - SURVEY 8f row 3 starts
    at the decoded frames), `torchvision` / the augmentation module by empty stand-ins (the evaluation branch never touches
    them).  Everything from `to_rgb().to_ndarray()` on is the reference's own code.  Outputs are stored as sha256 of the fp32
    bytes + a strided sample: the oracle (oracle/preprocess_oracle.py) must reproduce them bit for bit."""
    import hashlib
    import importlib
    import tempfile

    videos = {}

    class _Frame:
        def __init__(self, arr, pts):
            self.arr, self.pts = arr, pts

        def to_rgb(self):
            return self

        def to_ndarray(self):
            return self.arr

    class _Container:
        def __init__(self, path):
            self.frames = videos[os.path.basename(path)]

        def decode(self, video=0):
            for i in range(self.frames.shape[0] - 1, -1, -1):      # out of order on purpose: the reference sorts by pts
                yield _Frame(self.frames[i], 40 * i)

        def close(self):
            pass

    av = types.ModuleType("av")
    av.open = lambda path: _Container(path)
    sys.modules["av"] = av
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
    pkg = types.ModuleType("video_dataset")
    pkg.__path__ = [os.path.join(REF, "video_dataset")]
    sys.modules["video_dataset"] = pkg
    tr = types.ModuleType("video_dataset.transform")
    tr.create_random_augment = tr.random_resized_crop = None
    sys.modules["video_dataset.transform"] = tr
    ds_mod = importlib.import_module("video_dataset.dataset")
    assert ds_mod.__file__.startswith(REF)

    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    # (n_frames, H, W, T, rate, size, spatial views, temporal views): tests/test_preprocess.py CASES + its multi-view cases
    cases = [(20, 240, 320, 8, 2, 224, 1, 1), (9, 320, 240, 8, 1, 224, 1, 1), (5, 256, 256, 8, 1, 224, 1, 1),
             (40, 360, 640, 16, 2, 224, 1, 1), (12, 224, 224, 8, 1, 224, 1, 1), (10, 181, 333, 4, 3, 96, 1, 1),
             (30, 240, 320, 8, 2, 224, 1, 10), (30, 320, 240, 8, 2, 224, 3, 10), (12, 224, 400, 8, 2, 224, 3, 1)]
    out = {"cases": np.array(cases, dtype=np.int64)}
    with tempfile.TemporaryDirectory() as tmp:
        for i, (n, h, w, T, rate, size, sv, tvw) in enumerate(cases):
            seed = n * 1000 + h if sv == 1 and tvw == 1 else n + h          # the seeds tests/test_preprocess.py uses
            rng = np.random.default_rng(seed)
            videos[f"v{i}.mp4"] = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
            lst = os.path.join(tmp, f"list{i}.csv")
            with open(lst, "w") as f:
                f.write(f"v{i}.mp4,{i % 3}\n")
            ds = ds_mod.VideoDataset(list_path=lst, data_root=tmp, num_spatial_views=sv, num_temporal_views=tvw, random_sample=False,
                                     num_frames=T, sampling_rate=rate, spatial_size=size, mean=mean, std=std, is_train=False)
            frames, label, name = ds[0]
            assert tuple(frames.shape) == (3, T, size, size) and label == i % 3 and name == f"v{i}"
            a = frames.contiguous().numpy()
            out[f"sha256_{i}"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
            out[f"sample_{i}"] = a.reshape(-1)[::max(1, a.size // 4096)][:4096].copy()
            print("preprocess case", i, (n, h, w, T, rate, size, sv, tvw), "->", a.shape, hashlib.sha256(a.tobytes()).hexdigest()[:16])
    path = os.path.join(REPO, "tests", "golden", "preprocess_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def dump_tokens(txt_mod):
    from gava_clip_amd.tokenizer import read_class_names, prompt_texts
    for fn in ("updrs_3cls_classes.txt", "k400_classes.txt"):
        names = read_class_names(os.path.join(CLASSES, fn))
        texts = prompt_texts(names, 8)
        ids = torch.cat([txt_mod.tokenize(t) for t in texts]).numpy().astype(int).tolist()
        path = os.path.join(REPO, "tests", "golden", "tokens_" + fn.replace("_classes.txt", "") + ".json")
        with open(path, "w") as f:
            json.dump({"texts": texts, "ids": ids}, f)
        print("wrote", path, len(ids))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if "--preprocess" in sys.argv:    # video_dataset/dataset.py only; the model modules are not imported
        run_preprocess_cases()
        sys.exit(0)
    mod, txt_mod = import_reference()
    C3 = os.path.join(CLASSES, "updrs_3cls_classes.txt")
    if "--c3-full" in sys.argv or "--c5-full" in sys.argv:   # BASELINE configs c3 / c5 (per GPU) at their full batch
        from gava_clip_amd.config import VIT_B16_T16, VIT_L14_T32
        if "--c3-full" in sys.argv:
            run_case(mod, VIT_B16_T16, os.path.join(CLASSES, "k400_classes.txt"), 32, "c3_full", False, wseed=0, xseed=4243, compact=True)
        if "--c5-full" in sys.argv:
            run_case(mod, VIT_L14_T32, C3, 32, "c5_full", False, wseed=0, xseed=4244, compact=True)
        sys.exit(0)
    if "--c2-full" in sys.argv:      # BASELINE config c2 at its full batch: 64 clips through the reference (logits + features only)
        run_case(mod, VIT_B16_T8, C3, 64, "c2_full", False, wseed=0, xseed=4242, compact=True)
        sys.exit(0)
    if "--round4" in sys.argv:       # twelve more weight + input seeds at c1 (round 4: the weight-lo parity modes are judged on 24 c1 seeds)
        for sidx in range(12, 24):
            run_case(mod, VIT_B16_T8, C3, 2, f"c1_b16_s{sidx}", False, wseed=sidx, xseed=1234 + sidx, compact=True)
        sys.exit(0)
    if "--more-seeds" in sys.argv:   # further weight + input seeds at c1 (statistics of the logits error; compact fixtures)
        for sidx in range(4, 12):
            run_case(mod, VIT_B16_T8, C3, 2, f"c1_b16_s{sidx}", False, wseed=sidx, xseed=1234 + sidx, compact=True)
        sys.exit(0)
    if "--round3" in sys.argv:      # only the fixtures added in round 3: more seeds at c1, one clip at c3 and c5 shapes
        from gava_clip_amd.config import VIT_B16_T16, VIT_L14_T32
        for sidx in (1, 2, 3):      # other weights AND other clips than c1_b16.npz (wseed 0, xseed 1234)
            run_case(mod, VIT_B16_T8, C3, 2, f"c1_b16_s{sidx}", False, wseed=sidx, xseed=1234 + sidx, compact=True)
        # c3's clip 0 (16 frames, 400 classes: 400 sequential text passes upstream, VitaCLIP_model.py:282-290); the
        # input is the clip tests/test_gpu_forward.py puts first in its 32-clip batch
        run_case(mod, VIT_B16_T16, os.path.join(CLASSES, "k400_classes.txt"), 1, "c3_clip0", False, xseed=3, compact=True)
        # c5's clip 0 (ViT-L/14, 32 frames), likewise
        run_case(mod, VIT_L14_T32, C3, 1, "c5_clip0", False, xseed=5, compact=True)
        sys.exit(0)
    if "--round2" in sys.argv:      # only the fixtures added in round 2 (the others are unchanged)
        run_kapt_case(mod, TINY, C3, 2, "tiny_kapt_nodisc", init="cntn_split_uni")
        run_zeroshot_case(mod, TINY, C3, 2, "tiny_zeroshot")
        run_sigmoid_case(mod, TINY, C3, 2, "tiny_sigmoid")
        sys.exit(0)
    dump_tokens(txt_mod)
    run_case(mod, TINY, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 2, "tiny", True)
    run_case(mod, VIT_B16_T8, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 2, "c1_b16", False)
    run_grad_case(mod, TINY, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 2, "tiny_grads", False)
    run_grad_case(mod, TINY, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 2, "tiny_aux_grads", True)
    run_grad_case(mod, VIT_B16_T8, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 1, "b16_grads", False, sampled=True)
    run_kapt_case(mod, TINY, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 2, "tiny_kapt")
    run_kapt_case(mod, TINY, os.path.join(CLASSES, "updrs_3cls_classes.txt"), 2, "tiny_kapt_desc", descriptors=True)
    run_kapt_case(mod, TINY, C3, 2, "tiny_kapt_nodisc", init="cntn_split_uni")
    run_zeroshot_case(mod, TINY, C3, 2, "tiny_zeroshot")
    run_sigmoid_case(mod, TINY, C3, 2, "tiny_sigmoid")
