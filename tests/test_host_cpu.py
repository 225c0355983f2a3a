"""CPU: host-side logic and the C-ABI library surface (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from gava_clip_amd.config import TINY, VIT_B16_T8, VIT_L14_T32, param_shapes
from helpers import REPO, CLASSES_3, CLASSES_400, model_kwargs, synth_torch_state


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from gava_clip_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    header = open(os.path.join(REPO, "include", "gava_hip.h")).read()
    declared = set(re.findall(r"\b(gava_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    from gava_clip_amd.build import abi_hash, needs_build
    assert lib.gava_abi_version() == abi_hash()
    assert not needs_build()          # the content hash stored next to the library matches csrc/ + the header


def test_a_library_built_from_another_header_is_refused(tmp_path, monkeypatch):
    """gava_abi_version() is a hash of include/gava_hip.h baked in at build time; hip.load() must refuse a library whose
    hash differs from the header in the tree (a stale .so would be called with shifted structs)."""
    import subprocess
    from gava_clip_amd import build as B, hip
    stale = str(tmp_path / "libgava_stale.so")
    # a one-symbol stand-in is enough: load() checks the version before it binds anything else
    src = tmp_path / "stale.c"
    src.write_text("int gava_abi_version(void) { return %d; }\n" % ((B.abi_hash() ^ 0x5a5a) & 0x7fffffff))
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", stale])
    monkeypatch.setattr(hip, "LIB_PATH", stale)
    monkeypatch.setattr(hip, "_lib", None)
    with pytest.raises(hip.GavaError, match="another include/gava_hip.h"):
        hip.load()
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(hip.GavaError, match="missing"):
        hip.load()


def test_ctypes_structs_match_header_sizes():
    """sizeof of every ABI struct as the C compiler sees it == ctypes layout."""
    import subprocess, tempfile
    from gava_clip_amd import hip
    names = {"gava_gemm_args": hip.GemmArgs, "gava_layernorm_args": hip.LayerNormArgs,
             "gava_attention_args": hip.AttentionArgs, "gava_vision_layer": hip.VisionLayer,
             "gava_vision_model": hip.VisionModel, "gava_vision_layer8": hip.VisionLayer8, "gava_text_layer": hip.TextLayer, "gava_text_model": hip.TextModel,
             "gava_preprocess_args": hip.PreprocessArgs, "gava_layernorm_bwd_args": hip.LayerNormBwdArgs,
             "gava_attention_bwd_args": hip.AttentionBwdArgs}
    src = '#include <stdio.h>\n#include "gava_hip.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    sizes = dict(zip(out[::2], map(int, out[1::2])))
    for n, cls in names.items():
        assert ctypes.sizeof(cls) == sizes[n], n


def test_load_refuses_a_ctypes_mirror_that_is_out_of_step(monkeypatch):
    """gava_struct_sizes: load() compares sizeof of every ABI struct as the library was compiled with the ctypes mirrors and
    refuses a mismatch (the ABI hash ties library and header, this ties header and mirrors) - without a GPU."""
    from gava_clip_amd import hip
    hip.load()

    class Short(ctypes.Structure):
        _fields_ = hip.GemmArgs._fields_[:-1]
    monkeypatch.setattr(hip, "GemmArgs", Short)
    monkeypatch.setattr(hip, "_lib", None)
    with pytest.raises(hip.GavaError, match="out of step"):
        hip.load()
    monkeypatch.undo()
    hip._lib = None
    hip.load()


@pytest.mark.parametrize("cfg,cls_file,n_cls", [(TINY, CLASSES_3, 3), (VIT_B16_T8, CLASSES_400, 400)])
def test_state_dict_keys_match_reference(cfg, cls_file, n_cls):
    from gava_clip_amd import VitaCLIP
    if cfg is VIT_B16_T8:
        cfg = type(cfg)(num_layers=1, text_layers=1)  # same key pattern, fewer layers to allocate
    m = VitaCLIP(**model_kwargs(cfg, cls_file))
    want = param_shapes(cfg, n_cls)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == dict(want)
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert "prompt_learner.ctx" in trainable and "logit_scale" in trainable
    assert "visual.time_embed" in trainable and "visual.global_prompts" in trainable
    assert all(("summary" in n or "local" in n or "global" in n or "time_embed" in n)
               for n in trainable if n.startswith("visual."))
    assert not any(n.startswith("textual.") for n in trainable)
    assert len(m.tokenized_prompts) == n_cls and tuple(m.tokenized_prompts[0].shape) == (1, 77)


def test_strict_load_and_no_cpu_fallback():
    from gava_clip_amd import VitaCLIP
    from gava_clip_amd.hip import GavaError
    m = VitaCLIP(**model_kwargs(TINY))
    m.load_state_dict(synth_torch_state(TINY, 3), strict=True)
    m.eval()
    with pytest.raises(GavaError):
        with torch.no_grad():
            m(torch.zeros(1, 3, TINY.num_frames, TINY.input_size, TINY.input_size))


def test_broken_reference_configs_are_rejected():
    from gava_clip_amd import VitaCLIP
    kw = model_kwargs(TINY)
    with pytest.raises(NotImplementedError):
        VitaCLIP(**{**kw, "use_global_prompts": False})
    with pytest.raises(NotImplementedError):
        VitaCLIP(**{**kw, "text_prompt_CSC": False})
    with pytest.raises(NotImplementedError):          # KAPT without split/uni does not run upstream either
        VitaCLIP(**{**kw, "text_prompt_init": "cntn_disc"})
    with pytest.raises(AssertionError):               # knowledge files missing: same assertion as kapt_head.py:61
        VitaCLIP(**{**kw, "text_prompt_init": "cntn_split_uni_disc"})


def test_kapt_prompt_construction_matches_reference(golden_dir, tmp_path, monkeypatch):
    """Knowledge-aware prompts on the synthetic knowledge files the golden was generated with: token ids, state_dict
    keys, the shifted embedding order and the context rows (ctx + per-class MLP of the entity embeddings)."""
    import os
    import torch
    from gava_clip_amd import VitaCLIP, synth
    g = np.load(os.path.join(golden_dir, "tiny_kapt.npz"))
    synth.synth_knowledge_files(str(tmp_path), "updrs", 3, ["v1", "v2", "v3"])
    monkeypatch.chdir(tmp_path)
    m = VitaCLIP(**{**model_kwargs(TINY), "text_prompt_init": "cntn_split_uni_disc", "knowledge_version": ["v1", "v2", "v3"]})
    pl = m.prompt_learner
    assert pl.n_kv == 3 and np.array_equal(torch.cat(m.tokenized_prompts).numpy(), g["tokens"])
    keys = [k for k in m.state_dict() if "context_prompt_learner" in k]
    assert keys == list(synth.synth_kapt_state(TINY, 3).keys())
    sd = {**{k: torch.from_numpy(v) for k, v in synth.synth_state_dict(TINY, 3).items()},
          **{k: torch.from_numpy(v) for k, v in synth.synth_kapt_state(TINY, 3).items()}}
    m.load_state_dict(sd, strict=True)
    eff, tok = pl.embedding_token_ids(), torch.cat(pl.tokenized_prompts)
    n = TINY.text_num_prompts
    assert torch.equal(eff[:, 0], tok[:, 0]) and torch.equal(eff[:, 1 + n:], tok[:, 1:-n])
    full = pl.full_context()
    assert full.shape == (9, n, TINY.text_width)
    e = pl.context_prompt_learner.cntn_embeds[1]
    want = pl.ctx[1].unsqueeze(0) + pl.context_prompt_learner.projector[1](e).unsqueeze(1)
    assert torch.allclose(full[3:6], want.expand(-1, n, -1))


def test_flop_model_matches_survey():
    from gava_clip_amd.flops import vision_flops_per_clip, text_flops_per_prompt, forward_flops
    assert abs(vision_flops_per_clip(VIT_B16_T8) / 1e9 - 298.593) < 0.01
    assert abs(text_flops_per_prompt(VIT_B16_T8) / 1e9 - 5.9595) < 0.001
    assert abs(forward_flops(VIT_B16_T8, 64, 3) / 1e9 - 19127.8) < 0.5
    assert abs(vision_flops_per_clip(VIT_L14_T32) / 1e9 - 5631.77) < 0.1


def test_kapt_descriptor_prompts_are_ragged_like_the_reference(golden_dir, tmp_path, monkeypatch):
    """KAPT with use_descriptor=True (kapt_head.py:65-88): class c gets one prompt per line of descriptor_<c>.txt -
    2, 3 and 1 prompts here; token ids equal the reference's, context rows are ctx[c] + MLP_c(descriptor embedding)."""
    import os
    import torch
    from gava_clip_amd import VitaCLIP, synth
    g = np.load(os.path.join(golden_dir, "tiny_kapt_desc.npz"))
    synth.synth_descriptor_files(str(tmp_path), "updrs", (2, 3, 1))
    monkeypatch.chdir(tmp_path)
    m = VitaCLIP(**{**model_kwargs(TINY), "text_prompt_init": "cntn_split_uni_disc", "knowledge_version": ["v1", "v2", "v3"],
                    "use_descriptor": True})
    pl = m.prompt_learner
    assert pl.kv_counts == [2, 3, 1] and pl.n_kv is None
    assert np.array_equal(torch.cat(m.tokenized_prompts).numpy(), g["tokens"])
    assert [k for k in m.state_dict() if "context_prompt_learner" in k] == list(synth.synth_kapt_state(TINY, 3).keys())
    full = pl.full_context()
    assert full.shape == (6, TINY.text_num_prompts, TINY.text_width)
    A = m._class_mean_matrix(torch.device("cpu"))
    assert A.shape == (6, 3) and torch.allclose(A.sum(0), torch.ones(3)) and float(A[2:5, 1].min()) == pytest.approx(1 / 3)
    with pytest.raises(NotImplementedError):
        VitaCLIP(**{**model_kwargs(TINY), "text_prompt_init": "cntn_split_uni_disc", "token_wise_mlp": True})
