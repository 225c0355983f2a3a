"""CPU: the oracle (oracle/vita_oracle.py) against the golden vectors produced by the imported
reference (tools/gen_golden.py).  This is what pins the oracle; tolerance 2e-5 relative to the
tensor's max (fp32 summation-order noise, measured < 1e-6)."""
import json
import os

import numpy as np
import pytest
import torch

from gava_clip_amd import synth
from gava_clip_amd.config import TINY, VIT_B16_T8, param_shapes
from gava_clip_amd.tokenizer import tokenize, read_class_names, prompt_texts
from oracle.vita_oracle import Oracle
from helpers import CLASSES_3, CLASSES_400, synth_torch_state, rel_to_max

TOL = 2e-5


@pytest.mark.parametrize("name,path", [("updrs_3cls", CLASSES_3), ("k400", CLASSES_400)])
def test_tokenizer_matches_reference_ids(golden_dir, name, path):
    g = json.load(open(os.path.join(golden_dir, f"tokens_{name}.json")))
    texts = prompt_texts(read_class_names(path), 8)
    assert texts == g["texts"]
    ids = tokenize(texts)
    assert ids.dtype == np.int32 and ids.shape == (len(texts), 77)
    assert np.array_equal(ids, np.array(g["ids"]))


def test_tokenizer_edge_cases():
    with pytest.raises(RuntimeError):
        tokenize("word " * 100)
    t = tokenize("word " * 100, truncate=True)
    assert t[0, -1] == 49407 and t[0, 0] == 49406
    e = tokenize("")
    assert list(e[0, :3]) == [49406, 49407, 0]


def test_synth_is_deterministic_and_matches_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny.npz"))
    x = synth.synth_clip(2, TINY.num_frames, TINY.input_size)
    assert abs(float(x.astype(np.float64).sum()) - g["x_checksum"][0]) < 1e-6
    assert abs(float(np.abs(x.astype(np.float64)).sum()) - g["x_checksum"][1]) < 1e-6
    a = synth.synth_param("visual.proj", (128, 128), TINY)
    b = synth.synth_param("visual.proj", (128, 128), TINY)
    assert np.array_equal(a, b)


def _run(cfg, golden):
    sd = synth_torch_state(cfg, 3)
    x = torch.from_numpy(synth.synth_clip(2, cfg.num_frames, cfg.input_size))
    return Oracle(cfg, sd, golden["tokens"]).forward(x, trace=True)


def test_oracle_tiny_all_intermediates(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny.npz"))
    r = _run(TINY, g)
    for k in ("logits", "video_features", "text_features", "summary"):
        assert rel_to_max(r[k].numpy(), g[k]) < TOL, k
    for i in range(TINY.num_layers):
        assert rel_to_max(r["trace"][f"block{i}"].numpy(), g[f"block{i}"]) < TOL
        assert rel_to_max(r["trace"][f"summ{i}"].numpy(), g[f"summ{i}"]) < TOL
    assert np.allclose(r["logits"].softmax(-1).numpy(), g["scores"], atol=1e-6)


def test_oracle_c1_vit_b16(golden_dir):
    """BASELINE config c1: ViT-B/16, B=2, T=8, 224^2, 3 classes."""
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    g = np.load(os.path.join(golden_dir, "c1_b16.npz"))
    r = _run(VIT_B16_T8, g)
    for k in ("logits", "video_features", "text_features", "summary"):
        assert rel_to_max(r[k].numpy(), g[k]) < TOL, k
    cls = np.stack([r["trace"][f"block{i}"][:, 0].numpy() for i in range(12)])
    assert rel_to_max(cls, g["cls_rows"]) < TOL
    row7 = np.stack([r["trace"][f"block{i}"][:, 1 + 7].numpy() for i in range(12)])
    assert rel_to_max(row7, g["patch_row7"]) < TOL
    assert np.array_equal(r["logits"].argmax(-1).numpy(), g["logits"].argmax(-1))


@pytest.mark.parametrize("name", ["c1_b16_s1", "c1_b16_s2", "c1_b16_s3", "c1_b16_s6", "c1_b16_s11", "c3_clip0", "c5_clip0"])
def test_oracle_vs_round3_reference_fixtures(golden_dir, name):
    """The round-3 reference runs (tools/gen_golden.py --round3): three more weight + input seeds at c1, clip 0 of c3
    (16 frames, 400 classes) and of c5 (ViT-L/14, 32 frames).  Pins the oracle at the shapes the full-size GPU tests
    compare against."""
    from helpers import golden_case
    from gava_clip_amd.tokenizer import read_class_names, prompt_texts
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    cfg, class_file, n_cls, B, wseed, xseed = golden_case(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    assert int(g["wseed"]) == wseed and int(g["xseed"]) == xseed
    x = synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=xseed)
    assert abs(float(x.astype(np.float64).sum()) - g["x_checksum"][0]) < 1e-6
    tokens = tokenize(prompt_texts(read_class_names(class_file), cfg.text_num_prompts))
    r = Oracle(cfg, synth_torch_state(cfg, n_cls, wseed), tokens).forward(torch.from_numpy(x))
    for k in ("logits", "video_features", "text_features", "summary"):
        assert rel_to_max(r[k].numpy(), g[k]) < TOL, k
    assert np.array_equal(r["logits"].argmax(-1).numpy(), g["logits"].argmax(-1))


def test_oracle_gradients_match_reference_tiny(golden_dir):
    """The oracle under torch autograd against the gradients the REFERENCE computed for the same loss
    (tests/golden/tiny_grads.npz, tools/gen_golden.py run_grad_case): pins the checker of the HIP backward."""
    g = np.load(os.path.join(golden_dir, "tiny_grads.npz"))
    sd = synth_torch_state(TINY, 3)
    p = {k: v.clone().float() for k, v in sd.items()}
    names = [k[len("grad."):] for k in g.files if k.startswith("grad.")]
    for k in names:
        p[k].requires_grad_()
    o = Oracle(TINY, p, torch.from_numpy(np.load(os.path.join(golden_dir, "tiny.npz"))["tokens"]))
    x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size))
    vf, _ = o.vision(x)
    tf = o.text(o.prompts())
    logits = p["logit_scale"].exp() * (vf / vf.norm(dim=-1, keepdim=True)) @ (tf / tf.norm(dim=-1, keepdim=True)).t()
    assert rel_to_max(logits.detach().numpy(), g["logits"]) < 1e-5
    (logits * torch.from_numpy(g["w_logits"])).sum().backward()
    for k in names:
        ref = g["grad." + k]
        got = p[k].grad.numpy()
        if k.endswith("k_proj.bias"):          # exactly zero in exact arithmetic: both sides are rounding noise
            assert np.abs(got).max() < 1e-6 and np.abs(ref).max() < 1e-6
            continue
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
