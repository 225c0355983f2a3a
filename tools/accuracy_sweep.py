#!/usr/bin/env python
"""Logits error of the HIP path against every reference fixture tests/golden/c1_b16*.npz (and c3_clip0 / c5_clip0 with --all),
both operand dtypes: the statistics behind DESIGN "Numerics".  GPU box.

    python tools/accuracy_sweep.py [--all]"""
import glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from gava_clip_amd import VitaCLIP, synth
from gava_clip_amd import config as C
from helpers import CLASSES_3, CLASSES_400, model_kwargs, synth_torch_state, mixed_violation

cases = []
for f in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "c1_b16*.npz"))):
    g = np.load(f)
    cases.append((os.path.basename(f)[:-4], C.VIT_B16_T8, CLASSES_3, 3, 2, int(g["wseed"]) if "wseed" in g.files else 0,
                  int(g["xseed"]) if "xseed" in g.files else 1234, g))
if "--all" in sys.argv:
    cases.append(("c3_clip0", C.VIT_B16_T16, CLASSES_400, 400, 1, 0, 3, np.load(os.path.join(REPO, "tests", "golden", "c3_clip0.npz"))))
    cases.append(("c5_clip0", C.VIT_L14_T32, CLASSES_3, 3, 1, 0, 5, np.load(os.path.join(REPO, "tests", "golden", "c5_clip0.npz"))))
models = {}
rows = {"fp16": [], "bf16": []}
for name, cfg, cls, n_cls, B, wseed, xseed, g in cases:
    key = (cfg, cls)
    if key not in models:
        models.clear()
        models[key] = VitaCLIP(**model_kwargs(cfg, cls)).cuda().eval()
    m = models[key]
    m.load_state_dict(synth_torch_state(cfg, n_cls, wseed), strict=True)
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=xseed)).cuda()
    for prec in ("fp16", "bf16"):
        m.set_operand_dtype(prec)
        with torch.no_grad():
            lg = m(x)[0].float().cpu().numpy()
        d = np.abs(lg - g["logits"])
        nw = d.max() / np.abs(g["logits"]).max()
        rows[prec].append((name, float(np.abs(g["logits"]).max()), float(d.max()), float(nw), mixed_violation(lg, g["logits"]),
                           bool((lg.argmax(-1) == g["logits"].argmax(-1)).all())))
    m.set_operand_dtype("fp16")
for prec in ("fp16", "bf16"):
    print(f"== {prec} operands")
    for r in rows[prec]:
        print(f"  {r[0]:12s} max|ref| {r[1]:6.3f}  max|d| {r[2]:.2e}  norm-wise {r[3]:.2e}  mixed/bound {r[4]:5.2f}  argmax {'ok' if r[5] else 'DIFFERS'}")
    nw = np.array([r[3] for r in rows[prec]])
    print(f"  norm-wise over {len(nw)} fixtures: max {nw.max():.2e}  median {np.median(nw):.2e};  worst mixed/bound {max(r[4] for r in rows[prec]):.2f}")
