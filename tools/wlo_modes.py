#!/usr/bin/env python
"""The weight-lo parity modes against the default: c2 forward time (same process, interleaved rounds) and the logits error on
the reference fixtures.  GPU box.

    python tools/wlo_modes.py [--modes fp16,fp16+wlo,fp16+wlo8] [--no-time] [--all]     (--all: also c2_full / c3 / c5 fixtures)"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from gava_clip_amd import VitaCLIP, synth
from gava_clip_amd import config as C
from helpers import CLASSES_3, GOLDEN_LOGIT_CASES, golden_case, model_kwargs, synth_torch_state, mixed_violation

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="fp16,fp16+wlo")
ap.add_argument("--no-time", action="store_true")
ap.add_argument("--all", action="store_true")
a = ap.parse_args()
modes = a.modes.split(",")

if not a.no_time:
    cfg = C.VIT_B16_T8
    model = VitaCLIP(**model_kwargs(cfg, CLASSES_3)).cuda().eval()
    model.load_state_dict(synth_torch_state(cfg, 3, 0), strict=True)
    x = torch.from_numpy(synth.synth_clip(64, cfg.num_frames, cfg.input_size, seed=4242)).cuda()

    def run(n=20):
        with torch.no_grad():
            for _ in range(5):
                model(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                model(x)
            e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms = {m: [] for m in modes}
    for rep in range(3):
        for m in modes:
            model.set_operand_dtype(m)
            ms[m].append(run())
    for m in modes:
        t = float(np.median(ms[m]))
        print(f"c2 forward, {m:10s}: {t:7.3f} ms = {64 / t * 1e3:7.1f} clips/s   (rounds {['%.2f' % v for v in ms[m]]})", flush=True)
    del model, x

names = [n for n in GOLDEN_LOGIT_CASES if a.all or n.startswith("c1_")]
rows = {m: [] for m in modes}
models = {}
for name in names:
    cfg, class_file, n_cls, B, wseed, xseed = golden_case(name)
    g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    key = (cfg, class_file)
    if key not in models:
        models.clear()
        models[key] = VitaCLIP(**model_kwargs(cfg, class_file)).cuda().eval()
    model = models[key]
    model.load_state_dict(synth_torch_state(cfg, n_cls, wseed), strict=True)
    x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=xseed)).cuda()
    line = f"{name:10s} max|ref| {np.abs(g['logits']).max():5.3f}"
    for m in modes:
        model.set_operand_dtype(m)
        with torch.no_grad():
            lg = model(x)[0].float().cpu().numpy()
        d = np.abs(lg - g["logits"])
        nw = float(d.max() / np.abs(g["logits"]).max())
        vf = float(np.abs(model.last["video_features"].cpu().numpy() - g["video_features"]).max() / np.abs(g["video_features"]).max())
        rows[m].append((nw, mixed_violation(lg, g["logits"]), vf))
        line += f" | {m}: norm-wise {nw:.2e} mixed {rows[m][-1][1]:4.2f} video {vf:.2e}"
    print(line, flush=True)
for m in modes:
    r = np.array(rows[m])
    print(f"{m:10s}: norm-wise max {r[:, 0].max():.2e} median {np.median(r[:, 0]):.2e}; worst mixed/bound {r[:, 1].max():.2f}; video features max {r[:, 2].max():.2e}")
