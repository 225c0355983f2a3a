#!/usr/bin/env python
"""Same-process A/B of whole-forward switches at c2:  python tools/ab_forward.py  (ms/forward per setting)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs

cfg = C.VIT_B16_T8
cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt")
torch.manual_seed(0)
model = VitaCLIP(**model_kwargs(cfg, cls_path)).cuda().eval()
x = torch.randn(64, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")


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


ref = None
for rep in range(2):
    for name, side, cache in (("text on side stream", True, False), ("text on main stream", False, False), ("text cached", False, True)):
        model.text_on_side_stream, model.cache_text_features = side, cache
        model._text_cache = None
        ms = run()
        with torch.no_grad():
            lg = model(x)[0]
        ref = lg if ref is None else ref
        print(f"{name}: {ms:.3f} ms/forward  max|dlogits| vs first setting {float((lg - ref).abs().max()):.2e}", flush=True)
