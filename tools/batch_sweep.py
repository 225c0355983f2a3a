#!/usr/bin/env python
"""clips/s of the eval forward vs clips per forward (is there a cache-residency sweet spot below B = 64?)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs
cfg = C.VIT_B16_T8
m = VitaCLIP(**model_kwargs(cfg, os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt"))).cuda().eval()
for B in (8, 16, 24, 32, 48, 64, 96, 128):
    x = torch.randn(B, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")
    with torch.no_grad():
        for _ in range(4): m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): m(x)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"B={B:4d}: {ms:8.3f} ms/forward  {B / ms * 1e3:8.1f} clips/s", flush=True)
