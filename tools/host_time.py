#!/usr/bin/env python
"""Host enqueue time vs GPU time of one forward at c2 (is the launch path ever the bottleneck?)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs
cfg = C.VIT_B16_T8
cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt")
model = VitaCLIP(**model_kwargs(cfg, cls_path)).cuda().eval()
x = torch.randn(64, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")
with torch.no_grad():
    for _ in range(5): model(x)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter(); model(x); host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"host enqueue per forward: median {sorted(host)[10]*1e3:.2f} ms, min {min(host)*1e3:.2f}, max {max(host)*1e3:.2f}; "
      f"20 forwards enqueued in {(t1-t0)*1e3:.1f} ms, finished after {(t2-t0)*1e3:.1f} ms")
