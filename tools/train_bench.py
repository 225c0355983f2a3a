#!/usr/bin/env python
"""One training step (forward + loss.backward()) of the drop-in model at c2-like sizes: ms per phase, worst gradient sanity.
    python tools/train_bench.py [--B 64] [--iters 3]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
cfg = C.VIT_B16_T8
cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt")
torch.manual_seed(0)
model = VitaCLIP(**model_kwargs(cfg, cls_path)).cuda().train()
x = torch.randn(a.B, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")
y = torch.randint(0, 3, (a.B,), device="cuda")
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)


def step():
    t0 = time.perf_counter()
    logits = model(x)[0]
    loss = torch.nn.functional.cross_entropy(logits, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return float(loss), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3


for i in range(a.iters + 1):
    loss, f, b, o = step()
    print(f"step {i}: loss {loss:.4f}  forward {f:.1f} ms  backward {b:.1f} ms  optimizer {o:.1f} ms  "
          f"-> {a.B / ((f + b + o) / 1e3):.0f} clips/s   peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
