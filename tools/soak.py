#!/usr/bin/env python
"""Short soak: alternate eval forwards and training steps at c2 size; checks finiteness, bitwise-stable eval logits
between identical calls, falling loss and flat memory.  python tools/soak.py [--rounds 5]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs

ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--B", type=int, default=32)
a = ap.parse_args()
cfg = C.VIT_B16_T8
cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt")
torch.manual_seed(0)
m = VitaCLIP(**model_kwargs(cfg, cls_path)).cuda()
x = torch.randn(a.B, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")
y = torch.randint(0, 3, (a.B,), device="cuda")
opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=2e-4)
losses, mems = [], []
for r in range(a.rounds):
    m.eval()
    with torch.no_grad():
        l1 = m(x)[0]
        for _ in range(10):
            l2 = m(x)[0]
    assert torch.equal(l1, l2) and bool(torch.isfinite(l1).all()), "eval logits unstable"
    m.train()
    for _ in range(5):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x)[0], y)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    losses.append(float(loss.detach())); mems.append(torch.cuda.max_memory_allocated() / 2 ** 30)
    print(f"round {r}: loss {losses[-1]:.4f}  peak mem {mems[-1]:.2f} GiB", flush=True)
assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
assert losses[-1] < losses[0], losses
assert mems[-1] - mems[1] < 0.5, mems
print("soak ok")
