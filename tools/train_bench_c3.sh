#!/bin/bash
# training step at the c3 shape (T = 16, 32 clips, 400 classes): python tools/train_bench.py has c2 only
cd "$(dirname "$0")/.."
python - <<'PY'
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs, CLASSES_400
cfg = C.VIT_B16_T16
m = VitaCLIP(**model_kwargs(cfg, CLASSES_400)).cuda().train()
x = torch.randn(32, 3, 16, 224, 224, device="cuda"); y = torch.randint(0, 400, (32,), device="cuda")
opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-5)
for i in range(4):
    t0 = time.perf_counter(); loss = torch.nn.functional.cross_entropy(m(x)[0], y); torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True); loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter(); opt.step(); torch.cuda.synchronize()
    print(f"step {i}: forward {(t1-t0)*1e3:.1f} ms backward {(t2-t1)*1e3:.1f} ms loss {float(loss.detach()):.4f}", flush=True)
PY
