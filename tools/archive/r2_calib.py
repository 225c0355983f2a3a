#!/usr/bin/env python
"""Calibration: the persistent 256^2 GEMM (v3, plain h16 epilogue) and the vendor library (torch.matmul -> hipBLASLt) on
square shapes and on the forward's shapes, random operands, same process (interleaved rounds)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip

def t(fn, iters=10, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

g = torch.Generator(device="cuda").manual_seed(1)
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (100864, 3072, 768), (100864, 2304, 768), (100864, 768, 3072), (100864, 768, 768),
          (8192, 8192, 768), (16384, 4096, 768)]
for (M, N, K) in shapes:
    A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).half()
    O = torch.empty(M, N, device="cuda", dtype=torch.float16)
    O2 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    res = {}
    for rnd in range(3):
        for name, fn in (("ours", lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16, prec=0)),
                         ("vendor", lambda: torch.matmul(A, W.t(), out=O2))):
            res.setdefault(name, []).append(t(fn))
    fl = 2.0 * M * N * K
    err = float((O.float() - O2.float()).abs().max())
    print(f"M={M} N={N} K={K}: ours {min(res['ours']):.4f} ms = {fl/min(res['ours'])/1e9:.0f} TF/s | vendor {min(res['vendor']):.4f} ms = {fl/min(res['vendor'])/1e9:.0f} TF/s | max diff {err:.3g}", flush=True)
