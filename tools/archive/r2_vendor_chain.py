#!/usr/bin/env python
"""What the box sustains on the forward's GEMM sequence alone: the four big products of a ViT-B/16 block at c2
(R = 100864 rows), 12 blocks back to back, (a) with the vendor library behind torch.matmul (fp16 in, fp16 out, no bias, no
epilogue work at all) and (b) with this repo's plain kernels (bias + 16-bit or fp32 store, no LayerNorm fold, no residual).
The forward's own GEMM time (with LayerNorm fold, residual stream, x16 copy, row statistics, QuickGELU) is printed by
bench.py's kernel table; the difference to (a) is what the fused epilogues cost on a power-limited chip."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip
d = torch.device("cuda")
R, D, F = 100864, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, scale=1.0: (torch.randn(*s, device=d, generator=g) * scale).half()
X, H = rn(R, D), rn(R, F)
Wq, Wo, W1, W2 = rn(3 * D, D, scale=D ** -0.5), rn(D, D, scale=D ** -0.5), rn(F, D, scale=D ** -0.5), rn(D, F, scale=F ** -0.5)
Oq, Oo, O1, O2 = (torch.empty(R, n, dtype=torch.float16, device=d) for n in (3 * D, D, F, D))
O32 = torch.empty(R, D, device=d)
bq, bo, b1, b2 = (torch.zeros(n, device=d) for n in (3 * D, D, F, D))
flops = 12 * 2.0 * R * (3 * D * D + D * D + 2 * D * F)


def vendor():
    for _ in range(12):
        torch.matmul(X, Wq.t(), out=Oq); torch.matmul(X, Wo.t(), out=Oo); torch.matmul(X, W1.t(), out=O1); torch.matmul(H, W2.t(), out=O2)


def ours():
    for _ in range(12):
        hip.gemm(X, Wq, bq, Oq, epilogue=hip.EPI_H16, prec=0)
        hip.gemm(X, Wo, bo, O32, epilogue=hip.EPI_F32, prec=0)
        hip.gemm(X, W1, b1, O1, epilogue=hip.EPI_H16, prec=0)
        hip.gemm(H, W2, b2, O32, epilogue=hip.EPI_F32, prec=0)


def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for r in range(3):
    tv, to = timeit(vendor), timeit(ours)
    print(f"round {r}: vendor chain {tv:.3f} ms = {flops / tv / 1e9:.0f} TF/s | plain kernels of this repo {to:.3f} ms = {flops / to / 1e9:.0f} TF/s"
          f"   (48 GEMMs, {flops / 1e12:.2f} TFLOP)")
