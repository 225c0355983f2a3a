#!/usr/bin/env python
"""One process, interleaved rounds: the folded-LayerNorm consumers against the plain GEMMs of the same shape."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip
d = "cuda"; R, D, F = 100864, 768, 3072
g = torch.Generator(device=d).manual_seed(1)
rn = lambda *s, scale=1.0, dtype=torch.float16: (torch.randn(*s, device=d, generator=g) * scale).to(dtype)
Xn = rn(R, D)
part = torch.rand(R + 288, 4, 2, device=d, generator=g) * 40 + 200
stats = torch.cat([rn(R + 256, 1, scale=0.1, dtype=torch.float32), 1 + rn(R + 256, 1, scale=0.1, dtype=torch.float32).abs()], 1).contiguous()
def t(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for N, epi, kw, name in ((3 * D, hip.EPI_H16, dict(scale_cols=D, scale=0.125), "qkv"), (F, hip.EPI_H16_QGELU, {}, "fc1")):
    W = rn(N, D, scale=D ** -0.5); b = rn(N, dtype=torch.float32); O = torch.empty(R, N, dtype=torch.float16, device=d)
    fs, ft = W.float().sum(1).contiguous(), rn(N, dtype=torch.float32)
    fns = {"plain": lambda: hip.gemm(Xn, W, b, O, epilogue=epi, prec=0, **kw),
           "stats": lambda: hip.gemm(Xn, W, None, O, epilogue=epi, prec=0, fold_stats=stats, fold_s=fs, fold_t=ft, **kw),
           "partials": lambda: hip.gemm(Xn, W, None, O, epilogue=epi, prec=0, fold_partials=part, fold_s=fs, fold_t=ft, **kw)}
    res = {k: [] for k in fns}
    for k in fns:
        for _ in range(10): fns[k]()
    for rnd in range(5):
        for k in fns: res[k].append(t(fns[k]))
    print(name, {k: "%.4f (min %.4f)" % (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}, flush=True)
