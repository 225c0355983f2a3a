#!/usr/bin/env python
"""Latency of the prompt path's small GEMMs (B*T rows) stand-alone: a chain of dependent launches, as the forward issues them.
    GAVA_SMALL_NST=2|3|4 python tools/small_gemm_latency.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip
d = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s, scale=1.0, dtype=torch.float16: (torch.randn(*s, device=d, generator=g) * scale).to(dtype)
BT, D, F = 512, 768, 3072
cases = [("cls_proj  512x768x768  f32", BT, D, D, hip.EPI_F32), ("sqkv      512x2304x768 h16", BT, 3 * D, D, hip.EPI_H16),
         ("side kv   1032x1536x768 h16", 1032, 2 * D, D, hip.EPI_H16), ("cls fc1   512x3072x768 qgelu", BT, F, D, hip.EPI_H16_QGELU),
         ("cls fc2   512x768x3072 f32", BT, D, F, hip.EPI_F32)]
for name, M, N, K, epi in cases:
    A, W = rn(M, K), rn(N, K, scale=K ** -0.5)
    out = torch.empty(M, N, device=d, dtype=torch.float32 if epi == hip.EPI_F32 else torch.float16)
    fn = lambda: hip.gemm(A, W, None, out, epilogue=epi, prec=0)
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch (back to back, same stream)")
