#!/usr/bin/env python
"""Calibration of the k-loops: the persistent 256^2 GEMM (plain h16 epilogue), the ping-pong 256^2 kernel (GAVA_KERNEL_PP) and
the vendor library (torch.matmul -> hipBLASLt) on square shapes and on the forward's shapes, uniform random [-1, 1) operands,
ONE process, interleaved rounds, 10 warm-up launches per timing (the protocol of bench.py's kernel table; VERDICT r3 weak 4:
the two vendor yardsticks of earlier rounds differed in warm-up and in how many kernels shared the process).

    python tools/gemm_vs_vendor.py [--no-pp]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip


def t(fn, iters=20, warm=10):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device="cuda").manual_seed(1)
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (100864, 3072, 768), (100864, 2304, 768), (100864, 768, 3072), (100864, 768, 768),
          (8192, 8192, 768), (16384, 4096, 768)]
pp = "--no-pp" not in sys.argv
for (M, N, K) in shapes:
    A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).half()
    O = torch.empty(M, N, device="cuda", dtype=torch.float16)
    O2 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    O3 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    runs = [("ours256", lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16, prec=0, kernel=hip.KERNEL_256)),
            ("vendor", lambda: torch.matmul(A, W.t(), out=O2))]
    if pp:
        runs.append(("pingpong", lambda: hip.gemm(A, W, None, O3, epilogue=hip.EPI_H16, prec=0, kernel=hip.KERNEL_PP)))
    res = {}
    for rnd in range(3):
        for name, fn in runs:
            res.setdefault(name, []).append(t(fn))
    fl = 2.0 * M * N * K
    line = f"M={M} N={N} K={K}:"
    for name, _ in runs:
        ms = sorted(res[name])[1]
        line += f" {name} {ms:.4f} ms = {fl / ms / 1e9:.0f} TF/s |"
    line += f" max diff ours-vendor {float((O.float() - O2.float()).abs().max()):.3g}"
    if pp:
        line += f", pingpong-ours {float((O3.float() - O.float()).abs().max()):.3g}"
    print(line, flush=True)
