#!/usr/bin/env python
"""Does a small kernel on a second stream get CUs while the persistent 256x256 GEMM runs?  (GAVA_CU_RESERVE=n leaves
n CUs out of the GEMM's grid.)  Prints the small kernel's elapsed time beside the GEMM and alone."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip
d = torch.device("cuda")
prec, dt = hip.PREC_F16, torch.float16
R, D = 100864, 768
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn(R, D, device=d, generator=g)).to(dt)
W = (torch.randn(3 * D, D, device=d, generator=g) * D ** -0.5).to(dt)
b = torch.randn(3 * D, device=d, generator=g)
O = torch.empty(R, 3 * D, dtype=dt, device=d)
X = torch.randn(512, D, device=d, generator=g)
gm, bt = torch.ones(D, device=d), torch.zeros(D, device=d)
Xo = torch.empty(512, D, dtype=dt, device=d)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
big = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_H16, prec=prec, scale_cols=D, scale=0.125)
small = lambda: hip.layernorm(X, gm, bt, out16=Xo, prec=prec)
for _ in range(3):
    big(); small()
torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
res = []
for trial in range(5):
    e0, e1, f0, f1, go = ev(), ev(), ev(), ev(), ev()
    with torch.cuda.stream(sa):
        go.record()
        e0.record(); big(); e1.record()
    with torch.cuda.stream(sb):
        sb.wait_event(go)
        torch.cuda._sleep(200000)          # ~0.1 ms: the GEMM is resident by now
        f0.record(); small(); f1.record()
    torch.cuda.synchronize()
    res.append((e0.elapsed_time(e1), f0.elapsed_time(f1), e0.elapsed_time(f1)))
a0, a1 = ev(), ev()
a0.record(); small(); a1.record(); torch.cuda.synchronize()
print("reserve", os.environ.get("GAVA_CU_RESERVE"), "gemm ms / small ms / small end after gemm start:",
      ["%.3f / %.3f / %.3f" % r for r in res], "small alone %.3f" % a0.elapsed_time(a1))
