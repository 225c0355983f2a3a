#!/usr/bin/env python
"""Per-stage cycle shares of the persistent GEMM (v3) from in-kernel s_memtime stamps: vmcnt wait, barrier wait, stage
body and epilogue, per wave group (gava_debug_set_buffer).  The stamps are compiled in only with -DGAVA_STAMPS (they cost
scalar registers the product kernels need): build the variant first,
    tools/ab_build.sh stamps -DGAVA_STAMPS  &&  GAVA_HIP_LIB=gava_clip_amd/libgava_hip_stamps.so python tools/gemm_stamps.py fc2"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GAVA_GEMM_VARIANT", "3")   # the persistent v3 kernel carries the stamps
import torch
from gava_clip_amd import hip
lib = hip.load()
lib.gava_debug_set_buffer.argtypes = [C.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "fc2"
R, D, F = 100864, 768, 3072
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s, scale=1.0, dtype=torch.float16: (torch.randn(*s, device="cuda", generator=g) * scale).to(dtype)
if which == "fc2":
    A, W, b, O = rn(R, F), rn(D, F, scale=F ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=0, resid=O)
elif which == "qkv":
    A, W, b, O = rn(R, D), rn(3 * D, D, scale=D ** -0.5), rn(3 * D, dtype=torch.float32), torch.empty(R, 3 * D, dtype=torch.float16, device="cuda")
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_H16, prec=0, scale_cols=D, scale=0.125)
elif which in ("fc1fold", "qkvfold"):   # consumers with LayerNorm folded in
    N = F if which == "fc1fold" else 3 * D
    Rp = (R + 255) // 256 * 256
    A, W, O = rn(R, D), rn(N, D, scale=D ** -0.5), torch.empty(R, N, dtype=torch.float16, device="cuda")
    st = torch.cat([rn(Rp, 1, scale=0.1, dtype=torch.float32), 1 + rn(Rp, 1, scale=0.1, dtype=torch.float32).abs()], 1).contiguous()
    fs_, ft_ = W.float().sum(1).contiguous(), rn(N, dtype=torch.float32)
    if which == "fc1fold":
        fn = lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16_QGELU, prec=0, fold_stats=st, fold_s=fs_, fold_t=ft_)
    else:
        fn = lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16, prec=0, scale_cols=D, scale=0.125, fold_stats=st, fold_s=fs_, fold_t=ft_)
elif which in ("fc1part", "qkvpart"):   # consumers in the product's mode: (mean, rstd) derived from the producers' partials
    N = F if which == "fc1part" else 3 * D
    Rp = (R + 255) // 256 * 256
    A, W, O = rn(R, D), rn(N, D, scale=D ** -0.5), torch.empty(R, N, dtype=torch.float16, device="cuda")
    part = torch.rand(Rp + 32, 4, 2, dtype=torch.float32, device="cuda", generator=g) * 40 + 200
    fs_, ft_ = W.float().sum(1).contiguous(), rn(N, dtype=torch.float32)
    if which == "fc1part":
        fn = lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16_QGELU, prec=0, fold_partials=part, fold_s=fs_, fold_t=ft_)
    else:
        fn = lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16, prec=0, scale_cols=D, scale=0.125, fold_partials=part, fold_s=fs_, fold_t=ft_)
elif which in ("outfold", "fc2fold", "out"):
    K = F if which == "fc2fold" else D
    Rp = (R + 255) // 256 * 256
    A, W, b, O = rn(R, K), rn(D, K, scale=K ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    x16, rsum = torch.empty(Rp, D, dtype=torch.float16, device="cuda"), torch.empty(Rp, D // 64, 2, dtype=torch.float32, device="cuda")
    if which == "out":
        fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=0, resid=O)
    else:
        fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=0, resid=O, x16_out=x16, rowsum_out=rsum)
else:
    A, W, b, O = rn(R, D), rn(F, D, scale=D ** -0.5), rn(F, dtype=torch.float32), torch.empty(R, F, dtype=torch.float16, device="cuda")
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_H16_QGELU, prec=0)
for _ in range(10): fn()
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
lib.gava_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
fn(); torch.cuda.synchronize()
lib.gava_debug_set_buffer(None)
d = dbg.view(256, 8, 8).double()
for grp, name in ((slice(0, 4), "group0 (waves 0-3)"), (slice(4, 8), "group1 (waves 4-7)")):
    x = d[:, grp].reshape(-1, 8)
    G = x[:, 5].mean()
    tot = x[:, :5].sum(1).mean()
    print(f"{which} {name}: stages/wave {G:.0f}; per stage cycles: vmcnt-wait {x[:,0].mean()/G:.0f}  barrier {x[:,1].mean()/G:.0f}  body {x[:,2].mean()/G:.0f}  epilogue {x[:,4].mean()/G:.0f}  total {tot/G:.0f}"
          + (f"  [epilogue: bias wait {x[:,6].mean()/G:.0f}, row groups {x[:,7].mean()/G:.0f}]" if x[:,6].sum() > 0 else ""))
