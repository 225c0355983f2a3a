#!/usr/bin/env python
"""Stand-alone launcher for one hot kernel shape (for rocprofv3 --pmc passes and A/B timing).

    python tools/kernel_bench.py fc1|fc2|qkv|out|attn|ln [--iters 5] [--B 64] [--prec fp16]
"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip
from gava_clip_amd import config as _config

ap = argparse.ArgumentParser()
ap.add_argument("which")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--prec", default="fp16")
ap.add_argument("--rows", type=int, default=0, help="override the row count R of the GEMM / LayerNorm cases (tile-quantisation probes)")
ap.add_argument("--cfg", default="VIT_B16_T8", help="config name in gava_clip_amd.config (VIT_L14_T32 for the 320-key attention class)")
a = ap.parse_args()
cfg = getattr(_config, a.cfg)
prec = hip.PREC_NAMES[a.prec]
dt = hip.h16_dtype(prec)
d = torch.device("cuda")
D, F, H, T, G, n1 = cfg.feature_dim, cfg.mlp_dim, cfg.num_heads, cfg.num_frames, cfg.num_global_prompts, cfg.tokens_main
BT = a.B * T
R = BT * n1
if a.rows:
    R = a.rows
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s, scale=1.0, dtype=dt: (torch.randn(*s, device=d, generator=g) * scale).to(dtype)
if a.which == "fc1":
    A, W, b, O = rn(R, D), rn(F, D, scale=D ** -0.5), rn(F, dtype=torch.float32), torch.empty(R, F, dtype=dt, device=d)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_H16_QGELU, prec=prec); fl = 2.0 * R * F * D
elif a.which == "fc2":
    A, W, b, O = rn(R, F), rn(D, F, scale=F ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=prec, resid=O); fl = 2.0 * R * F * D
elif a.which == "fc2nores":
    A, W, b, O = rn(R, F), rn(D, F, scale=F ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=prec); fl = 2.0 * R * F * D
elif a.which == "dhid":     # backward: dHID = dX . W2, QuickGELU' fused (pre-activations as aux)
    A, W, O = rn(R, D), rn(F, D, scale=D ** -0.5), torch.empty(R, F, dtype=dt, device=d)
    PRE = rn(R, F)
    fn = lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16_QGELU_BWD, prec=prec, aux=PRE); fl = 2.0 * R * F * D
elif a.which == "qkv":
    A, W, b, O = rn(R, D), rn(3 * D, D, scale=D ** -0.5), rn(3 * D, dtype=torch.float32), torch.empty(R, 3 * D, dtype=dt, device=d)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_H16, prec=prec, scale_cols=D, scale=0.125); fl = 2.0 * R * 3 * D * D
elif a.which == "out":
    A, W, b, O = rn(R, D), rn(D, D, scale=D ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=prec, resid=O); fl = 2.0 * R * D * D
elif a.which in ("fc1part", "qkvpart"):      # consumers that derive (mean, rstd) from the producers' partials themselves
    N = F if a.which == "fc1part" else 3 * D
    Rp = (R + 255) // 256 * 256
    A, W, O = rn(R, D), rn(N, D, scale=D ** -0.5), torch.empty(R, N, dtype=dt, device=d)
    part = torch.rand(Rp + 32, 4, 2, dtype=torch.float32, device=d, generator=g) * 40 + 200
    fs, ft = W.float().sum(1).contiguous(), rn(N, dtype=torch.float32)
    epi = hip.EPI_H16_QGELU if a.which == "fc1part" else hip.EPI_H16
    kw = {} if a.which == "fc1part" else dict(scale_cols=D, scale=0.125)
    fn = lambda: hip.gemm(A, W, None, O, epilogue=epi, prec=prec, fold_partials=part, fold_s=fs, fold_t=ft, **kw); fl = 2.0 * R * N * D
elif a.which in ("fc2part", "outpart"):      # producers with the row sums pre-reduced per 256-column tile
    K = F if a.which == "fc2part" else D
    Rp = (R + 255) // 256 * 256
    A, W, b, O = rn(R, K), rn(D, K, scale=K ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    x16, part = torch.empty(Rp, D, dtype=dt, device=d), torch.empty(Rp + 32, 4, 2, dtype=torch.float32, device=d)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=prec, resid=O, x16_out=x16, rowsum_out=part, rowsum_reduced=True); fl = 2.0 * R * K * D
elif a.which in ("fc2pair", "outpair"):      # producers with the residual stream as a 16-bit pair (what the big-batch forward launches)
    K = F if a.which == "fc2pair" else D
    Rp = (R + 255) // 256 * 256
    A, W, b = rn(R, K), rn(D, K, scale=K ** -0.5), rn(D, dtype=torch.float32)
    X = rn(R, D, dtype=torch.float32)
    hi = torch.zeros(Rp, D, dtype=dt, device=d); hi[:R] = X.to(dt)
    lo = torch.zeros(Rp, D, dtype=torch.float16, device=d); lo[:R] = (X - hi[:R].float()).half()
    part = torch.empty(Rp + 32, 4, 2, dtype=torch.float32, device=d)
    fn = lambda: hip.gemm(A, W, b, None, epilogue=hip.EPI_F32, prec=prec, resid16=hi, resid_lo=lo, x16_out=hi, xlo_out=lo, rowsum_out=part,
                          rowsum_reduced=True, M=R); fl = 2.0 * R * K * D
elif a.which in ("fc1fold", "qkvfold"):      # consumers of the LayerNorm folding
    N = F if a.which == "fc1fold" else 3 * D
    Rp = (R + 255) // 256 * 256
    A, W, O = rn(R, D), rn(N, D, scale=D ** -0.5), torch.empty(R, N, dtype=dt, device=d)
    st = torch.cat([rn(Rp, 1, scale=0.1, dtype=torch.float32), 1 + rn(Rp, 1, scale=0.1, dtype=torch.float32).abs()], 1).contiguous()
    fs, ft = W.float().sum(1).contiguous(), rn(N, dtype=torch.float32)
    epi = hip.EPI_H16_QGELU if a.which == "fc1fold" else hip.EPI_H16
    kw = {} if a.which == "fc1fold" else dict(scale_cols=D, scale=0.125)
    fn = lambda: hip.gemm(A, W, None, O, epilogue=epi, prec=prec, fold_stats=st, fold_s=fs, fold_t=ft, **kw); fl = 2.0 * R * N * D
elif a.which in ("fc2fold", "outfold"):      # producers: fp32 residual stream + x16 copy + row-sum partials
    K = F if a.which == "fc2fold" else D
    Rp = (R + 255) // 256 * 256
    A, W, b, O = rn(R, K), rn(D, K, scale=K ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    x16, rsum = torch.empty(Rp, D, dtype=dt, device=d), torch.empty(Rp, D // 64, 2, dtype=torch.float32, device=d)
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=prec, resid=O, x16_out=x16, rowsum_out=rsum); fl = 2.0 * R * K * D
elif a.which == "rowstats":
    Rp = (R + 255) // 256 * 256
    rsum = torch.randn(Rp, D // 64, 2, device=d, generator=g).abs()
    fn = lambda: hip.row_stats(rsum, D); fl = 0.0
elif a.which == "patch":
    n, Kp = 196, 768
    x = torch.randn(a.B, 3, T, 224, 224, device=d, generator=g)
    W, b, pos, tim = rn(D, Kp, scale=Kp ** -0.5), rn(D, dtype=torch.float32), rn(n + 1, D, dtype=torch.float32), rn(T, D, dtype=torch.float32)
    O = torch.zeros(BT * (n + 1), D, device=d)
    fn = lambda: hip.gemm(None, W, b, O, epilogue=hip.EPI_F32_PATCH, prec=prec, pos=pos, time=tim, n_patches=n, T=T,
                          M=BT * n, frames=x, frame_size=224, patch=16)
    fl = 2.0 * BT * n * D * Kp
elif a.which == "patchu8":   # the patch embedding reading decoded uint8 videos (360 x 640, 32 frames each) itself
    from gava_clip_amd.preprocess import ClipPreprocessor
    n, Kp = 196, 768
    pre = ClipPreprocessor(num_frames=T, sampling_rate=2, spatial_size=224)
    vids = [torch.randint(0, 256, (32, 360, 640, 3), dtype=torch.uint8, device=d, generator=g) for _ in range(a.B)]
    desc, keep = hip.clip_descriptors(vids, T=T, rate=2, size=224)
    lut = pre.lut(d)
    W, b, pos, tim = rn(D, Kp, scale=Kp ** -0.5), rn(D, dtype=torch.float32), rn(n + 1, D, dtype=torch.float32), rn(T, D, dtype=torch.float32)
    O = torch.zeros(BT * (n + 1), D, device=d)
    fn = lambda: hip.gemm(None, W, b, O, epilogue=hip.EPI_F32_PATCH, prec=prec, pos=pos, time=tim, n_patches=n, T=T,
                          M=BT * n, frame_size=224, patch=16, clips=desc, clip_lut=lut)
    fl = 2.0 * BT * n * D * Kp
elif a.which in ("patchify", "patchifyu8", "patchA"):   # the two-pass patch embedding: patch matrix, then an ordinary GEMM
    n, Kp = 196, 768
    W, b, pos, tim = rn(D, Kp, scale=Kp ** -0.5), rn(D, dtype=torch.float32), rn(n + 1, D, dtype=torch.float32), rn(T, D, dtype=torch.float32)
    O = torch.zeros(BT * (n + 1), D, device=d)
    A = torch.empty(BT * n, Kp, dtype=dt, device=d)
    if a.which == "patchifyu8":
        from gava_clip_amd.preprocess import ClipPreprocessor
        pre = ClipPreprocessor(num_frames=T, sampling_rate=2, spatial_size=224)
        vids = [torch.randint(0, 256, (32, 360, 640, 3), dtype=torch.uint8, device=d, generator=g) for _ in range(a.B)]
        desc, keep = hip.clip_descriptors(vids, T=T, rate=2, size=224)
        lut = pre.lut(d)
        fn = lambda: hip.patchify(A, B=a.B, T=T, size=224, patch=16, prec=prec, clips=desc, clip_lut=lut)
        fl = 0.0
    else:
        x = torch.randn(a.B, 3, T, 224, 224, device=d, generator=g)
        hip.patchify(A, B=a.B, T=T, size=224, patch=16, prec=prec, x=x)
        if a.which == "patchify":
            fn = lambda: hip.patchify(A, B=a.B, T=T, size=224, patch=16, prec=prec, x=x); fl = 0.0
        else:
            fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32_PATCH, prec=prec, pos=pos, time=tim, n_patches=n, T=T, M=BT * n)
            fl = 2.0 * BT * n * D * Kp
elif a.which == "attn":
    QKV, side, O = rn(R, 3 * D), rn(G + 2 * BT, 2 * D), torch.empty(R, D, dtype=dt, device=d)
    fn = lambda: hip.attention(QKV[:, :D], QKV[:, D:2 * D], QKV[:, 2 * D:], O, batch=BT, heads=H, n_q=n1, n_kmain=n1,
                               prec=prec, side_k=side[:, :D], side_v=side[:, D:], n_g=G, T=T, has_summary=True)
    fl = 4.0 * BT * H * n1 * cfg.attn_keys() * 64
elif a.which == "prep":
    from gava_clip_amd.preprocess import ClipPreprocessor
    pre = ClipPreprocessor(num_frames=T, sampling_rate=2, spatial_size=224)
    vids = [torch.randint(0, 256, (32, 360, 640, 3), dtype=torch.uint8, device=d, generator=g) for _ in range(a.B)]
    X = torch.empty(a.B, 3, T, 224, 224, device=d)
    def fn():
        for b, v in enumerate(vids):
            pre(v, out=X[b])
    fl = 0.0
    nbytes = a.B * T * 224 * 224 * (3 * 4 + 3 * (360 / 224) ** 2)   # fp32 out + the source pixels under the crop
elif a.which == "ln":
    X, gm, bt, O = rn(R, D, dtype=torch.float32), rn(D, dtype=torch.float32), rn(D, dtype=torch.float32), torch.empty(R, D, dtype=dt, device=d)
    fn = lambda: hip.layernorm(X, gm, bt, out16=O, prec=prec); fl = 0.0
else:
    raise SystemExit("unknown kernel")
for _ in range(15):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
if a.which == "prep":
    print(f"prep: {ms:.4f} ms per {a.B}-clip batch ({a.B} launches), {nbytes / ms / 1e6:.1f} GB/s algorithmic")
else:
    print(f"{a.which}: {ms:.4f} ms/launch, {fl / ms / 1e9:.1f} TFLOP/s (B={a.B}, {a.prec})")
