#!/usr/bin/env python
"""Wall-clock timeline of the persistent 256^2 GEMM (diagnostic build: tools/ab_build.sh tl -DGAVA_TIMELINE): per workgroup
and tile, when the k-loop starts (its first stage has landed: everything the previous epilogue left in flight has drained),
when the epilogue starts and when its last instruction is issued.  10 ns ticks (s_memrealtime).

    GAVA_HIP_LIB=gava_clip_amd/libgava_hip_tl.so python tools/gemm_timeline.py outpart|fc2part|fc1part"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gava_clip_amd import hip
lib = hip.load()
lib.gava_debug_set_buffer.argtypes = [C.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "outpart"
R, D, F = 100864, 768, 3072
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s, scale=1.0, dtype=torch.float16: (torch.randn(*s, device="cuda", generator=g) * scale).to(dtype)
Rp = (R + 255) // 256 * 256
if which in ("outpart", "fc2part"):
    K = F if which == "fc2part" else D
    A, W, b, O = rn(R, K), rn(D, K, scale=K ** -0.5), rn(D, dtype=torch.float32), rn(R, D, dtype=torch.float32)
    x16, part = torch.empty(Rp, D, dtype=torch.float16, device="cuda"), torch.empty(Rp + 32, 4, 2, dtype=torch.float32, device="cuda")
    fn = lambda: hip.gemm(A, W, b, O, epilogue=hip.EPI_F32, prec=0, resid=O, x16_out=x16, rowsum_out=part, rowsum_reduced=True)
else:
    A, W, O = rn(R, D), rn(F, D, scale=D ** -0.5), torch.empty(R, F, dtype=torch.float16, device="cuda")
    part = torch.rand(Rp + 32, 4, 2, dtype=torch.float32, device="cuda", generator=g) * 40 + 200
    fs_, ft_ = W.float().sum(1).contiguous(), rn(F, dtype=torch.float32)
    fn = lambda: hip.gemm(A, W, None, O, epilogue=hip.EPI_H16_QGELU, prec=0, fold_partials=part, fold_s=fs_, fold_t=ft_)
for _ in range(10): fn()
dbg = torch.zeros(256 * 2 * 32, dtype=torch.int64, device="cuda")
lib.gava_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
fn(); torch.cuda.synchronize()
lib.gava_debug_set_buffer(None)
d = dbg.view(256, 2, 32).cpu().numpy().astype(np.float64)
t0 = d[:, :, 0][d[:, :, 0] > 0].min()
split = os.environ.get("TL_SPLIT") == "1"      # stagger builds: even and odd slots (blockIdx >> 3) of every XCD apart
slot = (np.arange(256) >> 3) & 1
if os.environ.get("TL_SPLIT_XCD") == "1":      # GAVA_STAGGER_MODE=2: XCDs 0-3 against XCDs 4-7
    slot = ((np.arange(256) & 7) >= 4).astype(int)
for grp, par in [(g_, p_) for g_ in (0, 1) for p_ in ((0, 1) if split else (None,))]:
    x = d[:, grp] if par is None else d[slot == par, grp]
    nt = x[:, 31].astype(int)
    print(f"== {which} wave group {grp}{'' if par is None else ' slot parity %d' % par}: tiles per workgroup {np.bincount(nt)[1:].tolist() if nt.max() else nt[:4]}")
    rows = []
    for j in range(min(int(nt.max()), 10)):        # the kernel stamps the first ten tiles of a workgroup
        sel = nt > j
        ks, es, ee = x[sel, 3 * j] - t0, x[sel, 3 * j + 1] - t0, x[sel, 3 * j + 2] - t0
        nxt = np.where(nt[sel] > j + 1, x[sel, 3 * (j + 1)] - t0, x[sel, 30] - t0)     # next tile's k-loop start, or the kernel end
        print(f"  tile {j}: n={sel.sum():3d}  k-loop start {ks.mean() / 100:7.2f} us (sd {ks.std() / 100:5.2f})   k-loop {np.mean(es - ks) / 100:6.2f} us   "
              f"epilogue issue {np.mean(ee - es) / 100:6.2f} us   drain until next k-loop / end {np.mean(nxt - ee) / 100:6.2f} us   tile total {np.mean(nxt - ks) / 100:6.2f} us")
    print(f"  kernel end: {(x[:, 30] - t0).mean() / 100:.2f} us (min {(x[:, 30] - t0).min() / 100:.2f}, max {(x[:, 30] - t0).max() / 100:.2f})")
