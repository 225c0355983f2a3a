#!/usr/bin/env python
"""Cycle shares of the attention kernel phases from in-kernel stamps (diagnostic)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gava_clip_amd import hip
lib = hip.load()
lib.gava_debug_set_buffer.argtypes = [C.c_void_p]
D, H, T, G, n1 = 768, 12, 8, 8, 197
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 512     # 16 frames x 12 heads = 192 workgroups: one per CU, waves alone on their SIMD
R = BT * n1
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g).half()
QKV, side, O = rn(R, 3 * D), rn(G + 2 * BT, 2 * D), torch.empty(R, D, dtype=torch.float16, device="cuda")
fn = lambda: hip.attention(QKV[:, :D], QKV[:, D:2 * D], QKV[:, 2 * D:], O, batch=BT, heads=H, n_q=n1, n_kmain=n1, prec=0,
                           side_k=side[:, :D], side_v=side[:, D:], n_g=G, T=T, has_summary=True)
for _ in range(10): fn()
dbg = torch.zeros(2 * 4096 * 4 * 4, dtype=torch.int64, device="cuda")
lib.gava_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
fn(); torch.cuda.synchronize()
lib.gava_debug_set_buffer(None)
if BT * H >= 1024 and os.environ.get("GAVA_ATTN_PERSIST", "1") != "0" and os.environ.get("GAVA_HIP_LIB", "").endswith("stamps.so"):
    # persistent kernel (stamps build): per-wave sums over its problems
    d = dbg[:256 * 8 * 8].view(256, 8, 8).double()
    npb = d[..., 7].mean()
    for name, sl in (("waves 0-6 (one pair each)", slice(0, 7)), ("wave 7 (no pair)", slice(7, 8))):
        x = d[:, sl].reshape(-1, 8)
        print("%s, cycles per problem: issue next loads %.0f | compute %.0f | wait prefetch %.0f | stores %.0f | barrier %.0f | total %.0f; clock %.2f GHz" % (
            name, *(x[:, i].mean() / npb for i in range(5)), x[:, 5].mean() / npb, x[:, 5].mean() / x[:, 6].mean() * 0.1))
    for w in range(8):
        x = d[:, w]
        print("  wave %d: issue %.0f compute %.0f wait %.0f stores %.0f barrier %.0f" % (w, *(x[:, i].mean() / npb for i in range(5))))
    sys.exit(0)
nwg = min(4096, BT * H)
ph = dbg[4096 * 16:].view(4096, 4, 4)[:nwg].double()
d = dbg[:4096 * 16].view(4096, 4, 4)[:nwg].double()
if ph.sum() > 0:   # -DGAVA_ATTN_STAMPS build: phases of a wave's first query-tile pair
    print("first pair, cycles: K Q^T %.0f | mask + row max %.0f | exp %.0f | P V (+ row sums) %.0f" % tuple(ph[..., i].mean() for i in range(4)))
print("per wave cycles: issue+wait K/V,Q loads %.0f | LDS write + barrier %.0f | compute %.0f (%.1f q-tiles => %.0f / q-tile)" % (
    d[..., 0].mean(), d[..., 1].mean(), d[..., 2].mean(), d[..., 3].mean(), (d[..., 2] / d[..., 3]).mean()))
