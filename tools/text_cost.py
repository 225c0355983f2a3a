#!/usr/bin/env python
"""What the text tower costs the c2 forward's critical path (GPU box): the same model timed with the text tower on its own stream
(default), with the eval-time text-feature cache on (no text tower in the forward) and with the text tower on the main stream
(serial), alternating rounds in one process.

    python tools/text_cost.py [--config VIT_B16_T8] [--B 64] [--rounds 4]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="VIT_B16_T8"); ap.add_argument("--B", type=int, default=64); ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--classes", default=os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt"))
a = ap.parse_args()
cfg = getattr(C, a.config)
torch.manual_seed(0)
m = VitaCLIP(**model_kwargs(cfg, a.classes)).cuda().eval()
x = torch.randn(a.B, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")


def timed(n=30):
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


modes = {"side stream (default)": dict(cache_text_features=False, text_on_side_stream=True),
         "cached (no text tower)": dict(cache_text_features=True, text_on_side_stream=True),
         "main stream (serial)": dict(cache_text_features=False, text_on_side_stream=False)}
modes["side stream, lowest priority"] = dict(cache_text_features=False, text_on_side_stream=True, _prio="low")
modes["side stream, highest priority"] = dict(cache_text_features=False, text_on_side_stream=True, _prio="high")
try:
    import ctypes
    lo_hi = (ctypes.c_int(), ctypes.c_int())
    hipl = ctypes.CDLL("libamdhip64.so")
    hipl.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_hi[0]), ctypes.byref(lo_hi[1]))
    print("stream priority range (least, greatest):", lo_hi[0].value, lo_hi[1].value)
    least, greatest = lo_hi[0].value, lo_hi[1].value
except Exception as e:
    print("priority range unavailable:", e); least, greatest = 0, -1
res = {k: [] for k in modes}
for r in range(a.rounds):
    for k, kw in modes.items():
        for kk, v in kw.items():
            if not kk.startswith("_"):
                setattr(m, kk, v)
        pr = kw.get("_prio")
        # default modes: the model picks its own text stream again (model._overlapping_stream tries candidates until one runs beside
        # the main stream); priority modes: a raw new stream, as before
        m._text_stream = torch.cuda.Stream(device=x.device, priority=(least if pr == "low" else greatest)) if pr else None
        m._text_cache = None
        res[k].append(timed())
        if not pr and kw["text_on_side_stream"] and not kw["cache_text_features"]:
            print(f"round {r}: text stream candidates tried {m.last.get('text_stream_candidates')}")
for k, v in res.items():
    print(f"{k:26s} median {sorted(v)[len(v) // 2]:.3f} ms  ({' '.join('%.3f' % t for t in v)})")
