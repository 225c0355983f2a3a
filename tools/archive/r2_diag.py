#!/usr/bin/env python
"""Round-2 diagnostics on the GPU box: (1) split-precision CLS path of the last block on/off against the c1 golden
(video features, logits); (2) which gradient is missing / non-finite under autocast + GradScaler."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from gava_clip_amd import VitaCLIP, synth
from gava_clip_amd.config import TINY, VIT_B16_T8
from helpers import CLASSES_3, model_kwargs, synth_torch_state, rel_to_max

g = np.load(os.path.join(REPO, "tests/golden/c1_b16.npz"))
m = VitaCLIP(**model_kwargs(VIT_B16_T8, CLASSES_3)); m.load_state_dict(synth_torch_state(VIT_B16_T8, 3), strict=True)
m = m.cuda().eval(); m.debug_taps = True
x = torch.from_numpy(synth.synth_clip(2, 8, 224)).cuda()
for sp in (True, False):
    m.split_last_block = sp
    with torch.no_grad():
        lg = m(x)[0].cpu().numpy()
    vf = m.last["video_features"].cpu().numpy()
    d = np.abs(lg - g["logits"])
    cls = m.last["cls_rows"].cpu().numpy()
    print(f"split_last_block={sp}: logits max-abs {d.max():.3e} rel-to-max {d.max()/np.abs(g['logits']).max():.3e} elementwise {(d/np.abs(g['logits'])).max():.3e} "
          f"rms {np.sqrt((d**2).mean()):.3e}; video rel-to-max {rel_to_max(vf, g['video_features']):.3e} rms {np.sqrt(((vf-g['video_features'])**2).mean()):.3e}; "
          f"last-layer cls rel {rel_to_max(cls[11], g['cls_rows'][11]):.3e} rms {np.sqrt(((cls[11]-g['cls_rows'][11])**2).mean()):.3e}")

m = VitaCLIP(**{**model_kwargs(TINY, CLASSES_3), "use_fp16": True}); m.load_state_dict(synth_torch_state(TINY, 3), strict=True)
m = m.cuda().train()
x = torch.from_numpy(synth.synth_clip(2, TINY.num_frames, TINY.input_size)).cuda()
with torch.autocast("cuda", dtype=torch.float16):
    logits = m(x)[0]
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor([0, 2], device="cuda"))
print("autocast logits dtype", logits.dtype, "loss", float(loss))
scaler = torch.amp.GradScaler("cuda")
scaler.scale(loss).backward()
for n, p in m.named_parameters():
    if p.requires_grad:
        if p.grad is None: print("  NONE  ", n)
        elif not torch.isfinite(p.grad).all(): print("  NONFINITE", n, float(p.grad.float().abs().nan_to_num(0, 0, 0).max()))
print("scale", scaler.get_scale())
