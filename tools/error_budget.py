#!/usr/bin/env python
"""Where the logits error of a 16-bit-operand vision tower comes from (CPU, the pinned oracle).

The oracle (oracle/vita_oracle.py, pinned to the reference fixtures) is run on the c1 reference fixtures with the roundings
of the MFMA path applied SELECTIVELY to the vision tower's GEMM / attention operands (the text tower stays exact: the HIP
text tower is within 5e-6 of the reference), and the logits are compared with the REFERENCE's (tests/golden/c1_b16*.npz):

    all16        every operand (activations and weights) rounded to the 16-bit type
    act16        activations only (weights exact fp32)
    w16          weights only (activations, i.e. everything else, exact fp32)
    w16:<fam>    only the weights of one family: fc2 | out | qkv | fc1 | patch | side (cls_proj + summary attention)
    wlo16        act16 + weights as hi + lo 16-bit pairs (what `operand_dtype="fp16+wlo"` computes: A.[W_hi | W_lo]^T)
    wlo8         act16 + the lo product at 8 bits: bf8(act16) . e4m3(2^s (W - W_hi))  (`operand_dtype="fp16+wlo8"`)
    wlo8t        the same with the activation's bf8 copy made by TRUNCATION of the fp16 (a byte permute) instead of rounding

Output: one row per fixture, norm-wise error max|d| / max|ref| per mode, then max / median per mode.

    python tools/error_budget.py [--dtype fp16|bf16] [--seeds 0,5,6] [--modes all16,act16,w16,wlo16,wlo8]
"""
import argparse, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from gava_clip_amd import synth, tokenizer
from gava_clip_amd import config as C
from oracle.vita_oracle import Oracle
from helpers import CLASSES_3, synth_torch_state

FAMILIES = {"mlp.fc2.": "fc2", "attn.out_proj.": "out", "attn.q_proj.": "qkv", "attn.k_proj.": "qkv", "attn.v_proj.": "qkv",
            "mlp.fc1.": "fc1", "patch_embed.": "patch", "cls_proj.": "side", "summary_attn_layer.": "side"}


def family(name):
    if not name.startswith("visual."):
        return None
    if "summary_attn_layer." in name or "cls_proj." in name:
        return "side"
    for k, v in FAMILIES.items():
        if k in name:
            return v
    return None


class BudgetOracle(Oracle):
    """mode: see the module docstring.  dt: torch.float16 / torch.bfloat16."""

    def __init__(self, cfg, params, tok, mode, dt):
        super().__init__(cfg, params, tok, operand_dtype=None)
        self.mode, self.dt = mode, dt
        self.fam = {}
        for k, v in params.items():
            if v.dim() >= 2:
                self.fam[v.data_ptr()] = family(k)
        self.in_vision = False
        self._wcache = {}

    def r16(self, t):
        return t.to(self.dt).to(torch.float32)

    def _r(self, t):      # attention operands (q, k, P, v): activations
        if self.in_vision and self.mode in ("all16", "act16", "wlo16", "wlo8", "wlo8t"):
            return self.r16(t)
        return t

    def vision(self, x):
        self.in_vision = True
        try:
            return super().vision(x)
        finally:
            self.in_vision = False

    def linear(self, x, w, b=None):
        fam = self.fam.get(w.data_ptr())
        mode = self.mode
        if not self.in_vision or fam is None:        # text tower, final projection (split precision in the HIP path): exact
            y = x @ w.t()
        elif mode == "all16":
            y = self.r16(x) @ self.r16(w).t()
        elif mode == "act16":
            y = self.r16(x) @ w.t()
        elif mode == "w16":
            y = x @ self.r16(w).t()
        elif mode.startswith("w16:"):
            y = x @ (self.r16(w) if fam == mode[4:] else w).t()
        elif mode == "wlo16":
            hi = self.r16(w)
            y = self.r16(x) @ (hi + self.r16(w - hi)).t()
        elif mode in ("wlo8", "wlo8t"):
            key = w.data_ptr()
            if key not in self._wcache:
                hi = self.r16(w)
                lo = w - hi
                # per-tensor power-of-two scale: the largest |lo| lands in [128, 256) of e4m3's range (max 448)
                s = 2.0 ** (7 - int(np.floor(np.log2(float(lo.abs().max()) + 1e-45))))
                lo8 = (lo * s).to(torch.float8_e4m3fn).to(torch.float32) / s
                self._wcache[key] = (hi, lo8)
            hi, lo8 = self._wcache[key]
            a16 = self.r16(x)
            if mode == "wlo8t" and self.dt == torch.float16:
                a8 = (a16.to(torch.float16).view(torch.int16) & -256).view(torch.float16).to(torch.float32)   # high byte of the fp16 = e5m2
            else:
                a8 = a16.to(torch.float8_e5m2).to(torch.float32)
            y = a16 @ hi.t() + a8 @ lo8.t()
        else:
            raise ValueError(mode)
        return y if b is None else y + b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--seeds", default="")
    ap.add_argument("--modes", default="all16,act16,w16,w16:fc2,w16:out,w16:qkv,w16:fc1,w16:patch,w16:side,wlo16,wlo8,wlo8t")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    modes = args.modes.split(",")
    cfg = C.VIT_B16_T8
    tok = tokenizer.tokenize(tokenizer.prompt_texts(tokenizer.read_class_names(CLASSES_3), cfg.text_num_prompts))
    files = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "c1_b16*.npz")),
                   key=lambda f: int(np.load(f)["wseed"]) if "wseed" in np.load(f).files else 0)
    want = set(int(s) for s in args.seeds.split(",")) if args.seeds else None
    table = {}
    print(f"# operand type {args.dtype}; norm-wise logits error max|d| / max|ref| against the REFERENCE fixtures")
    print("fixture      " + " ".join(f"{m:>9s}" for m in modes))
    for f in files:
        g = np.load(f)
        wseed = int(g["wseed"]) if "wseed" in g.files else 0
        xseed = int(g["xseed"]) if "xseed" in g.files else 1234
        if want is not None and wseed not in want:
            continue
        params = synth_torch_state(cfg, 3, wseed)
        x = torch.from_numpy(synth.synth_clip(2, cfg.num_frames, cfg.input_size, seed=xseed))
        ref = g["logits"]
        row = []
        for m in modes:
            o = BudgetOracle(cfg, params, tok, m, dt)
            lg = o.forward(x)["logits"].numpy()
            row.append(float(np.abs(lg - ref).max() / np.abs(ref).max()))
        table[os.path.basename(f)[:-4]] = row
        print(f"{os.path.basename(f)[:-4]:12s} " + " ".join(f"{v:9.2e}" for v in row), flush=True)
    a = np.array(list(table.values()))
    print("max          " + " ".join(f"{v:9.2e}" for v in a.max(0)))
    print("median       " + " ".join(f"{v:9.2e}" for v in np.median(a, 0)))


if __name__ == "__main__":
    main()
