#!/usr/bin/env python
"""Round-3 experiment: two independent half-batch forwards on two streams (two host threads), every persistent GEMM limited
to half the CUs (GAVA_CU_RESERVE=128), against the ordinary 64-clip forward.  Question: do the HBM-bound stretches of one
chain (residual-GEMM epilogues, attention) run under the MFMA-bound stretches of the other when they sit on disjoint CUs?

    python tools/r3_two_chains.py [--chains 2] [--B 64] [--iters 20]
env: GAVA_CU_RESERVE, GAVA_SIDE_STREAM as for any forward."""
import argparse, os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=2)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
cfg = C.VIT_B16_T8
cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt")
torch.manual_seed(0)
models, xs, streams = [], [], []
for c in range(a.chains):
    torch.manual_seed(0)
    m = VitaCLIP(**model_kwargs(cfg, cls_path)).cuda().eval()
    m.text_on_side_stream = a.chains == 1
    models.append(m)
    xs.append(torch.randn(a.B // a.chains, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda"))
    streams.append(torch.cuda.Stream())


def run(c, n):
    with torch.no_grad(), torch.cuda.stream(streams[c]):
        for _ in range(n):
            models[c](xs[c])


def all_chains(n):
    ts = [threading.Thread(target=run, args=(c, n)) for c in range(a.chains)]
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()


all_chains(5)
t0 = time.perf_counter()
all_chains(a.iters)
dt = time.perf_counter() - t0
print(f"chains={a.chains} B={a.B} reserve={os.environ.get('GAVA_CU_RESERVE', '-')}: {1e3 * dt / a.iters:.3f} ms per {a.B} clips, {a.B * a.iters / dt:.1f} clips/s")
