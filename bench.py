#!/usr/bin/env python
"""bench.py — clips/sec of the GaVA-CLIP video-frame forward path on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one VitaCLIP.forward() (vision tower + text tower + RCCL all-gather of the clip
embeddings + similarity head) over one synthetic batch of 64 clips per GPU (BASELINE.json
configs[1] / configs[3]; weak scaling).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (contract in the task statement), extended with
  roofline      — the kernel INSTANTIATION with the largest share of the forward's GPU time (out_proj + fc2 of the vision blocks:
                  gemm256_kernel<PrecF16, EPI_F32, RES, ...>) timed stand-alone with HIP events, its members priced on their own roofs
  cpu_baseline  — the oracle (CPU port of the reference) timed on the host cores, N=1 only
  kernels       — stand-alone HIP-event timings of every hot kernel shape of one layer
  parity_modes  — clips/s of the weight-lo modes (operand_dtype fp16+wlo8 / fp16+wlo), whose accuracy block sits in `accuracy`
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch  # noqa: E402

T_START = time.perf_counter()


def log(msg):
    """progress to stderr (the JSON line on stdout stays alone)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def kernel_source_hash():
    """sha256 over the kernel sources the per-layer kernels are built from (recorded by tools/make_traffic.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "common.h", "attention.hip"):
        h.update(open(os.path.join(REPO, "gava_clip_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("GAVA_CPU_THREADS", "16"))))   # a 1-GPU box's CPU share is 16


PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

CONFIGS = {
    # name: (VitaConfig name, clips per GPU, class file, description)
    "c2": ("VIT_B16_T8", 64, "updrs_3cls_classes.txt", "ViT-B/16, 8 frames, 224^2, batch 64/GPU, 3 classes"),
    "c3": ("VIT_B16_T16", 32, "k400_classes.txt", "ViT-B/16, 16 frames, 224^2, batch 32/GPU, 400 classes"),
    "c5": ("VIT_L14_T32", 32, "updrs_3cls_classes.txt", "ViT-L/14, 32 frames, 224^2, batch 32/GPU, 3 classes"),
    "c1": ("VIT_B16_T8", 2, "updrs_3cls_classes.txt", "ViT-B/16, 8 frames, 224^2, batch 2, 3 classes"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)     # SURVEY 8d protocol: >= 10 warm-up + >= 50 timed forwards
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--prec", default=os.environ.get("GAVA_PREC", "fp16"), choices=["fp16", "bf16", "fp16+wlo", "fp16+wlo8"],
                    help="MFMA operand mode of the timed forward; the +wlo modes are the parity modes (weight-lo pass, DESIGN 'Numerics')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the stand-alone kernel timings")
    ap.add_argument("--no-alt", action="store_true", help="skip the short run in the other operand dtype")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the logits-vs-reference-golden check")
    ap.add_argument("--accuracy-c5", action="store_true", help="include the full c3 batch and the ViT-L/14 fixtures (c5_clip0, c5_full) in the accuracy leg")
    return ap.parse_args()


def timed_steps(fn, steps, warmup, dist=None):
    """W untimed + exactly K timed calls, bracketed by barrier + synchronize; seconds (max over ranks)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def event_time_ms(fn, iters=20, warmup=10):
    """Average duration of fn's launches, HIP events on the stream the kernels run on (torch's current).  Ten warm-up
    launches: the first few after an idle gap run at a boosted clock and would flatter a 0.4 ms kernel by 10 %."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def train_roofline(cfg, B):
    """SURVEY 8f row 1 evidence: the dominant kernel of the BACKWARD (by GPU time, profiles/r04_train_step_c2_kernel_stats.csv)
    is the dgrad GEMM with the fp32 gradient-accumulator epilogue - gemm256_kernel<PrecBF16, EPI_F32, no residual>: dX = dH . W_fc1
    (M = rows, N = D, K = F) and dX = dQKV . W_qkv (N = D, K = 3D), one launch each per block.  Timed stand-alone with HIP events
    in the form training.py launches them (bf16 gradient operands, transposed weight copies), priced on the MFMA roof with
    the algorithmic FLOPs 2 M N K."""
    from gava_clip_amd import hip
    d = torch.device("cuda")
    D, F = cfg.feature_dim, cfg.mlp_dim
    R = B * cfg.num_frames * cfg.tokens_main
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s, scale=1.0: (torch.randn(*s, device=d, generator=g) * scale).to(torch.bfloat16)
    dhid, dqkv = rn(R, F, scale=1e-3), rn(R, 3 * D, scale=1e-3)
    w1t, wqt = rn(D, F, scale=F ** -0.5), rn(D, 3 * D, scale=(3 * D) ** -0.5)
    dx = torch.empty(R, D, dtype=torch.float32, device=d)
    rows = []
    for name, A, W, K in (("dgrad fc1^T  [R,F]x[D,F]^T", dhid, w1t, F), ("dgrad qkv^T  [R,3D]x[D,3D]^T", dqkv, wqt, 3 * D)):
        fn = lambda A=A, W=W: hip.gemm(A, W, None, dx, epilogue=hip.EPI_F32, prec=hip.PREC_BF16)
        ms = sorted(event_time_ms(fn) for _ in range(3))[1]
        fl = 2.0 * R * D * K
        rows.append({"kernel": name, "ms_per_launch": round(ms, 4), "flops_per_launch": fl, "achieved": round(fl / ms / 1e9, 1),
                     "frac": round(fl / ms / 1e9 / PEAK_MFMA_TFLOPS, 4), "launches_per_step": cfg.num_layers - 1})
    n = sum(r["launches_per_step"] for r in rows)
    avg_ms = sum(r["ms_per_launch"] * r["launches_per_step"] for r in rows) / n
    avg_fl = sum(r["flops_per_launch"] * r["launches_per_step"] for r in rows) / n
    walk = "true" if hip.load().gava_gemm_aligned_walk(R, D, 0) else "false"
    return {"bound": "mfma", "kernel": f"gemm256_kernel<PrecBF16, 2, false, false, false, {walk}, 0, true, false> = the dgrad GEMMs with the fp32 gradient-accumulator epilogue",
            "achieved": round(avg_fl / avg_ms / 1e9, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(avg_fl / avg_ms / 1e9 / PEAK_MFMA_TFLOPS, 4),
            "ms_per_launch": round(avg_ms, 4), "launches_per_step": n, "members": rows,
            "timed": "HIP events around 20 back-to-back launches after 10 warm-up launches, median of 3",
            "first_thing_to_fix": "see DESIGN.md section 7 (round 4): the step's largest non-GEMM shares"}


def kernel_table(cfg, B, prec, fold=False, pair=False):
    """Stand-alone timings of the per-layer kernels at this config's shapes (through the C ABI)."""
    from gava_clip_amd import hip
    dt = hip.h16_dtype(prec)
    d = torch.device("cuda")
    D, F, H = cfg.feature_dim, cfg.mlp_dim, cfg.num_heads
    T, G, n1 = cfg.num_frames, cfg.num_global_prompts, cfg.tokens_main
    BT = B * T
    R = BT * n1
    g = torch.Generator(device="cuda").manual_seed(1)

    def rn(*s, scale=1.0, dtype=dt):
        return (torch.randn(*s, device=d, generator=g) * scale).to(dtype)

    Xn, MIX, HID = rn(R, D), rn(R, D), rn(R, F)
    X = rn(R, D, dtype=torch.float32)
    QKV = rn(R, 3 * D)
    Wqkv, Wo, W1, W2 = rn(3 * D, D, scale=D ** -0.5), rn(D, D, scale=D ** -0.5), rn(F, D, scale=D ** -0.5), rn(D, F, scale=F ** -0.5)
    bq, bo, b1, b2 = (rn(n, dtype=torch.float32) for n in (3 * D, D, F, D))
    gam, bet = rn(D, dtype=torch.float32), rn(D, dtype=torch.float32)
    side = rn(G + 2 * BT, 2 * D)
    rows = []

    pending = []

    def add(name, fn, flops, bytes_, inst=None, calls=0, key=None):
        """inst: the kernel instantiation as rocprofv3 names it (rows with the same inst are one line of a kernel-stats
        CSV); calls: launches of this shape per forward; key: the entry of profiles/traffic.json"""
        pending.append((name, fn, flops, bytes_, inst, calls, key))

    def measure():
        """Interleaved rounds in one process (every kernel sees the same thermal / clock history; a kernel timed alone
        right after an idle gap or first in a list reads up to 10 % off): 10 warm-up launches each, then 3 rounds of 20
        launches per kernel, median of the rounds."""
        for _, fn, *_ in pending:
            for _ in range(10):
                fn()
        samples = [[] for _ in pending]
        for _ in range(3):
            for i, (_, fn, *_) in enumerate(pending):
                samples[i].append(event_time_ms(fn, iters=20, warmup=2))
        for (name, fn, flops, bytes_, inst, calls, key), sm in zip(pending, samples):
            ms = sorted(sm)[1]
            rows.append(dict(kernel=name, ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1) if flops else None,
                             gbps=round(bytes_ / ms / 1e6, 1), flops=flops, bytes=bytes_, inst=inst, calls=calls, key=key))

    Rp = (R + 255) // 256 * 256
    rsum = torch.zeros(Rp, D // 64, 2, dtype=torch.float32, device=d)
    part = torch.rand(Rp + 32, 4, 2, dtype=torch.float32, device=d) * 40 + 200     # [rows][4] (sum, sum^2) per 256-column tile
    stats = torch.cat([rn(Rp, 1, scale=0.1, dtype=torch.float32), 1 + rn(Rp, 1, scale=0.1, dtype=torch.float32).abs()], 1).contiguous()
    s1, t1 = W1.float().sum(1).contiguous(), rn(F, dtype=torch.float32)
    sq, tq = Wqkv.float().sum(1).contiguous(), rn(3 * D, dtype=torch.float32)
    plain = {
        "qkv": ("gemm qkv  [R,D]x[3D,D] h16", lambda: hip.gemm(Xn, Wqkv, bq, QKV, epilogue=hip.EPI_H16, prec=prec, scale_cols=D, scale=0.125),
                2.0 * R * 3 * D * D, R * D * 2 + R * 3 * D * 2 + 3 * D * D * 2),
        "out": ("gemm out  [R,D]x[D,D] +res f32", lambda: hip.gemm(MIX, Wo, bo, X, epilogue=hip.EPI_F32, prec=prec, resid=X),
                2.0 * R * D * D, R * D * 2 + R * D * 8 + D * D * 2),
        "fc1": ("gemm fc1  [R,D]x[F,D] qgelu h16", lambda: hip.gemm(Xn, W1, b1, HID, epilogue=hip.EPI_H16_QGELU, prec=prec),
                2.0 * R * F * D, R * D * 2 + R * F * 2 + F * D * 2),
        "fc2": ("gemm fc2  [R,F]x[D,F] +res f32", lambda: hip.gemm(HID, W2, b2, X, epilogue=hip.EPI_F32, prec=prec, resid=X),
                2.0 * R * D * F, R * F * 2 + R * D * 8 + D * F * 2)}
    # the forms the inference forward launches when LayerNorm is folded into the consumer GEMMs (model.fold_layernorm)
    folded = {
        "qkv": ("gemm qkv  folded-LN consumer, h16", lambda: hip.gemm(Xn, Wqkv, None, QKV, epilogue=hip.EPI_H16, prec=prec, scale_cols=D,
                                                                      scale=0.125, fold_partials=part, fold_s=sq, fold_t=tq),
                2.0 * R * 3 * D * D, R * D * 2 + R * 3 * D * 2 + 3 * D * D * 2 + R * 8),
        "out": ("gemm out  +res f32 +x16 +row sums", lambda: hip.gemm(MIX, Wo, bo, X, epilogue=hip.EPI_F32, prec=prec, resid=X, x16_out=X16, rowsum_out=part,
                                                                      rowsum_reduced=True),
                2.0 * R * D * D, R * D * 2 + R * D * 10 + D * D * 2 + R * (D // 64) * 8),
        "fc1": ("gemm fc1  folded-LN consumer, qgelu h16", lambda: hip.gemm(Xn, W1, None, HID, epilogue=hip.EPI_H16_QGELU, prec=prec,
                                                                            fold_partials=part, fold_s=s1, fold_t=t1),
                2.0 * R * F * D, R * D * 2 + R * F * 2 + F * D * 2 + R * 8),
        "fc2": ("gemm fc2  +res f32 +x16 +row sums", lambda: hip.gemm(HID, W2, b2, X, epilogue=hip.EPI_F32, prec=prec, resid=X, x16_out=X16, rowsum_out=part,
                                                                      rowsum_reduced=True),
                2.0 * R * D * F, R * F * 2 + R * D * 10 + D * F * 2 + R * (D // 64) * 8)}
    X16 = torch.empty(R, D, dtype=dt, device=d)
    if fold and pair:
        # ... and when the driver keeps the residual stream as a 16-bit pair (gava_vision_pair_stream: big batches): the producers read
        # and write hi / lo (8 bytes per element through the epilogue: pair in, pair out - the hi half IS the 16-bit copy)
        X16 = X.to(dt)
        XLO = (X - X16.float()).half()
        folded["out"] = ("gemm out  +res pair -> pair +row sums", lambda: hip.gemm(MIX, Wo, bo, None, epilogue=hip.EPI_F32, prec=prec, resid16=X16, resid_lo=XLO,
                                                                                 x16_out=X16, xlo_out=XLO, rowsum_out=part, rowsum_reduced=True),
                         2.0 * R * D * D, R * D * 2 + R * D * 8 + D * D * 2 + R * (D // 64) * 8)
        folded["fc2"] = ("gemm fc2  +res pair -> pair +row sums", lambda: hip.gemm(HID, W2, b2, None, epilogue=hip.EPI_F32, prec=prec, resid16=X16, resid_lo=XLO,
                                                                                 x16_out=X16, xlo_out=XLO, rowsum_out=part, rowsum_reduced=True),
                         2.0 * R * D * F, R * F * 2 + R * D * 8 + D * F * 2 + R * (D // 64) * 8)
    first, second = (folded, plain) if fold else (plain, None)
    # which instantiation each shape runs as (the name rocprofv3 prints; out_proj and fc2 share one), and how often the
    # inference forward launches it: with the fold, block 0's qkv and the last block run other kernels (forward.hip)
    Lyr = cfg.num_layers
    PN = "PrecF16" if prec == hip.PREC_F16 else "PrecBF16"
    # template arguments: precision, epilogue, residual, split output, LayerNorm fold, aligned tile walk, 8-bit lo mode, ping-pong
    # k-loop, residual stream as a 16-bit pair;
    # the walk is the launcher's own decision for this shape on this device (gava_gemm_aligned_walk), not assumed
    walk = "true" if hip.load().gava_gemm_aligned_walk(R, D, 0) else "false"
    pp_env = os.environ.get("GAVA_PP", "1")      # the launcher's switch: 1 (default) = ping-pong k-loop wherever an instantiation exists, 2 = fp32-output GEMMs only
    pp, ppc = ("true" if pp_env in ("1", "2") else "false"), ("true" if pp_env == "1" else "false")
    hl = "true" if (fold and pair) else "false"
    if fold and pair:
        pp = "true"          # the pair instantiations exist on the ping-pong loop only
    inst = {"qkv": f"gemm256_kernel<{PN}, 0, false, false, {'true' if fold else 'false'}, false, 0, {ppc}, false>",
            "fc1": f"gemm256_kernel<{PN}, 1, false, false, {'true' if fold else 'false'}, false, 0, {ppc}, false>",
            "out": f"gemm256_kernel<{PN}, 2, true, false, false, {walk}, 0, {pp}, {hl}>", "fc2": f"gemm256_kernel<{PN}, 2, true, false, false, {walk}, 0, {pp}, {hl}>"}
    calls = {"qkv": Lyr - 2 if fold else Lyr - 1, "fc1": Lyr - 1, "out": Lyr - 1, "fc2": Lyr - 1}
    # per-layer order of the forward; the kernels the forward launches come first, at the positions they always had
    add("layernorm f32->h16 [R,D]", lambda: hip.layernorm(X, gam, bet, out16=Xn, prec=prec), 0, R * D * 6, "layernorm_kernel", 0, None)
    add(*first["qkv"], inst["qkv"], calls["qkv"], "qkv")
    add("attention (frame,head) %dq x %dk" % (n1, cfg.attn_keys()),
        lambda: hip.attention(QKV[:, :D], QKV[:, D:2 * D], QKV[:, 2 * D:], MIX, batch=BT, heads=H, n_q=n1, n_kmain=n1, prec=prec,
                              side_k=side[:, :D], side_v=side[:, D:], n_g=G, T=T, has_summary=True),
        4.0 * BT * H * n1 * cfg.attn_keys() * 64, R * 3 * D * 2 + R * D * 2,
        f"attention_persist_kernel<{PN}, 14, 13>" if cfg.attn_keys() <= 224 else f"attention_kernel<{PN}, 20, false, true, 14>", Lyr - 1, "attn")
    add(*first["out"], inst["out"], calls["out"], "out")
    add(*first["fc1"], inst["fc1"], calls["fc1"], "fc1")
    add(*first["fc2"], inst["fc2"], calls["fc2"], "fc2")
    if fold:
        for k in ("qkv", "out", "fc1", "fc2"):   # the unfolded forms (training, GAVA_LN_FOLD=0), for comparison
            add(*second[k])
    # yardstick, not a target: the vendor library (torch.matmul -> hipBLASLt) on the fc1 and fc2 shapes, plain GEMM with 16-bit
    # output and no epilogue - inside the SAME interleaved rounds as every kernel above (timed on its own after
    # the table it read 0.505 ms, in an interleaved calibration 0.429-0.437)
    HIDv, Xv = torch.empty(R, F, dtype=dt, device=d), torch.empty(R, D, dtype=dt, device=d)
    add("vendor hipBLASLt, fc1 shape, no epilogue", lambda: torch.matmul(Xn, W1.t(), out=HIDv), 2.0 * R * F * D, R * D * 2 + R * F * 2 + F * D * 2, None, 0, "vendor_fc1")
    add("vendor hipBLASLt, fc2 shape, no epilogue", lambda: torch.matmul(HID, W2.t(), out=Xv), 2.0 * R * D * F, R * F * 2 + R * D * 2 + D * F * 2, None, 0, "vendor_fc2")
    measure()
    return rows


def accuracy_vs_golden(prec, include_c5=False):
    """BASELINE.json's third metric, "logits max-abs-err vs ref", over EVERY reference-run fixture with logits
    (tests/helpers.py GOLDEN_LOGIT_CASES: four weight + input seeds at c1, clip 0 of c3 with 400 classes, optionally clip 0 of
    c5): a model of the fixture's shape is loaded with the synthetic weights the REFERENCE produced the fixture from, and
    its logits are compared with the golden logits in the timed operand dtype and in the other one.  Reported per fixture
    and as a distribution (max / median over fixtures of the norm-wise error, max / median / 95th percentile over all
    logits of the element-wise error), next to the frozen written criteria of tests/helpers.py."""
    import numpy as np
    from gava_clip_amd import VitaCLIP, synth
    from helpers import (GOLDEN_LOGIT_CASES, golden_case, model_kwargs, synth_torch_state, mixed_violation,
                         LOGITS_RTOL, LOGITS_ATOL)
    # default: the twenty-four c1 seeds, the full c2 batch, clip 0 of c3; --accuracy-c5 adds the full c3 batch and the ViT-L/14 fixtures
    names = [n for n in GOLDEN_LOGIT_CASES if include_c5 or not (n.startswith("c5") or n == "c3_full")]
    # ... and the two weight-lo parity modes of round 4 (DESIGN "Numerics": the configurations that meet 1e-3 on every fixture)
    order = tuple(dict.fromkeys((prec, "bf16" if prec == "fp16" else "fp16", "fp16+wlo8", "fp16+wlo")))
    res = {"reference": "tests/golden/{%s}.npz (reference fp32 CPU forwards, tools/gen_golden.py)" % ",".join(names),
           "criteria": f"norm-wise max|d| <= 1e-3 max|ref|; element-wise |d| <= {LOGITS_RTOL} |ref| + {LOGITS_ATOL} (frozen, tests/helpers.py)"}
    per = {p_: {} for p_ in order}
    elem = {p_: [] for p_ in order}
    models = {}
    for name in names:
        cfg, class_file, n_cls, B, wseed, xseed = golden_case(name)
        g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
        log(f"accuracy: {name}")
        key = (cfg, class_file)
        if key not in models:
            models.clear()                       # one fixture shape resident at a time
            models[key] = VitaCLIP(**model_kwargs(cfg, class_file), operand_dtype=prec).cuda().eval()
        model = models[key]
        model.load_state_dict(synth_torch_state(cfg, n_cls, wseed), strict=True)
        x = torch.from_numpy(synth.synth_clip(B, cfg.num_frames, cfg.input_size, seed=xseed)).cuda()
        for p_ in order:
            model.set_operand_dtype(p_)
            with torch.no_grad():
                lg = model(x)[0].float().cpu().numpy()
            d = np.abs(lg - g["logits"])
            per[p_][name] = {"max_abs_err": float(d.max()), "rel_to_max_logit": float(d.max() / np.abs(g["logits"]).max()),
                             "max_elementwise_rel": float((d / np.abs(g["logits"])).max()),
                             "mixed_violation": round(mixed_violation(lg, g["logits"]), 3), "n_logits": int(lg.size),
                             "argmax_equal": bool((lg.argmax(-1) == g["logits"].argmax(-1)).all())}
            elem[p_].append((d / np.abs(g["logits"])).reshape(-1))
        model.set_operand_dtype(prec)
    models.clear()
    for p_ in order:
        nw = np.array([v["rel_to_max_logit"] for v in per[p_].values()])
        el = np.concatenate(elem[p_])
        res[p_] = {"fixtures": len(nw), "logits": int(el.size),
                   "normwise": {"max": float(nw.max()), "median": float(np.median(nw))},
                   "elementwise": {"max": float(el.max()), "p95": float(np.percentile(el, 95)), "median": float(np.median(el))},
                   "max_abs_err": float(max(v["max_abs_err"] for v in per[p_].values())),
                   "max_mixed_violation": float(max(v["mixed_violation"] for v in per[p_].values())),
                   "argmax_equal": bool(all(v["argmax_equal"] for v in per[p_].values())),
                   "per_fixture": per[p_]}
    return res


def cpu_baseline():
    """The oracle (CPU port of the reference forward) at BASELINE config c1: B=2, T=8, 3 classes, fp32."""
    import numpy as np
    from gava_clip_amd import synth
    from gava_clip_amd.config import VIT_B16_T8
    from gava_clip_amd.tokenizer import tokenize, read_class_names, prompt_texts
    from oracle.vita_oracle import Oracle   # timed as the CPU baseline, never on the product path
    from helpers import CLASSES_3, synth_torch_state
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle at c1 on {cores} threads")
    cfg = VIT_B16_T8
    sd = synth_torch_state(cfg, 3)
    tok = tokenize(prompt_texts(read_class_names(CLASSES_3), cfg.text_num_prompts))
    o = Oracle(cfg, sd, tok)
    x = torch.from_numpy(synth.synth_clip(2, 8, 224))
    t0 = time.perf_counter()
    o.forward(x)
    log(f"cpu_baseline: warm-up forward {time.perf_counter() - t0:.2f}s")
    ts = []
    for _ in range(8):                        # ~10 s of CPU work on the box's 16-core share
        t0 = time.perf_counter()
        o.forward(x)
        ts.append(time.perf_counter() - t0)
        log(f"cpu_baseline: forward {ts[-1]:.2f}s")
    best = min(ts)
    return dict(value=round(2 / best, 3), unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample="oracle/vita_oracle.py fp32, config c1 (B=2,T=8,224^2,3 classes), best of 8 forwards, "
                       f"{best:.3f} s/forward, torch {torch.__version__} CPU")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1"
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    dist = None
    backend_name = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("GAVA_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" only to rehearse on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        backend_name = dist.get_backend()

    import gava_clip_amd.config as C
    from gava_clip_amd import VitaCLIP, flops as fl
    from helpers import model_kwargs
    cname, B, cls_file, desc = CONFIGS[a.config]
    cfg = getattr(C, cname)
    cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", cls_file)
    torch.set_num_threads(host_cores())
    log(f"building {cname} model, {B} clips/GPU, {world} GPU(s), operands {a.prec}")
    torch.manual_seed(0)                      # identical random-init weights on every rank
    model = VitaCLIP(**model_kwargs(cfg, cls_path), operand_dtype=a.prec).cuda().eval()
    n_cls = len(model.tokenized_prompts)
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    x = torch.randn(B, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda", generator=gen)

    def step():
        with torch.no_grad():
            return model(x)

    log("model on device; first forward (packs weights)")
    step(); torch.cuda.synchronize()
    log("timed region")
    secs = timed_steps(step, a.steps, a.warmup, dist)
    log(f"{a.steps} steps in {secs:.3f}s")
    logits = step()[0]
    assert tuple(logits.shape) == (B * world, n_cls) and bool(torch.isfinite(logits).all())
    clips = world * B * a.steps
    value = clips / secs
    fwd_flops = fl.forward_flops(cfg, B, n_cls)
    exe_flops = fl.executed_flops(cfg, B, n_cls, text_rows=model.text_rows_per_prompt)
    gather = None
    if dist is not None:
        # the path's only exchange on its own: event pair (time.perf_counter for a CPU rendezvous) around 20 all-gathers of
        # the (B, E) clip embeddings, every rank taking part; max over ranks
        feats = torch.randn(B, cfg.embed_dim, device="cuda")
        for _ in range(5):
            model._gather(feats)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            model._gather(feats)
        torch.cuda.synchronize()
        gms = 1e3 * (time.perf_counter() - t0) / 20
        t = torch.tensor([gms], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather = {"ms": round(float(t.item()), 4), "bytes_per_rank": B * cfg.embed_dim * 4, "backend": dist.get_backend(),
                  "what": "all_gather_into_tensor of the (B, E) fp32 clip embeddings, 20 back-to-back, host clock incl. launch, max over ranks"}
        # No rank may sit in a collective while rank 0 measures alone: the group is torn down HERE, before any rank-0-only
        # leg, and every other rank leaves.  (The rank-0-only legs - kernel table, roofline, other dtype, training step,
        # accuracy, CPU baseline - belong to the N = 1 line and are skipped for N > 1.)
        dist.barrier()
        dist.destroy_process_group()
        dist = None
        model.gather_across_ranks = False
        if rank != 0:
            return

    out = {
        "metric": "clips/sec (%d-frame 224^2 %s VitaCLIP.forward)" % (cfg.num_frames, "ViT-L/14" if cfg.patch_size == 14 else "ViT-B/16"),
        "value": round(value, 2), "unit": "clips/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * secs / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.prec, "data": "synthetic",
        "config": {"workload": f"{a.config}: {desc}", "clips_per_gpu": B, "global_batch": B * world,
                   "frames": cfg.num_frames, "classes": n_cls, "weights": "random-init",
                   "parallelism": (f"clips sharded over {world} GPU(s); all-gather of (B,E) embeddings over "
                                   + {"nccl": "RCCL (torch.distributed backend nccl)", "gloo": "gloo (CPU rendezvous, staged through the host: rehearsal only)"}.get(backend_name, str(backend_name))) if world > 1 else "single GPU",
                   "text_tower": ("split-precision GEMMs (3 MFMA passes)" + (", fp32 softmax core" if getattr(model, "text_attention_fp32", False) else ""))
                                 if model.text_split_precision else a.prec},
        # two FLOP figures: the REFERENCE's dense work for this batch (what a drop-in replaces; includes rows
        # whose results the reference discards) and the work this path EXECUTES (flops.executed_flops: no prompt-row
        # q/out/MLP, CLS-only last block, text rows up to the last EOT).  Utilisation is quoted on the executed figure.
        "algorithmic_tflops": round(fwd_flops * world * a.steps / secs / 1e12, 1),
        "executed_tflops": round(exe_flops * world * a.steps / secs / 1e12, 1),
        "mfma_frac_executed": round(exe_flops * a.steps / secs / 1e12 / PEAK_MFMA_TFLOPS, 4),
        "mfma_frac_reference_flops": round(fwd_flops * a.steps / secs / 1e12 / PEAK_MFMA_TFLOPS, 4),
        "target": {"what": "north_star: >= 40 % MFMA utilisation at c2 = 3370 clips/s at the 2.5 PFLOP/s dense peak",
                   "met": bool(a.config == "c2" and value / world >= 3370.0)} if a.config == "c2" else None,
    }
    if gather is not None:
        out["gather"] = gather
    solo = world == 1      # the legs below run on one rank alone: N = 1 only
    if solo and not a.no_kernels:
        log("stand-alone kernel timings")
        fold = bool(getattr(model, "fold_layernorm", False))
        rows = kernel_table(cfg, B, model.prec, fold=fold, pair=bool(model.last.get("pair_stream")))
        out["kernels"] = [{k: r[k] for k in ("kernel", "ms", "tflops", "gbps")} for r in rows]
        # ---- roofline: the kernel INSTANTIATION with the largest share of the forward's GPU time (what the first row of a
        # rocprofv3 --kernel-trace --stats summary of this command shows; out_proj and fc2 of the vision blocks are one
        # instantiation).  Its members are priced separately: out_proj is HBM-bound, the others MFMA-bound.
        groups = {}
        for r in rows:
            if r["inst"] and r["calls"]:
                groups.setdefault(r["inst"], []).append(r)
        share = {k: sum(r["ms"] * r["calls"] for r in v) for k, v in groups.items()}
        dom = max(share, key=share.get)
        members = groups[dom]
        # HBM bytes / launch from the committed PMC passes (tools/make_traffic.py writes them together with a hash of the
        # kernel sources they were measured on): a file measured on other sources is stale and is NOT reported
        tj, traffic_src = {}, "profiles/traffic.json missing"
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel_source_sha256") == kernel_source_hash() and a.config == tj.get("config", "c2"):
                traffic_src = tj.get("source")
            else:
                tj, traffic_src = {}, "profiles/traffic.json is stale (kernel sources changed since its PMC passes, or another config): not reported"

        def traffic_of(key):
            e = tj.get("kernels", {}).get(key)
            return e.get("bytes_per_launch") if e else None

        # cross-check where they run: HIP-event pairs around every launch of a member inside a few more forwards of the
        # timed workload (gava_probe_fc1_*, on the stream the driver launches on)
        def in_forward_ms(which):
            try:
                import ctypes as C_
                from gava_clip_amd import hip as hip_
                lib_ = hip_.load()
                lib_.gava_probe_fc1_enable(which)
                samples = []
                try:
                    for _ in range(5):
                        step()
                        buf = (C_.c_float * 64)()
                        n_ = lib_.gava_probe_fc1_read(buf, 64)
                        samples += [buf[i] for i in range(n_) if buf[i] >= 0]
                finally:
                    lib_.gava_probe_fc1_enable(0)
                return sum(samples) / len(samples) if samples else None
            except Exception as e:   # the probe must never break the bench line
                log(f"in-forward probe skipped: {e}")
                return None
        PROBE = {"fc1": 1, "out": 2, "fc2": 3, "qkv": 4, "attn": 5}
        PEAK_HBM_GBPS, ACHIEVABLE_HBM_GBPS = 8000.0, 6300.0      # MI355X_MICROARCH.md "HBM"
        mem = []
        for r in members:
            infwd = in_forward_ms(PROBE[r["key"]])
            e = {"kernel": r["kernel"], "launches_per_forward": r["calls"], "ms_per_launch": r["ms"],
                 "in_forward_ms_per_launch": round(infwd, 4) if infwd else None,
                 "flops_per_launch": r["flops"], "algorithmic_bytes_per_launch": r["bytes"], "traffic": traffic_of(r["key"])}
            t_mfma, t_hbm = r["flops"] / (PEAK_MFMA_TFLOPS * 1e9), r["bytes"] / (PEAK_HBM_GBPS * 1e6)     # ms at the two roofs
            if t_hbm > t_mfma:
                e.update(bound="hbm", achieved=round(r["bytes"] / r["ms"] / 1e6, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                         frac=round(r["bytes"] / r["ms"] / 1e6 / PEAK_HBM_GBPS, 4),
                         frac_of_achievable=round(r["bytes"] / r["ms"] / 1e6 / ACHIEVABLE_HBM_GBPS, 4))
            else:
                e.update(bound="mfma", achieved=round(r["flops"] / r["ms"] / 1e9, 1), peak=PEAK_MFMA_TFLOPS, unit="TFLOP/s",
                         frac=round(r["flops"] / r["ms"] / 1e9 / PEAK_MFMA_TFLOPS, 4))
            mem.append(e)
        n_calls = sum(r["calls"] for r in members)
        avg_ms = sum(r["ms"] * r["calls"] for r in members) / n_calls                 # what a kernel-stats row averages
        avg_flops = sum(r["flops"] * r["calls"] for r in members) / n_calls
        infw = [e["in_forward_ms_per_launch"] for e in mem]
        tr = [e["traffic"] for e in mem]
        fwd_ms = 1e3 * secs / a.steps
        # `achieved` / `frac` are priced on the launches INSIDE the forward (the timed workload; what rocprofv3's kernel-stats row
        # averages), the stand-alone figure stays beside them; the group's `bound` is that of the member holding most of its time
        infw_avg = sum(x * r["calls"] for x, r in zip(infw, members)) / n_calls if all(infw) else None
        price_ms = infw_avg if infw_avg else avg_ms
        t_by_bound = {}
        for e, r in zip(mem, members):
            t_by_bound[e["bound"]] = t_by_bound.get(e["bound"], 0.0) + r["ms"] * r["calls"]
        out["roofline"] = {
            "bound": max(t_by_bound, key=t_by_bound.get), "bound_members": {r["key"]: e["bound"] for e, r in zip(mem, members)},
            "kernel": dom + " = " + " + ".join(r["kernel"].split("  ")[0].replace("gemm ", "") for r in members) + " of the vision blocks",
            "why": "largest share of the forward's GPU time among the kernel instantiations (launches x stand-alone ms): "
                   + ", ".join(f"{k.split('<')[0]}<{k.split(', ', 1)[1]} {100 * v / fwd_ms:.1f}%" for k, v in sorted(share.items(), key=lambda kv: -kv[1])),
            "achieved": round(avg_flops / price_ms / 1e9, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(avg_flops / price_ms / 1e9 / PEAK_MFMA_TFLOPS, 4),
            "priced_on": "in_forward_ms_per_launch" if infw_avg else "ms_per_launch",
            "standalone_frac": round(avg_flops / avg_ms / 1e9 / PEAK_MFMA_TFLOPS, 4),
            "flops_per_launch": avg_flops, "ms_per_launch": round(avg_ms, 4), "launches_per_forward": n_calls,
            "in_forward_ms_per_launch": round(infw_avg, 4) if infw_avg else None,
            "traffic": round(sum(x * r["calls"] for x, r in zip(tr, members)) / n_calls) if all(tr) else None,
            "algorithmic_bytes_per_launch": round(sum(r["bytes"] * r["calls"] for r in members) / n_calls),
            "traffic_source": traffic_src,
            "timed": "per-launch averages over the instantiation's launches in one forward; each member: HIP events around 20 back-to-back "
                     "launches, median of 3 rounds interleaved with the other per-layer kernels, after 10 warm-up launches; in_forward: an "
                     "event pair per launch inside 5 forwards of the timed workload (adds ~10-20 us of event latency per pair)",
            "members": mem}
        # the other per-layer kernels, priced the same way (the MFMA-bound consumers, the HBM-bound attention)
        others = []
        for r in rows:
            if r["inst"] and r["calls"] and r["inst"] != dom:
                t_mfma, t_hbm = r["flops"] / (PEAK_MFMA_TFLOPS * 1e9), r["bytes"] / (PEAK_HBM_GBPS * 1e6)
                hb = t_hbm > t_mfma
                others.append({"kernel": r["kernel"], "inst": r["inst"], "bound": "hbm" if hb else "mfma", "ms_per_launch": r["ms"],
                               "frac": round((r["bytes"] / r["ms"] / 1e6 / PEAK_HBM_GBPS) if hb else (r["flops"] / r["ms"] / 1e9 / PEAK_MFMA_TFLOPS), 4),
                               "share_of_forward": round(r["ms"] * r["calls"] / fwd_ms, 4), "traffic": traffic_of(r["key"]),
                               "algorithmic_bytes_per_launch": r["bytes"]})
        out["roofline_other_kernels"] = others
    if solo and not a.no_kernels:
        vy = {r["key"]: r for r in rows if r.get("key") in ("vendor_fc1", "vendor_fc2")}
        if vy:
            out["vendor_yardstick"] = {"what": "torch.matmul (hipBLASLt), plain 16-bit-output GEMM without epilogue, timed inside the kernel table's "
                                               "interleaved rounds (same warm-up, same process state as every kernel of the table)",
                                       **{k: {"ms": r["ms"], "tflops": r["tflops"]} for k, r in vy.items()}}
    if solo and not a.no_alt:
        log("alt operand dtype run")
        other = "bf16" if a.prec == "fp16" else "fp16"
        model.set_operand_dtype(other)
        s2 = timed_steps(step, max(3, a.steps // 4), 2, None) if world == 1 else None
        model.set_operand_dtype(a.prec)
        if s2 is not None:
            out["alt"] = {"dtype": other, "value": round(B * max(3, a.steps // 4) / s2, 2), "unit": "clips/s"}
        # the parity modes (weight-lo pass: every reference fixture within 1e-3 norm-wise, see `accuracy`): what they cost
        if world == 1:
            out["parity_modes"] = {}
            for mode in ("fp16+wlo8", "fp16+wlo"):
                model.set_operand_dtype(mode)
                n_ = max(3, a.steps // 4)
                s3 = timed_steps(step, n_, 2, None)
                out["parity_modes"][mode] = {"value": round(B * n_ / s3, 2), "unit": "clips/s", "ms_per_step": round(1e3 * s3 / n_, 3)}
            model.set_operand_dtype(a.prec)
    if rank == 0 and world == 1 and not a.no_train and a.config in ("c2", "c3"):
        # one training step (forward with saved block inputs, loss.backward() through the HIP
        # backward kernels, AdamW on the trainable subset), same batch.  Reported beside the headline, not part of it.
        log("training step")
        model.train()
        y = torch.randint(0, n_cls, (B,), device="cuda", generator=gen)
        opt = torch.optim.AdamW([q for q in model.parameters() if q.requires_grad], lr=1e-5)

        def train_step():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(model(x)[0], y).backward()
            opt.step()
        ts = timed_steps(train_step, 3, 2, None)
        model.eval()
        out["train_step"] = {"ms_per_step": round(1e3 * ts / 3, 1), "value": round(3 * B / ts, 1), "unit": "clips/s",
                             "what": "forward + backward (bf16 gradient operands) + AdamW, %d trainable parameters"
                                     % sum(q.numel() for q in model.parameters() if q.requires_grad)}
        try:
            del opt
            torch.cuda.empty_cache()
            out["train_roofline"] = train_roofline(cfg, B)
        except Exception as e:   # evidence leg: must never break the bench line
            log(f"train_roofline skipped: {e}")
    if rank == 0 and world == 1 and cname == "VIT_B16_T8" and not a.no_accuracy:
        out["accuracy"] = accuracy_vs_golden(a.prec, include_c5=a.accuracy_c5)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
