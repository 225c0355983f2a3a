#!/usr/bin/env python
"""Same-box A/B of whole-forward settings that are chosen by environment variables (one fresh process per setting and
round, alternating, so that box-to-box and thermal drift cancel):

    python tools/ab_env.py "base:" "noprefuse:GAVA_NO_PREFUSE=1" "lib2:GAVA_HIP_LIB=gava_clip_amd/libgava_hip_x.so" [--rounds 3] [--config c2]

Each child runs the bench's forward loop (tools/ab_env.py --child) and prints ms/forward."""
import os, subprocess, sys, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cname, B, steps):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import gava_clip_amd.config as C
    from gava_clip_amd import VitaCLIP
    from helpers import model_kwargs
    cfg = getattr(C, cname)
    cls_path = os.path.join(REPO, "gava_clip_amd", "data", "classes", "updrs_3cls_classes.txt")
    torch.manual_seed(0)
    model = VitaCLIP(**model_kwargs(cfg, cls_path)).cuda().eval()
    x = torch.randn(B, 3, cfg.num_frames, cfg.input_size, cfg.input_size, device="cuda")
    with torch.no_grad():
        for _ in range(10):
            model(x)
        torch.cuda.synchronize()
        step = lambda: model(x)
        if os.environ.get("GAVA_AB_GRAPH") == "1":      # the whole forward captured once into a hipGraph, replays timed
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                model(x)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = model(x)[0]
            step = g.replay
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record(); torch.cuda.synchronize()
        lg = out if os.environ.get("GAVA_AB_GRAPH") == "1" else model(x)[0]
    print("MS %.4f CHK %.6f" % (e0.elapsed_time(e1) / steps, float(lg.double().abs().sum())))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    if "--rounds" in sys.argv:
        args.remove(sys.argv[sys.argv.index("--rounds") + 1])
    cfg = {"c2": ("VIT_B16_T8", 64), "c3": ("VIT_B16_T16", 32), "c5": ("VIT_L14_T32", 32)}[
        sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "c2"]
    if "--config" in sys.argv:
        args.remove(sys.argv[sys.argv.index("--config") + 1])
    res = {}
    for r in range(rounds):
        for a in args:
            name, _, envs = a.partition(":")
            env = dict(os.environ)
            for kv in filter(None, envs.split(",")):
                k, _, v = kv.partition("=")
                env[k] = os.path.join(REPO, v) if k == "GAVA_HIP_LIB" and not os.path.isabs(v) else v
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg[0], str(cfg[1]), "30"], env=env,
                                 capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("MS")]
            if not line:
                print(name, "FAILED", out.stderr[-400:]); continue
            ms, chk = float(line[0].split()[1]), line[0].split()[3]
            res.setdefault(name, []).append(ms)
            print(f"round {r} {name}: {ms:.3f} ms/forward (logits checksum {chk})", flush=True)
    for name, v in res.items():
        print(f"== {name}: median {statistics.median(v):.3f} min {min(v):.3f} ms/forward over {len(v)} rounds")
