#!/usr/bin/env python
"""profiles/traffic.json from the PMC passes of tools/pmc.sh, one file per per-layer kernel in the form the forward launches
it (run on the GPU box: `bash tools/pmc.sh r3_fc1part fc1part --iters 3 > gpurun_out/r3final/pmc_fc1part.txt`, likewise
qkvpart / outpair / fc2pair / attn): FETCH_SIZE / WRITE_SIZE per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950 (FETCH_SIZE counts 128-B requests at 64 B -> x2; WRITE_SIZE exact for 16-B/lane stores), together
with the sha256 of the kernel sources the passes were measured on - bench.py refuses to report traffic measured on other
sources.

    python tools/make_traffic.py DIR [profiles/traffic.json]      # DIR holds pmc_{fc1part,qkvpart,outpair,fc2pair,attn}.txt
"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# bench.py key -> (pmc file tag, substring of the kernel name rocprofv3 prints, algorithmic bytes per launch at c2)
M, D, F = 100864, 768, 3072
CASES = {
    "fc1": ("fc1part", "gemm256_kernel<PrecF16, 1, false, false, true, false, 0, true, false>", M * D * 2 + F * D * 2 + M * F * 2 + M * 8),
    "qkv": ("qkvpart", "gemm256_kernel<PrecF16, 0, false, false, true, false, 0, true, false>", M * D * 2 + 3 * D * D * 2 + M * 3 * D * 2 + M * 8),
    # (the big-batch forward keeps the residual stream as a 16-bit pair: 8 bytes per element through the producers' epilogue)
    "out": ("outpair", "gemm256_kernel<PrecF16, 2, true, false, false, true>", M * D * 2 + M * D * 8 + D * D * 2 + M * (D // 64) * 8),
    "fc2": ("fc2pair", "gemm256_kernel<PrecF16, 2, true, false, false, true>", M * F * 2 + M * D * 8 + D * F * 2 + M * (D // 64) * 8),
    "attn": ("attn", "attention_persist_kernel<PrecF16, 14, 13>", M * 3 * D * 2 + M * D * 2),
}


def counters(path, pattern):
    cur, vals = None, {}
    for ln in open(path):
        if ln.startswith("void") or ln.startswith("(anon"):
            cur = ln.strip()
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=([0-9.e+-]+)", ln)
        if m and cur and pattern.split("<")[0] in cur and pattern.split("<", 1)[1].rstrip(">")[:30] in cur:
            vals[m.group(1)] = float(m.group(2))
    return vals


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "traffic.json")
    import bench
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes), tools/pmc.sh <kernel> --iters 3 "
                  f"per kernel ({os.path.basename(os.path.normpath(src))}/pmc_*.txt)",
        "config": "c2",
        "kernel_source_sha256": bench.kernel_source_hash(),
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane stores; "
                      "both count L2 misses to the fabric, Infinity-Cache hits included",
        "kernels": {},
    }
    for key, (tag, pattern, alg) in CASES.items():
        path = os.path.join(src, f"pmc_{tag}.txt")
        if not os.path.exists(path):
            print("missing", path)
            continue
        v = counters(path, pattern)
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            print("no counters for", key, "in", path)
            continue
        rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
        e = {"kernel": pattern, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
             "algorithmic_bytes_per_launch": alg, "ratio": round((rd + wr) / alg, 3)}
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
        if v.get("SQ_BUSY_CYCLES"):
            e["mfma_busy_frac_at_held_clock"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["SQ_BUSY_CYCLES"] * 4 * 8), 4)
        if "SQ_LDS_BANK_CONFLICT" in v:
            e["lds_bank_conflict_cycles"] = v["SQ_LDS_BANK_CONFLICT"]
        out["kernels"][key] = e
    assert out["kernels"], f"no kernel counters under {src}"
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
