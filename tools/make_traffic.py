#!/usr/bin/env python
"""profiles/traffic.json from the PMC passes of tools/pmc.sh (run on the GPU box, after `bash tools/pmc.sh fc1fold fc1fold
--iters 3 > gpurun_out/pmc_fc1fold.txt`): FETCH_SIZE / WRITE_SIZE of the roofline kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950, per launch, together with the sha256 of the kernel sources
the passes were measured on - bench.py refuses to report a traffic figure measured on other sources.

    python tools/make_traffic.py gpurun_out/pmc_fc1fold.txt [profiles/traffic.json]
"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "traffic.json")
    import bench
    cur, vals = None, {}
    for ln in open(src):
        if ln.startswith("void") or ln.startswith("(anon"):
            cur = ln.strip()
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=([0-9.e+-]+)", ln)
        if m and cur and "gemm256_kernel" in cur and ", 1, false, false, true>" in cur:      # EPI_H16_QGELU, FOLD
            vals[m.group(1)] = float(m.group(2))
    assert "FETCH_SIZE" in vals and "WRITE_SIZE" in vals, f"no fc1 FOLD kernel counters in {src}"
    rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
    M, N, K = 100864, 3072, 768
    out = {
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes), tools/pmc.sh fc1fold fc1fold --iters 3 ({os.path.basename(src)})",
        "kernel": "gemm256_kernel<PrecF16, EPI_H16_QGELU, FOLD> (vision fc1 with LayerNorm folded in, M=100864 N=3072 K=768)",
        "kernel_source_sha256": bench.kernel_source_hash(),
        "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane stores",
        "gemm_fc1_read_bytes_per_launch": rd, "gemm_fc1_write_bytes_per_launch": wr, "gemm_fc1_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": M * K * 2 + N * K * 2 + M * N * 2 + M * 8,
    }
    if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
        out["l2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
        out["mfma_busy_frac_at_held_clock"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["SQ_BUSY_CYCLES"] * 4 * 8) if vals["SQ_BUSY_CYCLES"] else None
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
