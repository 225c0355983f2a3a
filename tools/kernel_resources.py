#!/usr/bin/env python
"""Registers / scratch / LDS of every kernel of a source file as hipcc allocates them (no GPU needed):

    python tools/kernel_resources.py gemm.hip [-DFLAG ...] [--grep gemm256]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = [a for a in sys.argv[2:] if a.startswith("-D")]
pat = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DGAVA_ABI_HASH=1", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.join(REPO, "gava_clip_amd", "csrc", src), "-o", "/dev/null"] + flags
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for ln in err.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = dict(name=m.group(1)); rows.append(cur); continue
    for key, tag in (("sgpr", "TotalSGPRs"), ("vgpr", "VGPRs"), ("agpr", "AGPRs"), ("scratch", "ScratchSize \\[bytes/lane\\]"), ("lds", "LDS Size \\[bytes/block\\]"), ("occ", "Occupancy \\[waves/SIMD\\]")):
        m = re.search(r"\s" + tag + r": (\d+)", ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("(GemmParams)", "")
    if pat and pat not in name:
        continue
    print(f"{name[:100]:100s} sgpr {r.get('sgpr', -1):3d} vgpr {r.get('vgpr', -1):3d} agpr {r.get('agpr', 0):3d} scratch {r.get('scratch', -1):4d} lds {r.get('lds', -1):6d} occ {r.get('occ', -1)}")
