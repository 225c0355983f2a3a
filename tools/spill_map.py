#!/usr/bin/env python
"""Where a kernel's scratch traffic sits relative to its MFMAs (build container, no GPU):
    python tools/spill_map.py gemm.hip "gemm256_kernelI7PrecF16Li1ELb0ELb0ELb1ELb0ELb1E" [-DFLAG ...]
prints, per 100-line bin of the kernel's assembly, the number of MFMAs, scratch loads / stores, v_readlane / v_writelane."""
import collections, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
flags = [a for a in sys.argv[3:] if a.startswith("-D")]
out = "/tmp/spill_map.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DGAVA_ABI_HASH=1", "-S", "--cuda-device-only",
                os.path.join(REPO, "gava_clip_amd", "csrc", src), "-o", out] + flags, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and l.rstrip().endswith(tuple("E:")) or (l.startswith("_ZN") and pat in l and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
open("/tmp/spill_map_kernel.s", "w").write("\n".join(body))
bins = collections.defaultdict(lambda: collections.Counter())
for i, l in enumerate(body):
    for key, tag in (("mfma", "v_mfma"), ("sld", "scratch_load"), ("sst", "scratch_store"), ("rdl", "v_readlane"), ("wrl", "v_writelane"), ("bar", "s_barrier"),
                     ("glds", "global_load_lds"), ("vmcnt", "vmcnt")):
        if tag in l:
            bins[i // 100][key] += 1
print(f"{len(body)} lines; kernel text in /tmp/spill_map_kernel.s")
for b in sorted(bins):
    c = bins[b]
    if c["sld"] or c["sst"] or c["rdl"] or c["wrl"] or c["mfma"]:
        print(f"  lines {b * 100:5d}+: mfma {c['mfma']:3d}  scratch ld {c['sld']:3d} st {c['sst']:3d}  readlane {c['rdl']:3d} writelane {c['wrl']:3d}  barrier {c['bar']} glds {c['glds']} vmcnt {c['vmcnt']}")
