#!/usr/bin/env python
"""Timeline view of a rocprofv3 --kernel-trace CSV: busy time, idle gaps and the per-kernel table of the
last forward.  python tools/trace_gaps.py <kernel_trace.csv> [n_forwards_to_skip]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# one forward = from a patch-embed GEMM (EPI 3) to the next
import re
starts = [i for i, r in enumerate(rows) if re.search(r"gemm_kernel<\w+, 3,", r["Kernel_Name"])]
if len(starts) < 3:
    print("no forward boundaries found"); sys.exit(0)
a, b = starts[-2], starts[-1]
fw = rows[a:b]
t0, t1 = fw[0]["s"], rows[b]["s"]
print(f"forward wall {(t1 - t0) / 1e6:.3f} ms, {len(fw)} kernels")
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for r in fw:
    if cur_e is None or r["s"] > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = r["s"], r["e"]
    else:
        cur_e = max(cur_e, r["e"])
busy += cur_e - cur_s
print(f"GPU busy (union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for r in fw:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split(">(")[0][:90]
    agg[k][0] += r["e"] - r["s"]; agg[k][1] += 1
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"{t / 1e6:8.3f} ms  x{n:<4d} {k}")
