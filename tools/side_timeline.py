#!/usr/bin/env python
"""Is the prompt ("side") path on the critical path?  From a rocprofv3 --kernel-trace CSV of the bench command: per block of the
last forward, when the QKV GEMM ends, when the side stream's last kernel of that block (the K/V projection of the prompt rows)
ends, and when the attention starts.    python tools/side_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if re.search(r"gemm_kernel<\w+, 3,", r["k"])]
a, b = starts[-2], starts[-1]
fw = rows[a:b]
t0 = fw[0]["s"]
attn = [r for r in fw if r["k"].startswith("attention_persist_kernel")]
print(f"forward wall {(rows[b]['s'] - t0) / 1e6:.3f} ms; {len(attn)} full-width attention launches")
print("block  qkv_start  qkv_end  side_first_start  side_last_end  attn_start   (us from the forward's start)   attention waited for: ")
for i, at in enumerate(attn):
    prev_end = attn[i - 1]["e"] if i else t0
    win = [r for r in fw if prev_end <= r["s"] < at["s"]]
    qkv = [r for r in win if re.match(r"gemm256_kernel<\w+, 0, false, false, (true|false), false", r["k"])]
    side = [r for r in win if r["k"].startswith("gemm_kernel<") or r["k"].startswith("side_ln") or r["k"].startswith("attention_kernel<")]
    if not qkv or not side:
        continue
    q = qkv[-1]
    # the block's side chain starts after the previous block's fc2: take the kernels that start after the last producer GEMM before qkv
    prod = [r for r in win if re.match(r"gemm256_kernel<\w+, 2,", r["k"])]
    chain_from = prod[-1]["s"] if prod else prev_end
    chain = [r for r in side if r["s"] >= chain_from]
    if not chain:
        continue
    who = "side path" if chain[-1]["e"] > q["e"] else "qkv"
    u = lambda t: (t - t0) / 1e3
    print(f"{i + 1:5d} {u(q['s']):10.1f} {u(q['e']):8.1f} {u(chain[0]['s']):17.1f} {u(chain[-1]['e']):14.1f} {u(at['s']):11.1f}   {who}  "
          f"(qkv {(q['e'] - q['s']) / 1e3:.0f} us, chain {(chain[-1]['e'] - chain[0]['s']) / 1e3:.0f} us over {len(chain)} kernels)")
