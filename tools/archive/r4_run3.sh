set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py -x -q -s -k "odd_storage or full_size_fixtures or c3_c5_full_batch or c2_full" > $O/t_fwd_new.log 2>&1 || (tail -40 $O/t_fwd_new.log; exit 1)
grep -E "^\[|passed|failed" $O/t_fwd_new.log | cut -c1-220
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_c2_a.json 2> $O/bench_c2_a.err || (tail -20 $O/bench_c2_a.err; exit 1)
python - <<PY
import json
d=json.load(open("$O/bench_c2_a.json"))
print("value", d["value"], "ms", d["ms_per_step"]); print("roofline", {k:d["roofline"][k] for k in ("bound","achieved","frac","standalone_frac","ms_per_launch","in_forward_ms_per_launch")})
print("parity_modes", d.get("parity_modes")); print("train", d.get("train_step"), d.get("train_roofline",{}).get("frac"))
for k in ("fp16","bf16","fp16+wlo8","fp16+wlo"):
    print(k, d["accuracy"][k]["normwise"], d["accuracy"][k]["max_mixed_violation"])
for r in d["kernels"][:8]: print(r["kernel"][:50], r["ms"], r.get("tflops"))
PY
