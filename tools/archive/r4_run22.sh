cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4final
S=$(date +%s); timeout -k 10 900 python bench.py > gpurun_out/r4final/bench_c2_d.json 2> gpurun_out/r4final/bench_c2_d.err; echo "bench wall $(( $(date +%s) - S )) s"; grep "bench +" gpurun_out/r4final/bench_c2_d.err | tail -12
python - <<PY
import json
d=json.load(open("gpurun_out/r4final/bench_c2_d.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic"], d["parity_modes"])
a=d["accuracy"]
for k in ("fp16","bf16","fp16+wlo8","fp16+wlo"): print(k, a[k]["fixtures"], a[k]["normwise"], a[k]["max_mixed_violation"])
PY
