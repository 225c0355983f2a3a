cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4final
timeout -k 10 900 python bench.py > gpurun_out/r4final/bench_c2_c.json 2> gpurun_out/r4final/bench_c2_c.err || tail -20 gpurun_out/r4final/bench_c2_c.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4final/bench_c2_c.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic"], r["algorithmic_bytes_per_launch"], r["in_forward_ms_per_launch"], r["traffic_source"][:60])
PY
