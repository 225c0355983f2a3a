mkdir -p gpurun_out/r3final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout -k 10 600 python bench.py > gpurun_out/r3final/bench_c2.json 2> gpurun_out/r3final/bench_c2.err || tail -20 gpurun_out/r3final/bench_c2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3final/bench_c2.json'))
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'], r['in_forward_ms_per_launch'])
PY
