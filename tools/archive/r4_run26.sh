cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4hl
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_stream_as_16_bit_pair or layernorm_and_join_rows" > gpurun_out/r4hl/ops.log 2>&1; echo "ops rc $?"; tail -3 gpurun_out/r4hl/ops.log
timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -q -s -m gpu -k "residual_stream_as_16_bit_pair or c2_full_batch_vs" > gpurun_out/r4hl/fwd.log 2>&1; echo "fwd rc $?"; grep "^\[" gpurun_out/r4hl/fwd.log; tail -3 gpurun_out/r4hl/fwd.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-accuracy > gpurun_out/r4hl/bench2.json 2> gpurun_out/r4hl/bench2.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r4hl/bench2.json').read().splitlines() if l.strip()][-1])
print(d['value'], d['ms_per_step'])
r=d['roofline']
print(r['frac'], r['standalone_frac'], r['in_forward_ms_per_launch'], r['ms_per_launch'])
for m in r['members']: print(m['kernel'], m['ms_per_launch'], m['in_forward_ms_per_launch'], m['bound'], m['frac'])
for k in d.get('kernels',[]): print(k['kernel'], k['ms'])
PY
