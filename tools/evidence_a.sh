#!/bin/bash
# Evidence bundle, part A (run on the GPU box from the repo root): GPU tests, the bench lines of c2 / c3 / c5, the 2-rank
# rehearsal of the bench's N > 1 leg (gloo rendezvous, both ranks on the one GPU), smoke().  Lands in gpurun_out/<tag>/.
#   bash tools/evidence_a.sh r4final        (SKIP_TESTS=1 leaves the test suite out)
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || tail -20 $O/bench_c2.err
timeout -k 10 400 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_c3.json 2> $O/bench_c3.err || tail -5 $O/bench_c3.err
timeout -k 10 400 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_c5.json 2> $O/bench_c5.err || tail -5 $O/bench_c5.err
GAVA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_c2_2rank_gloo.json 2> $O/bench_c2_2rank_gloo.err || tail -20 $O/bench_c2_2rank_gloo.err
python - <<PY
import json
for c in ("c2", "c3", "c5", "c2_2rank_gloo"):
    try:
        d = json.loads([l for l in open("$O/bench_%s.json" % c).read().splitlines() if l.strip()][-1])
        r = d.get("roofline") or {}
        print(c, d["value"], d["ms_per_step"], r.get("frac"), r.get("traffic"), d.get("parity_modes"), (d.get("train_step") or {}).get("ms_per_step"), d["config"].get("parallelism"))
    except Exception as e:
        print(c, "failed:", e)
PY
