#!/bin/bash
# round-2 measurement bundle (run on the GPU box from the repo root): GPU tests, stamps, kernel trace of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2; mkdir -p $O
TAG=${1:-m1}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; tail -3 $O/${TAG}_gpu_tests.log
fi
for k in ${STAMPS:-}; do timeout -k 10 120 python tools/gemm_stamps.py $k >> $O/${TAG}_stamps.log 2>&1; done
[ -f $O/${TAG}_stamps.log ] && grep -v amdgpu.ids $O/${TAG}_stamps.log
timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-train --no-alt --no-cpu-baseline --no-accuracy > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || tail -20 $O/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench.json").read().strip().splitlines()[-1])
print("BENCH", d["value"], "clips/s", d["ms_per_step"], "ms", "exec frac", d.get("mfma_frac_executed"))
for k in d.get("kernels", []): print("   ", k["ms"], k["tflops"], k["kernel"])
PY
if [ "${TRACE:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o c2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $O/${TAG}_prof.log 2>&1 || tail -20 $O/${TAG}_prof.log
  cd $R
  T=$(find $O/${TAG}_prof -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $T > $O/${TAG}_gaps.txt 2>&1; head -40 $O/${TAG}_gaps.txt
  S=$(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1); cp $S $O/${TAG}_kernel_stats.csv
  # keep the merged output small: the raw trace stays on the box
  rm -rf $O/${TAG}_prof
fi
