#!/bin/bash
# bench + rocprofv3 kernel summary for one config (c3 / c5): gpurun_out/r2/<tag>_bench_<cfg>.json, <tag>_<cfg>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2; mkdir -p $O
TAG=$1; CFG=$2
timeout -k 10 900 python bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-accuracy > $O/${TAG}_bench_$CFG.json 2> $O/${TAG}_bench_$CFG.err || tail -20 $O/${TAG}_bench_$CFG.err
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench_$CFG.json").read().strip().splitlines()[-1])
print("BENCH $CFG", d["value"], "clips/s", d["ms_per_step"], "ms; executed TF/s", d.get("executed_tflops"), "frac", d.get("mfma_frac_executed"), "ref-flops frac", d.get("mfma_frac_reference_flops"))
for k in d.get("kernels", []): print("   ", k["ms"], k["tflops"], k["kernel"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$CFG -o x -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $O/${TAG}_prof_$CFG.log 2>&1 || tail -5 $O/${TAG}_prof_$CFG.log
S=$(find $O/${TAG}_prof_$CFG -name "*kernel_stats.csv" | head -1); cp $S $O/${TAG}_${CFG}_kernel_stats.csv; rm -rf $O/${TAG}_prof_$CFG
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/${TAG}_${CFG}_kernel_stats.csv")))[:10]:
    print(r["Calls"], "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"], r["Name"].replace("(anonymous namespace)::","")[:100])
PY
