#!/bin/bash
# Round-2 evidence bundle (run on the GPU box from the repo root): everything lands in gpurun_out/r2final/ and the
# summaries are copied to profiles/ by hand afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2final; mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
fi
timeout -k 10 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || tail -20 $O/bench_c2.err
tail -c 400 $O/bench_c2.json; echo
timeout -k 10 600 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_c3.json 2> $O/bench_c3.err || tail -5 $O/bench_c3.err
timeout -k 10 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_c5.json 2> $O/bench_c5.err || tail -5 $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c5; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o x -- python3 $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $O/prof_$cfg.log 2>&1 || tail -5 $O/prof_$cfg.log
  S=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1); cp $S $O/bench_${cfg}_kernel_stats.csv
  T=$(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_gaps.py $T > $O/gaps_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
done
cd $R
for k in fc1part qkvpart outpart fc2part attn; do
  bash tools/pmc.sh r2_$k $k --iters 3 > $O/pmc_$k.txt 2>&1
  rm -rf $R/gpurun_out/pmc_r2_$k
  echo "== $k"; grep -E "^void|^\(anon|FETCH_SIZE|WRITE_SIZE|MFMA_BUSY|GRBM_GUI|BANK_CONFLICT|TCC_HIT|TCC_MISS|SQ_INSTS_MFMA|SQ_INSTS_VALU|SQ_BUSY_CYCLES" $O/pmc_$k.txt | cut -c1-110
done
python3 tools/make_traffic.py $O/pmc_fc1part.txt $O/traffic.json > /dev/null && tail -8 $O/traffic.json
# board power / clock while the forward loops (rocm-smi every 0.25 s)
( python3 - <<PY
import sys, os, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import gava_clip_amd.config as C
from gava_clip_amd import VitaCLIP
from helpers import model_kwargs
m = VitaCLIP(**model_kwargs(C.VIT_B16_T8, "$R/gava_clip_amd/data/classes/updrs_3cls_classes.txt")).cuda().eval()
x = torch.randn(64, 3, 8, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(400): m(x)
torch.cuda.synchronize()
PY
) &
LOOP=$!
sleep 6
for i in $(seq 1 16); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done > $O/power_watch.txt
wait $LOOP
head -4 $O/power_watch.txt
