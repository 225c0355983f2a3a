#!/bin/bash
# Evidence bundle, part B: rocprofv3 kernel summaries of the bench command (c2, c3, c5) and of one training step, PMC passes of
# the five per-layer kernels in the form the forward launches them -> traffic.json, shader clock while the forward loops.
#   bash tools/evidence_b.sh r4final
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c3 c5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o x -- python3 $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $O/prof_$cfg.log 2>&1 || tail -5 $O/prof_$cfg.log
  S=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1); cp $S $O/bench_${cfg}_kernel_stats.csv
  T=$(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_gaps.py $T > $O/gaps_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
  head -3 $O/bench_${cfg}_kernel_stats.csv | cut -c1-170
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o x -- python3 $R/tools/train_bench.py --B 64 --iters 3 > $O/train_step_c2.txt 2>&1 || tail -5 $O/train_step_c2.txt
S=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); cp $S $O/train_step_c2_kernel_stats.csv; rm -rf $O/prof_train
head -6 $O/train_step_c2_kernel_stats.csv | cut -c1-170
cd $R
for k in fc1part qkvpart outpair fc2pair attn; do
  bash tools/pmc.sh ${TAG}_$k $k --iters 3 > $O/pmc_$k.txt 2>&1
  rm -rf $R/gpurun_out/pmc_${TAG}_$k
  echo "== $k"; grep -E "^void|^\(anon|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|MFMA_BUSY|GRBM" $O/pmc_$k.txt | cut -c1-120
done
python3 tools/make_traffic.py $O $O/traffic.json > /dev/null && head -c 700 $O/traffic.json
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
sleep 8
for i in $(seq 1 16); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done > $O/clock_watch.txt
wait $LOOP
head -3 $O/clock_watch.txt
