#!/bin/bash
# rocprofv3 kernel summary of three c2 training steps (forward with kept activations + backward + AdamW)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -o x -- python3 $R/tools/train_bench.py --B 64 --iters 3 > $O/train_prof.log 2>&1 < /dev/null || { tail -5 $O/train_prof.log; exit 1; }
f=$(find $O/train_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/train_kernel_stats.csv && rm -rf $O/train_prof
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (4 steps incl. warm-up):", tot/4/1e6)
for r in rows[:22]:
    print("%6.2f ms/step x%-4d %7.1f us  %s" % (float(r["TotalDurationNs"])/4/1e6, int(r["Calls"])//4, float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:95]))
PY
