#!/bin/bash
# per-kernel time of the c2 forward under two settings of one env variable (rocprofv3 --stats, same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  name=${tag%%:*}; envs=${tag#*:}
  ( [ -n "$envs" ] && export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cmp_$name -o c2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $O/cmp_$name.log 2>&1 )
  S=$(find $O/cmp_$name -name "*kernel_stats.csv" | head -1); cp $S $O/cmp_${name}_kernel_stats.csv; rm -rf $O/cmp_$name
  grep -o '"ms_per_step": [0-9.]*' $O/cmp_$name.log
done
python3 - "$@" <<PY
import csv, sys
O="$O"
tabs={}
for tag in sys.argv[1:]:
    name=tag.split(":")[0]
    t={}
    for r in csv.DictReader(open(f"{O}/cmp_{name}_kernel_stats.csv")):
        k=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
        t[k]=(int(r["Calls"]), float(r["TotalDurationNs"])/13/1e6)   # 13 forwards (3 warm-up + 10)
    tabs[name]=t
names=list(tabs)
keys=sorted(set().union(*[set(t) for t in tabs.values()]), key=lambda k:-max(t.get(k,(0,0))[1] for t in tabs.values()))
print("ms per forward by kernel:", names)
for k in keys[:18]:
    print("  ".join(f"{tabs[n].get(k,(0,0))[1]:7.3f} x{tabs[n].get(k,(0,0))[0]//13:<3d}" for n in names), k)
print("  ".join(f"{sum(v[1] for v in tabs[n].values()):7.3f}     " for n in names), "TOTAL kernel time")
PY
