#!/bin/bash
# tile walk A/B: time + fabric fetch (FETCH_SIZE, TCC hit rate) of the consumers
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2; mkdir -p $O
for rep in 1 2; do for w in 0 1; do for k in fc1fold qkvfold fc1 qkv; do
  echo -n "walk $w  "; GAVA_TILE_WALK=$w python tools/kernel_bench.py $k --iters 20 2>&1 | grep -v amdgpu
done; done; done
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do for k in fc1fold qkvfold; do
  for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    GAVA_TILE_WALK=$w timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/walk_${w}_${k} -- python3 $R/tools/kernel_bench.py $k --iters 3 > $O/walk_pmc.log 2>&1 || echo "pass failed"
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/walk_${w}_${k}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("walk $w $k:", {c: round(sum(v)/len(v), 1) for c, v in sorted(agg.items())})
PY
  rm -rf $O/walk_${w}_${k}
done; done
