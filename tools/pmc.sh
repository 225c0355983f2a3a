#!/bin/bash
# usage: tools/pmc.sh <tag> <kernel_bench args...>   (run on the GPU box from the repo root)
# Separate rocprofv3 --pmc passes (counter groups do not fit one pass; never combined with traces).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/kernel_bench.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" in k or "attention_" in k or "layernorm_kernel" in k or "patchify" in k:
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
