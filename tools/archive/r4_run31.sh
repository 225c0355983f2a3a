mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r4dyn
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4dyn/trace -o x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $GRAFT_REPO_ROOT/gpurun_out/r4dyn/trace.log 2>&1
cd $GRAFT_REPO_ROOT; T=$(find gpurun_out/r4dyn/trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$T")))
print(rows[0].keys())
rows=[r for r in rows if "gemm256_kernel<PrecF16, 0, false, false, true, false, 0, false, false, true>" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=rows[-30:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print(r.get("Grid_Size"), r.get("Workgroup_Size"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id"), r.get("Stream_Id"))
PY
rm -rf gpurun_out/r4dyn/trace
