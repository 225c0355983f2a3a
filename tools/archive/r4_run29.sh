cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4dyn
timeout -k 10 500 python tools/ab_env.py "helper:" "nohelper:GAVA_QKV_HELPER=0" "claimsonly:GAVA_QKV_HELPER=2" --rounds 3 > gpurun_out/r4dyn/ab2.log 2>&1; tail -4 gpurun_out/r4dyn/ab2.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4dyn/trace -o x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $GRAFT_REPO_ROOT/gpurun_out/r4dyn/trace.log 2>&1
cd $GRAFT_REPO_ROOT; T=$(find gpurun_out/r4dyn/trace -name "*kernel_trace.csv" | head -1); python tools/side_timeline.py $T > gpurun_out/r4dyn/side_timeline.txt 2>&1; cat gpurun_out/r4dyn/side_timeline.txt; rm -rf gpurun_out/r4dyn/trace
