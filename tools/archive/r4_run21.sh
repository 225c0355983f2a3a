R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_side -o x -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt --no-kernels --no-train --no-accuracy > $O/prof_side.log 2>&1 || tail -5 $O/prof_side.log
T=$(find $O/prof_side -name "*kernel_trace.csv" | head -1)
python3 $R/tools/side_timeline.py $T > $O/side_timeline.txt 2>&1; cat $O/side_timeline.txt
rm -rf $O/prof_side
