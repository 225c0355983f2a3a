#!/bin/bash
# End-of-round evidence: GPU tests, the bench line, the rocprofv3 kernel summary of the same workload, PMC passes of the
# roofline kernel.  Run on the GPU box from the repo root; everything lands in gpurun_out/final/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err || { tail -20 $O/bench_c2.err; exit 1; }
tail -c 600 $O/bench_c2.json; echo
timeout -k 10 600 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_c3.json 2> $O/bench_c3.err || { tail -20 $O/bench_c3.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-kernels --no-train > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
cd $R
bash tools/pmc.sh fc1fold fc1fold --iters 3 > $O/pmc_fc1fold.txt 2>&1 || true
tail -30 $O/pmc_fc1fold.txt
