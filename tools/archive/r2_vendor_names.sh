#!/bin/bash
# names (macro tile, depth, workgroup) of the vendor kernels torch.matmul picks for the four big GEMM shapes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vendor_prof -o x -- python3 $R/tools/r2_vendor_names.py > $O/vendor_prof.log 2>&1 < /dev/null || { tail -5 $O/vendor_prof.log; exit 1; }
f=$(find $O/vendor_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-600 "$f" | head -8
