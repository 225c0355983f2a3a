#!/bin/bash
# round 3: throttled epilogue (s_sleep after every 64-column chunk) with and without a start offset of the CU's second workgroup
O=gpurun_out/r3; mkdir -p $O
L=gava_clip_amd/libgava_hip_abl.so
run() { echo "== $1 :: $(env $2 GAVA_HIP_LIB=$L timeout -k 10 120 python tools/kernel_bench.py $3 --iters 30 2>/dev/null | tail -1)"; }
{
run "base" "GAVA_PAIR_MODE=0" outpart
for s in 4 8 16 32 64; do
  run "sleep $s" "GAVA_PAIR_MODE=512 GAVA_PAIR_SLEEP=$s" outpart
  run "sleep $s + delay 15us" "GAVA_PAIR_MODE=513 GAVA_PAIR_SLEEP=$s GAVA_PAIR_DELAY=1500" outpart
  run "sleep $s + delay 30us hw" "GAVA_PAIR_MODE=515 GAVA_PAIR_SLEEP=$s GAVA_PAIR_DELAY=3000" outpart
done
run "base" "GAVA_PAIR_MODE=0" fc2part
for s in 16 64; do
  run "sleep $s" "GAVA_PAIR_MODE=512 GAVA_PAIR_SLEEP=$s" fc2part
  run "sleep $s + delay 40us" "GAVA_PAIR_MODE=513 GAVA_PAIR_SLEEP=$s GAVA_PAIR_DELAY=4000" fc2part
done
} 2>&1 | tee $O/pair_exp3.log
