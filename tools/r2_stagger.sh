#!/bin/bash
# out_proj / fc2 producers with late-start patterns (GAVA_STAGGER=mode,ticks[10 ns])
for st in "0,0" "1,500" "1,1000" "1,2000" "2,1000" "2,2000" "2,3000" "3,2000" "3,4000" "4,2000" "4,3000"; do
  for k in outfold out fc2fold; do
    echo -n "stagger $st  "; GAVA_STAGGER=$st python tools/kernel_bench.py $k --iters 20 2>&1 | grep -v amdgpu
  done
done
