#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
{
for r in 1 2; do
python tools/r3_two_chains.py --chains 1
GAVA_SIDE_STREAM=0 python tools/r3_two_chains.py --chains 1
GAVA_SIDE_STREAM=0 python tools/r3_two_chains.py --chains 2
GAVA_SIDE_STREAM=0 GAVA_CU_RESERVE=128 python tools/r3_two_chains.py --chains 2
GAVA_SIDE_STREAM=0 GAVA_CU_RESERVE=96 python tools/r3_two_chains.py --chains 2
GAVA_SIDE_STREAM=0 GAVA_CU_RESERVE=160 python tools/r3_two_chains.py --chains 2
done
} 2>&1 | grep -v Warning | tee $O/two_chains.log
