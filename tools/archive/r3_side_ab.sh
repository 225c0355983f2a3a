#!/bin/bash
# round 3: what the prompt ("side") path costs the main stream, and eager launches vs hipGraph replay.
# Same box, 3 alternating rounds, c2.
O=gpurun_out/r3; mkdir -p $O
python tools/ab_env.py "base:" "graph:GAVA_AB_GRAPH=1" "side_cus0:GAVA_SIDE_CUS=0" "side_cus8:GAVA_SIDE_CUS=8" "side_cus16:GAVA_SIDE_CUS=16" \
   "side_cus48:GAVA_SIDE_CUS=48" "one_stream:GAVA_SIDE_STREAM=0" --rounds 3 2>&1 | tee $O/side_ab.log
