#!/bin/bash
# GPU box: sustained rate of the two fp16 MFMA shapes with the board at its power cap, with rocm-smi samples beside
mkdir -p gpurun_out
for shape in 16 32; do
  ( for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/mfma_power_smi_$shape.log &
  timeout -k 10 60 ./tools/probes/mfma_power $shape 3.0 | tail -3
  wait
  grep -o "sclk clock level: [^G]*\|Power (W): [0-9.]*" gpurun_out/mfma_power_smi_$shape.log | paste - - | sed -n '6,10p'
done
