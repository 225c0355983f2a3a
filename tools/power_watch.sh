#!/bin/bash
# Samples board power and shader clock (rocm-smi, read-only) while a forward-only bench loop runs: is the forward power-capped?
mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/power_watch.log &
W=$!
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-kernels --no-cpu-baseline --no-alt --no-train 2>/dev/null | tail -1 | cut -c1-200
kill $W 2>/dev/null
sed -n '1p;10p;20p;24p;28p;32p;36p;40p' gpurun_out/power_watch.log | cut -c1-300
