#!/bin/bash
# Sample socket power / clocks with rocm-smi while a GEMM variant runs in a loop. usage: tools/power_probe.sh <shape> <variant|hipblaslt>
shape=$1; var=$2
if [ "$var" = "hipblaslt" ]; then args="--variants --hipblaslt"; else args="--variants $var"; fi
timeout 120 python tools/gemm_bench.py $args --iters 1500 --shapes $shape > /tmp/pp_$var.log 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' ; echo
  sleep 0.5
done
wait $pid
tail -2 /tmp/pp_$var.log
