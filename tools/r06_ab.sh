#!/bin/bash
# Generic round-6 A/B inside ONE gpurun call: tools/r06_ab.sh "<variant tags>" "<gemm_bench shapes>" [pytest selection]
# (variant "default" = the in-tree library; others = hirest_amd/lib/libhirest_hip.<tag>.so from tools/build_variant.sh)
tags=$1; shapes=$2; shift 2
mkdir -p gpurun_out
{
if [ -n "$1" ]; then echo "== parity (default build): $*"; timeout 1500 python -m pytest "$@" -m gpu -q 2>&1 | tail -4; fi
for round in 1 2; do for v in $tags; do
  if [ $v = default ]; then unset HIREST_LIB_VARIANT; else export HIREST_LIB_VARIANT=$v; fi
  echo "== build $v (round $round)"
  timeout 300 python tools/gemm_bench.py --variants 0 --iters 20 --shapes $shapes 2>&1 | grep -v amdgpu.ids | tail -6
done; done
for round in 1 2; do for v in $tags; do
  if [ $v = default ]; then unset HIREST_LIB_VARIANT; else export HIREST_LIB_VARIANT=$v; fi
  echo "== bench, build $v"
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matched-recall --no-secondary > gpurun_out/bb.log 2>&1
  python - <<PY
import json
x=json.loads(open("gpurun_out/bb.log").read().strip().splitlines()[-1])
print("frames/s %.0f"%x["value"], sorted([(e["tag"],e["dims"][1],e["dims"][2],round(e["avg_ms"],3)) for e in x["roofline"]["breakdown"][:6]]))
PY
done; done
} 2>&1 | tee gpurun_out/r06_ab.txt
