#!/bin/bash
# Round 6, two-array residual epilogue: flat (round 5) vs buffer (straight-line, counted waits), all in ONE gpurun call.
mkdir -p gpurun_out
{
echo "== parity of the default build (buffer epilogue)"
python tools/r06_dbg_epi.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_lnfold.py -m gpu -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for round in 1 2; do
for v in flat default; do
  if [ $v = default ]; then unset HIREST_LIB_VARIANT; else export HIREST_LIB_VARIANT=$v; fi
  echo "== build $v (round $round)"
  timeout 300 python tools/gemm_bench.py --variants 0 --iters 20 --shapes proj_stats2 fc2_stats2 2>&1 | tail -2
done; done
unset HIREST_LIB_VARIANT
for v in flat default flat default; do
  if [ $v = default ]; then unset HIREST_LIB_VARIANT; else export HIREST_LIB_VARIANT=$v; fi
  echo "== bench, build $v"
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matched-recall --no-secondary > gpurun_out/bb.log 2>&1
  python - <<PY
import json
x=json.loads(open("gpurun_out/bb.log").read().strip().splitlines()[-1])
print("frames/s %.0f"%x["value"], sorted([(e["tag"],e["dims"][1],e["dims"][2],round(e["avg_ms"],3)) for e in x["roofline"]["breakdown"][:6]]))
PY
done
} 2>&1 | tee gpurun_out/r06_epi_ab.txt
