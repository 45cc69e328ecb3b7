#!/bin/bash
# A/B inside one gpurun call: fc2 walking its rows backwards (Infinity-Cache reuse across kernels) on / off
for f in "0" "512" "0" "512"; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-dbg $f > gpurun_out/bb.log 2>&1; python - <<PY
import json
x=json.loads(open("gpurun_out/bb.log").read().strip().splitlines()[-1])
print("[dbg $f] frames/s %.0f"%x["value"], sorted([(e["kernel"][:4],e["tag"],e["dims"][1],e["dims"][2],round(e["avg_ms"],3)) for e in x["roofline"]["breakdown"][:7] if e["avg_ms"]>0.3]))
PY
done
