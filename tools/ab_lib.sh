#!/bin/bash
# A/B of two builds inside ONE gpurun call (box-to-box variance is 5-8 %): copy the baseline library to
# hirest_amd/lib/base.so.keep before rebuilding, then `gpurun -- bash tools/ab_lib.sh` alternates new / base twice.
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary > gpurun_out/bb.log 2>&1; python - <<PY
import json
x=json.loads(open("gpurun_out/bb.log").read().strip().splitlines()[-1])
print("$1 frames/s %.0f"%x["value"], sorted([(e["tag"],e["dims"][1],e["dims"][2],round(e["avg_ms"],3)) for e in x["roofline"]["breakdown"][:5]]))
PY
}
cp hirest_amd/lib/libhirest_hip.so /tmp/new.so
run new; cp hirest_amd/lib/base.so.keep hirest_amd/lib/libhirest_hip.so; run base; cp /tmp/new.so hirest_amd/lib/libhirest_hip.so; run new; cp hirest_amd/lib/base.so.keep hirest_amd/lib/libhirest_hip.so; run base
