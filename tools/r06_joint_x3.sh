#!/bin/bash
# Round 6: the joint model's split-operand precision — parity gates, then fp32 vs bf16x3 throughput, then kernel stats
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_joint.py -m gpu -q -x -k "moment_model_vs_reference or step_captioning_vs_reference" 2>&1 | grep -v amdgpu.ids | tail -4
python - <<'PY'
import json, sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import secondary_bench as sb
out = sb.measure(cpu=False, log=lambda m: None)
keep = {k: v for k, v in out.items() if k.startswith("moment_")}
for k, v in keep.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "roofline"}, "frac", round(v["roofline"]["frac"], 3))
json.dump(out, open("gpurun_out/secondary_r06.json", "w"), indent=1)
PY
bash tools/r06_joint_prof.sh 2>&1 | grep -v "^W2026\|fp32 kernel\|gemm_f32\|amdgpu.ids" | head -24
} 2>&1 | tee gpurun_out/r06_joint_x3.txt
