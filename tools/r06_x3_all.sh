#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_joint.py tests/test_gpu_x3.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5
python - <<'PY'
import json, sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import secondary_bench as sb
out = sb.measure(cpu=False, log=lambda m: None)
for k, v in out.items():
    if k.startswith("moment_") or k.startswith("step_captioning") or k == "train_step":
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_batch", "speedup_vs_fp32", "token_ids_equal_real_reference", "predictions_equal_real_reference", "indices_equal_real_reference", "boundaries_equal_real_reference")})
json.dump(out, open("gpurun_out/secondary_r06.json", "w"), indent=1)
PY
} 2>&1 | tee gpurun_out/r06_x3_all.txt
