#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_joint.py -m gpu -q -k "step_captioning_vs_reference or caption_batches" 2>&1 | grep -v amdgpu.ids | tail -6
python - <<'PY'
import json, sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import secondary_bench as sb
out = sb.measure(cpu=False, log=lambda m: None)
for k, v in out.items():
    if k.startswith("step_captioning"):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_batch", "speedup_vs_fp32", "token_ids_equal_real_reference")})
json.dump(out, open("gpurun_out/secondary_r06.json", "w"), indent=1)
PY
} 2>&1 | tee gpurun_out/r06_caption_x3.txt
