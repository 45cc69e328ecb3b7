#!/bin/bash
# VERDICT r5 item 4, first half: where did the training step's 3.07 -> 3.45 ms (profiles/r04 vs r05 bench lines) come from?  Same box, same build:
# the fp32 GEMM's split-K threshold at round 4's value (K >= 1024) and at round 5's (K >= 512), ring mode 0 / default.
mkdir -p gpurun_out
run() {
python - <<'PY'
import json, sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import secondary_bench as sb
out = sb.measure(cpu=False, log=lambda m: None)
t = out["train_step"]
print(os.environ.get("LABEL"), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items() if k in ("value", "videos_per_s", "with_fused_adamw_ms", "fused_adamw_ms_per_step")},
      [ (k, round(v, 3)) for k, v in t.items() if isinstance(v, float)][:8])
PY
}
{
for rep in 1 2; do
LABEL="split-K from K=512 (round 5 default)" run
LABEL="split-K from K=1024 (round 4)" HIREST_F32_SPLIT_MIN_K=1024 run
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_bisect.txt
echo "== hipBLASLt yardstick (torch.mm, plain bf16 out) next to pq256 with the same plain epilogue"
timeout 600 python tools/gemm_bench.py --variants 0 --iters 20 --hipblaslt --shapes qkv fc1_nogelu proj_plain fc2_plain 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_hipblaslt_yardstick.txt
