#!/bin/bash
# Why does the training step read 3.5 ms inside bench.py and 3.05 stand-alone?  The same measure() in four contexts.
mkdir -p gpurun_out
run() { python - "$@" <<'PY'
import json, sys, os, gc, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import secondary_bench as sb
cpu = sys.argv[1] == "cpu"
pre = sys.argv[2] if len(sys.argv) > 2 else ""
if pre == "tower":                      # what bench.py holds when it calls measure(): the g/14 model, its workspace, a 1024-frame batch
    import hirest_amd
    from hirest_amd import synth
    m = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to("cuda:0").eval(); m.init_random_(seed=1)
    x = torch.randn(1024, 3, 224, 224, device="cuda:0").to(torch.bfloat16)
    m.encode_image(x); torch.cuda.synchronize()
out = sb.measure(cpu=cpu, log=lambda m: None)
t = out["train_step"]
print(f"cpu_oracle_legs={cpu} before={pre or 'nothing'}: train_step {t['value']:.3f} ms, fused AdamW {t['ms_per_step_with_fused_adamw']:.3f} ms", flush=True)
PY
}
{
run nocpu
run cpu
run nocpu tower
run cpu tower
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_ctx.txt
