#!/usr/bin/env python3
"""precision='bf16x3' vision tower: frames/s of 1024-frame calls of EVA-CLIP-g/14 and the per-kernel breakdown from the library's
per-launch event records (the same mechanism bench.py's roofline uses).  --precision fp32 | bf16 for comparison."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd
from hirest_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--attn-waves", type=int, default=0, help="hirest_attention_x3_select_waves (0 auto, 3 / 4 / 8 / 9)")
ap.add_argument("--gemm-kernel", type=int, default=0, help="hirest_gemm_select_kernel (9 = the two-phase ping-pong kernel)")
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
lib.hirest_gemm_select_kernel(a.gemm_kernel)
lib.hirest_attention_x3_select_waves(a.attn_waves)
model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
model.init_random_(seed=1234)
model.set_precision(a.precision)
if a.chunk:
    model.visual.max_frames_per_call_x3 = model.visual.max_frames_per_call_f32 = model.visual.max_frames_per_call = a.chunk
gen = torch.Generator(device=dev); gen.manual_seed(99)
frames = torch.randn((a.frames, 3, 224, 224), device=dev, generator=gen).to(torch.bfloat16)
model.encode_image(frames[:min(a.frames, 256)]); model.encode_image(frames)
torch.cuda.synchronize()
lib.hirest_profile_enable(1)
t0 = time.perf_counter()
for _ in range(a.steps):
    out = model.encode_image(frames)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
recs = (_lib.ProfRecord * 100000)()
n = lib.hirest_profile_collect(recs, len(recs))
lib.hirest_profile_enable(0)
groups = {}
for i in range(max(n, 0)):
    r = recs[i]
    g = groups.setdefault((r.kind, r.tag, r.d0, r.d1, r.d2), [0, 0.0]); g[0] += 1; g[1] += r.ms
rows = []
for (kind, tag, d0, d1, d2), (cnt, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    e = {"kind": {0: "gemm", 1: "attention", 2: "rows"}.get(kind, kind), "tag": tag, "dims": [d0, d1, d2], "launches": cnt, "avg_ms": ms / cnt,
         "ms_per_step": ms / a.steps}
    if kind == 0:
        real_k = d2 // 2 if a.precision == "bf16x3" else d2
        e["algorithmic_tflops"] = 2.0 * d0 * d1 * real_k / (ms / cnt * 1e-3) / 1e12
        e["mfma_tflops"] = e["algorithmic_tflops"] * (3 if a.precision == "bf16x3" else 1)
    rows.append(e)
print(json.dumps({"precision": a.precision, "frames": a.frames, "frames_per_s": a.frames / dt, "ms_per_call": dt * 1e3,
                  "finite": bool(torch.isfinite(out).all()), "kernels": rows}, indent=1))
