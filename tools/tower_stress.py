#!/usr/bin/env python3
"""Race screen for the whole vision tower (folded-LayerNorm path incl. the LDS-DMA statistics prefetch): the same 1024-frame
batch is encoded repeatedly; every repeat must reproduce the first output bit for bit."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd  # noqa: E402
from hirest_amd import synth  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--repeats", type=int, default=20)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--precision", default="bf16", help="bf16 | bf16x3 | fp32 (the tower precision to screen)")
a = ap.parse_args()
dev = torch.device("cuda:0")
model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
model.init_random_(seed=7)
model.set_precision(a.precision)
g = torch.Generator(device=dev); g.manual_seed(11)
frames = torch.randn((a.frames, 3, 224, 224), device=dev, generator=g).to(torch.bfloat16 if a.precision == "bf16" else torch.float32)
ref = model.encode_image(frames)
bad = 0
for r in range(a.repeats):
    out = model.encode_image(frames)
    if not torch.equal(out, ref):
        bad += 1
        print(f"repeat {r}: {int((out != ref).any(dim=1).sum())} rows differ, max |diff| {(out - ref).abs().max().item():.3e}", flush=True)
print("RESULT:", a.precision, "clean" if bad == 0 else f"{bad} of {a.repeats} repeats differ", f"(finite: {bool(torch.isfinite(ref).all())})")
sys.exit(1 if bad else 0)
