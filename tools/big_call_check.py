#!/usr/bin/env python3
"""Tower calls larger than the bench's 1024 frames (activations past 2^32 bytes): a 2048- and a 3072-frame call must reproduce
the rows of 1024-frame calls bit for bit.   python tools/big_call_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd  # noqa: E402

dev = torch.device("cuda:0")
model, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_g_14", pretrained="synth:3", precision="bf16")
model = model.to(dev).eval()
gen = torch.Generator(device=dev); gen.manual_seed(5)
full = torch.randn((3072, 3, 224, 224), device=dev, generator=gen, dtype=torch.bfloat16)
model.visual.max_frames_per_call = 1024
ref = model.encode_image(full)
for n in (2048, 3072):
    model.visual.max_frames_per_call = n
    out = model.encode_image(full[:n])
    print(n, "frames in one call: finite", bool(torch.isfinite(out).all()), " equal to 1024-frame calls:", bool(torch.equal(out, ref[:n])),
          " max |diff|", (out - ref[:n]).abs().max().item(), flush=True)
