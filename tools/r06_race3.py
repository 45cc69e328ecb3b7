#!/usr/bin/env python3
"""Does a tower call read memory nobody wrote?  The same 2 x 1024-frame sweep with every fresh allocation (workspace, outputs) carved out of
memory pre-filled with 0x00 / 0xFF / 0x7F bytes must give identical rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd
from hirest_amd import retrieval, synth
dev = torch.device("cuda:0")
V, F = 64, 32
ref = None
for poison in (0x00, 0xFF, 0x7F, 0xFF):
    torch.cuda.empty_cache()
    junk = torch.full((int(os.environ.get("POISON_GB", "60")) << 30,), poison, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    del junk                                              # stays in the caching allocator: the next allocations are carved out of it
    model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
    model.init_random_(seed=1234)
    ids = synth.c3_device_names(V)
    src = retrieval.FrameSource(ids, lambda lo, hi: synth.c3_device_block(lo, hi, F, dev), videos_per_call=32)
    rows = retrieval.corpus_block_rows(model, src, 0, 1, F).clone()
    small = model.encode_image(synth.c3_device_block(0, 2, F, dev).reshape(64, 3, 224, 224)).float().clone()      # a 64-frame call too
    torch.cuda.synchronize()
    if ref is None:
        ref = (rows, small)
    else:
        dr = (rows != ref[0]).any(dim=1).nonzero().flatten().tolist()
        ds = (small != ref[1]).any(dim=1).nonzero().flatten().tolist()
        print(f"poison 0x{poison:02X}: pooled rows differing {dr[:16]} ({len(dr)}) nan {bool(torch.isnan(rows).any())}; 64-frame call rows differing {ds[:8]} ({len(ds)})", flush=True)
    del model, src, rows, small
print("done")
