#!/usr/bin/env python3
"""Does a tower call's result depend on what ran before it?  The first 4 call-blocks (32 videos x 32 frames each) of the c3 corpus, encoded
in order several times; every repeat must equal the first."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd
from hirest_amd import retrieval, synth
dev = torch.device("cuda:0")
V, F = 128, 32
model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
model.init_random_(seed=1234)
ids = synth.c3_device_names(V)
src = retrieval.FrameSource(ids, lambda lo, hi: synth.c3_device_block(lo, hi, F, dev), videos_per_call=32)
ref = None
for r in range(int(os.environ.get("REPS", "6"))):
    rows = retrieval.corpus_block_rows(model, src, 0, 1, F)
    fe = [model.encode_image(synth.c3_device_block(lo, lo + 32, F, dev).reshape(1024, 3, 224, 224)).float() for lo in range(0, V, 32)]
    fe = torch.cat(fe)
    torch.cuda.synchronize()
    if ref is None:
        ref = (rows, fe)
    else:
        dr = (rows != ref[0]).any(dim=1).nonzero().flatten().tolist()
        df = (fe != ref[1]).any(dim=1).nonzero().flatten().tolist()
        print(f"repeat {r}: pooled rows differing {dr[:16]} ({len(dr)}); frame embeddings differing {df[:16]} ({len(df)})"
              + (f" max |d| {(fe - ref[1]).abs().max().item():.3e}" if df else ""), flush=True)
print("done")
