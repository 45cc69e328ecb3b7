#!/usr/bin/env python3
"""Round 6: where the host time of train_step's FORWARD goes (cProfile by internal time), B = 5, T = 300, moment retrieval."""
import cProfile, json, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs, train_targets
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(synth.joint_state_dict(shapes, 31), strict=False); model = model.to(dev).train()
B, T = 5, 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"tb.{T}", B, T, 61)
st, et, seg, prev = train_targets(f"tb.{T}", B, T, 61, bounds)
pin = lambda t: t.pin_memory()
batch = dict(vis_feats=pin(vis), vis_mask=pin(vis_mask), asr_feats=pin(asr), text_feat=pin(text), tasks=["moment_retrieval"],
             moment_mask=pin(moment_mask), moment_retrieval_start_target=pin(st), moment_retrieval_end_target=pin(et))
for _ in range(5):
    model.train_step(batch)["loss"].backward()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    torch.cuda.synchronize(); model.train_step(batch)
print("forward enqueue %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    torch.cuda.synchronize(); model.train_step(batch)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
