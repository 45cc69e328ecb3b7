#!/usr/bin/env python3
"""How much of the training step's wall time is the GPU's?  The whole loop iteration (train_step -> backward -> clip_grad_norm_ ->
AdamW) is captured into one hipGraph and replayed: a replay has no host work between kernels, so its time is what the kernels and
their dependencies need.  Measurement only (the replay re-uses one dropout seed and one batch).   python tools/train_graph_probe.py"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hirest_amd  # noqa: E402
from hirest_amd import synth  # noqa: E402
from hirest_amd.synth import joint_inputs, train_targets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--batch", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False)
model = model.to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
B, T = a.batch, a.frames
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"tb.{T}", B, T, 61)
st, et, seg, prev = train_targets(f"tb.{T}", B, T, 61, bounds)
batch = dict(vis_feats=vis, vis_mask=vis_mask, asr_feats=asr, text_feat=text, tasks=["moment_retrieval"], moment_mask=moment_mask,
             moment_retrieval_start_target=st, moment_retrieval_end_target=et)
batch = {n: (v.to(dev) if isinstance(v, torch.Tensor) else v) for n, v in batch.items()}     # resident: no copies in the graph


def make_step(opt):
    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.train_step(batch)["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss
    return step


def wall(step, reps):
    step(); step(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


for name, kw in (("AdamW(fused=True)", dict(fused=True)), ("AdamW(fused=True, capturable=True)", dict(fused=True, capturable=True))):
    opt = torch.optim.AdamW(params, lr=1e-5, **kw)
    print(f"{name}: eager loop {wall(make_step(opt), a.reps):.2f} ms/step", flush=True)
opt = torch.optim.AdamW(params, lr=1e-5, fused=True, capturable=True)
step = make_step(opt)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:400])
    sys.exit(0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.reps)
print(f"hipGraph replay of the whole iteration: {best:.2f} ms/step (loss {float(loss):.4f})")
