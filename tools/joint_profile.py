#!/usr/bin/env python3
"""Moment segmentation (20 iterations) at B = 5, T = 300 a few times: run under rocprofv3 --kernel-trace --stats to see where an iteration goes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
if os.environ.get("JOINT_PRECISION"):
    model.set_precision(os.environ["JOINT_PRECISION"])
B, T = int(os.environ.get("JOINT_B", "5")), 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
task = sys.argv[1] if len(sys.argv) > 1 else "moment_segmentation"
batch = {"tasks": [task], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "asr_feats": asr.to(dev), "text_feat": text.to(dev),
         "moment_bound_frames": bounds, "moment_mask": moment_mask.to(dev)}
model.test_step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    model.test_step(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"{task} B={B} T={T}: {dt * 1e3:.2f} ms per batch = {B / dt:.1f} videos/s")
