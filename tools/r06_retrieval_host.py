#!/usr/bin/env python3
"""Round 6: host vs GPU share of a B = 5, T = 300 moment-retrieval test_step (cProfile by cumulative time; the '.cpu()' line is the wait for the GPU)."""
import cProfile, json, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "joint_schema.json"))).items()}
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(synth.joint_state_dict(shapes, 31), strict=False)
model = model.to(dev).eval()
B, T = int(os.environ.get("JOINT_B", "5")), 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
task = sys.argv[1] if len(sys.argv) > 1 else "moment_retrieval"
batch = {"tasks": [task], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": moment_mask.to(dev), "asr_feats": asr.to(dev),
         "text_feat": text.to(dev), "moment_bound_frames": bounds}
for _ in range(5):
    model.test_step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 100
for _ in range(n):
    model.test_step(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{task} B={B} precision={model.precision}: {dt * 1e3:.3f} ms per batch = {B / dt:.0f} videos/s")
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    model.test_step(batch)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
