#!/usr/bin/env python3
"""Round 6: where a B = 32 captioning test_step spends its time outside the word loop — cProfile of the host side (cumulative), plus wall time
per batch.   CAPTION_B=32 HIREST_JOINT_PRECISION=bf16x3 python tools/r06_caption_host.py 3"""
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
B, T = int(os.environ.get("CAPTION_B", "32")), 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
mm15 = torch.zeros_like(moment_mask); mm15[:, 10:25] = 1
batch = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": mm15,
         "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
beams = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(3):
    model.test_step(batch, num_beams=beams, return_ids=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    model.test_step(batch, num_beams=beams, return_ids=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"captioning B={B} beams={beams} precision={model.precision}: {dt * 1e3:.2f} ms per batch = {B / dt:.1f} captions/s")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    model.test_step(batch, num_beams=beams, return_ids=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
