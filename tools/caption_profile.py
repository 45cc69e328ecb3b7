#!/usr/bin/env python3
"""Where the step-captioning time goes: wall time per batch vs the sum of kernel durations (run under rocprofv3 --stats)."""
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
B, T = int(os.environ.get("CAPTION_B", "5")), 300            # CAPTION_B=32: the reference's default evaluation batch (args.py:27)
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
mm15 = torch.zeros_like(moment_mask); mm15[:, 10:25] = 1
batch = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": mm15,
         "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
beams = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if os.environ.get("HIREST_ROWS_LN_MODE"):
    from hirest_amd import _lib
    _lib.check(_lib.load().hirest_gemm_f32_rows_ln_mode(int(os.environ["HIREST_ROWS_LN_MODE"])), "mode")
model.test_step(batch, num_beams=beams)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = int(os.environ.get("CAPTION_REPS", "10"))
for _ in range(reps):
    model.test_step(batch, num_beams=beams)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"captioning B={B} beams={beams}: {dt * 1e3:.1f} ms per batch = {B / dt:.1f} captions/s")
