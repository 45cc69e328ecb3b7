import json, os, sys, time, types
import torch
sys.path.insert(0, "/root/repo")
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs
shapes = {k: tuple(v) for k, v in json.load(open("/root/repo/tests/golden/joint_schema.json")).items()}
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(synth.joint_state_dict(shapes, 31), strict=False)
model = model.to(dev).eval()
B, T = 32, 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
mm15 = torch.zeros_like(moment_mask); mm15[:, 10:25] = 1
batch = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": mm15, "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
for W in (48, 24, 12, 2):
    model.args = types.SimpleNamespace(max_words=W, max_frames_step_captioning=20)
    for _ in range(3): model.test_step(batch, num_beams=3, return_ids=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): model.test_step(batch, num_beams=3, return_ids=True)
    torch.cuda.synchronize(); print(W, "words:", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms per batch")
