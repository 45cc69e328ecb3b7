#!/usr/bin/env python3
"""Step captioning, one batch: time before the beam search (trim, fusion, encoder, cross-attention K / V), inside it (set-up, the
word loop, the read-out) — with a device synchronisation at every boundary, so the parts add up to slightly more than the batch."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hirest_amd
from hirest_amd import synth, _lib
from hirest_amd.synth import joint_inputs
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
B, T = 5, 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
mm15 = torch.zeros_like(moment_mask); mm15[:, 10:25] = 1
batch = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": mm15,
         "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
beams = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(3):
    model.test_step(batch, num_beams=beams)
lib = _lib.load()
acc = {"search": 0.0, "words": 0.0, "calls": 0}
inner = type(model)._beam_search_cached
def timed_search(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = inner(self, *a, **k)
    torch.cuda.synchronize(); acc["search"] += time.perf_counter() - t0
    return r
type(model)._beam_search_cached = timed_search
step = lib.hirest_caption_beam_step
first = [None]
def timed_step(*a):
    if first[0] is None:
        first[0] = time.perf_counter()
    acc["calls"] += 1
    t0 = time.perf_counter(); r = step(*a); acc["words"] += time.perf_counter() - t0
    return r
lib.hirest_caption_beam_step = timed_step
reps = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    model.test_step(batch, num_beams=beams)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"beams={beams}: batch {dt*1e3:.2f} ms (with the extra synchronisations); beam search {acc['search']/reps*1e3:.2f} ms of it, "
      f"{acc['calls']/reps:.0f} words per batch, host time inside the per-word C call {acc['words']/acc['calls']*1e6:.1f} us")
