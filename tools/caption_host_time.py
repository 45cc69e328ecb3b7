"""Host time spent inside the two C calls of a step-captioning word (are the ~20 launches per word host- or GPU-bound?)."""
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
model.test_step(batch, num_beams=5)
lib = _lib.load()
acc = {}
class W:
    def __init__(self, name, f): self.name, self.f = name, f
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.f(*a); acc[self.name] = acc.get(self.name, 0.0) + time.perf_counter() - t0; return r
for n in ("hirest_caption_decode_logits", "hirest_caption_beam_tail"):
    setattr(lib, n, W(n, getattr(lib, n)))
torch.cuda.synchronize(); t0 = time.perf_counter()
model.test_step(batch, num_beams=5)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"batch {dt*1e3:.2f} ms; host time inside the C calls per word: " + ", ".join(f"{k} {v/48*1e6:.1f} us" for k, v in acc.items()))
