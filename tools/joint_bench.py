#!/usr/bin/env python3
"""Joint-model (MomentModel) timing at the C4 sizes of SURVEY 8d: B=5, T in {120, 300, 571, 1855}: videos/s for
moment retrieval and the 20-iteration moment segmentation on the GPU, next to the fp32 CPU oracle."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd  # noqa: E402
from hirest_amd import synth  # noqa: E402
from hirest_amd.synth import joint_inputs
from oracle import ref_cpu as O  # noqa: E402

shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
torch.set_num_threads(min(os.cpu_count(), 32))
for T in (120, 300, 571, 1855):
    B = 5
    vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
    bmr = {"tasks": ["moment_retrieval"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": moment_mask.to(dev),
           "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
    bsg = {"tasks": ["moment_segmentation"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "asr_feats": asr.to(dev),
           "text_feat": text.to(dev), "moment_bound_frames": bounds}
    out = {}
    mm15 = torch.zeros_like(moment_mask); mm15[:, 10:25] = 1        # 15-frame moments -> 20 trimmed frames
    bcp = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": mm15,
           "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
    for name, batch, reps in (("retrieval", bmr, 20), ("segmentation", bsg, 5), ("captioning", bcp, 2)):
        pred = model.test_step(batch)["prediction"]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            model.test_step(batch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        out[name] = (B / dt, pred)
    line = (f"T={T:5d}  GPU retrieval {out['retrieval'][0]:8.1f} videos/s   segmentation {out['segmentation'][0]:7.1f} videos/s"
            f"   captioning(beam 5, 48 words) {out['captioning'][0]:6.1f} captions/s")
    if T <= 571:
        t0 = time.perf_counter(); p_cpu, _, _ = O.moment_retrieval(sd, vis, text, asr, vis_mask, moment_mask); t_mr = time.perf_counter() - t0
        t0 = time.perf_counter(); s_cpu, _ = O.moment_segmentation(sd, vis, text, asr, vis_mask, bounds); t_sg = time.perf_counter() - t0
        line += f"   | CPU oracle {B / t_mr:6.1f} / {B / t_sg:5.2f} videos/s   exact: {p_cpu == out['retrieval'][1]} / {s_cpu == out['segmentation'][1]}"
    print(line, flush=True)
