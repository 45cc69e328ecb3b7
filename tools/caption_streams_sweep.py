#!/usr/bin/env python3
"""captions/s of MomentModel.caption_batches against the number of batches in flight (and with / without hipGraph replay), on the
configs[4] operating point (B = 5, 15-frame moments, beam 5 / 3, 48 words; the c5 / c3 golden inputs)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs, CAPTION_CASES
dev = torch.device("cuda:0")
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
sd["clip4cap_model.decoder.classifier.cls.predictions.bias"][102] += 1.5
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
for case in ("c5", "c3"):
    B, T, beams, lens = CAPTION_CASES[case]
    vis, asr, text, vm, _, _ = joint_inputs(f"cap.{case}", B, T, 47)
    mm = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        mm[b, 5 + b:5 + b + lens[b]] = 1
    batch = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vm.to(dev), "asr_feats": asr.to(dev), "text_feat": text.to(dev),
             "moment_mask": mm}
    nb = 12
    for graphs in (True, False):
        for ns in (1, 2, 3, 4, 6):
            model.caption_batches([batch] * max(ns, 2), num_beams=beams, streams=ns, graphs=graphs, merge=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.caption_batches([batch] * nb, num_beams=beams, streams=ns, graphs=graphs, merge=False)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / nb
            print(f"beam {beams}  graphs {int(graphs)}  batches in flight {ns}:  {dt * 1e3:6.2f} ms per batch  {B / dt:7.1f} captions/s", flush=True)
