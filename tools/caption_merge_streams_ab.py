import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
B, T = 5, 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"jb.{T}", B, T, 43)
mm15 = torch.zeros_like(moment_mask); mm15[:, 10:25] = 1
batch = {"tasks": ["step_captioning"], "vis_feats": vis.to(dev), "vis_mask": vis_mask.to(dev), "moment_mask": mm15, "asr_feats": asr.to(dev), "text_feat": text.to(dev)}
for nb in (12, 24):
    many = [batch] * nb
    for beams in (5, 3):
        for streams, graphs in ((1, True), (2, True), (1, False), (2, False)):
            model.caption_batches(many, num_beams=beams, streams=streams, graphs=graphs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2): model.caption_batches(many, num_beams=beams, streams=streams, graphs=graphs)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
            print(f"{nb} batches of 5, beam {beams}, streams {streams}, graphs {graphs}: {nb * B / dt:7.1f} captions/s", flush=True)
