"""Host side of the joint-model training step: enqueue time per phase with the GPU idle at the start (is the step host- or GPU-bound?),
wall time per step, and cProfile tables of the main thread and of MomentLoss.backward (which runs on autograd's thread)."""
import json, os, sys, time, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
import hirest_amd
from hirest_amd import synth
from hirest_amd.synth import joint_inputs, train_targets
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
sd = synth.joint_state_dict(shapes, 31)
dev = torch.device("cuda:0")
model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
model.load_state_dict(sd, strict=False); model = model.to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-5)
B, T = 5, 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"tb.{T}", B, T, 61)
st, et, seg, prev = train_targets(f"tb.{T}", B, T, 61, bounds)
pin = lambda t: t.pin_memory()                       # as DataLoader(pin_memory=True) delivers them (hirest_dataset.py:614,624)
batch = dict(vis_feats=pin(vis), vis_mask=pin(vis_mask), asr_feats=pin(asr), text_feat=pin(text), tasks=["moment_retrieval"],
             moment_mask=pin(moment_mask), moment_retrieval_start_target=pin(st), moment_retrieval_end_target=pin(et))
import cProfile, pstats
def step(parts=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model.train_step(batch)["loss"]
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if parts is not None:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): parts[i] += d
for _ in range(5): step()
torch.cuda.synchronize()
# host time with the GPU never the bottleneck: synchronise before every step, time only the enqueue
parts = [0.0] * 4; n = 20
for _ in range(n):
    torch.cuda.synchronize(); step(parts)
print("host enqueue per step (GPU idle at start): forward %.2f backward %.2f clip %.2f adamw %.2f = %.2f ms" % (*[p / n * 1e3 for p in parts], sum(parts) / n * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); print("wall per step %.2f ms" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(n): step()
torch.cuda.synchronize(); pr.disable()
print("---- main thread (forward, clip, optimizer; the backward runs on autograd's thread)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
from hirest_amd import train
inner = train.MomentLoss.backward
prb = cProfile.Profile()
def profiled(ctx, gloss):
    prb.enable()
    try:
        return inner(ctx, gloss)
    finally:
        prb.disable()
train.MomentLoss.backward = staticmethod(profiled)
for _ in range(n): step()
torch.cuda.synchronize()
print("---- MomentLoss.backward")
pstats.Stats(prb).sort_stats("tottime").print_stats(22)
