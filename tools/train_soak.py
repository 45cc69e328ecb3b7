#!/usr/bin/env python3
"""Soak / race screen of the joint-model training loop: `steps` optimizer steps cycling through the three tasks (train mode, dropout
on, pinned batches, the reference's optimizer calls), once with the weight-gradient GEMMs on the side stream and once on one stream,
from the same seeds.  Every loss must be finite and the two trajectories must stay together (1e-3): they are not bit-identical —
the loss scalar is an atomic sum over blocks and captioning's tied embedding gradient an atomic scatter-add — while the gradients of
a single step are (tests/test_gpu_train.py).   python tools/train_soak.py [steps]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hirest_amd
from hirest_amd import synth, train
from hirest_amd.synth import joint_inputs, train_targets, caption_targets

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 90
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
dev = torch.device("cuda:0")
B, T = 5, 300
vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"tb.{T}", B, T, 61)
st, et, seg, prev = train_targets(f"tb.{T}", B, T, 61, bounds)
cap_mask = torch.zeros(B, T, dtype=torch.long)
for b_ in range(B):
    cap_mask[b_, 10 + b_:10 + b_ + 15 + 3 * b_] = 1
pin = lambda t: t.pin_memory()
common = {"vis_feats": pin(vis), "vis_mask": pin(vis_mask), "asr_feats": pin(asr), "text_feat": pin(text)}
batches = [dict(common, tasks=["moment_retrieval"], moment_mask=pin(moment_mask), moment_retrieval_start_target=pin(st), moment_retrieval_end_target=pin(et)),
           dict(common, tasks=["moment_segmentation"], moment_mask=pin(moment_mask), prev_boundary_mask=pin(prev), moment_segmentation_target=pin(seg)),
           dict(common, tasks=["step_captioning"], moment_mask=cap_mask, target_text=caption_targets(f"tb.{T}", B, 48, 61))]

def run(side):
    train.SIDE_STREAM_DW = side
    torch.manual_seed(3)
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    model.load_state_dict(synth.joint_state_dict(shapes, 31), strict=False)
    model = model.to(dev).train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    losses = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = model.train_step(batches[i % 3])["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(loss.detach())
    return torch.stack(losses).cpu()

a, b = run(True), run(False)
train.SIDE_STREAM_DW = True
assert torch.isfinite(a).all() and torch.isfinite(b).all()
print(f"{steps} steps x 2 runs: losses finite; max |side stream - single stream| over all steps {(a - b).abs().max().item():.3e}; "
      f"retrieval loss {a[0].item():.4f} -> {a[3 * ((steps - 1) // 3)].item():.4f}, segmentation {a[1].item():.4f} -> "
      f"{a[3 * ((steps - 2) // 3) + 1].item():.4f}, captioning {a[2].item():.4f} -> {a[3 * ((steps - 3) // 3) + 2].item():.4f}")
assert (a - b).abs().max().item() < 1e-3
