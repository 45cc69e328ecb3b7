#!/usr/bin/env python3
"""Joint-model training step timing: train_step -> backward -> clip_grad_norm_ -> AdamW at B = 5 (args.py default
batch) and the C4 frame counts, per task, next to the fp32 CPU oracle under torch autograd (the same loss restated in
oracle/ref_cpu.py, 32 threads).   python tools/train_bench.py [--frames 120 300 571]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hirest_amd  # noqa: E402
from hirest_amd import synth  # noqa: E402
from hirest_amd.synth import joint_inputs, train_targets, caption_targets
from oracle import ref_cpu as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="*", default=[120, 300, 571])
    ap.add_argument("--batch", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--fused", action="store_true", help="also time the loop with torch.optim.AdamW(fused=True)")
    ap.add_argument("--tasks", nargs="*", default=None, help="subset of moment_retrieval moment_segmentation step_captioning")
    ap.add_argument("--pageable", action="store_true", help="leave the batch tensors in pageable memory (default: pinned, as DataLoader(pin_memory=True))")
    a = ap.parse_args()
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
    sd = synth.joint_state_dict(shapes, 31)
    dev = torch.device("cuda:0")
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).train()
    # trainer_base.py:56 builds torch.optim.AdamW with its defaults (the for-each implementation: ~6 passes over the 63 M parameters);
    # --fused adds a second line with the same class' fused=True option, which is the caller's choice, not a kernel of this repo
    params = [p for p in model.parameters() if p.requires_grad]
    opts = {"AdamW": torch.optim.AdamW(params, lr=1e-5)}
    if a.fused:
        opts["AdamW(fused=True)"] = torch.optim.AdamW(params, lr=1e-5, fused=True)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    B = a.batch
    for T in a.frames:
        vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"tb.{T}", B, T, 61)
        st, et, seg, prev = train_targets(f"tb.{T}", B, T, 61, bounds)
        cap_mask = torch.zeros(B, T, dtype=torch.long)
        for b_ in range(B):
            cap_mask[b_, 10 + b_:10 + b_ + 15 + 3 * b_] = 1
        common = {"vis_feats": vis, "vis_mask": vis_mask, "asr_feats": asr, "text_feat": text}
        batches = {
            "moment_retrieval": dict(common, tasks=["moment_retrieval"], moment_mask=moment_mask, moment_retrieval_start_target=st,
                                     moment_retrieval_end_target=et),
            "moment_segmentation": dict(common, tasks=["moment_segmentation"], moment_mask=moment_mask, prev_boundary_mask=prev,
                                        moment_segmentation_target=seg),
            "step_captioning": dict(common, tasks=["step_captioning"], moment_mask=cap_mask,
                                    target_text=caption_targets(f"tb.{T}", B, 48, 61)),
        }
        if a.tasks:
            batches = {k: v for k, v in batches.items() if k in a.tasks}
        if not a.pageable:                               # the reference's loaders deliver pinned batches (hirest_dataset.py:614,624)
            batches = {k: {n: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for n, v in b.items()} for k, b in batches.items()}
        for oname, opt in opts.items():
            line = f"T={T:4d} B={B} {oname}:"
            for task, batch in batches.items():
                def step():
                    opt.zero_grad(set_to_none=True)
                    loss = model.train_step(batch)["loss"]
                    loss.backward()
                    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                    opt.step()
                    return loss
                step(); step(); torch.cuda.synchronize()
                dts = []
                for _ in range(3):                                # best of three groups of `reps` steps (one-off allocator growth is not the step)
                    t0 = time.perf_counter()
                    for _ in range(a.reps):
                        step()
                    torch.cuda.synchronize()
                    dts.append((time.perf_counter() - t0) / a.reps)
                dt = min(dts)
                line += f"  {task} {dt * 1e3:6.1f} ms/step ({B / dt:6.0f} videos/s)"
            print(line, flush=True)
        if T <= 300:   # CPU oracle under autograd, retrieval loss only (the other two scale alike)
            psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
            t0 = time.perf_counter()
            loss = O.moment_retrieval_loss(psd, vis, text, asr, vis_mask, moment_mask, st, et)
            loss.backward()
            dt = time.perf_counter() - t0
            print(f"        CPU oracle + torch autograd ({torch.get_num_threads()} threads), moment_retrieval forward + backward: "
                  f"{dt * 1e3:.0f} ms ({B / dt:.1f} videos/s)", flush=True)


if __name__ == "__main__":
    main()
