#!/usr/bin/env python3
"""BASELINE configs[2] at full size on ONE GPU: V = 4096 videos x 32 frames (131 072 frames) of the SURVEY 8d corpus
(video v's frames = base_v + 0.1 * noise_f), 546 real-prompt queries, EVA-CLIP-g/14 with synthetic weights.

Pass 1 encodes the corpus in one sweep; pass 2 re-encodes it as 8 rank-sized blocks (shard_range of the 8-GPU run, each
block encoded separately and concatenated in rank order = what the all-gather assembles) and checks that pooled rows,
scores and top-10 lists are bit-identical — the 1-GPU == 8-GPU self-consistency of SURVEY 8d without an 8-GPU node.
Prints one JSON object (throughput, invariance verdict, score statistics)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd  # noqa: E402
from hirest_amd import retrieval, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=4096)
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--block", type=int, default=32, help="videos per encode call (32 x 32 = one 1024-frame tower call)")
a = ap.parse_args()
dev = torch.device("cuda:0")
model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
model.init_random_(seed=1234)
model.visual.max_frames_per_call = 1024
prompts = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "test_prompts.json")))
tokens = hirest_amd.tokenize(prompts).to(dev)
texts = retrieval.encode_texts(model, tokens)


def video_block(lo, hi):
    """frames of videos [lo, hi): seeded per video, so any partition of the corpus regenerates identical inputs"""
    out = torch.empty((hi - lo, a.frames, 3, 224, 224), device=dev, dtype=torch.bfloat16)
    for i, v in enumerate(range(lo, hi)):
        g = torch.Generator(device=dev); g.manual_seed(100000 + v)
        base = torch.randn((1, 3, 224, 224), device=dev, generator=g)
        out[i] = (base + 0.1 * torch.randn((a.frames, 3, 224, 224), device=dev, generator=g)).to(torch.bfloat16)
    return out


def encode_range(lo, hi):
    rows = []
    for s in range(lo, hi, a.block):
        rows.append(retrieval.encode_videos(model, video_block(s, min(hi, s + a.block))))
    return torch.cat(rows) if rows else torch.empty((0, texts.shape[1]), device=dev)


torch.cuda.synchronize(); t0 = time.perf_counter()
pooled = encode_range(0, a.videos)
torch.cuda.synchronize(); t1 = time.perf_counter()
scores, val, idx = retrieval.retrieve(texts, pooled, 10)
parts = []
for r in range(a.ranks):
    lo, hi, per = retrieval.shard_range(a.videos, r, a.ranks)
    parts.append(encode_range(lo, hi))
sharded = torch.cat(parts)
scores2, val2, idx2 = retrieval.retrieve(texts, sharded, 10)
torch.cuda.synchronize(); t2 = time.perf_counter()
top2 = scores.topk(2, dim=1).values
print(json.dumps({
    "workload": f"{a.videos} videos x {a.frames} frames, {len(prompts)} queries, EVA-CLIP-g/14 bf16, 1 GPU",
    "frames": a.videos * a.frames, "seconds_single_sweep_incl_input_generation": t1 - t0,
    "frames_per_s_incl_input_generation": a.videos * a.frames / (t1 - t0),
    "rank_blocks": a.ranks, "pooled_rows_bit_identical": bool(torch.equal(pooled, sharded)),
    "scores_bit_identical": bool(torch.equal(scores, scores2)), "top10_identical": bool(torch.equal(idx, idx2)),
    "pooled_row_norm_min_max": [pooled.norm(dim=1).min().item(), pooled.norm(dim=1).max().item()],
    "score_min_max": [scores.min().item(), scores.max().item()],
    "median_top1_margin": (top2[:, 0] - top2[:, 1]).median().item(),
    "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9}))
