#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the EVA-CLIP-g shape (B frames x 16 heads, 257 tokens, dh 88)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import ops  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, nargs="*", default=[2, 3])
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dbg", type=int, nargs="*", default=[0], help="hirest_attention_debug_mode bit sets to time (v3 only)")
ap.add_argument("--skew", type=int, nargs="*", default=[0], help="hirest_attention_set_skew values to time")
ap.add_argument("--pace", type=int, nargs="*", default=[0], help="hirest_attention_set_pace values to time (producer-wave kernel)")
ap.add_argument("--stagger", type=int, nargs="*", default=[0], help="hirest_attention_set_stagger values to time")
ap.add_argument("--map", type=int, nargs="*", default=[1], help="hirest_attention_set_mapping values to time (1 = one head per workgroup)")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N, H, dh = a.frames, 257, 16, 88
g = torch.Generator(device=dev); g.manual_seed(0)
qkv = torch.randn((B * N, 3 * H * dh), device=dev, generator=g).to(torch.bfloat16)
out = torch.empty((B * N, H * dh), device=dev, dtype=torch.bfloat16)
from hirest_amd import _lib  # noqa: E402
for v, dbg, skew, mp, pc, sg in [(v, d, k, m, pc, sg) for v in a.variants for d in a.dbg for k in a.skew for m in a.map for pc in (a.pace if v >= 6 else a.pace[:1]) for sg in a.stagger]:
    _lib.load().hirest_attention_set_stagger(sg)
    _lib.load().hirest_attention_set_pace(pc)
    _lib.load().hirest_attention_set_mapping(mp)
    _lib.load().hirest_attention_debug_mode(dbg)
    _lib.load().hirest_attention_set_skew(skew)
    ops.attention_select_kernel(v)
    for _ in range(15):                                   # clocks / power state settle (the first timing in a process reads 10 % high)
        ops.attention(qkv, out, B, N, H, dh, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.attention(qkv, out, B, N, H, dh, False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"attention variant {v} dbg {dbg} skew {skew} map {mp} pace {pc} stagger {sg}: {ms:.3f} ms  {4.0 * B * H * N * N * dh / ms / 1e9:.0f} TFLOP/s (algorithmic)", flush=True)
