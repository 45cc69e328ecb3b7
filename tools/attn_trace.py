#!/usr/bin/env python3
"""Where a head step of the persistent attention kernel spends its time: workgroup 0 of the debug instantiation stamps the shader
clock at the phase boundaries of every step (hirest_attention_debug_mode bit 8); this prints, per wave, the mean cycles of each
phase over the steps, and the step length.  Timing tool: the stamps (s_memtime + lgkmcnt(0)) perturb the kernel by a few per cent."""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import ops, _lib  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=5)
ap.add_argument("--map", type=int, default=0)
ap.add_argument("--skew", type=int, default=12)
ap.add_argument("--pace", type=int, default=3)
ap.add_argument("--extra-dbg", type=int, default=0, help="knock-out bits to combine with the trace bit")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N, H, dh = 1024, 257, 16, 88
g = torch.Generator(device=dev); g.manual_seed(0)
qkv = torch.randn((B * N, 3 * H * dh), device=dev, generator=g).to(torch.bfloat16)
out = torch.empty((B * N, H * dh), device=dev, dtype=torch.bfloat16)
lib = _lib.load()
lib.hirest_attention_set_pace(a.pace); lib.hirest_attention_set_skew(a.skew); lib.hirest_attention_set_mapping(a.map); ops.attention_select_kernel(a.variant)
lib.hirest_attention_debug_mode(256 | a.extra_dbg)
for _ in range(5):
    ops.attention(qkv, out, B, N, H, dh, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.attention(qkv, out, B, N, H, dh, False); e1.record(); torch.cuda.synchronize()
STEPS, WAVES, SLOTS = 64, 12, 12
buf = np.zeros(STEPS * WAVES * SLOTS, dtype=np.int64)
assert lib.hirest_attention_debug_trace_read(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
lib.hirest_attention_debug_mode(0)
t = buf.reshape(STEPS, WAVES, SLOTS).astype(np.float64)
nsteps = 64 if a.map else 16
nw = 9
print(f"variant {a.variant} map {a.map} skew {a.skew} extra dbg {a.extra_dbg}: launch {e0.elapsed_time(e1):.3f} ms; workgroup 0, steps 1..{nsteps - 1}; cycles (shader clock)")
names = ["wait vmcnt(0)+barrier A (10->0)", "DMA issue (0->1)", "tile1 S^T (1->2)", "tile1 softmax (2->3)", "tile1 wait V / barrier B (3->4)",
         "tile1 P.V + store (4->5)", "tile2 S^T (5->6)", "tile2 softmax (6->7)", "tile2 V check (7->8)", "tile2 P.V + store (8->9)", "step (10->10')"]
pairs = [(10, 0), (0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9)]
print(f"{'phase':36s}" + "".join(f"  wave{w:2d}" for w in range(nw)))
for name, (s0, s1) in zip(names, pairs):
    row = []
    for w in range(nw):
        has2 = w + nw < 17
        if s0 >= 5 and s0 <= 8 and not has2:
            row.append(float("nan")); continue
        end = t[1:nsteps, w, s1] if not (s1 == 9 and False) else None
        start = t[1:nsteps, w, s0]
        if s1 == 5 and not has2:   # one-tile wave: tile 1 ends at stamp 9
            end = t[1:nsteps, w, 9]
        row.append(float(np.mean(end - start)))
    print(f"{name:36s}" + "".join(f"{v:8.0f}" for v in row))
step = [float(np.mean(t[2:nsteps, w, 10] - t[1:nsteps - 1, w, 10])) for w in range(nw)]
print(f"{names[-1]:36s}" + "".join(f"{v:8.0f}" for v in step))
first = t[1:nsteps, :nw, 0].min(axis=1)
print("barrier A release spread over waves (max - min of stamp 0):", float(np.mean(t[1:nsteps, :nw, 0].max(axis=1) - first)))
