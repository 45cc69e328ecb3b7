#!/usr/bin/env python3
"""Where a 32-key tile of the split-operand attention (attention_x3.hip) spends its time: workgroup 0 stamps the shader clock at the phase
boundaries of every tile (hirest_attention_x3_debug_trace); prints mean cycles per phase and wave over tiles 1..7 of a 257-token head."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import ops, _lib  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--waves", type=int, default=0)
ap.add_argument("--frames", type=int, default=1024)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, T, H, dh = a.frames, 257, 16, 88
D = H * dh
g = torch.Generator(device=dev); g.manual_seed(0)
qkv = torch.randn((B * T, 3 * D), device=dev, generator=g)
out = torch.empty((B * T, D), device=dev)
lib = _lib.load()
lib.hirest_attention_x3_select_waves(a.waves)
trace = torch.zeros((16, 16, 8), dtype=torch.int64, device=dev)


def run():
    _lib.check(lib.hirest_attention_x3_qkv(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, out.data_ptr(), B, T, T, H, dh,
                                           dh ** -0.5, ops.stream_ptr()), "attention_x3")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
plain = e0.elapsed_time(e1)
lib.hirest_attention_x3_debug_trace(trace.data_ptr())
run(); torch.cuda.synchronize()
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.hirest_attention_x3_debug_trace(None)
t = trace.cpu().double().numpy()
nw = int((t[1, :, 0] > 0).sum())
print(f"waves per workgroup {nw}; launch {plain:.3f} ms ({e0.elapsed_time(e1):.3f} ms with the stamps); workgroup 0, tiles 1..7, cycles")
names = ["wait at barrier (0->1)", "store tile t+1 to LDS, fetch t+2 (1->2)", "S^T: 18 MFMAs + fragment reads (2->3)", "softmax (3->4)", "P.V: 18 MFMAs + reads (4->5)",
         "tile (0->0')"]
print(f"{'phase':46s}" + "".join(f"  wave{w:2d}" for w in range(nw)))
for i, name in enumerate(names[:5]):
    print(f"{name:46s}" + "".join(f"{(t[1:8, w, i + 1] - t[1:8, w, i]).mean():8.0f}" for w in range(nw)))
    if i == 1:   # the staging step in three parts: loads of tile t+1 land (1->6), split + LDS stores (6->7), fetch of t+2 issued (7->2)
        for sub, (a0, a1) in (("   of which: waiting for the tile's loads", (1, 6)), ("             split + LDS stores", (6, 7)), ("             issuing the next fetch", (7, 2))):
            print(f"{sub:46s}" + "".join(f"{(t[1:8, w, a1] - t[1:8, w, a0]).mean():8.0f}" for w in range(nw)))
print(f"{names[5]:46s}" + "".join(f"{(t[2:9, w, 0] - t[1:8, w, 0]).mean():8.0f}" for w in range(nw)))
