#!/usr/bin/env python3
"""The training step's fp32 products by operand layout (hirest_gemm_f32_layouts): the same logical [M, N, K] problem with row-major
operands (the forward's form), a k-major W (dX = dY W) and both k-major (dW = dY^T X) — what the k-major staging costs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
def run(M, N, K, ak, wk, reps=40):
    a = torch.randn((K, M) if ak else (M, K), device=dev); w = torch.randn((K, N) if wk else (N, K), device=dev); out = torch.empty(M, N, device=dev)
    nb = lib.hirest_gemm_f32_layouts_workspace_bytes(M, N, K)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    f = lambda: _lib.check(lib.hirest_gemm_f32_layouts(a.data_ptr(), M if ak else K, int(ak), w.data_ptr(), N if wk else K, int(wk), None, None, 0, out.data_ptr(), N,
                                                       M, N, K, 0, ws.data_ptr() if nb else None, nb, ops.stream_ptr()), "layouts")
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, (M, N, K) in (("dW so/cq/co", (768, 768, 1500)), ("dW qkv", (2304, 768, 1500)), ("dW ff1", (3072, 768, 1500)), ("dW ff2", (768, 3072, 1500)),
                        ("dX so", (1500, 768, 768)), ("dX qkv", (1500, 768, 2304)), ("dX ff1", (1500, 768, 3072)), ("dX ff2", (1500, 3072, 768))):
    ts = [run(M, N, K, ak, wk) for ak, wk in ((0, 0), (0, 1), (1, 1))]
    print(f"{name:12s} M {M:5d} N {N:5d} K {K:5d}: row-major {ts[0]:6.1f} us ({2.0 * M * N * K / ts[0] / 1e6:5.1f} TF)   W k-major {ts[1]:6.1f} us   both k-major {ts[2]:6.1f} us", flush=True)
