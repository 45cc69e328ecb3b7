#!/usr/bin/env python3
"""The 64 x 64 fp32 GEMM by operand path at the joint model's / fp32 towers' shapes: register prefetch (two slabs ahead; with its split-K form
where hirest_gemm_f32_ws takes it) against the LDS-DMA ring (4 slots, two blocks per CU).  Bits must agree."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops
from hirest_amd.moment_model import MomentModel
lib = _lib.load(); dev = torch.device("cuda:0")
def timeit(f, reps=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in ((1500, 768, 768), (1500, 2304, 768), (1500, 3072, 768), (1500, 768, 3072), (1500, 512, 1024), (9600, 768, 768), (9600, 3072, 768),
                  (9600, 768, 3072), (65792, 1408, 1408), (65792, 6144, 1408), (65792, 1408, 6144), (3000, 384, 384), (3000, 1536, 384)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
    res, ref = [], None
    for mode, name in ((1, "registers"), (2, "ring"), (0, "auto")):
        lib.hirest_gemm_f32_ring_mode(mode)
        f = lambda: MomentModel._gemm(a, w, b, out=out, act=1)          # hirest_gemm_f32_ws: the split form where it applies (mode 1 / large K)
        t = timeit(f)
        if ref is None: ref = out.clone()
        res.append(f"{name} {t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.1f} TF){'' if torch.equal(out, ref) else ' BITS DIFFER'}")
    lib.hirest_gemm_f32_ring_mode(0)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    print(f"M {M:6d} N {N:5d} K {K:5d} ({tiles / 256:5.2f} tiles/CU): " + "   ".join(res), flush=True)
