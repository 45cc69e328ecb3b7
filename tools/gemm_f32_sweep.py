#!/usr/bin/env python3
"""fp32 GEMM (hirest_gemm_f32, no workspace: the plain 64 x 64 kernel for M > 256) over tile counts and depths: time per launch from
HIP events over back-to-back launches, TFLOP/s, and the marginal time per 32-deep slab — what a block costs alone on its CU, with
four per CU, and per launch.     python tools/gemm_f32_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
def run(M, N, K, reps=50):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    f = lambda: lib.hirest_gemm_f32(a.data_ptr(), K, w.data_ptr(), K, None, None, 0, None, 0, out.data_ptr(), N, M, N, K, 0, ops.stream_ptr())
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N) in ((512, 2048), (1024, 2048), (1024, 4096), (1024, 8192), (1500, 768), (1500, 2304), (1500, 3072)):
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    ts = {K: run(M, N, K) for K in (192, 768, 3072)}
    print(f"M {M:5d} N {N:5d}: {tiles:5d} tiles ({tiles / 256:4.2f} per CU)  " +
          "  ".join(f"K {K}: {t:6.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF" for K, t in ts.items()) +
          f"   per slab {(ts[768] - ts[192]) / 18:5.2f} / {(ts[3072] - ts[768]) / 72:5.2f} us")
