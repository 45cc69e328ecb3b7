#!/usr/bin/env python3
"""The LM-head product (N = 30 528, K = 768) at the row counts of a merged beam search (60 - 160 rows) through each fp32 GEMM
kernel hirest_gemm_f32 can take: 0 automatic, 1 the 64x64 kernel, 2 automatic without the 16-column kernel (= split-K 32x32)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, synth
from hirest_amd.moment_model import MomentModel
dev = torch.device("cuda:0")
lib = _lib.load()
N, K = 30528, 768
w = synth.tensor("lm.w", (N, K), 0.05, 3).to(dev)
b = synth.tensor("lm.b", (N,), 0.3, 3).to(dev)
for M in (25, 60, 96, 100, 160, 256):
    a = synth.tensor("lm.a", (M, K), 1.0, 3).to(dev)
    out = torch.empty((M, N), device=dev)
    line = []
    ref = None
    for mode in (0, 1, 2):
        _lib.check(lib.hirest_gemm_f32_select_kernel(mode), "select")
        for _ in range(3):
            MomentModel._gemm(a, w, b, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            MomentModel._gemm(a, w, b, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        if ref is None:
            ref = out.clone()
        line.append(f"mode {mode}: {dt * 1e6:7.1f} us ({2.0 * M * N * K / dt / 1e12:5.1f} TF){'' if torch.equal(out, ref) else ' BITS DIFFER'}")
    lib.hirest_gemm_f32_select_kernel(0)
    print(f"M={M:4d}  " + "  ".join(line), flush=True)
