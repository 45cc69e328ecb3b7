#!/usr/bin/env python3
"""Race screen for the persistent GEMM: random shapes / epilogues, results must equal the simple 128x128 kernel bit for bit
(all variants use the same k order per output element), repeated launches must be identical to each other."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--variant", type=int, default=6, help="kernel under test (hirest_gemm_select_kernel)")
ap.add_argument("--full", action="store_true", help="the four production shapes at M = 263168 instead of random shapes")
ap.add_argument("--dbg", type=int, default=0, help="hirest_gemm_debug_mode bits for the kernel under test (e.g. 262144 = bit 18: team walk)")
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
epis = [(_lib.EPI_BIAS_BF16, torch.bfloat16), (_lib.EPI_BIAS_GELU_BF16, torch.bfloat16), (_lib.EPI_BIAS_RESID_F32, torch.float32),
        (_lib.EPI_BIAS_F32, torch.float32)]
bad = 0
for c in range(a.cases):
    M = int(rng.integers(300, 40000)); N = int(rng.integers(64, 1600)) * 4; K = int(rng.integers(1, 40)) * 64
    if c % 5 == 0: M, N = 263168 // 8, [1408, 4224, 6144][c // 5 % 3]          # production-like panels
    if a.full: M, N, K = [(263168, 4224, 1408), (263168, 1408, 1408), (263168, 6144, 1408), (263168, 1408, 6144)][c % 4]
    epi, odt = epis[c % len(epis)]
    if a.full: epi, odt = [epis[0], epis[2], epis[1], epis[2]][c % 4]
    g = torch.Generator(device=dev); g.manual_seed(1000 + c)
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev, generator=g)
    base = torch.randn((M, N), device=dev, generator=g).to(odt)
    def run(variant):
        _lib.load().hirest_gemm_debug_mode(a.dbg if variant == a.variant else 0)
        ops.gemm_select_kernel(variant)
        out = base.clone()
        ops.gemm(A, W, bias, out, epi)
        return out
    ref = run(1)
    ok = True
    for r in range(a.repeats):
        got = run(a.variant)
        if not torch.equal(got, ref):
            ok = False
            d = (got.float() - ref.float()).abs()
            print(f"MISMATCH case {c} M={M} N={N} K={K} epi={epi} repeat {r}: {int((d > 0).sum())} elements, max {d.max().item():.3e}", flush=True)
            break
    bad += not ok
    if c % 10 == 9: print(f"{c + 1} cases, {bad} bad", flush=True)
ops.gemm_select_kernel(0)
_lib.load().hirest_gemm_debug_mode(0)
print("RESULT:", "clean" if bad == 0 else f"{bad} mismatching cases")
sys.exit(1 if bad else 0)
