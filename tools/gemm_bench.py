#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels at the EVA-CLIP-g layer shapes (random bf16 operands).
   python tools/gemm_bench.py [--variants 1 2 3] [--frames 1024] [--iters 10] [--shapes fc1 fc2 qkv proj]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops  # noqa: E402

SHAPES = {"qkv": (4224, 1408, _lib.EPI_BIAS_BF16), "proj": (1408, 1408, _lib.EPI_BIAS_RESID_F32),
          "fc1": (6144, 1408, _lib.EPI_BIAS_GELU_BF16), "fc2": (1408, 6144, _lib.EPI_BIAS_RESID_F32),
          "fc1_nogelu": (6144, 1408, _lib.EPI_BIAS_BF16), "fc2_plain": (1408, 6144, _lib.EPI_BIAS_BF16),
          "proj_plain": (1408, 1408, _lib.EPI_BIAS_BF16), "fc2_f32": (1408, 6144, _lib.EPI_BIAS_F32),
          # LayerNorm-fold epilogues (same operands, so the difference to qkv / fc1 / proj / fc2 is the epilogue's cost)
          "qkv_fold": (4224, 1408, _lib.EPI_LNFOLD_BF16), "fc1_fold": (6144, 1408, _lib.EPI_LNFOLD_GELU_BF16),
          "proj_stats": (1408, 1408, _lib.EPI_BIAS_RESID_LNSTATS_F32), "fc2_stats": (1408, 6144, _lib.EPI_BIAS_RESID_LNSTATS_F32),
          # the same with the residual stream as bf16 hi + bf16 lo (out = lo, aux0 = hi, both in / out)
          "proj_stats2": (1408, 1408, _lib.EPI_BIAS_RESID2_LNSTATS), "fc2_stats2": (1408, 6144, _lib.EPI_BIAS_RESID2_LNSTATS)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=15, help="untimed launches per measurement (clocks / power state settle)")
    ap.add_argument("--shapes", nargs="*", default=["qkv", "proj", "fc1", "fc2", "fc1_nogelu"])
    ap.add_argument("--alias", action="store_true", help="lda=ldw=0: every row aliases row 0 (operands stay cache-resident): "
                    "kernel-structure ceiling without the memory system")
    ap.add_argument("--hipblaslt", action="store_true", help="also time torch.mm (hipBLASLt, no epilogue) as a yardstick")
    ap.add_argument("--lda-pad", type=int, default=0, help="extra elements in A's row stride (L2 set-conflict experiment)")
    ap.add_argument("--ldw-pad", type=int, default=0, help="extra elements in W's row stride")
    ap.add_argument("--dbg", type=int, default=0, help="hirest_gemm_debug_mode bits (timing experiments)")
    ap.add_argument("--nk", type=int, nargs=3, action="append", default=[], metavar=("N", "K", "EPI"),
                    help="extra shape (repeatable), named n<N>k<K>e<EPI>")
    ap.add_argument("--reverse", action="store_true", help="HIREST_GEMM_REVERSE: walk the tile list backwards")
    ap.add_argument("--a-scale", type=float, default=1.0, help="A = randn * scale + offset (does operand distribution move the time?)")
    ap.add_argument("--a-offset", type=float, default=0.0)
    ap.add_argument("--w-scale", type=float, default=0.02, help="W = randn * scale; --a-scale 0 --w-scale 0 = all-zero operands: the "
                    "same instruction stream with (almost) no switching activity in the matrix pipe, i.e. how far the power cap is "
                    "from the kernel's own limit")
    a = ap.parse_args()
    for n, k, e in a.nk:
        SHAPES[f"n{n}k{k}e{e}"] = (n, k, e)
        a.shapes = [x for x in a.shapes] + [f"n{n}k{k}e{e}"]
    _lib.load().hirest_gemm_debug_mode(a.dbg)
    dev = torch.device("cuda:0")
    M = a.frames * 257
    g = torch.Generator(device=dev); g.manual_seed(0)
    for name in a.shapes:
        N, K, epi = SHAPES[name]
        A = (torch.randn((M, K + a.lda_pad), device=dev, generator=g) * a.a_scale + a.a_offset).to(torch.bfloat16)
        W = (torch.randn((N, K + a.ldw_pad), device=dev, generator=g) * a.w_scale).to(torch.bfloat16)
        bias = torch.randn((N,), device=dev, generator=g)
        out = torch.zeros((M, N), device=dev, dtype=torch.float32 if epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_BIAS_F32, _lib.EPI_BIAS_RESID_LNSTATS_F32) else torch.bfloat16)
        aux0 = aux1 = None
        if epi in (_lib.EPI_LNFOLD_BF16, _lib.EPI_LNFOLD_GELU_BF16):
            aux0 = torch.cat([torch.randn((M + 1, 1), device=dev, generator=g) * 0.1, torch.rand((M + 1, 1), device=dev, generator=g) + 0.5], 1)[:M].contiguous()
            aux0 = torch.cat([aux0, aux0[:1]])[:M]
            aux1 = torch.randn((N,), device=dev, generator=g)
        elif epi in (_lib.EPI_BIAS_RESID_LNSTATS_F32, _lib.EPI_BIAS_RESID2_LNSTATS):
            aux0 = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
            aux1 = torch.empty((M, (N + 63) // 64, 2), device=dev)
        import ctypes as C
        lib = _lib.load()

        def run():
            if not a.alias and not a.lda_pad and not a.ldw_pad:
                return ops.gemm(A, W, bias, out, epi, aux0=aux0, aux1=aux1, flags=1 if a.reverse else 0)
            lda = 0 if a.alias else K + a.lda_pad
            args = _lib.GemmArgs.make(A.data_ptr(), lda, W.data_ptr(), 0 if a.alias else K + a.ldw_pad, bias.data_ptr(), out.data_ptr(), N, M, N, K, epi, None, 0)
            _lib.check(lib.hirest_gemm_bf16(C.byref(args), ops.stream_ptr()), "gemm")
        for v in a.variants:
            ops.gemm_select_kernel(v)
            for _ in range(a.warmup):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{'[alias] ' if a.alias else ''}{'[dbg%d] ' % a.dbg if a.dbg else ''}{'[lda+%d ldw+%d] ' % (a.lda_pad, a.ldw_pad) if a.lda_pad or a.ldw_pad else ''}{name:11s} M={M} N={N} K={K} variant={v}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
        if a.hipblaslt:
            Wt = W[:N, :K].t()
            A = A[:, :K] if a.lda_pad else A
            for _ in range(2):
                torch.mm(A, Wt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                torch.mm(A, Wt)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{name:11s} M={M} N={N} K={K} torch.mm (hipBLASLt, plain bf16 out): {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
        ops.gemm_select_kernel(0)
        del A, W, out


if __name__ == "__main__":
    main()
