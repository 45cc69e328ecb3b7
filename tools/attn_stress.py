#!/usr/bin/env python3
"""Race screen for the persistent attention kernel (default form: nine compute waves + the producer wave): repeated launches on the
same input must be bit-identical, equal to the form without the producer wave bit for bit, and equal to the register-staged v1
kernel within bf16 rounding."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
bad = 0
for (B, N, H, dh) in [(1024, 257, 16, 88), (256, 257, 16, 88), (100, 200, 12, 64), (64, 257, 16, 88)]:
    g = torch.Generator(device=dev); g.manual_seed(B)
    qkv = torch.randn((B * N, 3 * H * dh), device=dev, generator=g).to(torch.bfloat16)
    outs = []
    ops.attention_select_kernel(3)
    v3 = torch.empty((B * N, H * dh), device=dev, dtype=torch.bfloat16)
    ops.attention(qkv, v3, B, N, H, dh, False)
    ops.attention_select_kernel(ops.ATTENTION_DEFAULT_KERNEL)
    ref = torch.empty((B * N, H * dh), device=dev, dtype=torch.bfloat16)
    ops.attention(qkv, ref, B, N, H, dh, False)
    if not torch.equal(ref, v3):
        bad += 1
        print(f"MISMATCH B={B} N={N}: producer-wave form differs from v3 in {(v3 != ref).sum().item()} elements", flush=True)
    for r in range(40):
        out = torch.empty_like(ref)
        ops.attention(qkv, out, B, N, H, dh, False)
        if not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH B={B} N={N} repeat {r}: {(out != ref).sum().item()} elements", flush=True)
            break
    ops.attention_select_kernel(1)
    v1 = torch.empty_like(ref)
    ops.attention(qkv, v1, B, N, H, dh, False)
    d = (v1.float() - ref.float()).abs().max().item()
    print(f"B={B} N={N} H={H} dh={dh}: 40 repeats identical: {bad == 0}; max |default - v1| = {d:.4f}", flush=True)
    if d > 0.05: bad += 1
ops.attention_select_kernel(ops.ATTENTION_DEFAULT_KERNEL)
print("RESULT:", "clean" if bad == 0 else f"{bad} problems")
sys.exit(1 if bad else 0)
