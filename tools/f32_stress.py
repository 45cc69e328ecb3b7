#!/usr/bin/env python3
"""Randomised bit-equality stress of the round-3 fp32 kernels against the kernels they replace (GPU):
   hirest_gemm_f32 automatic dispatch (16-column LDS-DMA kernel, split form via hirest_gemm_f32_ws) vs the forced 64x64 kernel;
   hirest_gemm_f32_ln vs hirest_layernorm + hirest_gemm_f32;  hirest_attention_f32_decode vs gather + hirest_attention_f32_qkv;
   hirest_gemm_f32_layouts (k-major operands in place) vs transposed copies + hirest_gemm_f32.
   python tools/f32_stress.py [cases]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops
from hirest_amd.moment_model import MomentModel

lib, dev = _lib.load(), torch.device("cuda:0")
st = ops.stream_ptr()
rng = random.Random(1234)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
g = torch.Generator(device="cpu"); g.manual_seed(7)
rand = lambda *s, scale=1.0: (torch.rand(*s, generator=g) * 2 - 1).mul_(scale).to(dev)
bad = 0
for case in range(n):
    M = rng.choice([1, 5, 15, 16, 17, 25, 32, 33, 64, 100, 240, 256, 300, 1500])
    N = 4 * rng.randint(1, 1000) if rng.random() < 0.8 else rng.choice([768, 2304, 3072, 30528])
    K = 32 * rng.randint(1, 100) if rng.random() < 0.8 else rng.choice([768, 3072, 1504, 16, 48, 1040])
    if M * N > 8e6 or N * K > 3e7: N = 768
    act = rng.randint(0, 3)
    a, w = rand(M, K), rand(N, K, scale=0.05)
    b = rand(N, scale=0.3) if rng.random() < 0.8 else None
    r = rand(M, N) if rng.random() < 0.5 else None
    outs = []
    for mode in (0, 1):
        _lib.check(lib.hirest_gemm_f32_select_kernel(mode), "select")
        outs.append(MomentModel._gemm(a, w, b, resid=r, act=act))
    lib.hirest_gemm_f32_select_kernel(0)
    if not torch.equal(outs[0], outs[1]):
        bad += 1; print("GEMM MISMATCH", M, N, K, act, b is not None, r is not None, float((outs[0] - outs[1]).abs().max()))
print(f"gemm_f32: {n} random problems, {bad} mismatches")
bad_ln = 0
for case in range(n // 3):
    M = rng.randint(1, 32); K = rng.choice([256, 512, 768, 1024]); N = 4 * rng.randint(1, 800) if rng.random() < 0.85 else 30528
    if N >= 8192 and K != 768: N = 768
    act = rng.randint(0, 3)
    x, w, b = rand(M, K, scale=2.0), rand(N, K, scale=0.05), rand(N, scale=0.3)
    gm, be = 1.0 + rand(K, scale=0.2), rand(K, scale=0.2)
    ln = torch.empty((M, K), device=dev)
    _lib.check(lib.hirest_layernorm(x.data_ptr(), K, None, gm.data_ptr(), be.data_ptr(), 1e-12, ln.data_ptr(), K, 1, M, K, st), "ln")
    ref = MomentModel._gemm(ln, w, b, act=act)
    out = torch.empty((M, N), device=dev); ln2 = torch.empty((M, K), device=dev)
    want_ln = N < 8192
    _lib.check(lib.hirest_gemm_f32_ln(x.data_ptr(), K, None, None, None, gm.data_ptr(), be.data_ptr(), 1e-12, ln2.data_ptr() if want_ln else None, K,
                                      w.data_ptr(), K, b.data_ptr(), None, 0, out.data_ptr(), N, M, N, K, act, st), "gemm_ln")
    if not torch.equal(out, ref) or (want_ln and not torch.equal(ln2, ln)):
        bad_ln += 1; print("GEMM_LN MISMATCH", M, N, K, act)
print(f"gemm_f32_ln: {n // 3} random problems, {bad_ln} mismatches")
bad_at = 0
for case in range(n // 3):
    R, H = rng.randint(1, 40), rng.choice([1, 4, 12]); D = 64 * H
    t_hist = rng.randint(0, 150); newkey = rng.random() < 0.7 or t_hist == 0
    addc = rng.choice([0.0, -10000.0])
    T = t_hist + (1 if newkey else 0)
    qkv = rand(R, 3 * D, scale=1.5); kh = rand(R, max(t_hist, 1), D, scale=1.5); vh = rand(R, max(t_hist, 1), D, scale=1.5)
    parent = torch.tensor([rng.randrange(R) for _ in range(R)], dtype=torch.int32, device=dev) if rng.random() < 0.7 else None
    src = parent.long() if parent is not None else torch.arange(R, device=dev)
    kc, vc = kh[src][:, :t_hist], vh[src][:, :t_hist]
    if newkey:
        kc = torch.cat([kc, qkv[:, None, D:2 * D]], 1); vc = torch.cat([vc, qkv[:, None, 2 * D:]], 1)
    kc, vc = kc.contiguous(), vc.contiguous()
    ref = torch.empty((R, D), device=dev)
    _lib.check(lib.hirest_attention_f32_qkv(qkv.data_ptr(), 3 * D, kc.data_ptr(), vc.data_ptr(), D, ref.data_ptr(), R, 1, T, H, 64, 0.125, addc, 0.0, st), "attn")
    out = torch.empty((R, D), device=dev); ko = torch.zeros((R, T, D), device=dev); vo = torch.zeros((R, T, D), device=dev)
    _lib.check(lib.hirest_attention_f32_decode(qkv.data_ptr(), 3 * D, kh.data_ptr() if t_hist else None, vh.data_ptr() if t_hist else None, D,
                                               parent.data_ptr() if parent is not None else None, t_hist,
                                               qkv.data_ptr() + 4 * D if newkey else None, qkv.data_ptr() + 8 * D if newkey else None, 3 * D,
                                               ko.data_ptr(), vo.data_ptr(), out.data_ptr(), R, H, 0.125, addc, 0.0, st), "decode")
    if not (torch.equal(out, ref) and torch.equal(ko, kc) and torch.equal(vo, vc)):
        bad_at += 1; print("ATTENTION MISMATCH", R, H, t_hist, newkey, addc)
print(f"attention_f32_decode: {n // 3} random problems, {bad_at} mismatches")
# backward products with k-major operands read in place vs zero-padded transposed copies
from hirest_amd import train
bad_lay = 0
for case in range(n // 3):
    R = rng.choice([37, 100, 240, 1498, 1500, rng.randint(5, 2000)]); O = 16 * rng.randint(1, 200); I = 4 * rng.randint(1, 800)
    if O * I > 3e6: I = 768
    dy, x, w = rand(R, O), rand(R, I), rand(O, I, scale=0.05)
    outs = []
    for flag in (True, False):
        train.LAYOUT_GEMM = flag
        outs.append((train._K.grad_input(dy, w), train._K.grad_weight(dy, x)))
    train.LAYOUT_GEMM = True
    if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
        bad_lay += 1; print("LAYOUTS MISMATCH", R, O, I)
print(f"gemm_f32_layouts (dX, dW): {n // 3} random problems, {bad_lay} mismatches")
sys.exit(1 if bad + bad_ln + bad_at + bad_lay else 0)
