import torch, sys
sys.path.insert(0, '/root/repo')
from hirest_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (R, C) in [(1500, 768), (37, 64), (240, 3072)]:
    x = torch.randn(R, C, device=dev)
    on, ot = ops.split2_both(x)
    ref_n = ops.split2(x)
    assert torch.equal(on, ref_n), "normal split"
    Rp = (R + 31) // 32 * 32
    xt = torch.zeros(C, Rp, device=dev); xt[:, :R] = x.t()
    ref_t = ops.split2(xt.contiguous())
    assert torch.equal(ot, ref_t), "transposed split"
    bn, bt = ops.split2_both(x, blocked=True)
    assert torch.equal(bn.data.permute(1, 0, 2).reshape(R, 2 * C), ref_n), "blocked normal split"
    assert torch.equal(bt.data.permute(1, 0, 2).reshape(C, 2 * Rp), ref_t), "blocked transposed split"
    print("split ok", R, C)
import os
from hirest_amd import _lib
_lib.load().hirest_gemm_debug_mode(int(os.environ.get('GDBG', '0')))
BLOCKED = os.environ.get('BLOCKED', '1') == '1'
for (M, N, K) in [(1500, 768, 768), (1500, 3072, 768), (1500, 768, 3072), (300, 2304, 768), (768, 3072, 1536), (240, 30528, 768)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    a2 = ops.split2(a); w2 = ops.split2(w)
    if BLOCKED:
        a2, _ = ops.split2_both(a, True, False, blocked=True); w2, _ = ops.split2_both(w, True, False, blocked=True)
    out = ops.gemm_x3(a2, w2, b)
    ref = (a.double() @ w.double().t() + b.double())
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    r = torch.randn(M, N, device=dev); r0 = r.clone()
    ops.gemm_x3(a2, w2, None, resid_out=r)
    err2 = (r.double() - (r0.double() + a.double() @ w.double().t())).abs().max().item() / ref.abs().max().item()
    # timing: 20 calls replayed from a hipGraph (the host needs ~25 us per call, more than these kernels take)
    outb = torch.empty(M, N, device=dev)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3): ops.gemm_x3(a2, w2, b, resid_out=None)
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): ops.gemm_x3(a2, w2, b)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"gemm_x3 blocked={BLOCKED} {M}x{N}x{K}: rel err {err:.2e} / resid {err2:.2e}; {e0.elapsed_time(e1)/100*1e3:.1f} us")

# the exact-fp32 kernels of the training path on the same shapes (hirest_gemm_f32_ws), same timing method
from hirest_amd import train
train.GEMM_PRECISION = "fp32"
for (M, N, K) in [(1500, 768, 768), (1500, 3072, 768), (1500, 768, 3072), (1500, 2304, 768)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3): train._K.gemm(a, w, b)
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): train._K.gemm(a, w, b)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"gemm_f32 {M}x{N}x{K}: {e0.elapsed_time(e1)/100*1e3:.1f} us")
