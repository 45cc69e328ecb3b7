"""LM head alone (hirest_gemm_f32_ln, N = 30 528, K = 768) with its 94 MB of weights coming from HBM: the caches are flushed with a 256-MB fill before every call."""
import sys, torch
sys.path.insert(0, "/root/repo")
from hirest_amd import _lib, ops, synth
lib, st = _lib.load(), ops.stream_ptr()
dev = torch.device("cuda:0")
N, K = 30528, 768
w = synth.tensor("lm.w", (N, K), 0.05, 3).to(dev); bias = synth.tensor("lm.b", (N,), 0.3, 3).to(dev)
g = torch.ones(K, device=dev); be = torch.zeros(K, device=dev)
junk = torch.empty((64, 1024, 1024), device=dev)   # 256 MB: flush the caches between repetitions
for M in (25, 15):
    x = synth.tensor("lm.x", (M, K), 2.0, 3).to(dev); out = torch.empty((M, N), device=dev)
    for eps, name in ((1e-12, "row-major W"),):          # (the as-if-tiled addressing that was timed against it — 54 -> 48 us — is not in the kernel any more)
        ts = []
        for rep in range(6):
            junk.fill_(rep)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.hirest_gemm_f32_ln(x.data_ptr(), K, None, None, None, g.data_ptr(), be.data_ptr(), eps, None, 0, w.data_ptr(), K, bias.data_ptr(),
                                              None, 0, out.data_ptr(), N, M, N, K, 0, st), "lm")
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"M={M} {name}: " + " ".join(f"{t:.1f}" for t in ts) + " us")
