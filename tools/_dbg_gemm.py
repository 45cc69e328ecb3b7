import ctypes as C, sys, time, torch
sys.path.insert(0, '/root/repo')
from hirest_amd import _lib, ops
lib = _lib.load()
lib.hirest_dbg_f32.argtypes = [C.c_int]; lib.hirest_dbg_f32.restype = C.c_int
dev = torch.device("cuda:0")
def run(M, N, K, dbg, reps=50):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    lib.hirest_dbg_f32(dbg)
    f = lambda: lib.hirest_gemm_f32(a.data_ptr(), K, w.data_ptr(), K, None, None, 0, None, 0, out.data_ptr(), N, M, N, K, 0, ops.stream_ptr())
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in ((1500, 3072, 768), (1500, 2304, 768), (1500, 768, 768), (4096, 3072, 768)):
    print(M, N, K, " ".join(f"dbg{d}={run(M, N, K, d):.1f}us" for d in (0, 1, 2, 3, 4, 8, 11, 15)), f" ideal {2.0*M*N*K/150e12*1e6:.1f}us")
