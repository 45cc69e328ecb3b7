import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import _lib, ops
dev = torch.device("cuda:0")
for (M, N, K) in [(2570, 1408, 1408), (2056, 1408, 6144), (1999, 1056, 704), (2560, 1408, 1408), (2568, 1408, 1408)]:
    g = torch.Generator(device=dev); g.manual_seed(M + N + 1)
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev, generator=g)
    x0 = torch.randn((M, N), device=dev, generator=g) * 3 + 0.7
    hi = x0.to(torch.bfloat16); lo = (x0 - hi.float()).to(torch.bfloat16)
    ref = hi.float() + lo.float()
    xb_ref = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    G = (N + 63) // 64
    part_ref = torch.empty((M, G, 2), device=dev)
    ops.gemm(A, W, bias, ref, _lib.EPI_BIAS_RESID_LNSTATS_F32, aux0=xb_ref, aux1=part_ref)
    part = torch.full((M, G, 2), float("nan"), device=dev)
    ops.gemm(A, W, bias, lo, _lib.EPI_BIAS_RESID2_LNSTATS, aux0=hi, aux1=part)
    torch.cuda.synchronize()
    bad_hi = (hi != xb_ref).nonzero()
    bad_pt = ((part != part_ref) | torch.isnan(part)).nonzero()
    lo_ref = (ref - hi.float()).to(torch.bfloat16)
    bad_lo = (lo != lo_ref).nonzero()
    print(M, N, K, "hi mismatches", bad_hi.shape[0], "rows", sorted(set(bad_hi[:, 0].tolist()))[:20], "cols", sorted(set(bad_hi[:, 1].tolist()))[:20])
    print("   part mismatches", bad_pt.shape[0], "rows", sorted(set(bad_pt[:, 0].tolist()))[:20], "groups", sorted(set(bad_pt[:, 1].tolist()))[:24])
    print("   lo mismatches", bad_lo.shape[0], "rows", sorted(set(bad_lo[:, 0].tolist()))[:20])
