"""precision='bf16x3' (csrc/tower_x3.hip, HIREST_GEMM_X3): the reference's fp32 forward with each linear layer's product formed from
bf16 hi + lo splits of both fp32 operands on the bf16 matrix pipe (w_hi a_lo + w_lo a_hi + w_hi a_hi, fp32 accumulation).

Bars: the split format is exact by construction (hi = bf16(x), lo = bf16(x - hi)); a product carries ~16 mantissa bits, so a GEMM is
within ~1e-5 of the fp64 product relative to the row's magnitude (the plain bf16 GEMM: ~4e-3); the towers agree with the REAL
reference's fp32 outputs (tests/golden/eva_tiny.npz, eva_g14.npz, eva_g14_c3.npz) to ~1e-5 and reproduce its retrieval ranks."""
import json
import os

import numpy as np
import pytest
import torch

from hirest_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _split_ref(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    rows, D = x.shape
    out = torch.empty((rows, D // 32, 2, 32), dtype=torch.bfloat16, device=x.device)
    out[:, :, 0] = hi.reshape(rows, D // 32, 32)
    out[:, :, 1] = lo.reshape(rows, D // 32, 32)
    return out.reshape(rows, 2 * D)


@pytest.mark.parametrize("rows,D", [(7, 32), (300, 1408), (257, 6144), (1025, 704)])
def test_split2_format_and_gelu(dev, rows, D):
    from hirest_amd import ops
    x = synth.tensor(f"x3.split.{rows}.{D}", (rows, D), 1.7, 3).to(dev)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -3e4, 1.0, 1.00390625, 65504.0, -1e-3], device=dev)
    got = ops.split2(x)
    assert torch.equal(got.view(torch.int16), _split_ref(x).view(torch.int16))
    # hi + lo reproduces x to 2^-17 relative (two 8-bit significands)
    back = got.reshape(rows, D // 32, 2, 32).float().sum(dim=2).reshape(rows, D)
    assert ((back - x).abs() <= x.abs() * 2.0 ** -16 + 1e-37).all()
    # act = 1: nn.GELU() (erf form) in fp32 before the split
    g = torch.nn.functional.gelu(x.double().cpu()).float().to(dev)
    got_g = ops.split2(x, gelu=True).reshape(rows, D // 32, 2, 32).float().sum(dim=2).reshape(rows, D)
    assert (got_g - g).abs().max().item() <= 3e-6 * max(1.0, g.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 132, 128), (1000, 1408, 704), (2570, 4224, 1408), (1028, 1408, 6144), (64, 6144, 1408)])
def test_gemm_x3_against_fp64(dev, M, N, K):
    """A W^T from split operands vs the fp64 product: error relative to |a|.|w| per element ~2^-17, three orders below the bf16 GEMM's;
    the residual epilogue adds into x; results do not depend on the rest of the batch (rows are independent)."""
    from hirest_amd import ops
    a = synth.tensor(f"x3.a.{M}.{K}", (M, K), 1.0, 5).to(dev)
    w = synth.tensor(f"x3.w.{N}.{K}", (N, K), 0.05, 6).to(dev)
    bias = synth.tensor(f"x3.b.{N}", (N,), 0.1, 7).to(dev)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.double().abs() @ w.double().abs().t())                       # per-element magnitude of the summed products
    a2, w2 = ops.split2(a), ops.split2(w)
    got = ops.gemm_x3(a2, w2, bias)
    err = ((got.double() - ref).abs() / scale).max().item()
    rms = ((got.double() - ref) / ref.abs().mean()).pow(2).mean().sqrt().item()
    bf = (a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + bias.double())
    err_bf = ((bf - ref).abs() / scale).max().item()
    f32 = (a @ w.t() + bias).double()
    rms32 = ((f32 - ref) / ref.abs().mean()).pow(2).mean().sqrt().item()
    print(f"gemm_x3 {M}x{N}x{K}: max err / sum|a||w| {err:.2e} (bf16 operands: {err_bf:.2e}); rms err / mean|ref| {rms:.2e} (torch fp32: {rms32:.2e})")
    assert err < 2.0 ** -16 and err < err_bf / 50
    x = synth.tensor(f"x3.x.{M}.{N}", (M, N), 1.0, 8).to(dev)
    want = x.double() + ref
    out = ops.gemm_x3(a2, w2, bias, resid_out=x.clone())
    assert ((out.double() - want).abs() / (scale + x.abs().double())).max().item() < 2.0 ** -16
    if M >= 300:                                                             # a row's result does not depend on its neighbours
        part = ops.gemm_x3(a2[100:300].contiguous(), w2, bias)
        assert torch.equal(part, got[100:300])
    if N % 32 == 0:                                                          # GELU + split in the epilogue == the separate pass on the fp32 output
        fused = ops.gemm_x3(a2, w2, bias, gelu_split=True)
        sep = ops.split2(got, gelu=True)
        back = lambda t: t.reshape(M, N // 32, 2, 32).float().sum(dim=2).reshape(M, N)
        assert (back(fused) - back(sep)).abs().max().item() <= 2.0 ** -16 * max(1.0, back(sep).abs().max().item())
        same = (fused.view(torch.int16) == sep.view(torch.int16)).float().mean().item()
        print(f"   gelu + split epilogue: {100 * same:.3f} % of the bf16 words equal to the separate pass")


def test_layernorm_split2_equals_layernorm_then_split(dev):
    from hirest_amd import _lib, ops
    lib = _lib.load()
    for rows, D, eps in ((514, 1408, 1e-6), (77, 704, 1e-6), (5, 2048, 1e-5)):
        x = synth.tensor(f"x3.ln.{rows}.{D}", (rows, D), 2.0, 9, mean=0.3).to(dev)
        g = synth.tensor("x3.ln.g", (D,), 0.2, 10, mean=1.0).to(dev)
        b = synth.tensor("x3.ln.b", (D,), 0.2, 11).to(dev)
        y = torch.empty_like(x)
        ops.layernorm(x, g, b, eps, y)
        out = torch.empty((rows, 2 * D), dtype=torch.bfloat16, device=dev)
        _lib.check(lib.hirest_layernorm_split2(x.data_ptr(), D, g.data_ptr(), b.data_ptr(), eps, out.data_ptr(), 2 * D, rows, D,
                                               ops.stream_ptr()), "layernorm_split2")
        assert torch.equal(out.view(torch.int16), ops.split2(y).view(torch.int16))


def test_eva_tiny_bf16x3_vs_reference(dev, golden_dir):
    import hirest_amd
    g = np.load(os.path.join(golden_dir, "eva_tiny.npz"))
    seed = int(g["seed"])
    model, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained=f"synth:{seed}", precision="bf16x3")
    model = model.to(dev).eval()
    assert model.visual.precision == "bf16x3" and model.text.precision == "fp32"
    img = synth.frames("eva_tiny.img", (int(g["n_img"]), 3, 224, 224), seed + 1).to(dev)
    ref = torch.from_numpy(g["image_embed"])
    got = model.encode_image(img)
    err = (got.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"tiny tower, bf16x3 vs reference: max |diff| / max |ref| = {err:.2e}")
    assert err <= 1e-4
    # batch invariance and uint8 / bf16 inputs go through the same front end as the fp32 tower
    big = torch.cat([img.flip(0), img, img[:1]], 0)
    assert torch.equal(model.encode_image(big)[img.shape[0]:2 * img.shape[0]], got)
    assert tuple(model.encode_image(img[:0]).shape) == (0, ref.shape[1])
    model.set_precision("fp32")
    f32 = model.encode_image(img)
    print(f"tiny tower, bf16x3 vs the exact-fp32 kernels: {(got - f32).abs().max().item() / f32.abs().max().item():.2e}")


def test_eva_g14_bf16x3_vs_reference_and_c3_ranks(dev, golden_dir):
    """EVA-CLIP-g/14, 40 layers: image embeddings vs the reference's fp32 outputs, then BASELINE configs[2]'s pinned sub-corpus (256
    videos x 4 frames x 546 real prompts, eva_g14_c3.npz): score error, top-1 flips against the reference's ranking, margin table."""
    import hirest_amd
    from hirest_amd import retrieval
    g = np.load(os.path.join(golden_dir, "eva_g14.npz"))
    seed = int(g["seed"])
    model, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_g_14", pretrained=f"synth:{seed}", precision="bf16x3")
    model = model.to(dev).eval()
    img = synth.frames("eva_g14.img", (int(g["n_img"]), 3, 224, 224), seed + 1).to(dev)
    ref = torch.from_numpy(g["image_embed"])
    got = model.encode_image(img)
    err = (got.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"EVA-g/14 bf16x3 vs reference image_embed: max |diff| / max |ref| = {err:.2e}")
    assert err <= 2e-4
    c = np.load(os.path.join(golden_dir, "eva_g14_c3.npz"))
    V, F = int(c["V"]), int(c["F"])
    assert int(c["seed"]) == seed
    frames = synth.c3_corpus(V, F)
    names = synth.c3_names(V)
    tok = torch.from_numpy(c["tokens"].astype(np.int64))
    pooled = retrieval.encode_videos(model, frames.to(dev))
    texts = retrieval.encode_texts(model, tok.to(dev))
    scores, _, idx = retrieval.retrieve(texts, pooled, 10, retrieval.tie_rank_from_names(names, dev))
    idx, scores = idx.cpu().long(), scores.cpu()
    ref_scores = torch.from_numpy(c["scores"])
    gt = torch.from_numpy(c["top10"][:, 0].astype(np.int64))
    margin = torch.from_numpy(c["margin"])
    serr = (scores - ref_scores).abs().max().item()
    flips = idx[:, 0] != gt
    same10 = int((idx == torch.from_numpy(c["top10"].astype(np.int64))).all(dim=1).sum())
    print(f"C3 @ g/14, precision='bf16x3': max |score error| {serr:.2e}, top-1 flips {int(flips.sum())} of {gt.numel()} "
          f"(largest reference margin among them {margin[flips].max().item() if flips.any() else 0.0:.2e}; smallest reference margin "
          f"{margin.min().item():.2e}), identical top-10 lists {same10}, pooled min cosine "
          f"{torch.nn.functional.cosine_similarity(pooled.cpu(), torch.from_numpy(c['pooled']), dim=-1).min().item():.7f}")
    assert serr < 5e-5
    safe = margin > 2 * serr
    assert torch.equal(idx[safe, 0], gt[safe])                     # exact wherever the reference's own margin decides
    assert int(flips.sum()) <= int((~safe).sum())


@pytest.mark.parametrize("B,T,H,dh", [(2, 257, 16, 88), (1, 300, 12, 64), (3, 77, 4, 32), (1, 33, 2, 96), (2, 64, 3, 40), (1, 1, 1, 64)])
def test_attention_x3_against_fp64(dev, B, T, H, dh):
    """hirest_attention_x3_qkv (both products from bf16 hi + lo splits, softmax in fp32) against softmax(q k^T * scale) v in fp64 on packed
    q | k | v rows, and against the exact-fp32 kernel it replaces in the bf16x3 tower; ragged tails (257 = 8 x 32 + 1 keys), padded head
    widths (88 -> 96, 40 -> 64) and large score magnitudes included."""
    from hirest_amd import _lib, ops
    lib = _lib.load()
    D = H * dh
    qkv = synth.tensor(f"x3.attn.{B}.{T}.{H}.{dh}", (B * T, 3 * D), 1.0, 13).to(dev)
    qkv[:, :D] *= 3.0                                                  # peaked softmax rows as well as flat ones
    scale = dh ** -0.5
    q, k, v = (qkv[:, i * D:(i + 1) * D].double().reshape(B, T, H, dh).permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    out = torch.full((B * T, D), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.hirest_attention_x3_qkv(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, out.data_ptr(), B, T, T, H,
                                           dh, scale, ops.stream_ptr()), "hirest_attention_x3_qkv")
    f32 = torch.empty_like(out)
    _lib.check(lib.hirest_attention_f32_qkv(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, f32.data_ptr(), B, T, T, H,
                                            dh, scale, 0.0, 0.0, ops.stream_ptr()), "hirest_attention_f32_qkv")
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    err32 = (f32.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"attention x3 B={B} T={T} H={H} dh={dh}: max err / max |ref| {err:.2e} (exact-fp32 kernel: {err32:.2e})")
    assert torch.isfinite(out).all() and err < 3e-5
    if D % 32 == 0:                                                    # the split-operand output == split of the fp32 output, bit for bit
        out2 = torch.empty((B * T, 2 * D), dtype=torch.bfloat16, device=dev)
        _lib.check(lib.hirest_attention_x3_qkv_split2(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, out2.data_ptr(), B, T,
                                                      T, H, dh, scale, ops.stream_ptr()), "hirest_attention_x3_qkv_split2")
        assert torch.equal(out2.view(torch.int16), ops.split2(out).view(torch.int16))
    for waves in (3, 4, 8, 9):                                        # workgroup size changes who stages the tiles, not a query's arithmetic
        _lib.check(lib.hirest_attention_x3_select_waves(waves), "select_waves")
        try:
            o2 = torch.full((B * T, D), 7.0, dtype=torch.float32, device=dev)
            _lib.check(lib.hirest_attention_x3_qkv(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, o2.data_ptr(), B, T, T,
                                                   H, dh, scale, ops.stream_ptr()), "hirest_attention_x3_qkv")
        finally:
            lib.hirest_attention_x3_select_waves(0)
        assert torch.equal(o2, out), waves


def test_x3_tower_attention_ab(dev, golden_dir):
    """The tower with its split-operand attention (default) and with the exact-fp32 attention agree to the level of the split products."""
    import hirest_amd
    from hirest_amd import _lib
    g = np.load(os.path.join(golden_dir, "eva_tiny.npz"))
    seed = int(g["seed"])
    model, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained=f"synth:{seed}", precision="bf16x3")
    model = model.to(dev).eval()
    img = synth.frames("eva_tiny.img", (int(g["n_img"]), 3, 224, 224), seed + 1).to(dev)
    a = model.encode_image(img)
    _lib.load().hirest_vision_x3_select_attention(1)
    try:
        b = model.encode_image(img)
    finally:
        _lib.load().hirest_vision_x3_select_attention(0)
    ref = torch.from_numpy(g["image_embed"]).to(dev)
    _lib.load().hirest_vision_x3_select_attention(2)                 # GELU + split as a separate pass over an fp32 hidden activation
    try:
        c = model.encode_image(img)
    finally:
        _lib.load().hirest_vision_x3_select_attention(0)
    ea, eb, ec = ((x - ref).abs().max().item() / ref.abs().max().item() for x in (a, b, c))
    print(f"tiny bf16x3 tower vs reference: default {ea:.2e}, exact-fp32 attention {eb:.2e}, separate GELU pass {ec:.2e}")
    assert ea <= 1e-4 and eb <= 1e-4 and ec <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1500, 768, 768), (300, 2304, 768), (1500, 768, 3072), (131, 132, 64), (240, 30528, 768)])
def test_gemm_x3_small_problem_kernel(dev, M, N, K):
    """Problems of fewer than 256 tiles of 256 x 256 take gemm_t128x3 (8 waves, the two 16-deep chunks of a step on two wave groups):
    against fp64, bias and accumulate-into forms."""
    from hirest_amd import ops
    g = torch.Generator(device="cpu"); g.manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ w.double().t() + b.double()
    out = ops.gemm_x3(ops.split2(a), ops.split2(w), b)
    assert (out.double() - ref).abs().max().item() <= 2 ** -16 * ref.abs().max().item() * 4
    a2, w2 = ops.split2(a), ops.split2(w)
    r = torch.randn(M, N, generator=g).to(dev)
    r0 = r.clone()
    ops.gemm_x3(a2, w2, None, resid_out=r)
    assert (r.double() - (r0.double() + ref - b.double())).abs().max().item() <= 2 ** -16 * ref.abs().max().item() * 4
