"""GPU parity of the joint-model path (BASELINE configs[3]): fp32 kernels of csrc/joint.hip through the
C ABI vs fp64 references, and hirest_amd.MomentModel.test_step vs the REAL reference MomentModel's outputs
(tests/golden/joint_*.npz, joint_predictions.json).  Boundary indices / boundary lists must be exact."""
import json
import os
import sys

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


@pytest.mark.parametrize("kernel", [0, 1, 2], ids=["auto", "64x64", "auto-no-m16"])
@pytest.mark.parametrize("M,N,K,act", [(64, 64, 16, 0), (193, 768, 768, 1), (300, 2304, 768, 0), (77, 512, 1024, 2), (130, 768, 3072, 0),
                                       (25, 30528, 768, 0), (5, 768, 768, 1), (256, 100, 1504, 0), (33, 36, 48, 2), (15, 2304, 768, 3),
                                       (32, 20, 64, 0), (17, 768, 3072, 1)])
def test_gemm_f32(dev, M, N, K, act, kernel):
    """automatic mode: M <= 256 takes the 16-column kernel when K % 32 == 0 (N < 8192 above 32 rows), else the split-K 32x32 kernel,
    the rest the 64x64 kernel; mode 1 forces the 64x64 kernel, mode 2 is automatic without the 16-column kernel"""
    from hirest_amd import _lib
    from hirest_amd.moment_model import MomentModel
    _lib.check(_lib.load().hirest_gemm_f32_select_kernel(kernel), "select")
    try:
        _gemm_f32_case(dev, M, N, K, act, MomentModel)
    finally:
        _lib.load().hirest_gemm_f32_select_kernel(0)


@pytest.mark.parametrize("rows,V", [(25, 30528), (7, 30522), (3, 5), (2, 1030)])
def test_log_softmax_rows(dev, rows, V):
    """train.py:563-564 + the beam score add of beam.py:76"""
    import ctypes as C
    from hirest_amd import _lib, ops
    x = synth.tensor("ls.x", (rows, V), 3.0, 3).to(dev)
    add = synth.tensor("ls.a", (rows,), 2.0, 3).to(dev)
    out = torch.empty_like(x)
    _lib.check(_lib.load().hirest_log_softmax_f32(x.data_ptr(), V, add.data_ptr(), out.data_ptr(), V, rows, V, ops.stream_ptr()), "ls")
    ref = torch.log_softmax(x.double(), 1) + add.double()[:, None]
    assert (out.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()


def test_gemm_f32_kernels_share_their_summation_order(dev):
    """A row's result must not depend on how many rows the call has (batch invariance of every fp32 path: a sentence embedded
    alone or in a batch, a beam's hidden state recomputed or cached): the split-K kernel for M <= 256 and the 64x64 kernel add the
    same products in the same order, so they agree bit for bit — also across a call that is cut into pieces."""
    from hirest_amd import _lib
    from hirest_amd.moment_model import MomentModel
    lib = _lib.load()
    for M, N, K in ((200, 768, 768), (25, 30528, 768), (256, 132, 3072), (77, 512, 1040), (25, 768, 3072), (15, 2304, 768), (9, 36, 64),
                    (32, 3072, 768), (16, 768, 96), (1500, 768, 768), (1500, 768, 3072), (300, 512, 1040), (1500, 100, 528)):   # the last four: split form
        a = synth.tensor("jo.a", (M, K), 1.0, 2).to(dev)
        w = synth.tensor("jo.w", (N, K), 0.05, 2).to(dev)
        b = synth.tensor("jo.b", (N,), 0.3, 2).to(dev)
        outs = []
        for kernel in (0, 1, 2):
            _lib.check(lib.hirest_gemm_f32_select_kernel(kernel), "select")
            outs.append(MomentModel._gemm(a, w, b, act=1))
        lib.hirest_gemm_f32_select_kernel(0)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (M, N, K)
        big = torch.cat([a, a, a], 0)                              # 3 M > 256 rows: the 64x64 kernel in automatic mode
        if 3 * M > 256:
            assert torch.equal(MomentModel._gemm(big, w, b, act=1)[M:2 * M], outs[0]), (M, N, K)


@pytest.mark.parametrize("M,N,K,act", [(1500, 768, 768, 1), (300, 2304, 768, 0), (1000, 100, 64, 2), (4100, 772, 96, 3), (257, 64, 3072, 0), (700, 36, 2080, 1)])
def test_gemm_f32_ring_form_equals_the_register_prefetch_kernel(dev, M, N, K, act):
    """gemm_f32_ring_kernel (the 64 x 64 tile with its operands on an LDS-DMA ring; automatic from 64 tiles per CU on) against the
    register-prefetch kernel: same K quarters, same k order -> bit for bit, with bias / activation / residual / periodic epilogue operands,
    ragged M and N, K of 2 .. 96 slabs (quarters without a slab included)."""
    from hirest_amd import _lib
    from hirest_amd.moment_model import MomentModel
    lib = _lib.load()
    a = synth.tensor("rg.a", (M, K), 1.0, 4).to(dev)
    w = synth.tensor("rg.w", (N, K), 0.05, 4).to(dev)
    b = synth.tensor("rg.b", (N,), 0.3, 4).to(dev)
    r = synth.tensor("rg.r", (M, N), 1.0, 4).to(dev)
    pos = synth.tensor("rg.p", (50, N), 0.5, 4).to(dev)
    outs = []
    for mode in (1, 2, 0):
        _lib.check(lib.hirest_gemm_f32_ring_mode(mode), "ring mode")
        try:
            outs.append(MomentModel._gemm(a, w, b, resid=r, periodic=pos, period=50, act=act))
        finally:
            lib.hirest_gemm_f32_ring_mode(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert lib.hirest_gemm_f32_ring_mode(3) == -1


def _gemm_f32_case(dev, M, N, K, act, MomentModel):
    a = synth.tensor("jf.a", (M, K), 1.0, 1)
    w = synth.tensor("jf.w", (N, K), 0.05, 1)
    b = synth.tensor("jf.b", (N,), 0.3, 1)
    r = synth.tensor("jf.r", (M, N), 1.0, 1)
    pos = synth.tensor("jf.p", (50, N), 0.5, 1)
    ref = a.double() @ w.double().t() + b.double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.tanh(ref)
    elif act == 3:
        ref = ref * torch.sigmoid(1.702 * ref)
    ref = ref + r.double() + pos.double()[torch.arange(M) % 50]
    out = MomentModel._gemm(a.to(dev), w.to(dev), b.to(dev), resid=r.to(dev), periodic=pos.to(dev), period=50, act=act)
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,T,H", [(2, 64, 12), (1, 300, 12), (1, 33, 2), (1, 700, 3)])
def test_attention_f32_with_constant_shift(dev, B, T, H):
    from hirest_amd import _lib, ops
    D = H * 64
    qkv = synth.tensor(f"ja.{T}", (B * T, 3 * D), 1.0, 2)
    q, k, v = qkv.reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s32 = (q @ k.transpose(-1, -2)) / 8.0 + (-10000.0)          # fp32 on purpose: the shift quantises (hazard H3)
    ref = (torch.softmax(s32.double(), -1) @ v.double()).transpose(1, 2).reshape(B * T, D)
    out = torch.empty((B * T, D), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().hirest_attention_f32(qkv.to(dev).data_ptr(), out.data_ptr(), B, T, H, 64, 0.125, -10000.0,
                                                ops.stream_ptr()), "attention_f32")
    # q.k summation order differs from torch's by ~1e-6, which the 9.8e-4 quantisation can turn into one ulp(1e4)
    # on isolated scores: bound = a few 1e-3 relative on the probabilities
    assert (out.cpu().double() - ref).abs().max().item() < 5e-3 * v.abs().max().item()


@pytest.mark.parametrize("dh", [32, 20, 48])
def test_attention_f32_narrow_heads_equal_zero_padded_wide_ones(dev, dh):
    """Heads narrower than 64 take the narrowest instantiation that holds them (MiniLM's 32-wide heads: the 32-wide one) instead of
    being zero-padded to 64 by the caller: same products in the same order, so the context equals the padded run bit for bit, and
    the fp64 softmax within the usual bound."""
    from hirest_amd import _lib, ops
    lib, st = _lib.load(), ops.stream_ptr()
    B, T, H = 3, 45, 12
    qkv = synth.tensor(f"an.{dh}", (B * T, 3, H, dh), 1.0, 5)
    out = torch.empty((B * T, H * dh), dtype=torch.float32, device=dev)
    q_ = qkv.reshape(B * T, 3 * H * dh).to(dev)
    _lib.check(lib.hirest_attention_f32(q_.data_ptr(), out.data_ptr(), B, T, H, dh, dh ** -0.5, 0.0, st), "attention_f32 narrow")
    wide = torch.zeros((B * T, 3, H, 64)); wide[..., :dh] = qkv
    w_ = wide.reshape(B * T, 3 * H * 64).to(dev)
    outw = torch.empty((B * T, H * 64), dtype=torch.float32, device=dev)
    _lib.check(lib.hirest_attention_f32(w_.data_ptr(), outw.data_ptr(), B, T, H, 64, dh ** -0.5, 0.0, st), "attention_f32 padded")
    assert torch.equal(out.view(B * T, H, dh), outw.view(B * T, H, 64)[..., :dh])
    q, k, v = qkv.reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4).double()
    ref = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(B * T, H * dh)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5


def test_attention_f32_one_wave_blocks_equal_four_wave_blocks(dev):
    """Calls with at most 64 queries per sequence run 32-query one-wave blocks instead of 128-query four-wave blocks (the sentence
    encoder's short sentences): a query's arithmetic does not depend on the block it sits in, so 100 queries in one call (four-wave
    form) equal the same queries in two calls of 50 (one-wave form) bit for bit — same keys and values."""
    from hirest_amd import _lib, ops
    lib, st = _lib.load(), ops.stream_ptr()
    B, T, H, D = 2, 100, 12, 768                            # (100 queries = 4 waves: four-wave blocks; the 257-query case below: three-wave)
    qkv = synth.tensor("aw.qkv", (B, T, 3 * D), 1.0, 6).to(dev)
    full = torch.empty((B, T, D), dtype=torch.float32, device=dev)
    _lib.check(lib.hirest_attention_f32(qkv.data_ptr(), full.data_ptr(), B, T, H, 64, 0.125, -10000.0, st), "four-wave")
    for b in range(B):
        k, v = qkv[b, :, D:2 * D], qkv[b, :, 2 * D:]
        for lo in (0, 50):
            q = qkv[b, lo:lo + 50, :D]
            part = torch.empty((50, D), dtype=torch.float32, device=dev)
            _lib.check(lib.hirest_attention_f32_qkv(q.data_ptr(), 3 * D, k.data_ptr(), v.data_ptr(), 3 * D, part.data_ptr(), 1, 50, T, H, 64, 0.125,
                                                    -10000.0, 0.0, st), "one-wave")
            assert torch.equal(part, full[b, lo:lo + 50]), (b, lo)
    # 257 queries (9 waves) take three-wave blocks; the first 128 of them asked for alone take one four-wave block
    T2 = 257
    qkv2 = synth.tensor("aw.qkv2", (1, T2, 3 * D), 1.0, 6).to(dev)
    full2 = torch.empty((T2, D), dtype=torch.float32, device=dev)
    _lib.check(lib.hirest_attention_f32(qkv2.data_ptr(), full2.data_ptr(), 1, T2, H, 64, 0.125, 0.0, st), "three-wave")
    part2 = torch.empty((128, D), dtype=torch.float32, device=dev)
    _lib.check(lib.hirest_attention_f32_qkv(qkv2.data_ptr(), 3 * D, qkv2[0, :, D:].data_ptr(), qkv2[0, :, 2 * D:].data_ptr(), 3 * D, part2.data_ptr(),
                                            1, 128, T2, H, 64, 0.125, 0.0, 0.0, st), "four-wave")
    assert torch.equal(part2, full2[:128])


def _case(golden_dir, case):
    from hirest_amd.synth import joint_inputs
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, "joint_schema.json"))).items()}
    pred = json.load(open(os.path.join(golden_dir, "joint_predictions.json")))[case]
    g = np.load(os.path.join(golden_dir, f"joint_{case}.npz"))
    return shapes, pred, g, joint_inputs(f"joint.{case}", pred["B"], pred["T"], 41)


# c*: SURVEY 8d C4 sizes (B = 5; c300 = the notebook's own operating point), d300: args.py:27's default eval batch of 32; real-reference goldens
# precision: 'bf16x3' = the encoder's linear layers on split operands (csrc/joint_x3.hip, MomentModel.set_precision) — the same gates: logits
# within 2e-3, frame indices and boundary lists EXACT on every real-reference golden
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", ["a", "b", "c120", "c300", "c571", "c1855", "d300"])
def test_moment_model_vs_reference(dev, golden_dir, case, precision):
    import hirest_amd
    shapes, pred, g, (vis, asr, text, vis_mask, moment_mask, bounds) = _case(golden_dir, case)
    sd = synth.joint_state_dict(shapes, 31)

    class Args:
        visual_num_hidden_layers = 2
        moment_segmentation_difference_threshold = 0.5
        moment_segmentation_max_iterations = 20
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=Args(), clip_model=None)
    res = model.load_state_dict(sd, strict=False)
    assert not res.missing_keys                                   # every parameter we own is in the reference schema
    model = model.to(dev).eval().set_precision(precision)
    B, T = pred["B"], pred["T"]
    out = model.forward_moment_retrieval(vis.to(dev), text.to(dev), vis_mask.to(dev), moment_mask.to(dev), asr.to(dev))
    rows = list(g["rows"])
    fr = out["feats"][:, rows].cpu().numpy()
    assert np.abs(fr - g["feats_rows"]).max() / np.abs(g["feats_rows"]).max() < 1e-3
    valid = vis_mask.numpy() == 1
    assert np.abs(out["start_logits"].cpu().numpy() - g["start_logits"])[valid].max() < 2e-3
    assert np.abs(out["end_logits"].cpu().numpy() - g["end_logits"])[valid].max() < 2e-3
    batch = {"tasks": ["moment_retrieval"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask,
             "asr_feats": asr, "text_feat": text}
    assert model.test_step(batch)["prediction"] == pred["pred_moment_retrieval"]       # exact frame indices
    batch = {"tasks": ["moment_segmentation"], "vis_feats": vis, "vis_mask": vis_mask, "asr_feats": asr,
             "text_feat": text, "moment_bound_frames": bounds}
    res = model.test_step(batch, return_trace=True)
    assert np.abs(res["first_logits"].cpu().numpy() - g["seg_logits_iter0"]).max() < 2e-3
    assert res["prediction"] == pred["pred_segmentation"]                               # exact boundary lists (IoU 1.0)
    with pytest.raises(NotImplementedError):
        model.test_step({"tasks": ["something_else"]})


# c3 / c5: BASELINE configs[4]'s own operating point (B = 5, beam 3 / 5); d3 / d5: the reference's default evaluation batch
# (args.py:27 --eval_batch_size 32: 96 / 160 beam rows per word, all three trim branches in one batch)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", ["a", "b", "c3", "c5", "d3", "d5"])
def test_step_captioning_vs_reference(dev, golden_dir, case, precision):
    """BASELINE configs[4] in miniature: trim_feats + encoder + beam-searched decoder; token ids exact vs the
    REAL reference MomentModel.test_step (tests/golden/caption_predictions.json)."""
    import hirest_amd
    from hirest_amd.synth import joint_inputs
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, "joint_schema.json"))).items()}
    sd = synth.joint_state_dict(shapes, 31)
    sd["clip4cap_model.decoder.classifier.cls.predictions.bias"][102] += 1.5
    pred = json.load(open(os.path.join(golden_dir, "caption_predictions.json")))[case]
    g = np.load(os.path.join(golden_dir, f"caption_{case}.npz"))
    B, T = pred["B"], pred["T"]
    vis, asr, text, vis_mask, _, _ = joint_inputs(f"cap.{case}", B, T, 47)
    moment_mask = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        moment_mask[b, 5 + b:5 + b + pred["lens"][b]] = 1
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval().set_precision(precision)
    trimmed = model._trim(vis.to(dev), moment_mask.to(dev), 20)
    assert np.array_equal(trimmed[:, [0, 7, 19]].cpu().numpy(), g["trimmed_rows"])
    batch = {"tasks": ["step_captioning"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask,
             "asr_feats": asr, "text_feat": text}
    res = model.test_step(batch, num_beams=pred["beams"], return_ids=True)
    assert res["prediction"] == pred["prediction"]
    assert [" ".join(str(i) for i in h) for h in res["token_ids"]] == pred["prediction"]
    # the read-out of the finished search on the device (default, hirest_beam_backtrack) and on the host (BeamState) pick the same words
    assert model.caption_device_readout
    model.caption_device_readout = False
    assert model.test_step(batch, num_beams=pred["beams"], return_ids=True) == res
    model.caption_device_readout = True
    # default = decoding with the self-attention K / V of earlier positions kept per beam; recomputing the whole prefix every
    # step, as the reference does (train.py:547-566), gives the same tokens
    assert model.caption_kv_cache
    model.caption_fused_tail = False        # log-softmax, top-k and beam bookkeeping as separate kernels
    assert model.test_step(batch, num_beams=pred["beams"], return_ids=True)["token_ids"] == res["token_ids"]
    model.caption_fused_tail = True
    from hirest_amd import _lib
    _lib.check(_lib.load().hirest_caption_select(1), "select")    # LayerNorms / embedding as separate kernels
    try:
        assert model.test_step(batch, num_beams=pred["beams"], return_ids=True)["token_ids"] == res["token_ids"]
    finally:
        _lib.load().hirest_caption_select(0)
    model.caption_kv_cache = False
    assert model.test_step(batch, num_beams=pred["beams"], return_ids=True)["token_ids"] == res["token_ids"]
    # greedy decoding (one beam) and a wider beam through both paths
    for nb in (1, 7) if B <= 5 else (2,):
        model.caption_kv_cache = False
        ref = model.test_step(batch, num_beams=nb, return_ids=True)["token_ids"]
        model.caption_kv_cache = True
        assert model.test_step(batch, num_beams=nb, return_ids=True)["token_ids"] == ref, nb


@pytest.mark.parametrize("M", [25, 15, 32, 3])
def test_lm_head_tile_maxima(dev, M):
    """hirest_gemm_f32_ln_colmax: same output as hirest_gemm_f32_ln, and colmax[m][t] = max of out[m][16 t .. 16 t + 15]"""
    from hirest_amd import _lib, ops
    lib, st = _lib.load(), ops.stream_ptr()
    N, K = 30528, 768
    x = synth.tensor("tm.x", (M, K), 2.0, 3).to(dev)
    w = synth.tensor("tm.w", (N, K), 0.05, 3).to(dev)
    b = synth.tensor("tm.b", (N,), 0.3, 3).to(dev)
    g = (1.0 + synth.tensor("tm.g", (K,), 0.2, 3)).to(dev)
    be = synth.tensor("tm.be", (K,), 0.2, 3).to(dev)
    ref = torch.empty((M, N), device=dev); out = torch.empty((M, N), device=dev); cm = torch.full((M, N // 16), 7.0, device=dev)
    _lib.check(lib.hirest_gemm_f32_ln(x.data_ptr(), K, None, None, None, g.data_ptr(), be.data_ptr(), 1e-12, None, 0, w.data_ptr(), K, b.data_ptr(),
                                      None, 0, ref.data_ptr(), N, M, N, K, 0, st), "gemm_ln")
    _lib.check(lib.hirest_gemm_f32_ln_colmax(x.data_ptr(), K, g.data_ptr(), be.data_ptr(), 1e-12, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N,
                                             cm.data_ptr(), M, N, K, st), "gemm_ln_colmax")
    assert torch.equal(out, ref)
    assert torch.equal(cm, ref.reshape(M, N // 16, 16).max(-1).values)


@pytest.mark.parametrize("M,N", [(160, 30528), (96, 30528), (100, 30528), (60, 30528), (33, 30528), (256, 30528), (17, 8200), (130, 8200),
                                 (75, 1000), (330, 30528), (160, 16), (64, 8216)])
def test_lm_head_row_groups_equal_the_64x64_kernel(dev, M, N):
    """hirest_gemm_f32_rows_colmax (a merged beam search's LM head: row groups of 32 - 80 rows, A fragments in registers, W streamed
    through LDS-DMA rings, rotating reducer wave) against the 64x64 kernel: same summation order -> bit for bit, every row-tile /
    row-group split (mt 2..5, ng 1..5), a last tile cut by N, and the 16-column tile maxima; hirest_gemm_f32's automatic dispatch
    takes it for 33 .. 256 rows."""
    from hirest_amd import _lib, ops
    from hirest_amd.moment_model import MomentModel
    lib, st = _lib.load(), ops.stream_ptr()
    K = 768
    a = synth.tensor("rs.a", (M, K), 1.0, 3).to(dev)
    w = synth.tensor("rs.w", (N, K), 0.05, 3).to(dev)
    b = synth.tensor("rs.b", (N,), 0.3, 3).to(dev)
    _lib.check(lib.hirest_gemm_f32_select_kernel(1), "select")
    try:
        ref = MomentModel._gemm(a, w, b)
        ref0 = MomentModel._gemm(a, w, None)
    finally:
        lib.hirest_gemm_f32_select_kernel(0)
    nt = (N + 15) // 16
    out = torch.full((M, N), 3.0, device=dev); cm = torch.full((M, nt), 7.0, device=dev)
    _lib.check(lib.hirest_gemm_f32_rows_colmax(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N, cm.data_ptr(), M, N, K, st), "rows")
    assert torch.equal(out, ref)
    pad = torch.full((M, nt * 16), float("-inf"), device=dev); pad[:, :N] = ref
    assert torch.equal(cm, pad.reshape(M, nt, 16).max(-1).values)
    out.fill_(3.0)
    _lib.check(lib.hirest_gemm_f32_rows_colmax(a.data_ptr(), K, w.data_ptr(), K, None, out.data_ptr(), N, None, M, N, K, st), "rows, no bias")
    assert torch.equal(out, ref0)
    assert torch.equal(MomentModel._gemm(a, w, b), ref)               # automatic dispatch
    assert lib.hirest_gemm_f32_rows_colmax(a.data_ptr(), K, w.data_ptr(), K, None, out.data_ptr(), N, None, M, N, 512, st) == -2


def test_round3_fp32_kernels_randomised_bit_equality(dev):
    """tools/f32_stress.py: random shapes / options — hirest_gemm_f32's automatic dispatch (16-column LDS-DMA kernel, split form) vs the
    forced 64x64 kernel, hirest_gemm_f32_ln vs hirest_layernorm + hirest_gemm_f32, hirest_attention_f32_decode vs gather +
    hirest_attention_f32_qkv.  (This is the test that caught -ffp-contract contracting a shared inline function differently in two
    kernels: the fused multiply-adds of the shared arithmetic are written out since.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "f32_stress.py"), "90"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gemm_f32: 90 random problems, 0 mismatches" in r.stdout


@pytest.mark.parametrize("t_hist,newkey,addc", [(0, True, 0.0), (5, True, 0.0), (31, True, 0.0), (32, True, 0.0), (47, True, 0.0), (70, True, 0.0),
                                                 (20, False, -10000.0), (64, False, -10000.0)])
def test_attention_f32_decode_equals_gather_then_attention(dev, t_hist, newkey, addc):
    """hirest_attention_f32_decode (one wave per (row, head), history read through the parent row, vector FMAs in the MFMA's order)
    against the gathered history + hirest_attention_f32_qkv: context rows and the re-gathered K / V bit for bit"""
    from hirest_amd import _lib, ops
    lib, st = _lib.load(), ops.stream_ptr()
    R, H, D = 25, 12, 768
    T = t_hist + (1 if newkey else 0)
    qkv = synth.tensor(f"ad.q.{t_hist}", (R, 3 * D), 1.5, 4).to(dev)
    kh = synth.tensor(f"ad.k.{t_hist}", (R, max(t_hist, 1), D), 1.5, 4).to(dev)
    vh = synth.tensor(f"ad.v.{t_hist}", (R, max(t_hist, 1), D), 1.5, 4).to(dev)
    parent = ((torch.arange(R) * 7 + 3) % R).to(torch.int32).to(dev) if newkey else None
    # reference: gather the history by parent row, append the newest key, run the general kernel with Tq = 1
    src = parent.long() if parent is not None else torch.arange(R, device=dev)
    kcat = kh[src][:, :t_hist]; vcat = vh[src][:, :t_hist]
    if newkey:
        kcat = torch.cat([kcat, qkv[:, None, D:2 * D]], 1); vcat = torch.cat([vcat, qkv[:, None, 2 * D:]], 1)
    kcat, vcat = kcat.contiguous(), vcat.contiguous()
    ref = torch.empty((R, D), device=dev)
    _lib.check(lib.hirest_attention_f32_qkv(qkv.data_ptr(), 3 * D, kcat.data_ptr(), vcat.data_ptr(), D, ref.data_ptr(), R, 1, T, H, 64, 0.125,
                                            addc, 0.0, st), "attention")
    out = torch.empty((R, D), device=dev)
    ko = torch.full((R, T, D), 7.0, device=dev); vo = torch.full((R, T, D), 7.0, device=dev)
    _lib.check(lib.hirest_attention_f32_decode(qkv.data_ptr(), 3 * D, kh.data_ptr() if t_hist else None, vh.data_ptr() if t_hist else None, D,
                                               parent.data_ptr() if parent is not None else None, t_hist,
                                               qkv.data_ptr() + 4 * D if newkey else None, qkv.data_ptr() + 8 * D if newkey else None, 3 * D,
                                               ko.data_ptr(), vo.data_ptr(), out.data_ptr(), R, H, 0.125, addc, 0.0, st), "decode attention")
    assert torch.equal(out, ref)
    assert torch.equal(ko, kcat) and torch.equal(vo, vcat)


@pytest.mark.parametrize("M,N,K,act,embed", [(25, 30528, 768, 0, False), (15, 30528, 768, 0, False), (32, 8200, 768, 1, False), (25, 2304, 768, 0, True), (25, 768, 768, 0, False), (15, 3072, 768, 1, False), (7, 100, 256, 2, False),
                                             (32, 768, 1024, 3, True), (1, 36, 512, 0, False),
                                             # merged beam searches: 60 - 160 rows per word, 16-row blocks across blockIdx.y
                                             (160, 2304, 768, 0, True), (96, 3072, 768, 1, False), (100, 768, 768, 0, False), (256, 768, 768, 1, False),
                                             (33, 36, 256, 2, False), (160, 30528, 768, 0, False), (96, 8200, 768, 1, False), (300, 768, 768, 1, True),
                                             (40, 64, 768, 0, False), (129, 3072, 768, 1, False)])
def test_gemm_f32_ln_equals_layernorm_then_gemm(dev, M, N, K, act, embed):
    """hirest_gemm_f32_ln (LayerNorm / token + position embedding as the GEMM's prologue) against hirest_embedding_pos_fwd_f32 +
    hirest_layernorm + hirest_gemm_f32: output and the normalised rows bit for bit"""
    from hirest_amd import _lib, ops
    from hirest_amd.moment_model import MomentModel
    lib, st = _lib.load(), ops.stream_ptr()
    w = synth.tensor("gl.w", (N, K), 0.05, 3).to(dev)
    bias = synth.tensor("gl.b", (N,), 0.3, 3).to(dev)
    resid = synth.tensor("gl.r", (M, N), 1.0, 3).to(dev)
    g = (1.0 + synth.tensor("gl.g", (K,), 0.2, 3)).to(dev)
    be = synth.tensor("gl.be", (K,), 0.2, 3).to(dev)
    if embed:
        table = synth.tensor("gl.t", (500, K), 1.0, 3).to(dev)
        pos = synth.tensor("gl.p", (48, K), 0.5, 3).to(dev)
        ids = ((torch.arange(M) * 37 + 11) % 500).to(torch.int32).to(dev)
        pids = torch.full((M,), 9, dtype=torch.int32, device=dev)
        x = torch.empty((M, K), device=dev)
        _lib.check(lib.hirest_embedding_pos_fwd_f32(ids.data_ptr(), pids.data_ptr(), table.data_ptr(), pos.data_ptr(), x.data_ptr(), M, K, st), "emb")
    else:
        x = synth.tensor("gl.x", (M, K), 2.0, 3).to(dev)
    ln = torch.empty((M, K), device=dev)
    _lib.check(lib.hirest_layernorm(x.data_ptr(), K, None, g.data_ptr(), be.data_ptr(), 1e-12, ln.data_ptr(), K, 1, M, K, st), "ln")
    want_ln = N < 8192 or M > 32                                # the LM-head form for <= 32 rows (persistent blocks) is taken without ln_out
    for use_resid in (True, False):
        # with a residual: the 16-row blocks (up to 256 rows, layer widths); without one, above 32 rows and K = 768: the row-group
        # streaming kernel (any number of rows, any width)
        if use_resid and (M > 256 or (M > 32 and N >= 8192)):
            continue
        ref = MomentModel._gemm(ln, w, bias, resid=resid if use_resid else None, act=act)
        out = torch.full((M, N), 5.0, device=dev)
        ln2 = torch.full((M, K), 5.0, device=dev)
        _lib.check(lib.hirest_gemm_f32_ln(None if embed else x.data_ptr(), K, ids.data_ptr() if embed else None,
                                          table.data_ptr() if embed else None, pos[9].data_ptr() if embed else None, g.data_ptr(), be.data_ptr(),
                                          1e-12, ln2.data_ptr() if want_ln else None, K, w.data_ptr(), K, bias.data_ptr(),
                                          resid.data_ptr() if use_resid else None, N, out.data_ptr(), N, M, N, K, act, st), "gemm_ln")
        assert not want_ln or torch.equal(ln2, ln), use_resid
        assert torch.equal(out, ref), use_resid
        if not use_resid and M > 32 and K == 768:               # the row-group form's A/B modes (ring depth / blocks per CU): same bits
            for mode in (0, 1, 2):
                _lib.check(lib.hirest_gemm_f32_rows_ln_mode(mode), "mode")
                out.fill_(5.0)
                try:
                    _lib.check(lib.hirest_gemm_f32_ln(None if embed else x.data_ptr(), K, ids.data_ptr() if embed else None,
                                                      table.data_ptr() if embed else None, pos[9].data_ptr() if embed else None, g.data_ptr(),
                                                      be.data_ptr(), 1e-12, None, K, w.data_ptr(), K, bias.data_ptr(), None, N, out.data_ptr(), N,
                                                      M, N, K, act, st), "gemm_ln mode")
                finally:
                    lib.hirest_gemm_f32_rows_ln_mode(2)
                assert torch.equal(out, ref), mode
    if K != 768:                                                # (K = 768 takes the row-group streaming kernel for any number of rows above 32)
        assert lib.hirest_gemm_f32_ln(x.data_ptr(), K, None, None, None, g.data_ptr(), be.data_ptr(), 1e-12, None, 0, w.data_ptr(), K, None, None, 0,
                                      out.data_ptr(), N, 257, N, K, act, st) == -2     # more than 256 rows: HIREST_E_SHAPE


@pytest.mark.parametrize("B,beam,step", [(5, 5, 3), (5, 3, 0), (2, 7, 1), (3, 1, 2)])
def test_beam_tail_equals_log_softmax_topk_advance(dev, B, beam, step):
    """hirest_caption_beam_tail (two kernels, log-probabilities never materialised) against hirest_log_softmax_f32 +
    hirest_topk_f32_ws + hirest_beam_advance: every output bit for bit, with duplicated logits (ties), one sample already done
    and one beam emitting [SEP]"""
    import ctypes as C
    from hirest_amd import _lib, ops
    lib = _lib.load()
    Vp, R, max_steps, eos = 30528, B * beam, 8, 102
    x = synth.tensor(f"bt.{B}.{beam}", (R, Vp), 3.0, 5)
    x[:, 30522:] = -1.0e30                                     # padded vocabulary entries
    x[0, 777] = x[0, 12345] = x[0].max() + 1.0                 # an exact tie at the top of one row
    x[R - beam, eos] = x[R - beam].max() + 20.0                # the last sample's best continuation (from its beam 0) is [SEP]
    x = x.to(dev)
    add = synth.tensor("bt.add", (R,), 2.0, 5).to(dev)
    if step == 0:
        add = torch.full((B, beam), -3.0e38); add[:, 0] = 0.0; add = add.reshape(-1).to(dev)
    def state():
        done = torch.zeros((B,), dtype=torch.int32, device=dev)
        if B > 2:
            done[1] = 1
        return dict(scores=torch.zeros((R,), device=dev), tokens=torch.full((B, max_steps, beam), -7, dtype=torch.int32, device=dev),
                    backptr=torch.full((B, max_steps, beam), -7, dtype=torch.int32, device=dev),
                    n_steps=torch.zeros((B,), dtype=torch.int32, device=dev), done=done,
                    ids=torch.full((R,), -7, dtype=torch.int32, device=dev), parents=torch.full((R,), -7, dtype=torch.int32, device=dev),
                    nadd=torch.full((R,), -7.0, device=dev))
    a, b = state(), state()
    st = ops.stream_ptr()
    logp = torch.empty_like(x)
    _lib.check(lib.hirest_log_softmax_f32(x.data_ptr(), Vp, add.data_ptr(), logp.data_ptr(), Vp, R, Vp, st), "ls")
    tk = torch.empty(max(int(lib.hirest_topk_workspace_bytes(B, beam * Vp, beam)), 16), dtype=torch.uint8, device=dev)
    val = torch.empty((B, beam), device=dev); idx = torch.empty((B, beam), dtype=torch.int32, device=dev)
    _lib.check(lib.hirest_topk_f32_ws(logp.data_ptr(), None, B, beam * Vp, beam, idx.data_ptr(), val.data_ptr(), tk.data_ptr(), tk.numel(),
                                      st), "topk")
    _lib.check(lib.hirest_beam_advance(val.data_ptr(), idx.data_ptr(), B, beam, Vp, step, max_steps, eos, a["scores"].data_ptr(),
                                       a["tokens"].data_ptr(), a["backptr"].data_ptr(), a["n_steps"].data_ptr(), a["done"].data_ptr(),
                                       a["ids"].data_ptr(), a["parents"].data_ptr(), a["nadd"].data_ptr(), st), "advance")
    ws = torch.empty(max(int(lib.hirest_caption_beam_tail_workspace_bytes(B, beam, Vp)), 16), dtype=torch.uint8, device=dev)
    host = torch.full((B,), -1, dtype=torch.int32).pin_memory()
    _lib.check(lib.hirest_caption_beam_tail(x.data_ptr(), Vp, add.data_ptr(), B, beam, Vp, step, max_steps, eos, b["scores"].data_ptr(),
                                            b["tokens"].data_ptr(), b["backptr"].data_ptr(), b["n_steps"].data_ptr(), b["done"].data_ptr(),
                                            b["ids"].data_ptr(), b["parents"].data_ptr(), b["nadd"].data_ptr(), host.data_ptr(),
                                            ws.data_ptr(), ws.numel(), st), "tail")
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(host & 1, a["done"].cpu()) and bool(((host >> 1) == step + 1).all())   # stamped with the step
    assert a["done"][B - 1].item() == 1                         # [SEP] on the best beam finished the last sample


def test_clip_text_ids_path_equals_text_feat_path(dev, golden_dir):
    """The reference's only way to text features is clip_model.encode_text(batch['clip_text_ids']) inside test_step
    (modeling.py:286,364,568).  A MomentModel that built its own (tiny, 1024-d) EVA_CLIP the reference's way runs all three
    tasks from token ids; feeding the same encoder's output as batch['text_feat'] must give bit-identical predictions,
    and the ids really matter (other prompts -> other text features)."""
    import hirest_amd
    from hirest_amd.synth import joint_inputs
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, "joint_schema.json"))).items()}
    sd = synth.joint_state_dict(shapes, 31)
    sd["clip4cap_model.decoder.classifier.cls.predictions.bias"][102] += 1.5

    class Args:
        clip_model_name = "EVA_CLIP_tiny_e1024_test"
        clip_pretrained = "synth:11"
        visual_num_hidden_layers = 2
        moment_segmentation_difference_threshold = 0.5
        moment_segmentation_max_iterations = 20
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=Args())          # builds + freezes its CLIP
    res = model.load_state_dict(sd, strict=False)                                     # BEST.pth-style: no clip_model.* keys
    assert all(k.startswith("clip_model.") for k in res.missing_keys)
    model = model.to(dev).eval()
    B, T = 3, 64
    vis, asr, _, vis_mask, moment_mask, bounds = joint_inputs("ids.path", B, T, 43)
    prompts = json.load(open(os.path.join(golden_dir, "test_prompts.json")))[:B]
    ids = hirest_amd.tokenize(prompts)
    text = model.clip_model.encode_text(ids.to(dev)).float()
    assert tuple(text.shape) == (B, 1024)
    other = model.clip_model.encode_text(hirest_amd.tokenize(prompts[::-1]).to(dev)).float()
    assert not torch.equal(text, other)
    cap_mask = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate([7, 20, 37]):
        cap_mask[b, 5 + b:5 + b + n] = 1
    batches = {
        "moment_retrieval": {"vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask, "asr_feats": asr},
        "moment_segmentation": {"vis_feats": vis, "vis_mask": vis_mask, "asr_feats": asr, "moment_bound_frames": bounds},
        "step_captioning": {"vis_feats": vis, "vis_mask": vis_mask, "moment_mask": cap_mask, "asr_feats": asr},
    }
    for task, b in batches.items():
        kw = {"num_beams": 3} if task == "step_captioning" else {}
        from_ids = model.test_step(dict(b, tasks=[task], clip_text_ids=ids), **kw)["prediction"]
        from_feat = model.test_step(dict(b, tasks=[task], text_feat=text), **kw)["prediction"]
        assert from_ids == from_feat, task
        assert len(from_ids) == B
    # a model without a CLIP and without text_feat has nothing to encode the ids with
    bare = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=Args(), clip_model=None)
    bare.load_state_dict(sd, strict=False)
    bare = bare.to(dev).eval()
    with pytest.raises(RuntimeError):
        bare.test_step(dict(batches["moment_retrieval"], tasks=["moment_retrieval"], clip_text_ids=ids))


def test_caption_batches_with_three_in_flight_equal_sequential_calls(dev, golden_dir):
    """MomentModel.caption_batches: loader batches captioned concurrently (one HIP stream + host thread each) give exactly the token
    ids of one test_step call per batch — every call owns its buffers and every kernel is batch-invariant — including the batch the
    real reference captioned (caption_predictions.json c5)."""
    import hirest_amd
    from hirest_amd.synth import joint_inputs, CAPTION_CASES
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, "joint_schema.json"))).items()}
    sd = synth.joint_state_dict(shapes, 31)
    sd["clip4cap_model.decoder.classifier.cls.predictions.bias"][102] += 1.5
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    B, T, beams, lens = CAPTION_CASES["c5"]
    batches = []
    for i in range(7):
        vis, asr, text, vis_mask, _, _ = joint_inputs("cap.c5" if i == 0 else f"cap.pipe{i}", B, T, 47 + 3 * i)
        mm = torch.zeros(B, T, dtype=torch.long)
        for b in range(B):
            n = lens[b] if i == 0 else 6 + ((5 * i + 3 * b) % 30)         # shorter than, equal to and longer than the 20 trimmed frames
            mm[b, 5 + b:5 + b + n] = 1
        batches.append({"tasks": ["step_captioning"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": mm, "asr_feats": asr,
                        "text_feat": text})
    want = [model.test_step(b, num_beams=beams, return_ids=True) for b in batches]
    pred = json.load(open(os.path.join(golden_dir, "caption_predictions.json")))["c5"]
    assert want[0]["prediction"] == pred["prediction"]
    # a loader's last batch is usually smaller: it takes the eager path inside the pipelined run
    small = {k: (v[:3] if torch.is_tensor(v) else v) for k, v in batches[3].items()}
    batches.append(small)
    want.append(model.test_step(small, num_beams=beams, return_ids=True))
    for streams, graphs in ((3, True), (2, True), (8, False), (3, False)):      # word steps replayed from hipGraphs / issued eagerly
        got = model.caption_batches(batches, num_beams=beams, streams=streams, return_ids=True, graphs=graphs, merge=False)
        assert [g["token_ids"] for g in got] == [w["token_ids"] for w in want], (streams, graphs)
        assert [g["prediction"] for g in got] == [w["prediction"] for w in want]
    # round 5, the default: ONE beam search over the union of consecutive batches' beam rows (up to rows_in_flight; the decoder's
    # weights are streamed once per word for all of them), each loader batch handed its own slice: 160 rows = 6 + 2 batches,
    # 50 rows = pairs, 25 rows = nothing to merge, 1000 = all 38 videos (190 rows) in one search
    for rows, streams, graphs in ((None, 2, True), (50, 3, True), (50, 1, False), (25, 2, True), (1000, 1, True)):
        got = model.caption_batches(batches, num_beams=beams, streams=streams, return_ids=True, graphs=graphs, rows_in_flight=rows)
        assert [g["token_ids"] for g in got] == [w["token_ids"] for w in want], (rows, streams, graphs)
        assert [g["prediction"] for g in got] == [w["prediction"] for w in want]
    # more than 256 rows in one search (38 videos x 7 beams = 266): the LayerNorm-GEMMs in their row-group form only, same tokens
    want7 = [model.test_step(b, num_beams=7, return_ids=True)["token_ids"] for b in batches]
    got7 = model.caption_batches(batches, num_beams=7, return_ids=True, rows_in_flight=1000)
    assert [g["token_ids"] for g in got7] == want7
    assert model.caption_batches(batches[:1], num_beams=beams, streams=3, return_ids=True)[0]["token_ids"] == want[0]["token_ids"]
    assert model.caption_batches([], num_beams=beams) == []
    # early stop inside the graph path: with [SEP] made the likeliest word every search ends after its first word, two chunks late at most
    with torch.no_grad():
        dict(model.named_parameters())["clip4cap_model.decoder.classifier.cls.predictions.bias"][102] += 50.0
    short = [model.test_step(b, num_beams=beams, return_ids=True)["token_ids"] for b in batches[:4]]
    assert all(len(h) <= 2 for hyp in short for h in hyp)
    assert [g["token_ids"] for g in model.caption_batches(batches[:4], num_beams=beams, streams=2, return_ids=True)] == short
