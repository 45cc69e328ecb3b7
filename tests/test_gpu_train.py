"""SURVEY 8f-4 on the GPU: hirest_amd.MomentModel.train_step (csrc/train.hip + joint.hip through hirest_amd/train.py) against
the REAL reference's loss and gradients (tests/golden/train_*.npz, made by make_golden.py gen_train with the reference's
autograd), plus the training-loop contract of run.py:238-295 (loss.backward -> clip_grad_norm_ -> optimizer step)."""
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


def _setup(golden_dir, case, dev):
    import hirest_amd
    from hirest_amd.synth import joint_inputs, train_targets, caption_targets, TRAIN_CASES
    B, T = TRAIN_CASES[case]
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, "joint_schema.json"))).items()}
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    model.load_state_dict(synth.joint_state_dict(shapes, 31), strict=False)
    model = model.to(dev)
    vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs(f"train.{case}", B, T, 53)
    st, et, seg, prev = train_targets(f"train.{case}", B, T, 53, bounds)
    batch = {"tasks": ["moment_retrieval"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask, "asr_feats": asr,
             "text_feat": text, "moment_retrieval_start_target": st, "moment_retrieval_end_target": et}
    seg_batch = {"tasks": ["moment_segmentation"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask, "asr_feats": asr,
                 "text_feat": text, "prev_boundary_mask": prev, "moment_segmentation_target": seg}
    cap_mask = torch.zeros(B, T, dtype=torch.long)
    for b_, n_ in enumerate([7, 20, 37][:B]):
        cap_mask[b_, 5 + b_:5 + b_ + n_] = 1
    cap_batch = {"tasks": ["step_captioning"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": cap_mask, "asr_feats": asr,
                 "text_feat": text, "target_text": caption_targets(f"train.{case}", B, 48, 53)}
    return model, batch, seg_batch, cap_batch, np.load(os.path.join(golden_dir, f"train_{case}.npz"))


# gemm: train.GEMM_PRECISION — 'bf16x3' = the encoder blocks' forward and dX products on split operands (csrc/train_block.hip, precision 1);
# the same gates on every gradient tensor, the loss within 2e-4 instead of 1e-5 (its products carry ~16 bits)
@pytest.mark.parametrize("gemm", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", ["a", "b"])
def test_train_step_loss_and_gradients_vs_reference(dev, golden_dir, case, gemm, monkeypatch):
    from hirest_amd import train
    monkeypatch.setattr(train, "GEMM_PRECISION", gemm)
    model, batch, seg_batch, cap_batch, g = _setup(golden_dir, case, dev)
    model.eval()                                   # dropout off: the arithmetic the goldens pin
    worst = 0.0
    for b, prefix in ((batch, ""), (seg_batch, "seg."), (cap_batch, "cap.")):
        for p in model.parameters():
            p.grad = None
        loss = model.train_step(b)["loss"]
        assert loss.requires_grad and loss.dim() == 0
        ref = float(g[prefix + "loss"])
        assert abs(loss.item() - ref) <= (1e-5 if gemm == "fp32" else 2e-4) * abs(ref), (prefix, loss.item(), ref)
        loss.backward()
        names = [str(n) for n in g[prefix + "names"]]
        with_grad = {n for n, p in model.named_parameters() if p.grad is not None}
        assert with_grad == set(names)             # the same tensors the reference's backward reaches; the rest stay None
        named = dict(model.named_parameters())
        for i, n in enumerate(names):
            gr = named[n].grad.detach().double().cpu()
            norm = g[prefix + "norms"][i]
            assert torch.isfinite(gr).all(), n
            # (the key-bias gradient is identically zero in exact arithmetic — softmax ignores a constant added to every key — and
            # ~1e-8 of rounding noise in practice: the 1e-6 floor exempts it from a relative comparison)
            rel = max(0.0, abs(float(gr.norm()) - norm) - 1e-6) / (norm + 1e-12)
            worst = max(worst, rel)
            assert rel <= 1e-3, (prefix, n, rel)
            k = min(8, gr.numel())
            assert np.abs(gr.flatten()[:k].numpy() - g[prefix + "heads"][i][:k]).max() <= 1e-3 * norm + 1e-6, (prefix, n)
            key = prefix + "full." + n
            if key in g.files:
                assert np.abs(gr.numpy().reshape(g[key].shape) - g[key]).max() <= 1e-3 * np.abs(g[key]).max() + 1e-6, (prefix, n)
        print(f"case {case} {prefix or 'retrieval '}loss {loss.item():.7f} (reference {ref:.7f}), {len(names)} gradient tensors")
    print(f"[{gemm}] worst gradient-norm deviation {worst:.2e}")
    with pytest.raises(NotImplementedError):
        model.train_step({"tasks": ["something_else"]})


def test_training_loop_contract_and_dropout(dev, golden_dir):
    """run.py:238-295: results['loss'].backward(); clip_grad_norm_; optim.step(); grads reset.  Train mode switches the four
    dropout sites on (counter-based masks): the loss changes from call to call, stays finite, and a few AdamW steps on one batch
    reduce the eval-mode loss."""
    model, batch, _, cap_batch, g = _setup(golden_dir, "a", dev)
    trainable = [p for n, p in model.named_parameters() if p.requires_grad]
    optim = torch.optim.AdamW(trainable, lr=2e-4)
    model.eval()
    first = model.train_step(batch)["loss"].item()
    model.train()
    seen = set()
    for step in range(4):
        loss = model.train_step(batch)["loss"]
        assert torch.isfinite(loss)
        seen.add(round(loss.item(), 6))
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        assert torch.isfinite(total) and total > 0
        optim.step()
        for p in model.parameters():
            p.grad = None
    assert len(seen) == 4                          # different dropout masks / updated weights every step
    model.eval()
    last = model.train_step(batch)["loss"].item()
    print(f"eval-mode loss {first:.5f} -> {last:.5f} after 4 AdamW steps")
    assert last < first
    # the captioning task through the same loop (train mode: dropout in the decoder as well)
    model.train()
    c0 = None
    for step in range(3):
        loss = model.train_step(cap_batch)["loss"]
        assert torch.isfinite(loss)
        c0 = loss.item() if c0 is None else c0
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        optim.step()
        for p in model.parameters():
            p.grad = None
    model.eval()
    assert model.train_step(cap_batch)["loss"].item() < c0
    # predictions still come out of the updated weights through the inference path
    out = model.test_step(dict(batch, tasks=["moment_retrieval"]))["prediction"]
    assert len(out) == batch["vis_feats"].shape[0]


def test_inference_cache_follows_optimizer_updates(dev, golden_dir):
    """run.py:328-336 validates after every epoch: test_step -> train (in-place AdamW) -> test_step must run on the UPDATED
    weights everywhere, including the fused / padded copies the inference path caches (QKV concatenations, padded LM head,
    head biases).  The second test_step has to equal a model freshly loaded from the updated state_dict, bit for bit."""
    import hirest_amd
    model, batch, seg_batch, cap_batch, _ = _setup(golden_dir, "a", dev)
    model.eval()
    tb = dict(batch, tasks=["moment_retrieval"])
    cb = dict(cap_batch, tasks=["step_captioning"])
    before = model.test_step(tb)["prediction"]
    model.test_step(cb, num_beams=3, return_ids=True)         # fills the decoder part of the cache as well
    optim = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-3)
    for b in (batch, cap_batch, seg_batch, batch):
        model.train_step(b)["loss"].backward()
        optim.step()
        for p in model.parameters():
            p.grad = None
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, "joint_schema.json"))).items()}
    fresh = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=False)
    fresh = fresh.to(dev).eval()
    from hirest_amd.synth import joint_inputs, TRAIN_CASES
    bounds = joint_inputs("train.a", *TRAIN_CASES["a"], 53)[-1]
    sb = dict(batch, tasks=["moment_segmentation"], moment_bound_frames=bounds)
    for tb_, kw in ((tb, {}), (sb, {}), (cb, {"num_beams": 3, "return_ids": True})):
        got, want = model.test_step(tb_, **kw), fresh.test_step(tb_, **kw)
        assert json.dumps(got, default=str, sort_keys=True) == json.dumps(want, default=str, sort_keys=True), tb_["tasks"]
    fl = model.forward_moment_retrieval(batch["vis_feats"].to(dev), batch["text_feat"].to(dev), batch["vis_mask"].to(dev),
                                        batch["moment_mask"].to(dev), batch["asr_feats"].to(dev))
    ff = fresh.forward_moment_retrieval(batch["vis_feats"].to(dev), batch["text_feat"].to(dev), batch["vis_mask"].to(dev),
                                        batch["moment_mask"].to(dev), batch["asr_feats"].to(dev))
    assert torch.equal(fl["start_logits"], ff["start_logits"]) and torch.equal(fl["end_logits"], ff["end_logits"])
    assert before is not None


def test_tiled_training_attention_equals_the_row_kernels_with_dropout_on(dev, golden_dir):
    """The attention products as batched MFMA GEMMs (round 3) against the one-wave-per-row kernels they replace, in TRAIN mode:
    the counter-based dropout masks are regenerated inside operand loads / epilogues of five different products (P as the A
    operand, P^T as the A operand, dP in an epilogue), so the same seed must give the same loss and the same gradients up to
    the summation order — a mis-indexed mask in any one product shows up as an O(1) gradient difference."""
    from hirest_amd import _lib
    lib = _lib.load()
    model, batch, seg_batch, cap_batch, _ = _setup(golden_dir, "b", dev)
    model.train()
    out = {}
    try:
        for which in (0, 1):
            lib.hirest_attention_train_select(which)
            res = []
            for b in (batch, cap_batch):
                for p in model.parameters():
                    p.grad = None
                torch.manual_seed(1234)                          # _train draws the dropout seed from torch's CPU generator
                loss = model.train_step(b)["loss"]
                loss.backward()
                res.append((loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
            out[which] = res
    finally:
        lib.hirest_attention_train_select(1)
    for (l0, g0), (l1, g1) in zip(out[0], out[1]):
        assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
        assert g0.keys() == g1.keys()
        for n in g0:
            d = (g0[n] - g1[n]).norm().item()
            assert d <= 1e-4 * g0[n].norm().item() + 1e-7, (n, d, g0[n].norm().item())


@pytest.mark.gpu
def test_backward_products_in_place_equal_the_transposed_copies():
    """hirest_gemm_f32_layouts (k-major operands read in place by the 64x64 kernel) against the zero-padded transposed copies +
    hirest_gemm_f32: dX and dW bit for bit, including a row count that is no multiple of 16 and the split form"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from hirest_amd import train, synth
    dev = torch.device("cuda:0")
    for R, O, I in ((1500, 768, 768), (1500, 768, 3072), (1500, 3072, 768), (1498, 512, 384), (100, 2304, 768), (37, 64, 36), (240, 768, 30528)):
        dy = synth.tensor(f"lay.dy.{R}.{O}", (R, O), 1.0, 3).to(dev)
        x = synth.tensor(f"lay.x.{R}.{I}", (R, I), 1.0, 3).to(dev)
        w = synth.tensor(f"lay.w.{O}.{I}", (O, I), 0.05, 3).to(dev)
        outs = []
        for flag in (True, False):
            train.LAYOUT_GEMM = flag
            outs.append((train._K.grad_input(dy, w), train._K.grad_weight(dy, x)))
        train.LAYOUT_GEMM = True
        assert torch.equal(outs[0][0], outs[1][0]), ("dX", R, O, I)
        assert torch.equal(outs[0][1], outs[1][1]), ("dW", R, O, I)


@pytest.mark.gpu
def test_time_grid_on_device_equals_the_host_linspace_table():
    """hirest_joint_time_grid_f32 (the backward's weight for temporal_embed.0.weight, built without reading n_valid back) against
    train.time_grid, the torch.linspace restatement of modeling.py:176-193 — bit for bit, incl. n = 1 and n = T."""
    import ctypes as C
    from hirest_amd import _lib, ops, train
    lib = _lib.load()
    dev = torch.device("cuda:0")
    T = 97
    n = torch.tensor([1, 2, 3, 50, 96, 97, 31], dtype=torch.int32)
    want = train.time_grid(n, T)
    nd = n.to(dev)
    got = torch.full((n.numel(), T), 7.0, dtype=torch.float32, device=dev)
    assert lib.hirest_joint_time_grid_f32(nd.data_ptr(), n.numel(), T, got.data_ptr(), ops.stream_ptr()) == 0
    assert torch.equal(got.cpu(), want)


@pytest.mark.gpu
def test_grouped_column_sums_equal_the_single_calls():
    """hirest_weighted_colsum_grouped_f32 (one launch for a step's bias / LayerNorm / embedding gradients) against
    hirest_weighted_colsum_f32 item by item, bit for bit: plain, weighted, selected, strided, a single column, and more items than
    one launch takes (HIREST_COLSUM_GROUP_MAX)."""
    from hirest_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    cases = []
    for i in range(_lib.COLSUM_GROUP_MAX + 7):
        R, Cc = [(1500, 768), (25, 3072), (1500, 1), (5, 300 * 768), (240, 513), (1, 64), (97, 31)][i % 7]
        big = torch.randn((R, Cc + 3 * (i % 2)), generator=g).to(dev)
        x = big[:, :Cc]                                                        # (odd cases: row stride > C)
        wt = torch.randn((R,), generator=g).to(dev) if i % 3 == 1 else None
        sel = torch.randint(0, 2, (R,), generator=g, dtype=torch.int32).to(dev) if i % 3 == 2 else None
        cases.append((x, wt, sel, i % 2))
    want, got = [], []
    arr = (_lib.ColsumItem * len(cases))()
    for slot, (x, wt, sel, val) in zip(arr, cases):
        w = torch.empty((x.shape[1],), dtype=torch.float32, device=dev)
        assert lib.hirest_weighted_colsum_f32(x.data_ptr(), x.stride(0), wt.data_ptr() if wt is not None else None,
                                              sel.data_ptr() if sel is not None else None, val, x.shape[0], x.shape[1], w.data_ptr(),
                                              ops.stream_ptr()) == 0
        want.append(w)
        o = torch.full((x.shape[1],), 3.0, dtype=torch.float32, device=dev)
        got.append(o)
        slot.x, slot.ldx, slot.R, slot.C, slot.out = x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], o.data_ptr()
        slot.row_weight = wt.data_ptr() if wt is not None else None
        slot.row_select = sel.data_ptr() if sel is not None else None
        slot.select_value = val
    assert lib.hirest_weighted_colsum_grouped_f32(arr, len(cases), ops.stream_ptr()) == 0
    for i, (w, o) in enumerate(zip(want, got)):
        assert torch.equal(w, o), i


def test_weight_gradients_on_the_side_stream_equal_the_single_stream_backward(dev, golden_dir):
    """The dW GEMMs of a backward run on a second stream behind "dY is ready" events (train.SIDE_STREAM_DW): same kernels on the same
    operands, so every gradient must equal the single-stream backward bit for bit — in train mode (same seed, same dropout masks),
    for all three tasks, repeatedly (a missing dependency would show as a race)."""
    from hirest_amd import train
    model, batch, seg_batch, cap_batch, _ = _setup(golden_dir, "a", dev)
    model.train()
    def grads(b, side, seed):
        train.SIDE_STREAM_DW = side
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(seed)
        model.train_step(b)["loss"].backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    try:
        for b in (batch, seg_batch, cap_batch):
            want = grads(b, False, 11)
            for rep in range(4):
                got = grads(b, True, 11)
                assert got.keys() == want.keys()
                for n in want:
                    if n == "clip4cap_model.decoder.embeddings.word_embeddings.weight":
                        # the tied matrix: its embedding share is an atomic scatter-add (order of the adds is not fixed run to run)
                        assert torch.allclose(got[n], want[n], rtol=1e-4, atol=1e-8), (b["tasks"][0], n, rep)
                    else:
                        assert torch.equal(got[n], want[n]), (b["tasks"][0], n, rep)
    finally:
        train.SIDE_STREAM_DW = True


def test_encoder_blocks_issued_from_c_equal_the_per_kernel_calls(dev, golden_dir):
    """train.C_BLOCKS: one C call per encoder block and direction (csrc/train_block.hip) issues the same entry points on the same operands
    in the same order as the per-kernel Python calls — loss and every gradient equal bit for bit, in train mode (same seed, same dropout
    masks) and in eval mode, with and without the side stream, retrieval and segmentation."""
    from hirest_amd import train
    model, batch, seg_batch, _, _ = _setup(golden_dir, "a", dev)

    def grads(b, c_blocks, side, seed):
        train.C_BLOCKS, train.SIDE_STREAM_DW = c_blocks, side
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(seed)
        loss = model.train_step(b)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    try:
        for mode in ("train", "eval"):
            getattr(model, mode)()
            for b in (batch, seg_batch):
                for side in (True, False):
                    lw, want = grads(b, False, side, 23)
                    for rep in range(3):
                        lg, got = grads(b, True, side, 23)
                        assert torch.equal(lw, lg), (mode, b["tasks"][0], side, rep)
                        assert got.keys() == want.keys()
                        for n in want:
                            assert torch.equal(got[n], want[n]), (mode, b["tasks"][0], side, n, rep)
    finally:
        train.C_BLOCKS, train.SIDE_STREAM_DW = True, True


def test_c_issued_step_without_an_asr_branch(dev, golden_dir):
    """A model built with asr_dim <= 0 has no asr_enc_layer (modeling.py:38-43): the C-issued backward (asr_dim = 0 in hirest_train_fusion_bwd)
    equals the per-kernel path bit for bit there too, both tasks, train mode."""
    import hirest_amd
    from hirest_amd import train
    from hirest_amd.synth import joint_inputs, train_targets
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=-1, args=None, clip_model=None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert not any(k.startswith("asr_enc_layer") for k in shapes)
    model.load_state_dict(synth.joint_state_dict(shapes, 37), strict=False)
    model = model.to(dev).train()
    B, T = 3, 77
    vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs("train.noasr", B, T, 59)
    st, et, seg, prev = train_targets("train.noasr", B, T, 59, bounds)
    batches = [{"tasks": ["moment_retrieval"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask, "text_feat": text,
                "moment_retrieval_start_target": st, "moment_retrieval_end_target": et},
               {"tasks": ["moment_segmentation"], "vis_feats": vis, "vis_mask": vis_mask, "moment_mask": moment_mask, "text_feat": text,
                "prev_boundary_mask": prev, "moment_segmentation_target": seg}]

    def grads(b, c_blocks):
        train.C_BLOCKS = c_blocks
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(5)
        loss = model.train_step(b)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    try:
        for b in batches:
            lw, want = grads(b, False)
            lg, got = grads(b, True)
            assert torch.equal(lw, lg) and torch.isfinite(lw) and got.keys() == want.keys()
            for n in want:
                assert torch.equal(got[n], want[n]), (b["tasks"][0], n)
    finally:
        train.C_BLOCKS = True


def test_bf16x3_blocks_split_their_weights_themselves_to_the_same_bits(dev, golden_dir):
    """precision 1 of hirest_train_block: with the weights pre-split by one grouped launch (default) or split by each block call
    (hirest_split2_bf16 / hirest_split2_transposed_bf16 inside the call) the operands are the same bits, so loss and gradients are too."""
    from hirest_amd import train
    model, batch, seg_batch, _, _ = _setup(golden_dir, "a", dev)
    model.train()

    def grads(b, grouped):
        train.GROUPED_WEIGHT_SPLIT = grouped
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(9)
        loss = model.train_step(b)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    train.GEMM_PRECISION = "bf16x3"
    try:
        for b in (batch, seg_batch):
            lw, want = grads(b, True)
            lg, got = grads(b, False)
            assert torch.equal(lw, lg) and got.keys() == want.keys()
            for n in want:
                assert torch.equal(got[n], want[n]), (b["tasks"][0], n)
    finally:
        train.GEMM_PRECISION, train.GROUPED_WEIGHT_SPLIT = "fp32", True


def test_backward_refuses_parameters_updated_in_place_after_the_forward(dev, golden_dir):
    """The forward keeps fp32 parameters by reference and the backward multiplies by them again: forward A, forward B, backward A,
    optimizer.step(), backward B would back-propagate B through the updated weights.  The version counters recorded in the forward
    make that a loud error instead (autograd's own check does not see tensors kept outside save_for_backward)."""
    model, batch, seg_batch, _, _ = _setup(golden_dir, "a", dev)
    model.train()
    optim = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    la, lb = model.train_step(batch)["loss"], model.train_step(seg_batch)["loss"]
    la.backward()
    optim.step()
    with pytest.raises(RuntimeError, match="modified in place"):
        lb.backward()
    for p in model.parameters():
        p.grad = None
    model.train_step(seg_batch)["loss"].backward()          # a fresh forward after the update is fine
    assert any(p.grad is not None for p in model.parameters())
