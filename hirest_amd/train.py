"""Training step of the joint model on the hand-written fp32 kernels.

``MomentModel.train_step(batch)`` keeps the reference's contract (/root/reference/modeling.py:130-140,226-270,323-351 and
the loop in run.py:238-295): it returns ``{'loss': tensor}``; ``loss.backward()`` leaves the gradients of the 63 M trainable
parameters in ``param.grad`` so the caller's ``clip_grad_norm_`` / optimizer / ``DistributedDataParallel`` (RCCL all-reduce,
run.py:93) work unchanged.  torch contributes the autograd *bookkeeping* only: one ``autograd.Function`` per task whose
forward and backward are sequences of C-ABI kernel calls (csrc/joint.hip forward kernels, csrc/train.hip backward kernels,
``hirest_gemm_f32`` on transposed operands for every dX / dW product).

* moment_retrieval: fusion -> VisualModel (2 post-LN layers) -> start / end heads -> masked BCE, full backward.
* moment_segmentation: the same graph plus the boundary embedding, the segment head and the cross-entropy over the moment's
  frames (modeling.py:310-351), full backward.
* step_captioning: trim_feats -> fusion -> VisualModel on 20 frames -> 2-layer caption decoder (masked self-attention,
  cross-attention, tied LM head) -> CrossEntropyLoss(ignore_index=-1) over the vocabulary (modeling.py:476-527), full backward.

Dropout (VisualEmbeddings / attention probabilities / VisualSelfOutput / VisualOutput, p = 0.1 in train mode:
module_visual.py:116-183) uses a counter-based mask; ``model.eval()`` switches it off, which is also how the gradient parity
tests pin the arithmetic against the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import os

import torch

from . import _lib, ops

_V = "clip4cap_model.visual."
# dX / dW products can read their operands in place (hirest_gemm_f32_strided) instead of through zero-padded transposed copies.
# Measured (tools/train_bench.py, B = 5): retrieval step 4.4 -> 4.1 ms at T = 120 (fewer launches on a host-bound step), 5.5 -> 5.6
# at T = 300, captioning 8.4 -> 9.0: the strided kernel's element-wise staging loses what the 53 saved copies gain, so it is off.
STRIDED_GEMM = False
STRIDED_MAX_FLOP = 1.5e9   # when on: only products up to this size (above it hirest_gemm_f32's register-prefetching kernel wins)
# Since late round 3 the 64x64 kernel itself reads k-major operands (hirest_gemm_f32_layouts: vector loads along the contiguous
# dimension, transposed on the way into LDS): dX / dW without the 37 transposed copies per step, same bits as with them.
LAYOUT_GEMM = True


def _chk(code, what):
    _lib.check(code, what)


# (Round 4 built the step's Linear-layer products as split-operand bf16x3 GEMMs as well — 3.3 vs 2.8 ms per step: slower, the ~50 extra split
# launches cost more than the faster products return — and round 5 removed that opt-in path again: profiles/r04/train_x3_ab.txt)
SIDE_STREAM_DW = True      # weight-gradient GEMMs of a backward on a second stream (_K.side_open)
# One C call per encoder block and direction (csrc/train_block.hip: the same entry points in the same order, issued from C — the step was
# paced by ~150 launches per step through Python at 15-20 us each).  False / HIREST_TRAIN_C_BLOCKS=0: the per-kernel calls below (A/B, tests).
C_BLOCKS = os.environ.get("HIREST_TRAIN_C_BLOCKS", "1") != "0"
# Products of the encoder blocks: "fp32" (exact fp32 MFMA: the reference's own arithmetic, run.py without --fp16) or "bf16x3" (forward and
# dX products on bf16 hi + lo splits of both operands — ~16 mantissa bits per product at 3 bf16 MFMAs; dW, attention, LayerNorm, GELU and
# the residual stream stay fp32; the reference's counterpart is --fp16's autocast, run.py:549-551).  C_BLOCKS only.  A model switched with
# MomentModel.set_precision('bf16x3') trains at bf16x3 whatever this says.
GEMM_PRECISION = os.environ.get("HIREST_TRAIN_GEMM", "fp32")
GROUPED_WEIGHT_SPLIT = True   # bf16x3: the blocks' weights split both ways by one grouped launch per step; False: by each block call itself (tests)
_SIDE = {}


class _K:
    """Thin tensor-level wrappers over the training entry points (device fp32 contiguous in, fresh tensors out)."""

    @staticmethod
    def gemm(a, w, bias=None, resid=None, periodic=None, period=0, act=0):
        lib = _lib.load()
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        ws, wsb = ops.f32_gemm_workspace(a.device, lib.hirest_gemm_f32_workspace_bytes(M, N, K))
        _chk(lib.hirest_gemm_f32_ws(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else None,
                                    resid.data_ptr() if resid is not None else None, N,
                                    periodic.data_ptr() if periodic is not None else None, period,
                                    out.data_ptr(), N, M, N, K, act, ws, wsb, ops.stream_ptr()), "hirest_gemm_f32_ws")
        return out

    @staticmethod
    def transpose_pad(x):
        """[R, C] -> [C, Rp], Rp = R rounded up to 16, zero filled."""
        R, Cc = x.shape
        Rp = (R + 15) // 16 * 16
        out = torch.empty((Cc, Rp), dtype=torch.float32, device=x.device)
        _chk(_lib.load().hirest_transpose_pad_f32(x.data_ptr(), x.stride(0), R, Cc, out.data_ptr(), Rp, ops.stream_ptr()), "transpose")
        return out

    @staticmethod
    def strided(a, sam, sak, b, sbn, sbk, M, N, K, alpha=1.0):
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _chk(_lib.load().hirest_gemm_f32_strided(a.data_ptr(), sam, sak, b.data_ptr(), sbn, sbk, out.data_ptr(), N, M, N, K, alpha,
                                                 ops.stream_ptr()), "gemm_f32_strided")
        return out

    @staticmethod
    def layouts(a, lda, a_kmajor, w, ldw, w_kmajor, M, N, K, resid=None, stream=None):
        lib = _lib.load()
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        ws, wsb = ops.f32_gemm_workspace(a.device, lib.hirest_gemm_f32_layouts_workspace_bytes(M, N, K), 0 if stream is None else 1, stream)
        _chk(lib.hirest_gemm_f32_layouts(a.data_ptr(), lda, int(a_kmajor), w.data_ptr(), ldw, int(w_kmajor), None,
                                         resid.data_ptr() if resid is not None else None, resid.stride(0) if resid is not None else 0,
                                         out.data_ptr(), N, M, N, K, 0, ws, wsb, ops.stream_ptr() if stream is None else stream),
             "hirest_gemm_f32_layouts")
        return out

    # Weight gradients on a second stream.  dW = dY^T X is needed only when the backward returns, while dX = dY W is on the critical
    # path: inside a backward (side_open() .. side_join()) every dW GEMM is issued on a side stream behind an event that marks "dY is
    # ready", so it fills the CUs the dX chain leaves idle — the tails of its GEMM launches, its row kernels of a few blocks, the
    # latency-bound attention products.  The operands stay referenced until the join (the allocator must not hand their memory to the
    # main stream while the side stream reads it); outputs are read after the join only.
    _side = None

    @staticmethod
    def side_open(device):
        if not SIDE_STREAM_DW:
            return
        st = _SIDE.get(device.index)
        if st is None:
            st = _SIDE[device.index] = {"stream": torch.cuda.Stream(device=device), "events": [], "keep": [], "used": 0}
        st["used"], st["c_used"] = 0, False
        _K._side = st

    @staticmethod
    def side_join():
        st, _K._side = _K._side, None
        if st is not None and (st["used"] or st.get("c_used")):
            torch.cuda.current_stream().wait_stream(st["stream"])
            st["keep"].clear()

    @staticmethod
    def grad_input(dy, w, resid=None):
        """dX = dY @ W (+ resid: the gradient arriving over a residual connection, added in the GEMM's epilogue — the same bits as a
        separate add) for y = x W^T:  dY [R, N], W [N, K] -> [R, K].  The strided GEMM reads W column-wise in place (its
        B[n][k] = W[k][n]); STRIDED_GEMM = False: through a zero-padded transposed copy and hirest_gemm_f32 (round 2)."""
        R, O = dy.shape
        I = w.shape[1]
        if STRIDED_GEMM and 2.0 * R * I * O <= STRIDED_MAX_FLOP:
            dx = _K.strided(dy, dy.stride(0), 1, w, 1, w.stride(0), R, I, O)
        elif w.shape[0] % 16 != 0:
            raise RuntimeError(f"grad_input: out_features {w.shape[0]} must be a multiple of 16")
        elif LAYOUT_GEMM and I % 4 == 0 and dy.stride(0) % 4 == 0 and w.stride(0) % 4 == 0 and dy.stride(1) == 1 and w.stride(1) == 1:
            return _K.layouts(dy, dy.stride(0), False, w, w.stride(0), True, R, I, O, resid=resid)   # B(n = i, k = o) = W[o][i]: k-major
        else:
            dx = _K.gemm(dy, _K.transpose_pad(w))
        return dx if resid is None else _K.dropout_add(dx, resid, 0.0, 0)

    @staticmethod
    def grad_weight(dy, x, side=True):
        """dW = dY^T @ X (side = False: on the main stream even inside a backward — for a result the backward itself goes on using):  dY [R, N], X [R, K] -> [N, K]: both operands read column-wise in place (A[m][k] = dY[k][m],
        B[n][k] = X[k][n]); STRIDED_GEMM = False: two zero-padded transposed copies."""
        if STRIDED_GEMM and 2.0 * dy.shape[1] * x.shape[1] * dy.shape[0] <= STRIDED_MAX_FLOP:
            return _K.strided(dy, 1, dy.stride(0), x, 1, x.stride(0), dy.shape[1], x.shape[1], dy.shape[0])
        R, O = dy.shape
        I = x.shape[1]
        if (LAYOUT_GEMM and O % 4 == 0 and I % 4 == 0 and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dy.stride(1) == 1
                and x.stride(1) == 1):
            st = _K._side if side else None
            if st is None:
                return _K.layouts(dy, dy.stride(0), True, x, x.stride(0), True, O, I, R)    # A(m = o, k = r) = dY[r][o], B(n = i, k = r) = X[r][i]
            if st["used"] == len(st["events"]):
                st["events"].append(torch.cuda.Event())
            ev = st["events"][st["used"]]
            st["used"] += 1
            ev.record()                                                      # dY (and X) are complete on the main stream here
            st["stream"].wait_event(ev)
            st["keep"].append((dy, x))
            return _K.layouts(dy, dy.stride(0), True, x, x.stride(0), True, O, I, R, stream=st["stream"].cuda_stream)
        return _K.gemm(_K.transpose_pad(dy), _K.transpose_pad(x))

    # column sums asked for while a batch is open are recorded and run as ONE launch by flush_colsums(): a step has ~36 of them
    # (bias, LayerNorm and embedding gradients), each a few blocks.  Their results are only read after the backward returns.
    _pending = None

    _c_items = None          # (ColsumItem * 64)() the C block backward appends to, its counter, and the tensors those items point into
    _c_count = None
    _c_keep: list = []

    @staticmethod
    def open_colsums():
        _K._pending = []
        if _K._c_items is None:
            _K._c_items, _K._c_count = (_lib.ColsumItem * 64)(), C.c_int32(0)
        _K._c_count.value = 0
        _K._c_keep = []

    @staticmethod
    def flush_colsums():
        items, _K._pending = _K._pending, None
        nc = _K._c_count.value if _K._c_count is not None else 0
        if not items and not nc:
            return
        arr = (_lib.ColsumItem * (len(items) + nc))()
        for slot, (x, weight, select, value, out) in zip(arr, items):
            slot.x, slot.ldx, slot.R, slot.C = x.data_ptr(), x.stride(0), x.shape[0], x.shape[1]
            slot.row_weight = weight.data_ptr() if weight is not None else None
            slot.row_select = select.data_ptr() if select is not None else None
            slot.select_value, slot.out = value, out.data_ptr()
        if nc:                                                               # the block backwards' own sums (csrc/train_block.hip), same launch
            C.memmove(C.byref(arr, len(items) * C.sizeof(_lib.ColsumItem)), _K._c_items, nc * C.sizeof(_lib.ColsumItem))
            _K._c_count.value = 0
        _chk(_lib.load().hirest_weighted_colsum_grouped_f32(arr, len(items) + nc, ops.stream_ptr()), "colsum_grouped")
        _K._c_keep = []
        # (`items` kept the operands alive up to here; the caching allocator orders their reuse after this launch on the stream)

    @staticmethod
    def colsum(x, weight=None, select=None, value=0, out=None):
        R, Cc = x.shape
        if out is None:
            out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        if _K._pending is not None:
            _K._pending.append((x, weight, select, value, out))
            return out
        _chk(_lib.load().hirest_weighted_colsum_f32(x.data_ptr(), x.stride(0), weight.data_ptr() if weight is not None else None,
                                                    select.data_ptr() if select is not None else None, value, R, Cc, out.data_ptr(),
                                                    ops.stream_ptr()), "colsum")
        return out

    @staticmethod
    def layernorm(x, g, b, eps):
        out = torch.empty_like(x)
        return ops.layernorm(x, g, b, eps, out)

    @staticmethod
    def layernorm_bwd(x, dy, g, eps):
        dx, dyx = torch.empty_like(x), torch.empty_like(x)
        _chk(_lib.load().hirest_layernorm_bwd_f32(x.data_ptr(), dy.data_ptr(), g.data_ptr(), eps, dx.data_ptr(), dyx.data_ptr(),
                                                  x.shape[0], x.shape[1], ops.stream_ptr()), "layernorm_bwd")
        return dx, _K.colsum(dyx), _K.colsum(dy)

    @staticmethod
    def act(pre, act):
        y = torch.empty_like(pre)
        _chk(_lib.load().hirest_act_f32(pre.data_ptr(), y.data_ptr(), pre.numel(), act, ops.stream_ptr()), "act")
        return y

    @staticmethod
    def act_bwd(pre, dy, act):
        dx = torch.empty_like(dy)
        _chk(_lib.load().hirest_act_bwd_f32(pre.data_ptr(), dy.data_ptr(), dx.data_ptr(), dy.numel(), act, ops.stream_ptr()), "act_bwd")
        return dx

    @staticmethod
    def dropout_add(x, resid, p, seed):
        """resid + dropout(x) (resid None: dropout only; p = 0: plain add)."""
        if resid is None and p == 0.0:
            return x
        y = torch.empty_like(x)
        _chk(_lib.load().hirest_dropout_add_f32(x.data_ptr(), resid.data_ptr() if resid is not None else None, y.data_ptr(), x.numel(),
                                                float(p), int(seed) & 0xFFFFFFFF, ops.stream_ptr()), "dropout_add")
        return y


def _shaped(g, p):
    """g in p's shape (most gradients already are: a reshape call costs the backward ~3 us of host time each, 57 per step)."""
    return g if g.shape == p.shape else g.reshape(p.shape)


def _f32(t):
    t = t.detach()
    return t if t.dtype is torch.float32 and t.is_contiguous() else t.float().contiguous()


def _alias_versions(params):       # params: iterable of (name, parameter)
    """The forward keeps fp32-contiguous parameters by ALIAS (no copy: 63 M parameters per step), and the backward multiplies by
    them again (dX = dY W).  autograd's saved-tensor version check does not see tensors kept in a side dict, so the versions are
    recorded here and compared in the backward: an in-place update between a loss's forward and its backward (optimizer.step()
    between two losses' backwards, an EMA or clipping hook) would silently back-propagate through the updated weights."""
    return [(n, p, p._version) for n, p in params if p.dtype is torch.float32 and p.is_contiguous()]


def _check_alias_versions(saved):
    for n, p, v in saved:
        if p._version != v:
            raise RuntimeError(f"hirest_amd: parameter {n!r} was modified in place between this loss's forward and its backward "
                               "(the training kernels keep fp32 parameters by reference); call backward() before optimizer.step() "
                               "or recompute the loss")


def _scale_by_upstream(dl: torch.Tensor, gloss) -> None:
    """dl *= gloss, the upstream gradient autograd hands to backward (1 for a plain ``loss.backward()``; a GradScaler's or an
    accumulation factor otherwise), read by the kernel from device memory: no host synchronisation in the training step."""
    if gloss is None:
        return
    gl = gloss.detach().to(device=dl.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
    _chk(_lib.load().hirest_scale_by_device_scalar_f32(dl.data_ptr(), gl.data_ptr(), dl.numel(), ops.stream_ptr()), "scale_by_upstream")


# parameters on a task's graph, in the order the Function receives them / returns gradients for
_D = "clip4cap_model.decoder."


def encoder_param_names(model, boundary: bool) -> List[str]:
    names = ["clip_g_map.weight", "clip_g_map.bias",
             "clip4cap_model.normalize_video.visual_norm2d.weight", "clip4cap_model.normalize_video.visual_norm2d.bias",
             "clip_g_map_text.weight", "clip_g_map_text.bias"]
    if model.use_asr:
        names += ["asr_enc_layer.0.weight", "asr_enc_layer.0.bias", "asr_enc_layer.1.weight", "asr_enc_layer.1.bias"]
    names += ["temporal_embed.0.weight", "temporal_embed.0.bias", "temporal_embed.2.weight", "temporal_embed.2.bias",
              "mask_embed.weight"] + (["boundary_embed.weight"] if boundary else []) + [
              _V + "embeddings.word_embeddings.weight", _V + "embeddings.word_embeddings.bias",
              _V + "embeddings.position_embeddings.weight", _V + "embeddings.LayerNorm.weight", _V + "embeddings.LayerNorm.bias"]
    for i in range(len(model.clip4cap_model.visual.encoder.layer)):
        p = _V + f"encoder.layer.{i}."
        for leaf in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                     "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"):
            names += [p + leaf + ".weight", p + leaf + ".bias"]
    return names


def task_param_names(model, task: str) -> List[str]:
    names = encoder_param_names(model, task == "moment_segmentation")
    if task == "moment_segmentation":
        names += ["segment_predictor.0.weight", "segment_predictor.0.bias"]
    elif task == "moment_retrieval":
        names += ["start_predictor.0.weight", "start_predictor.0.bias", "end_predictor.0.weight", "end_predictor.0.bias"]
    else:   # step_captioning: the caption decoder (its input embedding is tied to the LM head: one parameter)
        names += [_D + "embeddings.word_embeddings.weight", _D + "embeddings.position_embeddings.weight",
                  _D + "embeddings.LayerNorm.weight", _D + "embeddings.LayerNorm.bias"]
        for i in range(len(model.clip4cap_model.decoder.decoder.layer)):
            p = _D + f"decoder.layer.{i}."
            for att in ("slf_attn", "enc_attn"):
                for leaf in ("att.query", "att.key", "att.value", "output.dense", "output.LayerNorm"):
                    names += [p + f"{att}.{leaf}.weight", p + f"{att}.{leaf}.bias"]
            for leaf in ("intermediate.dense", "output.dense", "output.LayerNorm"):
                names += [p + leaf + ".weight", p + leaf + ".bias"]
        cp = _D + "classifier.cls.predictions."
        names += [cp + "transform.dense.weight", cp + "transform.dense.bias", cp + "transform.LayerNorm.weight",
                  cp + "transform.LayerNorm.bias", cp + "bias"]
    return names


def _block_workspaces(dev, R, Hd, mlp, side_stream):
    """Split-form scratch of the fp32 GEMMs of one block (ops.f32_gemm_workspace: one buffer per stream), sized for the largest problem."""
    lib = _lib.load()
    fwd = max(lib.hirest_gemm_f32_workspace_bytes(R, n, k) for n, k in ((3 * Hd, Hd), (Hd, Hd), (mlp, Hd), (Hd, mlp)))
    bwd = max(lib.hirest_gemm_f32_layouts_workspace_bytes(R, n, k) for n, k in ((mlp, Hd), (Hd, mlp), (Hd, Hd), (Hd, 3 * Hd)))
    main = ops.f32_gemm_workspace(dev, max(fwd, bwd, 1))
    if side_stream is None:
        return main, (None, 0)
    dw = max(lib.hirest_gemm_f32_layouts_workspace_bytes(m, n, R) for m, n in ((Hd, mlp), (mlp, Hd), (Hd, Hd), (3 * Hd, Hd)))
    return main, ops.f32_gemm_workspace(dev, max(dw, 1), 1, side_stream)


def _split_block_weights(P, wqkvs, dev):
    """bf16x3: the four weights of every encoder block in the split operand format, as they are (forward: the GEMM's B operand) and transposed
    (backward: dX = dY W), by ONE grouped launch per 16 matrices (hirest_split2_grouped_bf16) — they change with every optimizer step.
    Returns, per block, {field of hirest_train_block: bf16 tensor}."""
    per, items, total = [], [], 0
    for i, wqkv in enumerate(wqkvs):
        p = _V + f"encoder.layer.{i}."
        blk = {}
        for name, w in (("wqkv", wqkv), ("wo", P[p + "attention.output.dense.weight"]), ("w1", P[p + "intermediate.dense.weight"]),
                        ("w2", P[p + "output.dense.weight"])):
            O, I = w.shape
            for field, tr, shape in ((name + "2", 0, (O, 2 * I)), (name + "T2", 1, (I, 2 * O))):
                blk[field] = (w, tr, shape, total)
                total += (shape[0] * shape[1] + 127) // 128 * 128
        per.append(blk)
    buf = torch.empty((total,), dtype=torch.bfloat16, device=dev)
    out = []
    for blk in per:
        o = {}
        for field, (w, tr, shape, off) in blk.items():
            t = buf[off:off + shape[0] * shape[1]].view(shape)
            o[field] = t
            items.append((w, tr, t))
        out.append(o)
    arr = (_lib.SplitItem * len(items))()
    for slot, (w, tr, t) in zip(arr, items):
        slot.x, slot.out, slot.ldx, slot.ldo = w.data_ptr(), t.data_ptr(), w.stride(0), t.shape[1]
        slot.rows, slot.cols, slot.transposed = w.shape[0], w.shape[1], tr
    _chk(_lib.load().hirest_split2_grouped_bf16(arr, len(items), ops.stream_ptr()), "split2_grouped")
    return out


def _block_forward(P, p, i, x, wqkv, bqkv, B, T, heads, drop, seed, x3=False, x2=None, wsplit=None):
    """One encoder block through hirest_train_block_forward (csrc/train_block.hip); keeps the descriptor and the activations for the backward."""
    lib = _lib.load()
    dev = x.device
    R, Hd = x.shape
    mlp = P[p + "intermediate.dense.weight"].shape[0]
    al = lambda n: (n + 63) // 64 * 64
    sizes = [("qkv", R * 3 * Hd), ("P", B * heads * T * T), ("cx", R * Hd), ("a_pre", R * Hd), ("aa", R * Hd), ("hpre", R * mlp), ("hh", R * mlp),
             ("x_pre", R * Hd), ("out", R * Hd)]
    flat = torch.empty((sum(al(n) for _, n in sizes),), dtype=torch.float32, device=dev)
    act, off = {}, 0
    for name, n in sizes:
        act[name] = flat[off:off + n]
        off += al(n)
    (ws, wsb), _ = _block_workspaces(dev, R, Hd, mlp, None)
    d = _lib.TrainBlock()
    d.struct_size = C.sizeof(_lib.TrainBlock)
    d.B, d.T, d.heads, d.width, d.mlp, d.precision = B, T, heads, Hd, mlp, int(x3)
    out2 = torch.empty((R, 2 * Hd), dtype=torch.bfloat16, device=dev) if x3 else None
    d.x2, d.out2 = (x2.data_ptr() if x3 and x2 is not None else None), (out2.data_ptr() if x3 else None)
    if x3 and wsplit is not None:
        for field, t in wsplit.items():                            # wqkv2 wo2 w12 w22 + their transposes (the backward's)
            setattr(d, field, t.data_ptr())
    d.ln_eps, d.drop = 1e-12, float(drop)
    d.seed_attn, d.seed_ao, d.seed_out = (seed + 10 + 4 * i) & 0xFFFFFFFF, (seed + 11 + 4 * i) & 0xFFFFFFFF, (seed + 12 + 4 * i) & 0xFFFFFFFF
    for field, t in (("wqkv", wqkv), ("bqkv", bqkv), ("wo", P[p + "attention.output.dense.weight"]), ("bo", P[p + "attention.output.dense.bias"]),
                     ("ln1_g", P[p + "attention.output.LayerNorm.weight"]), ("ln1_b", P[p + "attention.output.LayerNorm.bias"]),
                     ("w1", P[p + "intermediate.dense.weight"]), ("b1", P[p + "intermediate.dense.bias"]),
                     ("w2", P[p + "output.dense.weight"]), ("b2", P[p + "output.dense.bias"]),
                     ("ln2_g", P[p + "output.LayerNorm.weight"]), ("ln2_b", P[p + "output.LayerNorm.bias"]), ("x", x)):
        setattr(d, field, t.data_ptr())
    for name, _n in sizes:
        setattr(d, name, act[name].data_ptr())
    d.ws, d.ws_bytes = ws, wsb
    need = lib.hirest_train_block_forward_scratch_bytes(C.byref(d))
    scratch = torch.empty((need,), dtype=torch.uint8, device=dev)
    _chk(lib.hirest_train_block_forward(C.byref(d), scratch.data_ptr(), need, ops.stream_ptr()), "train_block_forward")
    return dict(desc=d, flat=flat, x=x, x2=x2, out2=out2, wsplit=wsplit, wqkv=wqkv, bqkv=bqkv, out=act["out"].reshape(R, Hd), dims=(R, Hd, mlp))


def _block_backward(P, p, Ly, dout, G):
    """Backward of _block_forward through hirest_train_block_backward: dX products on the current stream, dW on the open side stream, the
    twelve column sums appended to the open batch.  Returns d loss / d (block input)."""
    lib = _lib.load()
    d = Ly["desc"]
    R, Hd, mlp = Ly["dims"]
    dev = dout.device
    st = _K._side
    side = st["stream"].cuda_stream if st is not None else None
    (ws, wsb), (sws, swsb) = _block_workspaces(dev, R, Hd, mlp, side)
    d.ws, d.ws_bytes = ws, wsb
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    dwqkv, dbqkv = f32(3 * Hd, Hd), f32(3 * Hd)
    g = _lib.TrainBlockGrads()
    g.struct_size = C.sizeof(_lib.TrainBlockGrads)
    dout = dout.contiguous()
    dx = f32(R, Hd)
    g.dout, g.dx = dout.data_ptr(), dx.data_ptr()
    out = {"attention.output.dense.weight": f32(Hd, Hd), "intermediate.dense.weight": f32(mlp, Hd), "output.dense.weight": f32(Hd, mlp),
           "attention.output.dense.bias": f32(Hd), "intermediate.dense.bias": f32(mlp), "output.dense.bias": f32(Hd),
           "attention.output.LayerNorm.weight": f32(Hd), "attention.output.LayerNorm.bias": f32(Hd),
           "output.LayerNorm.weight": f32(Hd), "output.LayerNorm.bias": f32(Hd)}
    g.g_wqkv, g.g_bqkv = dwqkv.data_ptr(), dbqkv.data_ptr()
    for field, name in (("g_wo", "attention.output.dense.weight"), ("g_w1", "intermediate.dense.weight"), ("g_w2", "output.dense.weight"),
                        ("g_bo", "attention.output.dense.bias"), ("g_b1", "intermediate.dense.bias"), ("g_b2", "output.dense.bias"),
                        ("g_ln1_g", "attention.output.LayerNorm.weight"), ("g_ln1_b", "attention.output.LayerNorm.bias"),
                        ("g_ln2_g", "output.LayerNorm.weight"), ("g_ln2_b", "output.LayerNorm.bias")):
        setattr(g, field, out[name].data_ptr())
    if _K._pending is None:
        raise RuntimeError("hirest_amd.train: the block backward runs inside an open column-sum batch (_colsum_batched)")
    g.items, g.n_items, g.max_items = _K._c_items, C.pointer(_K._c_count), 64
    g.side_stream, g.side_ws, g.side_ws_bytes = side, sws, swsb
    need = lib.hirest_train_block_backward_scratch_bytes(C.byref(d))
    scratch = torch.empty((need,), dtype=torch.uint8, device=dev)
    g.scratch, g.scratch_bytes = scratch.data_ptr(), need
    _chk(lib.hirest_train_block_backward(C.byref(d), C.byref(g), ops.stream_ptr()), "train_block_backward")
    _K._c_keep.append((scratch, dout, Ly))                     # read by the grouped column sums (and the side stream) after this call returns
    if st is not None:
        st["c_used"] = True                                    # side_join waits for the side stream
        st["keep"].append((scratch, dout, Ly))
    for name, t in out.items():
        G[p + name] = t
    for k, nm in enumerate(("query", "key", "value")):
        G[p + f"attention.self.{nm}.weight"] = dwqkv[k * Hd:(k + 1) * Hd]
        G[p + f"attention.self.{nm}.bias"] = dbqkv[k * Hd:(k + 1) * Hd]
    return dx



def _encoder_forward(model, P, inp, S):
    """Fusion + VisualModel (modeling.py:155-210) keeping what the backward needs in S.  Returns feats [B*T, 768]."""
    lib = _lib.load()
    vis, text, asr = inp["vis"], inp["text"], inp.get("asr")
    vmask, mmask = inp["vis_mask"], inp["moment_mask"]
    B, T, _ = vis.shape
    R, E, Hd = B * T, 512, 768
    heads = model.heads
    drop, seed = S["drop"], S["seed"]
    boundary = inp.get("boundary_mask") is not None
    vis2 = vis.reshape(R, -1).float().contiguous()
    v0 = _K.gemm(vis2, P["clip_g_map.weight"], P["clip_g_map.bias"])
    gnv, bnv = P["clip4cap_model.normalize_video.visual_norm2d.weight"], P["clip4cap_model.normalize_video.visual_norm2d.bias"]
    v = _K.layernorm(v0, gnv, bnv, 1e-12)
    t = _K.gemm(text.float().contiguous(), P["clip_g_map_text.weight"], P["clip_g_map_text.bias"])
    tn = ops.pool_l2norm(t.unsqueeze(1).contiguous())
    if model.use_asr:
        asr2 = asr.reshape(R, -1).float().contiguous()
        a0 = _K.layernorm(asr2, P["asr_enc_layer.0.weight"], P["asr_enc_layer.0.bias"], 1e-5)
        a = _K.gemm(a0, P["asr_enc_layer.1.weight"], P["asr_enc_layer.1.bias"])
        S.update(asr2=asr2, a0=a0)
    else:
        a = torch.zeros((R, E), dtype=torch.float32, device=vis.device)
    n_valid = vmask.sum(dim=-1).to(torch.int32).contiguous()
    tin = torch.empty((R, E), dtype=torch.float32, device=vis.device)
    _chk(lib.hirest_joint_time_features(n_valid.data_ptr(), P["temporal_embed.0.weight"].data_ptr(), P["temporal_embed.0.bias"].data_ptr(),
                                        tin.data_ptr(), B, T, E, ops.stream_ptr()), "time_features")
    temporal = _K.gemm(tin, P["temporal_embed.2.weight"], P["temporal_embed.2.bias"])
    base = torch.empty((R, E), dtype=torch.float32, device=vis.device)
    _chk(lib.hirest_joint_base(v.data_ptr(), t.data_ptr(), a.data_ptr(), temporal.data_ptr(), base.data_ptr(), B, T, E, ops.stream_ptr()),
         "joint_base")
    mm32 = mmask.to(torch.int32).contiguous()
    bm32 = inp["boundary_mask"].to(torch.int32).contiguous() if boundary else None
    bemb = P["boundary_embed.weight"] if boundary else _f32(model.boundary_embed.weight)
    f = torch.empty_like(base)
    _chk(lib.hirest_joint_mask_add(base.data_ptr(), mm32.data_ptr(), bm32.data_ptr() if boundary else None, P["mask_embed.weight"].data_ptr(),
                                   bemb.data_ptr(), f.data_ptr(), R, E, ops.stream_ptr()), "mask_add")
    x0 = _K.gemm(f, P[_V + "embeddings.word_embeddings.weight"], P[_V + "embeddings.word_embeddings.bias"],
                 periodic=P[_V + "embeddings.position_embeddings.weight"], period=T)
    xe = _K.layernorm(x0, P[_V + "embeddings.LayerNorm.weight"], P[_V + "embeddings.LayerNorm.bias"], 1e-12)
    x = _K.dropout_add(xe, None, drop, seed + 1)
    S.update(B=B, T=T, vis2=vis2, v0=v0, v=v, t=t, tn=tn, tin=tin, f=f, x0=x0, mm32=mm32, bm32=bm32, n_valid=n_valid,
             text=text.float().contiguous(), boundary=boundary)
    layers = []
    nl = len(model.clip4cap_model.visual.encoder.layer)
    cats = []
    for i in range(nl):
        p = _V + f"encoder.layer.{i}."
        cats.append((torch.cat([P[p + "attention.self.query.weight"], P[p + "attention.self.key.weight"], P[p + "attention.self.value.weight"]], 0).contiguous(),
                     torch.cat([P[p + "attention.self.query.bias"], P[p + "attention.self.key.bias"], P[p + "attention.self.value.bias"]], 0).contiguous()))
    c_blocks = C_BLOCKS and LAYOUT_GEMM and not STRIDED_GEMM
    if GEMM_PRECISION not in ("fp32", "bf16x3"):
        raise ValueError(f"hirest_amd.train.GEMM_PRECISION = {GEMM_PRECISION!r}: 'fp32' or 'bf16x3'")
    x3 = c_blocks and (GEMM_PRECISION == "bf16x3" or getattr(model, "precision", "fp32") == "bf16x3")   # MomentModel.set_precision covers training too
    wsplit = _split_block_weights(P, [c[0] for c in cats], vis.device) if x3 and GROUPED_WEIGHT_SPLIT else None
    for i in range(nl):
        p = _V + f"encoder.layer.{i}."
        wqkv, bqkv = cats[i]
        if c_blocks:
            Ly = _block_forward(P, p, i, x, wqkv, bqkv, B, T, heads, drop, seed, x3, layers[-1].get("out2") if layers else None,
                                wsplit[i] if wsplit is not None else None)
            layers.append(Ly)
            x = Ly["out"]
            continue
        qkv = _K.gemm(x, wqkv, bqkv)
        Pm = torch.empty((B, heads, T, T), dtype=torch.float32, device=vis.device)
        cx = torch.empty((R, Hd), dtype=torch.float32, device=vis.device)
        _chk(lib.hirest_attention_train_fwd_f32(qkv.data_ptr(), Pm.data_ptr(), cx.data_ptr(), B, T, heads, Hd // heads,
                                                (Hd // heads) ** -0.5, -10000.0, drop, (seed + 10 + 4 * i) & 0xFFFFFFFF, ops.stream_ptr()),
             "attention_train_fwd")
        o = _K.gemm(cx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"])
        a_pre = _K.dropout_add(o, x, drop, seed + 11 + 4 * i)
        aa = _K.layernorm(a_pre, P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"], 1e-12)
        hpre = _K.gemm(aa, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"])
        hh = _K.act(hpre, 1)
        y = _K.gemm(hh, P[p + "output.dense.weight"], P[p + "output.dense.bias"])
        x_pre = _K.dropout_add(y, aa, drop, seed + 12 + 4 * i)
        xn = _K.layernorm(x_pre, P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], 1e-12)
        layers.append(dict(x=x, wqkv=wqkv, qkv=qkv, Pm=Pm, cx=cx, a_pre=a_pre, aa=aa, hpre=hpre, hh=hh, x_pre=x_pre))
        x = xn
    S["layers"] = layers
    return x


def _fusion_backward(model, P, S, dx, G):
    """The backward below the encoder blocks (embeddings + fusion) through hirest_train_fusion_backward (csrc/train_block.hip): the calls of
    _encoder_backward's tail, issued from C.  Parameter gradients go into G."""
    lib = _lib.load()
    B, T = S["B"], S["T"]
    R, E, Hd = B * T, 512, 768
    dev = dx.device
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    pos = P[_V + "embeddings.position_embeddings.weight"]
    asr = model.use_asr
    A = S["asr2"].shape[1] if asr else 0
    d = _lib.TrainFusionBwd()
    d.struct_size = C.sizeof(_lib.TrainFusionBwd)
    d.B, d.T, d.E, d.W, d.vis_dim, d.text_dim, d.asr_dim = B, T, E, Hd, S["vis2"].shape[1], S["text"].shape[1], A
    d.boundary, d.max_pos = int(bool(S["boundary"])), pos.shape[0]
    d.drop, d.seed_emb = float(S["drop"]), (S["seed"] + 1) & 0xFFFFFFFF
    for field, t in (("w_emb", P[_V + "embeddings.word_embeddings.weight"]), ("emb_ln_g", P[_V + "embeddings.LayerNorm.weight"]),
                     ("t2_w", P["temporal_embed.2.weight"]), ("norm_g", P["clip4cap_model.normalize_video.visual_norm2d.weight"]),
                     ("x0", S["x0"]), ("f", S["f"]), ("v", S["v"]), ("tn", S["tn"]), ("tin", S["tin"]), ("v0", S["v0"]), ("vis2", S["vis2"]),
                     ("t", S["t"]), ("text", S["text"]), ("mm32", S["mm32"]), ("n_valid", S["n_valid"])):
        setattr(d, field, t.data_ptr())
    if asr:
        d.asr1_w, d.asr0_g, d.a0, d.asr2 = (P["asr_enc_layer.1.weight"].data_ptr(), P["asr_enc_layer.0.weight"].data_ptr(), S["a0"].data_ptr(),
                                            S["asr2"].data_ptr())
    if S["boundary"]:
        d.bm32 = S["bm32"].data_ptr()
    dx = dx.contiguous()
    d.dx = dx.data_ptr()
    out = {_V + "embeddings.LayerNorm.weight": ("g_emb_ln_g", f32(Hd)), _V + "embeddings.LayerNorm.bias": ("g_emb_ln_b", f32(Hd)),
           _V + "embeddings.position_embeddings.weight": ("g_pos", f32(*pos.shape)),
           _V + "embeddings.word_embeddings.weight": ("g_w_emb", f32(Hd, E)), _V + "embeddings.word_embeddings.bias": ("g_b_emb", f32(Hd)),
           "mask_embed.weight": ("g_mask", f32(2, E)),
           "temporal_embed.2.weight": ("g_t2_w", f32(E, E)), "temporal_embed.2.bias": ("g_t2_b", f32(E)),
           "temporal_embed.0.weight": ("g_t0_w", f32(E, 1)), "temporal_embed.0.bias": ("g_t0_b", f32(E)),
           "clip4cap_model.normalize_video.visual_norm2d.weight": ("g_norm_g", f32(E)),
           "clip4cap_model.normalize_video.visual_norm2d.bias": ("g_norm_b", f32(E)),
           "clip_g_map.weight": ("g_vis_w", f32(E, S["vis2"].shape[1])), "clip_g_map.bias": ("g_vis_b", f32(E)),
           "clip_g_map_text.weight": ("g_text_w", f32(E, S["text"].shape[1])), "clip_g_map_text.bias": ("g_text_b", f32(E))}
    if S["boundary"]:
        out["boundary_embed.weight"] = ("g_bound", f32(2, E))
    if asr:
        out.update({"asr_enc_layer.1.weight": ("g_asr1_w", f32(E, A)), "asr_enc_layer.1.bias": ("g_asr1_b", f32(E)),
                    "asr_enc_layer.0.weight": ("g_asr0_g", f32(A)), "asr_enc_layer.0.bias": ("g_asr0_b", f32(A))})
    for name, (field, t) in out.items():
        setattr(d, field, t.data_ptr())
        G[name] = t
    st = _K._side
    side = st["stream"].cuda_stream if st is not None else None
    shapes_main = [(R, E, Hd), (R, E, E)] + ([(R, A, E)] if asr else [])                                 # dX products: (M, N, K)
    shapes_side = [(Hd, E, R), (E, E, R), (E, S["vis2"].shape[1], R), (E, S["text"].shape[1], B)] + ([(E, A, R)] if asr else [])
    d.ws, d.ws_bytes = ops.f32_gemm_workspace(dev, max(max(lib.hirest_gemm_f32_layouts_workspace_bytes(*m) for m in shapes_main), 1))
    if side is not None:
        d.side_stream = side
        d.side_ws, d.side_ws_bytes = ops.f32_gemm_workspace(dev, max(max(lib.hirest_gemm_f32_layouts_workspace_bytes(*m) for m in shapes_side), 1), 1, side)
    d.items, d.n_items, d.max_items = _K._c_items, C.pointer(_K._c_count), 64
    need = lib.hirest_train_fusion_backward_scratch_bytes(C.byref(d))
    if need == 0:
        raise RuntimeError("hirest_train_fusion_backward: unsupported shape")
    scratch = torch.empty((need,), dtype=torch.uint8, device=dev)
    d.scratch, d.scratch_bytes = scratch.data_ptr(), need
    _chk(lib.hirest_train_fusion_backward(C.byref(d), ops.stream_ptr()), "train_fusion_backward")
    _K._c_keep.append((scratch, dx, S))
    if st is not None:
        st["c_used"] = True
        st["keep"].append((scratch, dx, S))



def _encoder_backward(model, P, S, dx, G):
    """Backward of _encoder_forward: dx = d loss / d feats [B*T, 768]; parameter gradients go into G."""
    lib = _lib.load()
    B, T, drop, seed = S["B"], S["T"], S["drop"], S["seed"]
    R, E, Hd = B * T, 512, 768
    heads = model.heads
    dev = dx.device
    mm32 = S["mm32"]
    for i in reversed(range(len(S["layers"]))):
        p = _V + f"encoder.layer.{i}."
        Ly = S["layers"][i]
        if "desc" in Ly:
            dx = _block_backward(P, p, Ly, dx, G)
            continue
        dxp, G[p + "output.LayerNorm.weight"], G[p + "output.LayerNorm.bias"] = _K.layernorm_bwd(Ly["x_pre"], dx, P[p + "output.LayerNorm.weight"], 1e-12)
        dy = _K.dropout_add(dxp, None, drop, seed + 12 + 4 * i)           # through dropout(y); the residual branch gets dxp as is
        G[p + "output.dense.weight"] = _K.grad_weight(dy, Ly["hh"])
        G[p + "output.dense.bias"] = _K.colsum(dy)
        dh = _K.grad_input(dy, P[p + "output.dense.weight"])
        dhp = _K.act_bwd(Ly["hpre"], dh, 1)
        G[p + "intermediate.dense.weight"] = _K.grad_weight(dhp, Ly["aa"])
        G[p + "intermediate.dense.bias"] = _K.colsum(dhp)
        da = _K.grad_input(dhp, P[p + "intermediate.dense.weight"], resid=dxp)                        # + residual path
        dap, G[p + "attention.output.LayerNorm.weight"], G[p + "attention.output.LayerNorm.bias"] = \
            _K.layernorm_bwd(Ly["a_pre"], da, P[p + "attention.output.LayerNorm.weight"], 1e-12)
        do = _K.dropout_add(dap, None, drop, seed + 11 + 4 * i)
        G[p + "attention.output.dense.weight"] = _K.grad_weight(do, Ly["cx"])
        G[p + "attention.output.dense.bias"] = _K.colsum(do)
        dcx = _K.grad_input(do, P[p + "attention.output.dense.weight"])
        dS = torch.empty_like(Ly["Pm"])
        dqkv = torch.empty_like(Ly["qkv"])
        _chk(lib.hirest_attention_train_bwd_f32(Ly["qkv"].data_ptr(), Ly["Pm"].data_ptr(), dcx.data_ptr(), dS.data_ptr(), dqkv.data_ptr(),
                                                B, T, heads, Hd // heads, (Hd // heads) ** -0.5, drop, (seed + 10 + 4 * i) & 0xFFFFFFFF,
                                                ops.stream_ptr()), "attention_train_bwd")
        dwqkv = _K.grad_weight(dqkv, Ly["x"])
        dbqkv = _K.colsum(dqkv)
        for k, nm in enumerate(("query", "key", "value")):
            G[p + f"attention.self.{nm}.weight"] = dwqkv[k * Hd:(k + 1) * Hd]
            G[p + f"attention.self.{nm}.bias"] = dbqkv[k * Hd:(k + 1) * Hd]
        dx = _K.grad_input(dqkv, Ly["wqkv"], resid=dap)                                                # + residual path
    if C_BLOCKS and LAYOUT_GEMM and not STRIDED_GEMM and _K._pending is not None:
        return _fusion_backward(model, P, S, dx, G)
    # ---- embeddings
    dxe = _K.dropout_add(dx, None, drop, seed + 1)
    dx0, G[_V + "embeddings.LayerNorm.weight"], G[_V + "embeddings.LayerNorm.bias"] = \
        _K.layernorm_bwd(S["x0"], dxe, P[_V + "embeddings.LayerNorm.weight"], 1e-12)
    pos = P[_V + "embeddings.position_embeddings.weight"]
    dpos = torch.zeros_like(pos)
    _K.colsum(dx0.reshape(B, T * Hd), out=dpos[:T].reshape(-1))
    G[_V + "embeddings.position_embeddings.weight"] = dpos
    G[_V + "embeddings.word_embeddings.weight"] = _K.grad_weight(dx0, S["f"])
    G[_V + "embeddings.word_embeddings.bias"] = _K.colsum(dx0)
    df = _K.grad_input(dx0, P[_V + "embeddings.word_embeddings.weight"])
    # ---- fusion
    G["mask_embed.weight"] = torch.empty((2, E), dtype=torch.float32, device=dev)
    for k in (0, 1):
        _K.colsum(df, select=mm32.reshape(-1), value=k, out=G["mask_embed.weight"][k])
    if S["boundary"]:
        G["boundary_embed.weight"] = torch.empty((2, E), dtype=torch.float32, device=dev)
        for k in (0, 1):
            _K.colsum(df, select=S["bm32"].reshape(-1), value=k, out=G["boundary_embed.weight"][k])
    dv = torch.empty_like(df)
    dtn = torch.empty((B, E), dtype=torch.float32, device=dev)
    _chk(lib.hirest_joint_base_bwd_f32(df.data_ptr(), S["v"].data_ptr(), S["tn"].data_ptr(), dv.data_ptr(), dtn.data_ptr(), B, T, E,
                                       ops.stream_ptr()), "joint_base_bwd")
    # temporal embedding: Linear(1, E) -> tanh -> Linear(E, E) over the normalised time grid
    G["temporal_embed.2.weight"] = _K.grad_weight(df, S["tin"])
    G["temporal_embed.2.bias"] = _K.colsum(df)
    dpre = _K.act_bwd(S["tin"], _K.grad_input(df, P["temporal_embed.2.weight"]), 3)
    time = torch.empty((R,), dtype=torch.float32, device=dev)      # on the device: reading n_valid back would stall the step's enqueue
    _chk(lib.hirest_joint_time_grid_f32(S["n_valid"].data_ptr(), B, T, time.data_ptr(), ops.stream_ptr()), "time_grid")
    G["temporal_embed.0.weight"] = _K.colsum(dpre, weight=time).reshape(E, 1)
    G["temporal_embed.0.bias"] = _K.colsum(dpre)
    if model.use_asr:
        G["asr_enc_layer.1.weight"] = _K.grad_weight(df, S["a0"])
        G["asr_enc_layer.1.bias"] = _K.colsum(df)
        da0 = _K.grad_input(df, P["asr_enc_layer.1.weight"])
        _, G["asr_enc_layer.0.weight"], G["asr_enc_layer.0.bias"] = _K.layernorm_bwd(S["asr2"], da0, P["asr_enc_layer.0.weight"], 1e-5)
    dv0, G["clip4cap_model.normalize_video.visual_norm2d.weight"], G["clip4cap_model.normalize_video.visual_norm2d.bias"] = \
        _K.layernorm_bwd(S["v0"], dv, P["clip4cap_model.normalize_video.visual_norm2d.weight"], 1e-12)
    G["clip_g_map.weight"] = _K.grad_weight(dv0, S["vis2"])
    G["clip_g_map.bias"] = _K.colsum(dv0)
    dt = torch.empty_like(dtn)
    _chk(lib.hirest_l2norm_bwd_f32(S["t"].data_ptr(), dtn.data_ptr(), dt.data_ptr(), B, E, ops.stream_ptr()), "l2norm_bwd")
    G["clip_g_map_text.weight"] = _K.grad_weight(dt, S["text"])
    G["clip_g_map_text.bias"] = _K.colsum(dt)


def _colsum_batched(backward):
    """Run a backward with _K's column-sum batch and its side stream for the weight gradients open; an exception closes the batch
    without launching (nothing may stay recorded); the side stream is joined either way."""
    def wrapped(ctx, gloss):
        _K.open_colsums()
        _K.side_open(gloss.device if gloss is not None and gloss.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        try:
            out = backward(ctx, gloss)
            _K.flush_colsums()
            return out
        finally:
            _K._pending = None
            if _K._c_count is not None:
                _K._c_count.value = 0
            _K._c_keep = []
            _K.side_join()
    return wrapped


class MomentLoss(torch.autograd.Function):
    """The training losses of modeling.py:226-270 (moment retrieval: (BCE_start + BCE_end) / 2) and :323-351 (moment
    segmentation: cross-entropy over the moment's frames) with their backward, every step a kernel of this library."""

    @staticmethod
    def forward(ctx, model, inp: Dict[str, torch.Tensor], names: List[str], *params):
        lib = _lib.load()
        P = {n: _f32(p) for n, p in zip(names, params)}
        seg = inp["task"] == "moment_segmentation"
        S = {"drop": float(inp.get("dropout", 0.0)), "seed": int(inp.get("seed", 0)), "seg": seg}
        feats = _encoder_forward(model, P, inp, S)
        B, T, mm32 = S["B"], S["T"], S["mm32"]
        R, Hd = B * T, 768
        dev = feats.device
        loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        if seg:   # modeling.py:319, 343-344
            wsg = P["segment_predictor.0.weight"]
            bias3 = torch.cat([P["segment_predictor.0.bias"], torch.zeros(2, device=dev)]).contiguous()
            logits = torch.empty((1, R), dtype=torch.float32, device=dev)
            _chk(lib.hirest_linear_heads(feats.data_ptr(), R, Hd, 1, wsg.data_ptr(), None, None, bias3.data_ptr(), logits.data_ptr(),
                                         ops.stream_ptr()), "linear_heads")
            st = inp["segment_target"].to(torch.int32).contiguous()
            et = st
            dl = torch.empty_like(logits)
            _chk(lib.hirest_ce_masked_f32(logits.data_ptr(), mm32.data_ptr(), st.data_ptr(), B, T, 1.0, loss.data_ptr(), dl.data_ptr(),
                                          ops.stream_ptr()), "ce_masked")
        else:     # modeling.py:218-219, 249-263
            ws, we = P["start_predictor.0.weight"], P["end_predictor.0.weight"]
            bias3 = torch.cat([P["start_predictor.0.bias"], P["end_predictor.0.bias"], torch.zeros(1, device=dev)]).contiguous()
            logits = torch.empty((2, R), dtype=torch.float32, device=dev)
            _chk(lib.hirest_linear_heads(feats.data_ptr(), R, Hd, 2, ws.data_ptr(), we.data_ptr(), None, bias3.data_ptr(), logits.data_ptr(),
                                         ops.stream_ptr()), "linear_heads")
            st = inp["start_target"].to(torch.int32).contiguous()
            et = inp["end_target"].to(torch.int32).contiguous()
            dl = torch.empty_like(logits)
            for h_i, tgt in enumerate((st, et)):
                _chk(lib.hirest_bce_masked_f32(logits[h_i].data_ptr(), tgt.data_ptr(), mm32.data_ptr(), B, T, 0.5, loss.data_ptr(),
                                               dl[h_i].data_ptr(), ops.stream_ptr()), "bce_masked")
        S.update(feats=feats, logits=logits, st=st, et=et, P=P, names=names, model=model, versions=_alias_versions(zip(names, params)))
        ctx.S = S
        ctx.set_materialize_grads(False)
        return loss.reshape(())

    @staticmethod
    @_colsum_batched
    def backward(ctx, gloss):
        S = ctx.S
        if S is None:
            raise RuntimeError("hirest_amd: this loss was already back-propagated (the kernels' saved activations are released after "
                               "the first backward; retain_graph is not supported)")
        _check_alias_versions(S["versions"])
        lib = _lib.load()
        P, names, model = S["P"], S["names"], S["model"]
        B, T = S["B"], S["T"]
        R, Hd = B * T, 768
        feats, logits, mm32 = S["feats"], S["logits"], S["mm32"]
        dev = feats.device
        G: Dict[str, torch.Tensor] = {}
        g = 1.0                                                     # the upstream scale is applied on the device below (no host read)
        dl = torch.empty_like(logits)
        scratch = torch.zeros((1,), dtype=torch.float32, device=dev)
        dx = torch.empty((R, Hd), dtype=torch.float32, device=dev)
        if S["seg"]:
            _chk(lib.hirest_ce_masked_f32(logits.data_ptr(), mm32.data_ptr(), S["st"].data_ptr(), B, T, g, scratch.data_ptr(), dl.data_ptr(),
                                          ops.stream_ptr()), "ce_masked")
            _scale_by_upstream(dl, gloss)
            wsg = P["segment_predictor.0.weight"]
            G["segment_predictor.0.weight"] = _K.colsum(feats, weight=dl[0]).reshape(1, Hd)
            G["segment_predictor.0.bias"] = _K.colsum(dl[0].reshape(R, 1))
            _chk(lib.hirest_heads_bwd_f32(dl.data_ptr(), R, Hd, 1, wsg.data_ptr(), None, None, dx.data_ptr(), ops.stream_ptr()), "heads_bwd")
        else:
            for h_i, tgt in enumerate((S["st"], S["et"])):
                _chk(lib.hirest_bce_masked_f32(logits[h_i].data_ptr(), tgt.data_ptr(), mm32.data_ptr(), B, T, 0.5 * g, scratch.data_ptr(),
                                               dl[h_i].data_ptr(), ops.stream_ptr()), "bce_masked")
            _scale_by_upstream(dl, gloss)
            ws, we = P["start_predictor.0.weight"], P["end_predictor.0.weight"]
            G["start_predictor.0.weight"] = _K.colsum(feats, weight=dl[0]).reshape(1, Hd)
            G["end_predictor.0.weight"] = _K.colsum(feats, weight=dl[1]).reshape(1, Hd)
            G["start_predictor.0.bias"] = _K.colsum(dl[0].reshape(R, 1))
            G["end_predictor.0.bias"] = _K.colsum(dl[1].reshape(R, 1))
            _chk(lib.hirest_heads_bwd_f32(dl.data_ptr(), R, Hd, 2, ws.data_ptr(), we.data_ptr(), None, dx.data_ptr(), ops.stream_ptr()), "heads_bwd")
        _encoder_backward(model, P, S, dx, G)
        ctx.S = None
        return (None, None, None) + tuple(_shaped(G[n], P[n]) for n in names)


def _attn_fwd(q, ldq, k, v, ldkv, mask, B, Tq, Tk, heads, addc, drop, seed):
    lib = _lib.load()
    Pm = torch.empty((B, heads, Tq, Tk), dtype=torch.float32, device=q.device)
    cx = torch.empty((B * Tq, heads * 64), dtype=torch.float32, device=q.device)
    _chk(lib.hirest_attention_train_fwd_qkv_f32(q.data_ptr(), ldq, k.data_ptr(), v.data_ptr(), ldkv, mask.data_ptr() if mask is not None else None,
                                                Pm.data_ptr(), cx.data_ptr(), heads * 64, B, Tq, Tk, heads, 64, 0.125, addc, drop,
                                                seed & 0xFFFFFFFF, ops.stream_ptr()), "attention_train_fwd_qkv")
    return Pm, cx


class CaptionLoss(torch.autograd.Function):
    """train_step_captioning (modeling.py:476-527): trimmed 20-frame encoder -> teacher-forced 2-layer decoder
    (module_decoder.py:279-420) -> CrossEntropyLoss(ignore_index=-1) over the vocabulary, with its backward."""

    @staticmethod
    def forward(ctx, model, inp, names: List[str], *params):
        P = {n: _f32(p) for n, p in zip(names, params)}
        drop, seed = float(inp.get("dropout", 0.0)), int(inp.get("seed", 0))
        S = {"drop": drop, "seed": seed}
        enc = _encoder_forward(model, P, inp, S)                       # [B*F, 768]
        B, F = S["B"], S["T"]
        ids, amask, target = inp["input_ids"], inp["decoder_mask"], inp["output_ids"]
        L = ids.shape[1]
        R, Hd, heads = B * L, 768, model.heads
        dev = enc.device
        lib = _lib.load()
        ids32 = ids.to(torch.int32).to(dev).contiguous()
        We, pos = P[_D + "embeddings.word_embeddings.weight"], P[_D + "embeddings.position_embeddings.weight"]
        e0 = torch.empty((R, Hd), dtype=torch.float32, device=dev)
        _chk(lib.hirest_embedding_fwd_f32(ids32.data_ptr(), We.data_ptr(), pos.data_ptr(), e0.data_ptr(), R, L, Hd, ops.stream_ptr()), "embedding_fwd")
        e1 = _K.layernorm(e0, P[_D + "embeddings.LayerNorm.weight"], P[_D + "embeddings.LayerNorm.bias"], 1e-12)
        x = _K.dropout_add(e1, None, drop, seed + 101)
        # self-attention mask (module_decoder.py:388-397): -10000 on future keys and on padded keys; host index arithmetic
        future = torch.triu(torch.ones((L, L), dtype=torch.bool), diagonal=1)
        blocked = future.unsqueeze(0) | (amask.cpu() == 0).unsqueeze(1)
        smask = torch.where(blocked, torch.tensor(-10000.0), torch.tensor(0.0)).to(dev).contiguous()
        dl_layers = []
        for i in range(len(model.clip4cap_model.decoder.decoder.layer)):
            p = _D + f"decoder.layer.{i}."
            sa, ea = p + "slf_attn.", p + "enc_attn."
            wqkv = torch.cat([P[sa + "att.query.weight"], P[sa + "att.key.weight"], P[sa + "att.value.weight"]], 0).contiguous()
            bqkv = torch.cat([P[sa + "att.query.bias"], P[sa + "att.key.bias"], P[sa + "att.value.bias"]], 0).contiguous()
            qkv = _K.gemm(x, wqkv, bqkv)
            P1, c1 = _attn_fwd(qkv, 3 * Hd, qkv[:, Hd:], qkv[:, 2 * Hd:], 3 * Hd, smask, B, L, L, heads, 0.0, drop, seed + 110 + 8 * i)
            o1 = _K.gemm(c1, P[sa + "output.dense.weight"], P[sa + "output.dense.bias"])
            s_pre = _K.dropout_add(o1, x, drop, seed + 111 + 8 * i)
            sx = _K.layernorm(s_pre, P[sa + "output.LayerNorm.weight"], P[sa + "output.LayerNorm.bias"], 1e-12)
            q2 = _K.gemm(sx, P[ea + "att.query.weight"], P[ea + "att.query.bias"])
            wkv = torch.cat([P[ea + "att.key.weight"], P[ea + "att.value.weight"]], 0).contiguous()
            bkv = torch.cat([P[ea + "att.key.bias"], P[ea + "att.value.bias"]], 0).contiguous()
            kv = _K.gemm(enc, wkv, bkv)
            P2, c2 = _attn_fwd(q2, Hd, kv, kv[:, Hd:], 2 * Hd, None, B, L, F, heads, -10000.0, drop, seed + 112 + 8 * i)
            o2 = _K.gemm(c2, P[ea + "output.dense.weight"], P[ea + "output.dense.bias"])
            d_pre = _K.dropout_add(o2, sx, drop, seed + 113 + 8 * i)
            dx_ = _K.layernorm(d_pre, P[ea + "output.LayerNorm.weight"], P[ea + "output.LayerNorm.bias"], 1e-12)
            hpre = _K.gemm(dx_, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"])
            hh = _K.act(hpre, 1)
            y = _K.gemm(hh, P[p + "output.dense.weight"], P[p + "output.dense.bias"])
            x_pre = _K.dropout_add(y, dx_, drop, seed + 114 + 8 * i)
            xn = _K.layernorm(x_pre, P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], 1e-12)
            dl_layers.append(dict(x=x, wqkv=wqkv, qkv=qkv, P1=P1, c1=c1, s_pre=s_pre, sx=sx, q2=q2, wkv=wkv, kv=kv, P2=P2, c2=c2, d_pre=d_pre,
                                  d=dx_, hpre=hpre, hh=hh, x_pre=x_pre))
            x = xn
        cp = _D + "classifier.cls.predictions."
        tpre = _K.gemm(x, P[cp + "transform.dense.weight"], P[cp + "transform.dense.bias"])
        tg = _K.act(tpre, 1)
        tnorm = _K.layernorm(tg, P[cp + "transform.LayerNorm.weight"], P[cp + "transform.LayerNorm.bias"], 1e-12)
        V = We.shape[0]
        Vp = (V + 15) // 16 * 16                                       # reduction-dim granule of the dX GEMM; pad logits can never win
        Wp = torch.zeros((Vp, Hd), dtype=torch.float32, device=dev)
        Wp[:V] = We
        bp = torch.full((Vp,), -3.0e38, dtype=torch.float32, device=dev)
        bp[:V] = P[cp + "bias"]
        logits = _K.gemm(tnorm, Wp, bp)
        tgt32 = target.to(torch.int32).reshape(-1).contiguous().to(dev)
        n_valid = int((target >= 0).sum().item())
        loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        dlog = torch.empty_like(logits)
        _chk(lib.hirest_ce_rows_f32(logits.data_ptr(), Vp, tgt32.data_ptr(), R, Vp, 1.0, n_valid, loss.data_ptr(), dlog.data_ptr(), ops.stream_ptr()),
             "ce_rows")
        del dlog
        S.update(enc=enc, L=L, ids32=ids32, e0=e0, dl=dl_layers, xlast=x, tpre=tpre, tg=tg, tnorm=tnorm, Wp=Wp, logits=logits, tgt32=tgt32,
                 n_tok=n_valid, V=V, Vp=Vp, P=P, names=names, model=model, versions=_alias_versions(zip(names, params)))
        ctx.S = S
        ctx.set_materialize_grads(False)
        return loss.reshape(())

    @staticmethod
    @_colsum_batched
    def backward(ctx, gloss):
        S = ctx.S
        if S is None:
            raise RuntimeError("hirest_amd: this loss was already back-propagated (the kernels' saved activations are released after "
                               "the first backward; retain_graph is not supported)")
        _check_alias_versions(S["versions"])
        lib = _lib.load()
        P, names, model = S["P"], S["names"], S["model"]
        B, F, L, drop, seed = S["B"], S["T"], S["L"], S["drop"], S["seed"]
        R, Hd, heads, V, Vp = B * L, 768, model.heads, S["V"], S["Vp"]
        dev = S["enc"].device
        G: Dict[str, torch.Tensor] = {}
        g = 1.0                                                     # the upstream scale is applied on the device below (no host read)
        cp = _D + "classifier.cls.predictions."
        logits = S["logits"]
        dlog = torch.empty_like(logits)
        scratch = torch.zeros((1,), dtype=torch.float32, device=dev)
        _chk(lib.hirest_ce_rows_f32(logits.data_ptr(), Vp, S["tgt32"].data_ptr(), R, Vp, g, S["n_tok"], scratch.data_ptr(), dlog.data_ptr(),
                                    ops.stream_ptr()), "ce_rows")
        _scale_by_upstream(dlog, gloss)
        dWe = _K.grad_weight(dlog, S["tnorm"], side=False)              # [Vp, 768]: the LM head's share of the tied matrix (the embedding
                                                                        # scatter below adds into it on this stream)
        G[cp + "bias"] = _K.colsum(dlog)[:V]
        dtn = _K.grad_input(dlog, S["Wp"])
        dtg, G[cp + "transform.LayerNorm.weight"], G[cp + "transform.LayerNorm.bias"] = _K.layernorm_bwd(S["tg"], dtn, P[cp + "transform.LayerNorm.weight"], 1e-12)
        dtpre = _K.act_bwd(S["tpre"], dtg, 1)
        G[cp + "transform.dense.weight"] = _K.grad_weight(dtpre, S["xlast"])
        G[cp + "transform.dense.bias"] = _K.colsum(dtpre)
        dx = _K.grad_input(dtpre, P[cp + "transform.dense.weight"])
        denc = None
        for i in reversed(range(len(S["dl"]))):
            p = _D + f"decoder.layer.{i}."
            sa, ea = p + "slf_attn.", p + "enc_attn."
            Ly = S["dl"][i]
            dxp, G[p + "output.LayerNorm.weight"], G[p + "output.LayerNorm.bias"] = _K.layernorm_bwd(Ly["x_pre"], dx, P[p + "output.LayerNorm.weight"], 1e-12)
            dy = _K.dropout_add(dxp, None, drop, seed + 114 + 8 * i)
            G[p + "output.dense.weight"] = _K.grad_weight(dy, Ly["hh"])
            G[p + "output.dense.bias"] = _K.colsum(dy)
            dhp = _K.act_bwd(Ly["hpre"], _K.grad_input(dy, P[p + "output.dense.weight"]), 1)
            G[p + "intermediate.dense.weight"] = _K.grad_weight(dhp, Ly["d"])
            G[p + "intermediate.dense.bias"] = _K.colsum(dhp)
            dd = _K.grad_input(dhp, P[p + "intermediate.dense.weight"], resid=dxp)
            # cross-attention block
            ddp, G[ea + "output.LayerNorm.weight"], G[ea + "output.LayerNorm.bias"] = _K.layernorm_bwd(Ly["d_pre"], dd, P[ea + "output.LayerNorm.weight"], 1e-12)
            do2 = _K.dropout_add(ddp, None, drop, seed + 113 + 8 * i)
            G[ea + "output.dense.weight"] = _K.grad_weight(do2, Ly["c2"])
            G[ea + "output.dense.bias"] = _K.colsum(do2)
            dc2 = _K.grad_input(do2, P[ea + "output.dense.weight"])
            dS2 = torch.empty_like(Ly["P2"])
            dq2 = torch.empty_like(Ly["q2"])
            dkv = torch.empty_like(Ly["kv"])
            kv = Ly["kv"]
            _chk(lib.hirest_attention_train_bwd_qkv_f32(Ly["q2"].data_ptr(), Hd, kv.data_ptr(), kv[:, Hd:].data_ptr(), 2 * Hd, Ly["P2"].data_ptr(),
                                                        dc2.data_ptr(), Hd, dS2.data_ptr(), dq2.data_ptr(), Hd, dkv.data_ptr(),
                                                        dkv[:, Hd:].data_ptr(), 2 * Hd, B, L, F, heads, 64, 0.125, drop,
                                                        (seed + 112 + 8 * i) & 0xFFFFFFFF, ops.stream_ptr()), "attention_train_bwd_qkv")
            G[ea + "att.query.weight"] = _K.grad_weight(dq2, Ly["sx"])
            G[ea + "att.query.bias"] = _K.colsum(dq2)
            dwkv, dbkv = _K.grad_weight(dkv, S["enc"]), _K.colsum(dkv)
            G[ea + "att.key.weight"], G[ea + "att.value.weight"] = dwkv[:Hd], dwkv[Hd:]
            G[ea + "att.key.bias"], G[ea + "att.value.bias"] = dbkv[:Hd], dbkv[Hd:]
            denc = _K.grad_input(dkv, Ly["wkv"], resid=denc)
            ds = _K.grad_input(dq2, P[ea + "att.query.weight"], resid=ddp)
            # self-attention block
            dsp, G[sa + "output.LayerNorm.weight"], G[sa + "output.LayerNorm.bias"] = _K.layernorm_bwd(Ly["s_pre"], ds, P[sa + "output.LayerNorm.weight"], 1e-12)
            do1 = _K.dropout_add(dsp, None, drop, seed + 111 + 8 * i)
            G[sa + "output.dense.weight"] = _K.grad_weight(do1, Ly["c1"])
            G[sa + "output.dense.bias"] = _K.colsum(do1)
            dc1 = _K.grad_input(do1, P[sa + "output.dense.weight"])
            dS1 = torch.empty_like(Ly["P1"])
            dqkv = torch.empty_like(Ly["qkv"])
            qkv = Ly["qkv"]
            _chk(lib.hirest_attention_train_bwd_qkv_f32(qkv.data_ptr(), 3 * Hd, qkv[:, Hd:].data_ptr(), qkv[:, 2 * Hd:].data_ptr(), 3 * Hd,
                                                        Ly["P1"].data_ptr(), dc1.data_ptr(), Hd, dS1.data_ptr(), dqkv.data_ptr(), 3 * Hd,
                                                        dqkv[:, Hd:].data_ptr(), dqkv[:, 2 * Hd:].data_ptr(), 3 * Hd, B, L, L, heads, 64, 0.125,
                                                        drop, (seed + 110 + 8 * i) & 0xFFFFFFFF, ops.stream_ptr()), "attention_train_bwd_qkv")
            dwqkv, dbqkv = _K.grad_weight(dqkv, Ly["x"]), _K.colsum(dqkv)
            for k, nm in enumerate(("query", "key", "value")):
                G[sa + f"att.{nm}.weight"] = dwqkv[k * Hd:(k + 1) * Hd]
                G[sa + f"att.{nm}.bias"] = dbqkv[k * Hd:(k + 1) * Hd]
            dx = _K.grad_input(dqkv, Ly["wqkv"], resid=dsp)
        # decoder embeddings: LayerNorm, position table, and the input-embedding share of the tied matrix
        dxe = _K.dropout_add(dx, None, drop, seed + 101)
        de0, G[_D + "embeddings.LayerNorm.weight"], G[_D + "embeddings.LayerNorm.bias"] = \
            _K.layernorm_bwd(S["e0"], dxe, P[_D + "embeddings.LayerNorm.weight"], 1e-12)
        pos = P[_D + "embeddings.position_embeddings.weight"]
        dpos = torch.zeros_like(pos)
        _K.colsum(de0.reshape(B, L * Hd), out=dpos[:L].reshape(-1))
        G[_D + "embeddings.position_embeddings.weight"] = dpos
        _chk(lib.hirest_embedding_bwd_f32(S["ids32"].data_ptr(), de0.data_ptr(), dWe.data_ptr(), R, Hd, ops.stream_ptr()), "embedding_bwd")
        G[_D + "embeddings.word_embeddings.weight"] = dWe[:V]
        _encoder_backward(model, P, S, denc, G)
        ctx.S = None
        return (None, None, None) + tuple(_shaped(G[n], P[n]) for n in names)


def time_grid(n_valid: torch.Tensor, T: int) -> torch.Tensor:
    """modeling.py:176-193 as a [B, T] table: (linspace(0, 1, n) - 0.5) * 2 over the n valid frames, zero padded — built on the host
    with torch.linspace, as the reference builds it.  The training step does NOT call this (reading n_valid back would stall its
    enqueue): it uses hirest_joint_time_grid_f32, which tests/test_gpu_train.py holds to this table bit for bit."""
    rows = []
    for n in n_valid.cpu().tolist():
        rows.append(torch.cat([(torch.linspace(0, 1, int(n)) - 0.5) * 2, torch.zeros(T - int(n))]))
    return torch.stack(rows).to(n_valid.device)


def _dropout_seed() -> int:
    """A fresh seed per step from torch's CPU generator (so torch.manual_seed makes runs repeatable), mixed with the process rank:
    data-parallel ranks started from the same torch seed must not share their dropout masks."""
    s = int(torch.randint(0, 2 ** 31 - 1024, (1,)).item())
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        s = (s + 0x9E3779B1 * (dist.get_rank() + 1)) % (2 ** 31 - 1024)
    return s


def _parameters_by_name(model, names: List[str]):
    """The Parameters behind `names`, through (submodule, attribute) pairs resolved once per model: dict(model.named_parameters())
    walks the whole module tree (0.65 ms of host time per step, more with the CLIP towers attached).  Read through getattr every
    step, so a Parameter object swapped after the first step is still found; after replacing a whole SUBMODULE of a model that
    has already trained, drop the table (``model.__dict__.pop("_train_param_sites", None)``)."""
    cache = model.__dict__.setdefault("_train_param_sites", {})
    out = []
    for n in names:
        site = cache.get(n)
        if site is None:
            prefix, _, leaf = n.rpartition(".")
            site = cache[n] = (model.get_submodule(prefix) if prefix else model, leaf)
        out.append(getattr(site[0], site[1]))
    return out


def _train(model, batch, task) -> Dict[str, torch.Tensor]:
    dev = model.clip_g_map.weight.device
    if dev.type != "cuda":
        raise RuntimeError("hirest_amd.MomentModel trains on MI355X only (no CPU fallback); move the model to a GPU")
    with torch.no_grad():
        text = model._text_feat(batch, dev)
    inp = {"task": task, "vis": ops.to_device(batch["vis_feats"], dev), "text": text, "vis_mask": ops.to_device(batch["vis_mask"], dev),
           "moment_mask": ops.to_device(batch["moment_mask"], dev),
           "dropout": 0.1 if model.training else 0.0, "seed": _dropout_seed()}
    if model.use_asr:
        inp["asr"] = ops.to_device(batch["asr_feats"], dev)
    fn = MomentLoss
    if task == "moment_segmentation":
        inp["boundary_mask"] = ops.to_device(batch["prev_boundary_mask"], dev)
        inp["segment_target"] = ops.to_device(batch["moment_segmentation_target"], dev)
    elif task == "moment_retrieval":
        inp["start_target"] = ops.to_device(batch["moment_retrieval_start_target"], dev)
        inp["end_target"] = ops.to_device(batch["moment_retrieval_end_target"], dev)
    else:   # step_captioning (modeling.py:476-527): trimmed moment frames, all-ones masks, teacher-forcing triples of target_text
        args = model.args
        max_frames = int(getattr(args, "max_frames_step_captioning", 20)) if args is not None else 20
        B = inp["vis"].shape[0]
        rows = model._trim_rows(batch["moment_mask"], max_frames, dev)      # from the batch's own (CPU) mask: no device round trip
        inp["vis"] = model._trim(inp["vis"].float(), None, max_frames, idx=rows)
        if model.use_asr:
            inp["asr"] = model._trim(inp["asr"].float(), None, max_frames, idx=rows)
        ones = torch.ones((B, max_frames), dtype=torch.long, device=dev)
        inp["vis_mask"], inp["moment_mask"] = ones, ones
        tt = batch["target_text"]
        inp["input_ids"] = torch.tensor([list(t[5]) for t in tt], dtype=torch.long)
        inp["decoder_mask"] = torch.tensor([list(t[6]) for t in tt], dtype=torch.long)
        inp["output_ids"] = torch.tensor([list(t[7]) for t in tt], dtype=torch.long)
        fn = CaptionLoss
    names = task_param_names(model, task)
    return {"loss": fn.apply(model, inp, names, *_parameters_by_name(model, names))}


def train_moment_retrieval(model, batch) -> Dict[str, torch.Tensor]:
    """modeling.py:226-270.  ``batch`` keys as in the reference (hirest_dataset.py:409-531)."""
    return _train(model, batch, "moment_retrieval")


def train_moment_segmentation(model, batch) -> Dict[str, torch.Tensor]:
    """modeling.py:323-351: one teacher-forced step of the iterative segmentation (previous boundary -> next boundary)."""
    return _train(model, batch, "moment_segmentation")


def train_step_captioning(model, batch) -> Dict[str, torch.Tensor]:
    """modeling.py:476-527: ``batch['target_text'][i]`` = the reference's 9-tuple whose fields 5, 6, 7 are the decoder input
    ids, the decoder mask and the output ids (-1 = ignored) of one caption."""
    return _train(model, batch, "step_captioning")


def allreduce_gradients(parameters, group=None, bucket_bytes: int = 64 << 20) -> None:
    """Average ``.grad`` over the ranks of `group` (RCCL on GPU tensors, gloo on CPU tensors): what run.py:93's
    DistributedDataParallel is there for.  (The reference calls ``model.module.train_step`` directly, run.py:247-255, which
    bypasses DDP's forward and with it the reducer's bookkeeping; an explicit bucketed all-reduce after ``loss.backward()`` is
    the dependable form of the same exchange.)  Gradients are packed into flat fp32 buckets of up to `bucket_bytes` so the
    63 M parameters travel as a handful of large messages over xGMI instead of ~130 small ones.  Only parameters that
    received a gradient on SOME rank take part (agreed on with one all-reduced bitmask): a task leaves the other tasks' heads,
    ``moment_conv`` and — for the moment tasks — the whole caption decoder without a gradient, and those must stay ``None``
    as in a single-process run so that AdamW skips them (no weight decay, no moment decay, no step count)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    has = torch.tensor([1 if p.grad is not None else 0 for p in params], dtype=torch.int32, device=params[0].device)
    dist.all_reduce(has, op=dist.ReduceOp.MAX, group=group)
    params = [p for p, h in zip(params, has.tolist()) if h]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bucket])
        dist.all_reduce(flat, group=group)
        flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(flat[off:off + n].reshape(p.shape))
            off += n
        bucket, size = [], 0
    for p in params:
        bucket.append(p)
        size += p.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
