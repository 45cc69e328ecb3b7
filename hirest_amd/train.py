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
* step_captioning: not implemented (decoder backward), raises.

Dropout (VisualEmbeddings / attention probabilities / VisualSelfOutput / VisualOutput, p = 0.1 in train mode:
module_visual.py:116-183) uses a counter-based mask; ``model.eval()`` switches it off, which is also how the gradient parity
tests pin the arithmetic against the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from . import _lib, ops

_V = "clip4cap_model.visual."


def _chk(code, what):
    _lib.check(code, what)


class _K:
    """Thin tensor-level wrappers over the training entry points (device fp32 contiguous in, fresh tensors out)."""

    @staticmethod
    def gemm(a, w, bias=None, resid=None, periodic=None, period=0, act=0):
        lib = _lib.load()
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _chk(lib.hirest_gemm_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else None,
                                 resid.data_ptr() if resid is not None else None, N,
                                 periodic.data_ptr() if periodic is not None else None, period,
                                 out.data_ptr(), N, M, N, K, act, ops.stream_ptr()), "hirest_gemm_f32")
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
    def grad_input(dy, w):
        """dX = dY @ W for y = x W^T:  dY [R, N], W [N, K] -> [R, K]  (the GEMM's W-operand is W^T [K, N]; N % 16 == 0)."""
        if w.shape[0] % 16 != 0:
            raise RuntimeError(f"grad_input: out_features {w.shape[0]} must be a multiple of 16")
        return _K.gemm(dy, _K.transpose_pad(w))

    @staticmethod
    def grad_weight(dy, x):
        """dW = dY^T @ X:  dY [R, N], X [R, K] -> [N, K] (reduction over the zero-padded rows)."""
        return _K.gemm(_K.transpose_pad(dy), _K.transpose_pad(x))

    @staticmethod
    def colsum(x, weight=None, select=None, value=0):
        R, Cc = x.shape
        out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
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


def _f32(t):
    return t.detach().float().contiguous()


# parameters on a task's graph, in the order the Function receives them / returns gradients for
def task_param_names(model, task: str) -> List[str]:
    names = ["clip_g_map.weight", "clip_g_map.bias",
             "clip4cap_model.normalize_video.visual_norm2d.weight", "clip4cap_model.normalize_video.visual_norm2d.bias",
             "clip_g_map_text.weight", "clip_g_map_text.bias"]
    if model.use_asr:
        names += ["asr_enc_layer.0.weight", "asr_enc_layer.0.bias", "asr_enc_layer.1.weight", "asr_enc_layer.1.bias"]
    names += ["temporal_embed.0.weight", "temporal_embed.0.bias", "temporal_embed.2.weight", "temporal_embed.2.bias",
              "mask_embed.weight"] + (["boundary_embed.weight"] if task == "moment_segmentation" else []) + [
              _V + "embeddings.word_embeddings.weight", _V + "embeddings.word_embeddings.bias",
              _V + "embeddings.position_embeddings.weight", _V + "embeddings.LayerNorm.weight", _V + "embeddings.LayerNorm.bias"]
    for i in range(len(model.clip4cap_model.visual.encoder.layer)):
        p = _V + f"encoder.layer.{i}."
        for leaf in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                     "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"):
            names += [p + leaf + ".weight", p + leaf + ".bias"]
    if task == "moment_segmentation":
        names += ["segment_predictor.0.weight", "segment_predictor.0.bias"]
    else:
        names += ["start_predictor.0.weight", "start_predictor.0.bias", "end_predictor.0.weight", "end_predictor.0.bias"]
    return names


class MomentLoss(torch.autograd.Function):
    """The training losses of modeling.py:226-270 (moment retrieval: (BCE_start + BCE_end) / 2) and :323-351 (moment
    segmentation: cross-entropy over the moment's frames) with their backward, every step a kernel of this library."""

    @staticmethod
    def forward(ctx, model, inp: Dict[str, torch.Tensor], names: List[str], *params):
        lib = _lib.load()
        P = {n: _f32(p) for n, p in zip(names, params)}
        vis, text, asr = inp["vis"], inp["text"], inp.get("asr")
        vmask, mmask = inp["vis_mask"], inp["moment_mask"]
        B, T, _ = vis.shape
        R, E, Hd = B * T, 512, 768
        heads = model.heads
        drop = float(inp.get("dropout", 0.0))
        seed = int(inp.get("seed", 0))
        seg = inp["task"] == "moment_segmentation"
        S = {"B": B, "T": T, "drop": drop, "seed": seed, "seg": seg}
        vis2 = vis.reshape(R, -1).float().contiguous()
        # ---- fusion (modeling.py:158-199)
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
        f = torch.empty_like(base)
        bm32 = inp["boundary_mask"].to(torch.int32).contiguous() if seg else None
        bemb = P["boundary_embed.weight"] if seg else _f32(model.boundary_embed.weight)
        _chk(lib.hirest_joint_mask_add(base.data_ptr(), mm32.data_ptr(), bm32.data_ptr() if seg else None, P["mask_embed.weight"].data_ptr(),
                                       bemb.data_ptr(), f.data_ptr(), R, E, ops.stream_ptr()), "mask_add")
        S["bm32"] = bm32
        # ---- VisualModel (module_visual.py:104-264, 396-424)
        x0 = _K.gemm(f, P[_V + "embeddings.word_embeddings.weight"], P[_V + "embeddings.word_embeddings.bias"],
                     periodic=P[_V + "embeddings.position_embeddings.weight"], period=T)
        xe = _K.layernorm(x0, P[_V + "embeddings.LayerNorm.weight"], P[_V + "embeddings.LayerNorm.bias"], 1e-12)
        x = _K.dropout_add(xe, None, drop, seed + 1)
        S.update(vis2=vis2, v0=v0, v=v, t=t, tn=tn, tin=tin, f=f, x0=x0, mm32=mm32, n_valid=n_valid, text=text.float().contiguous())
        layers = []
        L = len(model.clip4cap_model.visual.encoder.layer)
        for i in range(L):
            p = _V + f"encoder.layer.{i}."
            wqkv = torch.cat([P[p + "attention.self.query.weight"], P[p + "attention.self.key.weight"], P[p + "attention.self.value.weight"]], 0).contiguous()
            bqkv = torch.cat([P[p + "attention.self.query.bias"], P[p + "attention.self.key.bias"], P[p + "attention.self.value.bias"]], 0).contiguous()
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
        feats = x
        # ---- heads + loss (modeling.py:218-219, 249-263 / 319, 343-344)
        loss = torch.zeros((1,), dtype=torch.float32, device=vis.device)
        if seg:
            wsg = P["segment_predictor.0.weight"]
            bias3 = torch.cat([P["segment_predictor.0.bias"], torch.zeros(2, device=vis.device)]).contiguous()
            logits = torch.empty((1, R), dtype=torch.float32, device=vis.device)
            _chk(lib.hirest_linear_heads(feats.data_ptr(), R, Hd, 1, wsg.data_ptr(), None, None, bias3.data_ptr(), logits.data_ptr(),
                                         ops.stream_ptr()), "linear_heads")
            st = inp["segment_target"].to(torch.int32).contiguous()
            et = st
            dl = torch.empty_like(logits)
            _chk(lib.hirest_ce_masked_f32(logits.data_ptr(), mm32.data_ptr(), st.data_ptr(), B, T, 1.0, loss.data_ptr(), dl.data_ptr(),
                                          ops.stream_ptr()), "ce_masked")
        else:
            ws, we = P["start_predictor.0.weight"], P["end_predictor.0.weight"]
            bias3 = torch.cat([P["start_predictor.0.bias"], P["end_predictor.0.bias"], torch.zeros(1, device=vis.device)]).contiguous()
            logits = torch.empty((2, R), dtype=torch.float32, device=vis.device)
            _chk(lib.hirest_linear_heads(feats.data_ptr(), R, Hd, 2, ws.data_ptr(), we.data_ptr(), None, bias3.data_ptr(), logits.data_ptr(),
                                         ops.stream_ptr()), "linear_heads")
            st = inp["start_target"].to(torch.int32).contiguous()
            et = inp["end_target"].to(torch.int32).contiguous()
            dl = torch.empty_like(logits)
            for h_i, tgt in enumerate((st, et)):
                _chk(lib.hirest_bce_masked_f32(logits[h_i].data_ptr(), tgt.data_ptr(), mm32.data_ptr(), B, T, 0.5, loss.data_ptr(),
                                               dl[h_i].data_ptr(), ops.stream_ptr()), "bce_masked")
        S.update(layers=layers, feats=feats, logits=logits, st=st, et=et, P=P, names=names, model=model)
        ctx.S = S
        ctx.set_materialize_grads(False)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        S = ctx.S
        lib = _lib.load()
        P, names, model = S["P"], S["names"], S["model"]
        B, T, drop, seed = S["B"], S["T"], S["drop"], S["seed"]
        R, E, Hd = B * T, 512, 768
        heads = model.heads
        dev = S["feats"].device
        G: Dict[str, torch.Tensor] = {}
        g = float(gloss.item()) if gloss is not None else 1.0       # upstream scale (GradScaler / accumulation); a host scalar
        logits, mm32 = S["logits"], S["mm32"]
        dl = torch.empty_like(logits)
        scratch = torch.zeros((1,), dtype=torch.float32, device=dev)
        feats = S["feats"]
        dx = torch.empty((R, Hd), dtype=torch.float32, device=dev)
        if S["seg"]:
            _chk(lib.hirest_ce_masked_f32(logits.data_ptr(), mm32.data_ptr(), S["st"].data_ptr(), B, T, g, scratch.data_ptr(), dl.data_ptr(),
                                          ops.stream_ptr()), "ce_masked")
            wsg = P["segment_predictor.0.weight"]
            G["segment_predictor.0.weight"] = _K.colsum(feats, weight=dl[0]).reshape(1, Hd)
            G["segment_predictor.0.bias"] = _K.colsum(dl[0].reshape(R, 1))
            _chk(lib.hirest_heads_bwd_f32(dl.data_ptr(), R, Hd, 1, wsg.data_ptr(), None, None, dx.data_ptr(), ops.stream_ptr()), "heads_bwd")
        else:
            for h_i, tgt in enumerate((S["st"], S["et"])):
                _chk(lib.hirest_bce_masked_f32(logits[h_i].data_ptr(), tgt.data_ptr(), mm32.data_ptr(), B, T, 0.5 * g, scratch.data_ptr(),
                                               dl[h_i].data_ptr(), ops.stream_ptr()), "bce_masked")
            ws, we = P["start_predictor.0.weight"], P["end_predictor.0.weight"]
            G["start_predictor.0.weight"] = _K.colsum(feats, weight=dl[0]).reshape(1, Hd)
            G["end_predictor.0.weight"] = _K.colsum(feats, weight=dl[1]).reshape(1, Hd)
            G["start_predictor.0.bias"] = _K.colsum(dl[0].reshape(R, 1))
            G["end_predictor.0.bias"] = _K.colsum(dl[1].reshape(R, 1))
            _chk(lib.hirest_heads_bwd_f32(dl.data_ptr(), R, Hd, 2, ws.data_ptr(), we.data_ptr(), None, dx.data_ptr(), ops.stream_ptr()), "heads_bwd")
        # ---- encoder layers, last to first
        L = len(S["layers"])
        for i in reversed(range(L)):
            p = _V + f"encoder.layer.{i}."
            Ly = S["layers"][i]
            dxp, G[p + "output.LayerNorm.weight"], G[p + "output.LayerNorm.bias"] = _K.layernorm_bwd(Ly["x_pre"], dx, P[p + "output.LayerNorm.weight"], 1e-12)
            dy = _K.dropout_add(dxp, None, drop, seed + 12 + 4 * i)           # through dropout(y); the residual branch gets dxp as is
            G[p + "output.dense.weight"] = _K.grad_weight(dy, Ly["hh"])
            G[p + "output.dense.bias"] = _K.colsum(dy)
            dh = _K.grad_input(dy, P[p + "output.dense.weight"])
            dhp = _K.act_bwd(Ly["hpre"], dh, 1)
            G[p + "intermediate.dense.weight"] = _K.grad_weight(dhp, Ly["aa"])
            G[p + "intermediate.dense.bias"] = _K.colsum(dhp)
            da = _K.dropout_add(_K.grad_input(dhp, P[p + "intermediate.dense.weight"]), dxp, 0.0, 0)      # + residual path
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
            dx = _K.dropout_add(_K.grad_input(dqkv, Ly["wqkv"]), dap, 0.0, 0)                               # + residual path
        # ---- embeddings
        dxe = _K.dropout_add(dx, None, drop, seed + 1)
        dx0, G[_V + "embeddings.LayerNorm.weight"], G[_V + "embeddings.LayerNorm.bias"] = \
            _K.layernorm_bwd(S["x0"], dxe, P[_V + "embeddings.LayerNorm.weight"], 1e-12)
        pos = P[_V + "embeddings.position_embeddings.weight"]
        dpos = torch.zeros_like(pos)
        dpos[:T] = _K.colsum(dx0.reshape(B, T * Hd)).reshape(T, Hd)
        G[_V + "embeddings.position_embeddings.weight"] = dpos
        G[_V + "embeddings.word_embeddings.weight"] = _K.grad_weight(dx0, S["f"])
        G[_V + "embeddings.word_embeddings.bias"] = _K.colsum(dx0)
        df = _K.grad_input(dx0, P[_V + "embeddings.word_embeddings.weight"])
        # ---- fusion
        G["mask_embed.weight"] = torch.stack([_K.colsum(df, select=mm32.reshape(-1), value=k) for k in (0, 1)])
        if S["seg"]:
            G["boundary_embed.weight"] = torch.stack([_K.colsum(df, select=S["bm32"].reshape(-1), value=k) for k in (0, 1)])
        dv = torch.empty_like(df)
        dtn = torch.empty((B, E), dtype=torch.float32, device=dev)
        _chk(lib.hirest_joint_base_bwd_f32(df.data_ptr(), S["v"].data_ptr(), S["tn"].data_ptr(), dv.data_ptr(), dtn.data_ptr(), B, T, E,
                                           ops.stream_ptr()), "joint_base_bwd")
        # temporal embedding: Linear(1, E) -> tanh -> Linear(E, E) over the normalised time grid
        G["temporal_embed.2.weight"] = _K.grad_weight(df, S["tin"])
        G["temporal_embed.2.bias"] = _K.colsum(df)
        dpre = _K.act_bwd(S["tin"], _K.grad_input(df, P["temporal_embed.2.weight"]), 3)
        time = time_grid(S["n_valid"], T).reshape(R).contiguous()
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
        ctx.S = None
        return (None, None, None) + tuple(G[n].reshape(P[n].shape) for n in names)


def time_grid(n_valid: torch.Tensor, T: int) -> torch.Tensor:
    """modeling.py:176-193 as a [B, T] table: (linspace(0, 1, n) - 0.5) * 2 over the n valid frames, zero padded.  Host index
    arithmetic (the reference builds it on the host too), uploaded once per step."""
    rows = []
    for n in n_valid.cpu().tolist():
        rows.append(torch.cat([(torch.linspace(0, 1, int(n)) - 0.5) * 2, torch.zeros(T - int(n))]))
    return torch.stack(rows).to(n_valid.device)


def _train(model, batch, task) -> Dict[str, torch.Tensor]:
    dev = model.clip_g_map.weight.device
    if dev.type != "cuda":
        raise RuntimeError("hirest_amd.MomentModel trains on MI355X only (no CPU fallback); move the model to a GPU")
    with torch.no_grad():
        text = model._text_feat(batch, dev)
    inp = {"task": task, "vis": batch["vis_feats"].to(dev), "text": text, "vis_mask": batch["vis_mask"].to(dev),
           "moment_mask": batch["moment_mask"].to(dev),
           "dropout": 0.1 if model.training else 0.0, "seed": int(torch.randint(0, 2 ** 31 - 64, (1,)).item())}
    if task == "moment_segmentation":
        inp["boundary_mask"] = batch["prev_boundary_mask"].to(dev)
        inp["segment_target"] = batch["moment_segmentation_target"].to(dev)
    else:
        inp["start_target"] = batch["moment_retrieval_start_target"].to(dev)
        inp["end_target"] = batch["moment_retrieval_end_target"].to(dev)
    if model.use_asr:
        inp["asr"] = batch["asr_feats"].to(dev)
    names = task_param_names(model, task)
    named = dict(model.named_parameters())
    return {"loss": MomentLoss.apply(model, inp, names, *[named[n] for n in names])}


def train_moment_retrieval(model, batch) -> Dict[str, torch.Tensor]:
    """modeling.py:226-270.  ``batch`` keys as in the reference (hirest_dataset.py:409-531)."""
    return _train(model, batch, "moment_retrieval")


def train_moment_segmentation(model, batch) -> Dict[str, torch.Tensor]:
    """modeling.py:323-351: one teacher-forced step of the iterative segmentation (previous boundary -> next boundary)."""
    return _train(model, batch, "moment_segmentation")


def allreduce_gradients(parameters, group=None, bucket_bytes: int = 64 << 20) -> None:
    """Average ``.grad`` over the ranks of `group` (RCCL on GPU tensors, gloo on CPU tensors): what run.py:93's
    DistributedDataParallel is there for.  (The reference calls ``model.module.train_step`` directly, run.py:247-255, which
    bypasses DDP's forward and with it the reducer's bookkeeping; an explicit bucketed all-reduce after ``loss.backward()`` is
    the dependable form of the same exchange.)  Gradients are packed into flat fp32 buckets of up to `bucket_bytes` so the
    63 M parameters travel as a handful of large messages over xGMI instead of ~130 small ones; a parameter without a gradient
    on this rank contributes zeros (every rank must present the same buckets)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    params = [p for p in parameters if p.requires_grad]
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
