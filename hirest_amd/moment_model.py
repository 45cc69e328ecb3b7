"""Drop-in for the inference side of the reference's joint model, ``MomentModel.test_step``
(/root/reference/modeling.py:141-153): moment retrieval (modeling.py:272-310) and iterative moment
segmentation (modeling.py:353-474) over precomputed 1-fps EVA-CLIP frame features + ASR features.

Same constructor arguments, batch dict keys (hirest_dataset.py:409-531: ``tasks``, ``vis_feats``,
``vis_mask``, ``moment_mask``, ``asr_feats``, ``clip_text_ids``, ``moment_bound_frames``) and result dict
(``prediction`` / ``raw_predictions``) as the reference; the parameter tree reproduces the reference's
state-dict keys (``clip_g_map``, ``asr_enc_layer.{0,1}``, ``temporal_embed.{0,2}``, ``mask_embed``,
``boundary_embed``, ``{start,end,segment}_predictor.0``, ``clip4cap_model.visual.*`` ...), so a reference
``BEST.pth`` loads with ``load_state_dict(strict=False)`` exactly as ``trainer_base.py:128-147`` does.

All math runs in the fp32 kernels of csrc/joint.hip (+ the LayerNorm kernel).  What differs from the
reference is where time goes: the 20 segmentation iterations run back to back on the device —
masks, softmax, arg-max, threshold walk and step list are device state, with ONE device->host copy at the
end instead of B x 20 ``.cpu().tolist()`` syncs — and the loop-invariant part of the fusion is hoisted.
Step captioning (modeling.py:556-632): trim_feats, the same fusion/encoder on 20 frames, then beam search over the
2-layer decoder — decoder math in the fp32 kernels, beam bookkeeping on the host as in the reference.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import json
import os
from copy import deepcopy
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib, ops

_HERE = os.path.dirname(os.path.abspath(__file__))


def _p(*shape):
    return nn.Parameter(torch.zeros(*shape))


class _Lin(nn.Module):
    def __init__(self, out_f, in_f):
        super().__init__()
        self.weight, self.bias = _p(out_f, in_f), _p(out_f)


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight, self.bias = _p(d), _p(d)


class _Emb(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = _p(n, d)


def _seq(**mods):
    m = nn.Module()
    for k, v in mods.items():
        m.add_module(k, v)
    return m


_BUILD_CLIP = "build"      # default of MomentModel(clip_model=...): construct the CLIP as the reference's __init__ does


class MomentModel(nn.Module):
    """modeling.py:18-129 (inference subset).

    ``MomentModel(n_frames, asr_dim, args)`` — the reference's three-argument call (run.py:52-56) — builds and freezes its
    own text encoder exactly as modeling.py:115-123 does: ``build_eva_model_and_transforms("EVA_CLIP_g_14",
    pretrained="./pretrained_weights/eva_clip_psz14.pt")``, ``.float()``, ``.eval()``, ``freeze_clip()``; a missing
    checkpoint raises ``FileNotFoundError`` there as it does in the reference.  The towers stay on the host until the
    model is moved; their kernel-ready device buffers are prepared on the first ``encode_text`` after ``.to(cuda)``.
    ``args.clip_model_name`` / ``args.clip_pretrained`` (not reference options; default to the two literals above)
    redirect the build, e.g. to a synthetic checkpoint offline.
    Pass ``clip_model=<EVA_CLIP>`` to share an already built encoder, or ``clip_model=None`` for a model that is fed
    ``batch['text_feat']`` (the commented-out alternative at modeling.py:284) and owns no CLIP."""

    def __init__(self, n_frames=-1, asr_dim=-1, args=None, clip_model=_BUILD_CLIP, max_position_embeddings=2048):
        super().__init__()
        self.args, self.n_frames, self.asr_dim = args, n_frames, asr_dim
        self.use_asr = asr_dim > 0
        E, H = 512, 768
        if self.use_asr:
            self.asr_enc_layer = _seq(**{"0": _LN(asr_dim), "1": _Lin(E, asr_dim)})
        self.temporal_embed = _seq(**{"0": _Lin(E, 1), "2": _Lin(E, E)})
        self.mask_embed, self.boundary_embed = _Emb(2, E), _Emb(2, E)
        self.start_predictor = _seq(**{"0": _Lin(1, H)})
        self.end_predictor = _seq(**{"0": _Lin(1, H)})
        self.segment_predictor = _seq(**{"0": _Lin(1, H)})
        vis = nn.Module()
        vis.embeddings = _seq(word_embeddings=_Lin(H, E), position_embeddings=_Emb(max_position_embeddings, H), LayerNorm=_LN(H))
        layers = []
        for _ in range(getattr(args, "visual_num_hidden_layers", 2) if args is not None else 2):
            lay = nn.Module()
            lay.attention = nn.Module()
            lay.attention.self = _seq(query=_Lin(H, H), key=_Lin(H, H), value=_Lin(H, H))
            lay.attention.output = _seq(dense=_Lin(H, H), LayerNorm=_LN(H))
            lay.intermediate = _seq(dense=_Lin(4 * H, H))
            lay.output = _seq(dense=_Lin(H, 4 * H), LayerNorm=_LN(H))
            layers.append(lay)
        vis.encoder = nn.Module()
        vis.encoder.layer = nn.ModuleList(layers)
        self.clip4cap_model = nn.Module()
        self.clip4cap_model.visual = vis
        self.clip4cap_model.normalize_video = _seq(visual_norm2d=_LN(E))
        # caption decoder (clip4caption/modules/module_decoder.py:279-406); LM head tied to the input embedding
        vocab = 30522
        dec = nn.Module()
        dec.embeddings = _seq(word_embeddings=_Emb(vocab, H), position_embeddings=_Emb(512, H), LayerNorm=_LN(H))
        dlayers = []
        for _ in range(getattr(args, "decoder_num_hidden_layers", 2) if args is not None else 2):
            lay = nn.Module()
            for nm in ("slf_attn", "enc_attn"):
                att = nn.Module()
                att.att = _seq(query=_Lin(H, H), key=_Lin(H, H), value=_Lin(H, H))
                att.output = _seq(dense=_Lin(H, H), LayerNorm=_LN(H))
                lay.add_module(nm, att)
            lay.intermediate = _seq(dense=_Lin(4 * H, H))
            lay.output = _seq(dense=_Lin(H, 4 * H), LayerNorm=_LN(H))
            dlayers.append(lay)
        dec.decoder = nn.Module()
        dec.decoder.layer = nn.ModuleList(dlayers)
        pred = nn.Module()
        pred.bias = _p(vocab)
        pred.transform = _seq(dense=_Lin(H, H), LayerNorm=_LN(H))
        pred.decoder = nn.Module()
        pred.decoder.weight = dec.embeddings.word_embeddings.weight       # tied (module_decoder.py:171-176)
        dec.classifier = nn.Module()
        dec.classifier.cls = nn.Module()
        dec.classifier.cls.predictions = pred
        self.clip4cap_model.decoder = dec
        self.tokenizer_vocab = None      # optional id -> token list (BERT vocab is not available offline)
        self.clip_g_map, self.clip_g_map_text = _Lin(E, 1024), _Lin(E, 1024)
        if isinstance(clip_model, str) and clip_model == _BUILD_CLIP:
            from .eva_clip import build_eva_model_and_transforms                 # modeling.py:114-123
            clip_model, self.clip_preprocess = build_eva_model_and_transforms(
                getattr(args, "clip_model_name", None) or "EVA_CLIP_g_14",
                pretrained=getattr(args, "clip_pretrained", None) or "./pretrained_weights/eva_clip_psz14.pt")
            print("Loaded EVA CLIP G")
            clip_model = clip_model.float()
            clip_model.eval()
        self.clip_model = clip_model
        self.freeze_clip()
        self.heads = 12
        self.caption_kv_cache = True   # step captioning keeps the decoder's self-attention K / V per beam (False: full-prefix recompute)
        self._cache = None
        # 'fp32' (default: the reference's own arithmetic, modeling.py:120 / run.py without --fp16) or 'bf16x3': the encoder's linear layers on
        # split operands (csrc/joint_x3.hip) — the counterpart of the reference's reduced-precision mode (torch.cuda.amp.autocast() under
        # --fp16, run.py:549-551), at 16 significand bits per product so that indices / boundaries / token ids stay the fp32 run's
        self.precision = "fp32"
        if os.environ.get("HIREST_JOINT_PRECISION"):
            self.set_precision(os.environ["HIREST_JOINT_PRECISION"])

    def set_precision(self, precision: str):
        """'fp32': every product exact fp32 (v_mfma_f32_*_f32).  'bf16x3': the VisualModel encoder's weight GEMMs (moment retrieval, the 20
        segmentation passes, the captioning encoder pass) as three bf16 MFMAs on hi + lo splits of both fp32 operands; attention, LayerNorm,
        GELU, residuals, fusion, heads and the caption decoder stay fp32."""
        if precision not in ("fp32", "bf16x3"):
            raise ValueError(f"MomentModel precision must be 'fp32' or 'bf16x3', got {precision!r}")
        self.precision = precision
        return self

    # ------------------------------------------------------------------ nn.Module plumbing
    def _apply(self, fn, *a, **k):
        self._cache = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._cache = None
        return super()._load_from_state_dict(*a, **k)

    def freeze_clip(self):   # modeling.py:126-129
        if self.clip_model is not None:
            for p in self.clip_model.parameters():
                p.requires_grad = False
            self.clip_model.eval()

    def _w(self):
        """fp32 contiguous device views + fused QKV weights, built once per parameter version.

        Most entries are views of the parameters and follow in-place optimizer updates; the fused / padded tensors
        (QKV concatenations, padded LM head, head biases, any non-fp32 parameter) are COPIES, so the cache remembers the
        ``_version`` counters of their sources and is rebuilt when one of them has moved (run.py:328-336 validates after
        every epoch of in-place AdamW steps)."""
        if self._cache is not None:
            if sum(p._version for p in self._cache["copied"]) == self._cache["copied_version"]:
                return self._cache
            self._cache = None
        dev = self.clip_g_map.weight.device
        if dev.type != "cuda":
            raise RuntimeError("hirest_amd.MomentModel runs on MI355X only (no CPU fallback); move the model to a GPU")
        f = lambda t: t.detach().float().contiguous()
        c = {"dev": dev}
        copied = []
        for name, prm in self.named_parameters():
            if not name.startswith("clip_model."):
                c[name] = f(prm)
                if c[name].data_ptr() != prm.data_ptr():
                    copied.append(prm)
        for i, lay in enumerate(self.clip4cap_model.visual.encoder.layer):
            s = lay.attention.self
            c[f"qkv_w.{i}"] = torch.cat([f(s.query.weight), f(s.key.weight), f(s.value.weight)], 0).contiguous()
            c[f"qkv_b.{i}"] = torch.cat([f(s.query.bias), f(s.key.bias), f(s.value.bias)], 0).contiguous()
        for i, lay in enumerate(self.clip4cap_model.decoder.decoder.layer):
            sa, ea = lay.slf_attn.att, lay.enc_attn.att
            c[f"dec_qkv_w.{i}"] = torch.cat([f(sa.query.weight), f(sa.key.weight), f(sa.value.weight)], 0).contiguous()
            c[f"dec_qkv_b.{i}"] = torch.cat([f(sa.query.bias), f(sa.key.bias), f(sa.value.bias)], 0).contiguous()
            c[f"dec_kv_w.{i}"] = torch.cat([f(ea.key.weight), f(ea.value.weight)], 0).contiguous()
            c[f"dec_kv_b.{i}"] = torch.cat([f(ea.key.bias), f(ea.value.bias)], 0).contiguous()
        # LM head: vocab padded to a multiple of 4 rows for the GEMM's 4-wide epilogue; pad logits are -3e38 so they
        # vanish in the log-softmax and can never enter the top-k
        we = f(self.clip4cap_model.decoder.embeddings.word_embeddings.weight)
        vb = f(self.clip4cap_model.decoder.classifier.cls.predictions.bias)
        padn = (-we.shape[0]) % 4
        c["lm_w"] = torch.cat([we, torch.zeros((padn, we.shape[1]), device=dev)], 0).contiguous() if padn else we
        c["lm_b"] = torch.cat([vb, torch.full((padn,), -3.0e38, device=dev)]).contiguous() if padn else vb
        # descriptor of the C-side decoder step (csrc/caption.hip): device pointers into the tensors above
        Dp = "clip4cap_model.decoder."
        nl = len(self.clip4cap_model.decoder.decoder.layer)
        layers = (_lib.CaptionLayer * nl)()
        for i in range(nl):
            p = Dp + f"decoder.layer.{i}."
            layers[i] = _lib.CaptionLayer(*[c[k].data_ptr() for k in (
                f"dec_qkv_w.{i}", f"dec_qkv_b.{i}", p + "slf_attn.output.dense.weight", p + "slf_attn.output.dense.bias",
                p + "slf_attn.output.LayerNorm.weight", p + "slf_attn.output.LayerNorm.bias",
                p + "enc_attn.att.query.weight", p + "enc_attn.att.query.bias", p + "enc_attn.output.dense.weight",
                p + "enc_attn.output.dense.bias", p + "enc_attn.output.LayerNorm.weight", p + "enc_attn.output.LayerNorm.bias",
                p + "intermediate.dense.weight", p + "intermediate.dense.bias", p + "output.dense.weight", p + "output.dense.bias",
                p + "output.LayerNorm.weight", p + "output.LayerNorm.bias")])
        cp = Dp + "classifier.cls.predictions."
        c["dec_layers"] = layers
        c["dec_desc"] = _lib.CaptionDecoder(
            nl, self.heads, 768, c[Dp + "decoder.layer.0.intermediate.dense.weight"].shape[0], c["lm_w"].shape[0],
            c[Dp + "embeddings.position_embeddings.weight"].shape[0],
            c[Dp + "embeddings.word_embeddings.weight"].data_ptr(), c[Dp + "embeddings.position_embeddings.weight"].data_ptr(),
            c[Dp + "embeddings.LayerNorm.weight"].data_ptr(), c[Dp + "embeddings.LayerNorm.bias"].data_ptr(), layers,
            c[cp + "transform.dense.weight"].data_ptr(), c[cp + "transform.dense.bias"].data_ptr(),
            c[cp + "transform.LayerNorm.weight"].data_ptr(), c[cp + "transform.LayerNorm.bias"].data_ptr(),
            c["lm_w"].data_ptr(), c["lm_b"].data_ptr(), None)
        c["head_bias"] = torch.cat([f(getattr(m, "0").bias) for m in
                                    (self.start_predictor, self.end_predictor, self.segment_predictor)]).contiguous()
        for lay in self.clip4cap_model.visual.encoder.layer:
            copied += [q for m in (lay.attention.self.query, lay.attention.self.key, lay.attention.self.value) for q in m.parameters()]
        for lay in self.clip4cap_model.decoder.decoder.layer:
            copied += [q for a in (lay.slf_attn.att, lay.enc_attn.att) for m in (a.query, a.key, a.value) for q in m.parameters()]
        copied += [self.clip4cap_model.decoder.embeddings.word_embeddings.weight, self.clip4cap_model.decoder.classifier.cls.predictions.bias]
        copied += [getattr(m, "0").bias for m in (self.start_predictor, self.end_predictor, self.segment_predictor)]
        c["copied"] = copied
        c["copied_version"] = sum(p._version for p in copied)
        self._cache = c
        return c

    def _x3(self):
        """Split-operand weights + the C-side descriptor of the encoder (hirest_joint_encoder_x3), built once per parameter version: every
        [N, K] fp32 weight becomes [N, 2K] bf16 (hirest_split2_bf16: per 32 k, hi | lo).  Lives in the weight cache, so an optimizer step or a
        load_state_dict rebuilds it with the rest."""
        c, lib = self._w(), _lib.load()
        if "x3" in c:
            return c["x3"]
        dev = c["dev"]

        def split(w):
            w = w.contiguous()
            out = torch.empty((w.shape[0], 2 * w.shape[1]), dtype=torch.bfloat16, device=dev)
            _lib.check(lib.hirest_split2_bf16(w.data_ptr(), w.shape[1], out.data_ptr(), 2 * w.shape[1], w.shape[0], w.shape[1], 0,
                                              ops.stream_ptr()), "hirest_split2_bf16")
            return out
        V = "clip4cap_model.visual."
        nl = len(self.clip4cap_model.visual.encoder.layer)
        keep, layers = [], (_lib.JointLayerX3 * nl)()
        for i in range(nl):
            p = V + f"encoder.layer.{i}."
            w2 = [split(c[f"qkv_w.{i}"]), split(c[p + "attention.output.dense.weight"]), split(c[p + "intermediate.dense.weight"]),
                  split(c[p + "output.dense.weight"])]
            keep += w2
            layers[i] = _lib.JointLayerX3(
                w2[0].data_ptr(), c[f"qkv_b.{i}"].data_ptr(), w2[1].data_ptr(), c[p + "attention.output.dense.bias"].data_ptr(),
                c[p + "attention.output.LayerNorm.weight"].data_ptr(), c[p + "attention.output.LayerNorm.bias"].data_ptr(),
                w2[2].data_ptr(), c[p + "intermediate.dense.bias"].data_ptr(), w2[3].data_ptr(), c[p + "output.dense.bias"].data_ptr(),
                c[p + "output.LayerNorm.weight"].data_ptr(), c[p + "output.LayerNorm.bias"].data_ptr())
        emb = split(c[V + "embeddings.word_embeddings.weight"])
        keep.append(emb)
        H = c[V + "embeddings.LayerNorm.weight"].shape[0]
        pos = c[V + "embeddings.position_embeddings.weight"]
        desc = _lib.JointEncoderX3(C.sizeof(_lib.JointEncoderX3), nl, self.heads, H, c[V + "encoder.layer.0.intermediate.dense.weight"].shape[0],
                                   c[V + "embeddings.word_embeddings.weight"].shape[1], pos.shape[0], 1e-12, -10000.0,
                                   emb.data_ptr(), c[V + "embeddings.word_embeddings.bias"].data_ptr(), pos.data_ptr(),
                                   c[V + "embeddings.LayerNorm.weight"].data_ptr(), c[V + "embeddings.LayerNorm.bias"].data_ptr(), layers)
        # the caption decoder's descriptor with the LM head's weights in the split format as well (csrc/caption.hip takes them at >= 64 rows)
        lm2 = split(c["lm_w"])
        keep.append(lm2)
        d0 = c["dec_desc"]
        dec = _lib.CaptionDecoder(*[getattr(d0, n) for n, _ in _lib.CaptionDecoder._fields_[:-1]], lm2.data_ptr())
        c["x3"] = {"desc": desc, "layers": layers, "keep": keep, "width": H, "dec_desc": dec}
        return c["x3"]

    def _dec_desc(self):
        """The C-side decoder descriptor of the current precision ('bf16x3': with the split LM head)."""
        return self._x3()["dec_desc"] if self.precision == "bf16x3" else self._w()["dec_desc"]

    def _encoder_x3(self, f2d: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """VisualModel.forward on split operands: ONE C call (csrc/joint_x3.hip) issues the embeddings and both blocks."""
        x3, lib = self._x3(), _lib.load()
        need = lib.hirest_joint_encoder_x3_workspace_bytes(C.byref(x3["desc"]), B, T)
        if need == 0:
            raise RuntimeError("hirest_joint_encoder_x3: unsupported encoder shape")
        ws = x3.get("ws")
        if ws is None or ws.numel() < need or ws.device != f2d.device:
            ws = x3["ws"] = torch.empty((need,), dtype=torch.uint8, device=f2d.device)
        out = torch.empty((B * T, x3["width"]), dtype=torch.float32, device=f2d.device)
        _lib.check(lib.hirest_joint_encoder_x3_forward(C.byref(x3["desc"]), f2d.data_ptr(), B, T, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                       ops.stream_ptr()), "hirest_joint_encoder_x3_forward")
        return out

    # ------------------------------------------------------------------ kernels
    @staticmethod
    def _gemm(a, w, bias, out=None, resid=None, periodic=None, period=0, act=0):
        lib = _lib.load()
        M, K = a.shape
        N = w.shape[0]
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        ws, wsb = ops.f32_gemm_workspace(a.device, lib.hirest_gemm_f32_workspace_bytes(M, N, K))
        _lib.check(lib.hirest_gemm_f32_ws(a.data_ptr(), K, w.data_ptr(), w.shape[1], bias.data_ptr() if bias is not None else None,
                                          resid.data_ptr() if resid is not None else None, N,
                                          periodic.data_ptr() if periodic is not None else None, period,
                                          out.data_ptr(), N, M, N, K, act, ws, wsb, ops.stream_ptr()), "hirest_gemm_f32_ws")
        return out

    @staticmethod
    def _ln(x, w, b, eps):
        out = torch.empty_like(x)
        return ops.layernorm(x, w, b, eps, out)

    def _encoder(self, f2d: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """VisualModel.forward (module_visual.py:396-424): embeddings + 2 post-LN layers, fp32."""
        c, lib = self._w(), _lib.load()
        V = "clip4cap_model.visual."
        x = self._gemm(f2d, c[V + "embeddings.word_embeddings.weight"], c[V + "embeddings.word_embeddings.bias"],
                       periodic=c[V + "embeddings.position_embeddings.weight"], period=T)
        x = self._ln(x, c[V + "embeddings.LayerNorm.weight"], c[V + "embeddings.LayerNorm.bias"], 1e-12)
        D = x.shape[1]
        for i in range(len(self.clip4cap_model.visual.encoder.layer)):
            p = V + f"encoder.layer.{i}."
            qkv = self._gemm(x, c[f"qkv_w.{i}"], c[f"qkv_b.{i}"])
            ctx = torch.empty_like(x)
            _lib.check(lib.hirest_attention_f32(qkv.data_ptr(), ctx.data_ptr(), B, T, self.heads, D // self.heads,
                                                (D // self.heads) ** -0.5, -10000.0, ops.stream_ptr()), "hirest_attention_f32")
            a = self._gemm(ctx, c[p + "attention.output.dense.weight"], c[p + "attention.output.dense.bias"], resid=x)
            a = self._ln(a, c[p + "attention.output.LayerNorm.weight"], c[p + "attention.output.LayerNorm.bias"], 1e-12)
            h = self._gemm(a, c[p + "intermediate.dense.weight"], c[p + "intermediate.dense.bias"], act=1)
            y = self._gemm(h, c[p + "output.dense.weight"], c[p + "output.dense.bias"], resid=a)
            x = self._ln(y, c[p + "output.LayerNorm.weight"], c[p + "output.LayerNorm.bias"], 1e-12)
        return x

    def _fusion_base(self, vis, text, asr, vis_mask) -> torch.Tensor:
        """Loop-invariant part of foward_moment_shared (modeling.py:158-195): v*t + asr + temporal."""
        c, lib = self._w(), _lib.load()
        B, T, _ = vis.shape
        E = 512
        v = self._gemm(vis.reshape(B * T, -1), c["clip_g_map.weight"], c["clip_g_map.bias"])
        v = self._ln(v, c["clip4cap_model.normalize_video.visual_norm2d.weight"],
                     c["clip4cap_model.normalize_video.visual_norm2d.bias"], 1e-12)
        tproj = self._gemm(text, c["clip_g_map_text.weight"], c["clip_g_map_text.bias"])
        if self.use_asr:
            a = self._ln(asr.reshape(B * T, -1).contiguous(), c["asr_enc_layer.0.weight"], c["asr_enc_layer.0.bias"], 1e-5)
            a = self._gemm(a, c["asr_enc_layer.1.weight"], c["asr_enc_layer.1.bias"])
        else:
            a = torch.zeros((B * T, E), dtype=torch.float32, device=vis.device)
        n_valid = vis_mask.sum(dim=-1).to(torch.int32).contiguous()
        tin = torch.empty((B * T, E), dtype=torch.float32, device=vis.device)
        _lib.check(lib.hirest_joint_time_features(n_valid.data_ptr(), c["temporal_embed.0.weight"].data_ptr(),
                                                  c["temporal_embed.0.bias"].data_ptr(), tin.data_ptr(), B, T, E,
                                                  ops.stream_ptr()), "hirest_joint_time_features")
        temporal = self._gemm(tin, c["temporal_embed.2.weight"], c["temporal_embed.2.bias"])
        base = torch.empty((B * T, E), dtype=torch.float32, device=vis.device)
        _lib.check(lib.hirest_joint_base(v.data_ptr(), tproj.data_ptr(), a.data_ptr(), temporal.data_ptr(), base.data_ptr(),
                                         B, T, E, ops.stream_ptr()), "hirest_joint_base")
        return base

    def _features(self, base, moment_mask_i32, boundary_mask_i32, B, T) -> torch.Tensor:
        c, lib = self._w(), _lib.load()
        f = torch.empty_like(base)
        _lib.check(lib.hirest_joint_mask_add(base.data_ptr(), moment_mask_i32.data_ptr(),
                                             boundary_mask_i32.data_ptr() if boundary_mask_i32 is not None else None,
                                             c["mask_embed.weight"].data_ptr(), c["boundary_embed.weight"].data_ptr(),
                                             f.data_ptr(), B * T, 512, ops.stream_ptr()), "hirest_joint_mask_add")
        return self._encoder_x3(f, B, T) if self.precision == "bf16x3" else self._encoder(f, B, T)

    def _heads(self, feats, which: List[str]) -> torch.Tensor:
        c, lib = self._w(), _lib.load()
        rows, D = feats.shape
        names = {"start": ("start_predictor.0.weight", 0), "end": ("end_predictor.0.weight", 1), "segment": ("segment_predictor.0.weight", 2)}
        ws = [c[names[w][0]] for w in which]
        bkey = "head_bias3." + ".".join(which)             # (lives in the weight cache: rebuilt with it when a parameter changes)
        bias3 = c.get(bkey)
        if bias3 is None:
            bias3 = torch.stack([c["head_bias"][names[w][1]] for w in which]).contiguous()
            bias3 = c[bkey] = torch.cat([bias3, torch.zeros(3 - len(which), device=bias3.device)]).contiguous()
        logits = torch.empty((len(which), rows), dtype=torch.float32, device=feats.device)
        _lib.check(lib.hirest_linear_heads(feats.data_ptr(), rows, D, len(which), ws[0].data_ptr(),
                                           ws[1].data_ptr() if len(ws) > 1 else None, ws[2].data_ptr() if len(ws) > 2 else None,
                                           bias3.data_ptr(), logits.data_ptr(), ops.stream_ptr()), "hirest_linear_heads")
        return logits

    # ------------------------------------------------------------------ reference interface
    def _text_feat(self, batch, device):
        if "text_feat" in batch:
            return ops.to_device(batch["text_feat"], device).float().contiguous()
        if self.clip_model is None:
            raise RuntimeError("MomentModel needs clip_model (encode_text) or batch['text_feat']")
        return self.clip_model.encode_text(batch["clip_text_ids"].to(device)).float().contiguous()

    def test_step(self, batch, **kwargs):
        task = batch["tasks"][0]
        dev = self.clip_g_map.weight.device
        # kernels launch on the model's device, whatever the process's current device is (the reference never calls
        # set_device in single-process runs: run.py:61-65)
        with torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext():
            if task == "moment_retrieval":
                return self.test_moment_retrieval(batch, **kwargs)
            elif task == "moment_segmentation":
                return self.test_moment_segmentation(batch, **kwargs)
            elif task == "step_captioning":
                return self.test_step_captioning(batch, **kwargs)
            else:
                raise NotImplementedError

    def train_step(self, batch):
        """modeling.py:130-140: ``{'loss': tensor}``; ``loss.backward()`` fills ``param.grad`` through the kernels of
        csrc/train.hip (hirest_amd/train.py), for all three tasks."""
        from . import train
        task = batch["tasks"][0]
        dev = self.clip_g_map.weight.device
        with torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext():
            if task == "moment_retrieval":
                return train.train_moment_retrieval(self, batch)
            elif task == "moment_segmentation":
                return train.train_moment_segmentation(self, batch)
            elif task == "step_captioning":
                return train.train_step_captioning(self, batch)
            else:
                raise NotImplementedError

    @torch.no_grad()
    def forward_moment_retrieval(self, video_feats, text_feat, video_mask=None, moment_mask=None, asr_feats=None):
        """modeling.py:212-224: returns {'start_logits','end_logits'} [B,T] (fp32, unmasked)."""
        B, T, _ = video_feats.shape
        video_feats = video_feats.float().contiguous()
        if video_mask is None:
            video_mask = torch.ones((B, T), dtype=torch.long, device=video_feats.device)
        base = self._fusion_base(video_feats, text_feat, asr_feats.float().contiguous() if asr_feats is not None else None, video_mask)
        feats = self._features(base, moment_mask.to(torch.int32).contiguous(), None, B, T)
        lg = self._heads(feats, ["start", "end"])
        return {"start_logits": lg[0].reshape(B, T), "end_logits": lg[1].reshape(B, T), "feats": feats.reshape(B, T, -1)}

    @torch.no_grad()
    def test_moment_retrieval(self, batch, **kwargs):
        lib = _lib.load()
        dev = self._w()["dev"]
        vis, vmask, mmask = batch["vis_feats"].to(dev), batch["vis_mask"].to(dev), batch["moment_mask"].to(dev)
        asr = batch["asr_feats"].to(dev) if self.use_asr else None
        out = self.forward_moment_retrieval(vis, self._text_feat(batch, dev), vmask, mmask, asr)
        B, T = vmask.shape
        m32 = vmask.to(torch.int32).contiguous()
        pred = torch.empty((2, B), dtype=torch.int32, device=dev)
        for i, k in enumerate(("start_logits", "end_logits")):
            _lib.check(lib.hirest_masked_argmax(out[k].contiguous().data_ptr(), m32.data_ptr(), -1e10, B, T,
                                                pred[i].data_ptr(), ops.stream_ptr()), "hirest_masked_argmax")
        return {"prediction": pred.t().cpu().tolist()}

    @torch.no_grad()
    def test_moment_segmentation(self, batch, threshold=0.15, return_trace=False, **kwargs):
        lib = _lib.load()
        dev = self._w()["dev"]
        vis, vmask = batch["vis_feats"].to(dev).float().contiguous(), batch["vis_mask"].to(dev)
        asr = batch["asr_feats"].to(dev).float().contiguous() if self.use_asr else None
        text = self._text_feat(batch, dev)
        B, T = vmask.shape
        starts = batch["moment_bound_frames"][:, 0].tolist()
        lasts = batch["moment_bound_frames"][:, 1].tolist()
        mm = torch.zeros((B, T), dtype=torch.int32)
        bm = torch.zeros((B, T), dtype=torch.int32)
        for b in range(B):
            mm[b, starts[b]:lasts[b] + 1] = 1
            bm[b, starts[b]] = 1
        mm, bm = mm.to(dev), bm.to(dev)
        thr = float(getattr(self.args, "moment_segmentation_difference_threshold", 0.5)) if self.args is not None else 0.5
        iters = int(getattr(self.args, "moment_segmentation_max_iterations", 20)) if self.args is not None else 20
        steps = torch.zeros((B, iters, 2), dtype=torch.int32, device=dev)
        nsteps = torch.zeros((B,), dtype=torch.int32, device=dev)
        base = self._fusion_base(vis, text, asr, vmask)
        first_logits = None
        for it in range(iters):                                   # no host sync inside the loop
            feats = self._features(base, mm, bm, B, T)
            logits = self._heads(feats, ["segment"])[0].contiguous()
            if it == 0 and return_trace:
                first_logits = logits.reshape(B, T).clone()
            _lib.check(lib.hirest_segmentation_step(logits.data_ptr(), mm.data_ptr(), bm.data_ptr(), B, T, thr,
                                                    steps.data_ptr(), nsteps.data_ptr(), iters, None, ops.stream_ptr()),
                       "hirest_segmentation_step")
        steps_h, n_h = steps.cpu().tolist(), nsteps.cpu().tolist()    # the only device->host copy
        preds = []
        for b in range(B):                                         # modeling.py:435-463, pure Python ints
            sp = [[starts[b], starts[b]]] + [list(s) for s in steps_h[b][:n_h[b]]] + [[lasts[b], lasts[b]]]
            sp.sort(key=lambda x: x[0])
            flat = [v for s in sp for v in s]
            while flat[-1] > lasts[b]:
                flat.pop(-1)
            temp = sorted(set(flat))
            keep, cur = [temp[0]], temp[0]
            for i in range(1, len(temp) - 1):
                if temp[i] - cur >= 5:
                    keep.append(temp[i])
                    cur = temp[i]
            preds.append(keep)
        res = {"raw_predictions": deepcopy(preds), "prediction": preds}
        if return_trace:
            res["first_logits"] = first_logits
        return res


    # ------------------------------------------------------------------ step captioning (modeling.py:529-632)
    @staticmethod
    def _trim_index(mask_row: List[int], max_frames: int) -> List[int]:
        """Row indices selected by trim_feats (modeling.py:529-554) for one sample; -1 = zero row."""
        sel = [i for i, m in enumerate(mask_row) if m == 1]
        N = len(sel)
        if N == 0:
            return [-1] * max_frames
        if max_frames < N:
            return sel[:max_frames]
        idx = []
        for j in range(N):
            idx += [sel[j]] * (((j + 1) * max_frames) // N - (j * max_frames) // N)
        return idx + [-1] * (max_frames - len(idx))

    def _trim_rows(self, moment_mask: torch.Tensor, max_frames: int, device):
        """The row-index table of trim_feats for a batch, on `device`: flat row numbers b * T + t into the [B * T, D] feature matrix,
        [B * max_frames] long, and — only when some sample selects fewer than max_frames rows — a [B * max_frames, 1] 0 / 1 column
        that zeroes the missing ones (else None).  Host index arithmetic on the mask as the collate function delivers it (a CPU
        tensor: no device round trip; a device tensor costs one copy back, not one per sample), uploaded once and shared by the
        visual and the ASR features."""
        idx = self._trim_index_table(moment_mask.cpu(), max_frames)              # [B, max_frames] int64, -1 = zero row
        B, T = moment_mask.shape
        flat = torch.from_numpy(np.maximum(idx, 0) + np.arange(B, dtype=np.int64)[:, None] * T).reshape(-1)
        keep = None
        if (idx < 0).any():
            keep = torch.from_numpy((idx >= 0).astype(np.float32).reshape(-1, 1)).to(device)
        return flat.to(device), keep

    @staticmethod
    def _trim_index_table(mask: torch.Tensor, max_frames: int) -> "np.ndarray":
        """_trim_index for every row of a [B, T] CPU mask at once (the per-sample list walk cost 0.5 ms of host time in front of a B = 32
        captioning batch, with the GPU idle): output position p of a sample with N <= max_frames selected frames takes selected frame
        ceil((p + 1) N / max_frames) - 1 — the closed form of the reference's repeat counts (j + 1) F // N - j F // N (modeling.py:529-554)."""
        m = mask.numpy() == 1
        B, T = m.shape
        F = int(max_frames)
        N = m.sum(axis=1).astype(np.int64)                                       # selected frames per sample
        sel = np.argsort(~m, axis=1, kind="stable")                              # selected frame numbers first, in order
        p = np.arange(F, dtype=np.int64)[None, :]
        Nc = np.maximum(N, 1)[:, None]
        j = np.where(N[:, None] > F, p, ((p + 1) * Nc + F - 1) // F - 1)         # more frames than slots: the first F; else repeats
        idx = np.take_along_axis(sel, np.minimum(j, T - 1), axis=1).astype(np.int64)
        idx[N == 0] = -1
        return idx

    def _trim(self, feats: torch.Tensor, moment_mask, max_frames: int, idx=None) -> torch.Tensor:
        if idx is None:
            idx = self._trim_rows(moment_mask, max_frames, feats.device)
        flat, keep = idx
        B, T, D = feats.shape
        out = feats.reshape(B * T, D).index_select(0, flat)                    # pure data movement
        if keep is not None:
            out = out * keep
        return out.reshape(B, max_frames, D)

    def _decoder_last_logprob(self, ids: torch.Tensor, enc_kv: List[torch.Tensor], row_add: torch.Tensor) -> torch.Tensor:
        """DecoderModel.forward on the whole prefix (no KV cache, like the reference), then log_softmax of the LAST
        position + row_add (train.py:547-566, beam.py:76).  ids [R,t] int64, enc_kv[i] [R,20,1536] -> [R, vocab]."""
        c, lib = self._w(), _lib.load()
        Dp = "clip4cap_model.decoder."
        R, t = ids.shape
        H, Dm = self.heads, 768
        x = torch.empty((R * t, Dm), dtype=torch.float32, device=ids.device)
        _lib.check(lib.hirest_embed_tokens(ids.contiguous().data_ptr(), c[Dp + "embeddings.word_embeddings.weight"].data_ptr(),
                                           c[Dp + "embeddings.position_embeddings.weight"].data_ptr(), x.data_ptr(), None,
                                           R, t, Dm, c[Dp + "embeddings.word_embeddings.weight"].shape[0], ops.stream_ptr()),
                   "hirest_embed_tokens")
        x = self._ln(x, c[Dp + "embeddings.LayerNorm.weight"], c[Dp + "embeddings.LayerNorm.bias"], 1e-12)
        scale = (Dm // H) ** -0.5
        for i in range(len(self.clip4cap_model.decoder.decoder.layer)):
            p = Dp + f"decoder.layer.{i}."
            qkv = self._gemm(x, c[f"dec_qkv_w.{i}"], c[f"dec_qkv_b.{i}"])
            ctx = torch.empty_like(x)
            _lib.check(lib.hirest_attention_f32_qkv(qkv.data_ptr(), 3 * Dm, qkv.data_ptr() + 4 * Dm, qkv.data_ptr() + 8 * Dm, 3 * Dm,
                                                    ctx.data_ptr(), R, t, t, H, Dm // H, scale, 0.0, -10000.0, ops.stream_ptr()),
                       "self attention")
            s1 = self._gemm(ctx, c[p + "slf_attn.output.dense.weight"], c[p + "slf_attn.output.dense.bias"], resid=x)
            s1 = self._ln(s1, c[p + "slf_attn.output.LayerNorm.weight"], c[p + "slf_attn.output.LayerNorm.bias"], 1e-12)
            q2 = self._gemm(s1, c[p + "enc_attn.att.query.weight"], c[p + "enc_attn.att.query.bias"])
            kv = enc_kv[i]
            Tk = kv.shape[1]
            _lib.check(lib.hirest_attention_f32_qkv(q2.data_ptr(), Dm, kv.data_ptr(), kv.data_ptr() + 4 * Dm, 2 * Dm, ctx.data_ptr(),
                                                    R, t, Tk, H, Dm // H, scale, -10000.0, 0.0, ops.stream_ptr()), "cross attention")
            d = self._gemm(ctx, c[p + "enc_attn.output.dense.weight"], c[p + "enc_attn.output.dense.bias"], resid=s1)
            d = self._ln(d, c[p + "enc_attn.output.LayerNorm.weight"], c[p + "enc_attn.output.LayerNorm.bias"], 1e-12)
            hmid = self._gemm(d, c[p + "intermediate.dense.weight"], c[p + "intermediate.dense.bias"], act=1)
            y = self._gemm(hmid, c[p + "output.dense.weight"], c[p + "output.dense.bias"], resid=d)
            x = self._ln(y, c[p + "output.LayerNorm.weight"], c[p + "output.LayerNorm.bias"], 1e-12)
        last = x.reshape(R, t, Dm)[:, -1, :].contiguous()                     # dec_output[:, -1, :] (train.py:562)
        cp = Dp + "classifier.cls.predictions."
        hh = self._gemm(last, c[cp + "transform.dense.weight"], c[cp + "transform.dense.bias"], act=1)
        hh = self._ln(hh, c[cp + "transform.LayerNorm.weight"], c[cp + "transform.LayerNorm.bias"], 1e-12)
        logits = self._gemm(hh, c["lm_w"], c["lm_b"])
        V = logits.shape[1]
        out = torch.empty_like(logits)
        _lib.check(lib.hirest_log_softmax_f32(logits.data_ptr(), V, row_add.data_ptr(), out.data_ptr(), V, R, V, ops.stream_ptr()),
                   "hirest_log_softmax_f32")
        return out

    def _beam_search_cached(self, beams, enc_kv_all, num_beams, max_words, return_ids):
        """Beam search without a host round trip per word: one C-side decoder step (csrc/caption.hip: ~35 kernels enqueued without
        returning to Python, each beam's self-attention K / V kept and re-gathered by parent beam), the top-k over beams x vocabulary
        and the beam bookkeeping (`hirest_beam_advance`: beam.py:70-92) all stay on the device.  The host only watches the "done" flags,
        two steps behind, to stop early.  The row set never shrinks: a finished sample's rows keep being computed and are ignored."""
        from .beam import BOS_ID, EOS_ID
        c, lib = self._w(), _lib.load()
        dev = c["dev"]
        B, F = enc_kv_all[0].shape[0], enc_kv_all[0].shape[1]
        R, nl, Dm = B * num_beams, len(enc_kv_all), 768
        desc = self._dec_desc()
        Vp = desc.vocab_padded
        enc = [kv.repeat_interleave(num_beams, 0).contiguous() for kv in enc_kv_all]             # [R, F, 1536], loop invariant
        enc_ptrs = (C.c_void_p * nl)(*[e.data_ptr() for e in enc])
        ws = torch.empty(lib.hirest_caption_step_workspace_bytes(C.byref(desc), R), dtype=torch.uint8, device=dev)
        cache = [torch.empty((2 * nl, R, max_words, Dm), dtype=torch.float32, device=dev) for _ in range(2)]   # ping-pong
        ptrs = [(C.c_void_p * (2 * nl))(*[cb[i].data_ptr() for i in range(2 * nl)]) for cb in cache]
        logp = torch.empty((R, Vp), dtype=torch.float32, device=dev)
        # device-side beam state (beam.py: scores, next_ys, prev_ks) and the inputs of the next step, in two packed buffers (one fill,
        # two small copies and one read-back per batch instead of a dozen): int32 = tokens | backptr | n_steps | done | ids | parents
        nt = B * max_words * num_beams
        ibuf = torch.zeros((2 * nt + 2 * B + 2 * R,), dtype=torch.int32, device=dev)
        tokens, backptr = ibuf[:nt].view(B, max_words, num_beams), ibuf[nt:2 * nt].view(B, max_words, num_beams)
        n_steps, done = ibuf[2 * nt:2 * nt + B], ibuf[2 * nt + B:2 * nt + 2 * B]
        ids, parents = ibuf[2 * nt + 2 * B:2 * nt + 2 * B + R], ibuf[2 * nt + 2 * B + R:]
        key = (B, num_beams, max_words, str(dev))
        init = self.__dict__.setdefault("_beam_init", {}).get(key)
        if init is None:                                  # constants of the search, built once per shape
            add0 = torch.full((B, num_beams), -3.0e38, dtype=torch.float32)    # first step: only beam 0 competes
            add0[:, 0] = 0.0                                                   # (beam.py:78)
            init = self._beam_init[key] = (
                torch.cat([torch.full((R,), BOS_ID, dtype=torch.int32), torch.arange(R, dtype=torch.int32)]).to(dev),
                torch.cat([add0.reshape(-1), torch.zeros(R)]).to(dev))
        ibuf[2 * nt + 2 * B:].copy_(init[0])
        fbuf = init[1].clone()                             # float32 = add | scores
        add, scores = fbuf[:R].view(B, num_beams), fbuf[R:]
        # the stamped done flags of every step, pinned, ONE TABLE PER CALL: a table cached per shape could still be written by kernels of
        # an earlier call that left its loop by an exception, or by a concurrent call of the same shape on another stream
        # (caption_batches), and a stale stamp would end this search early
        done_rows = torch.zeros((max_words, B), dtype=torch.int32).pin_memory()
        done_host = [done_rows[t] for t in range(max_words)]
        fused_tail = bool(getattr(self, "caption_fused_tail", True)) and num_beams <= 16 and Vp <= 32768    # (the tail kernels' limits)
        copied = None if fused_tail else [torch.cuda.Event() for _ in range(max_words)]
        need = lib.hirest_topk_workspace_bytes(B, num_beams * Vp, num_beams)
        tk_ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
        val = torch.empty((B, num_beams), dtype=torch.float32, device=dev)
        idx = torch.empty((B, num_beams), dtype=torch.int32, device=dev)
        st = ops.stream_ptr()
        if fused_tail:
            tail_ws = torch.empty(max(int(lib.hirest_caption_beam_tail_workspace_bytes(B, num_beams, Vp)), 16), dtype=torch.uint8,
                                  device=dev)
        for t in range(1, max_words + 1):
            if fused_tail:
                # one C call per word: the decoder step up to the LM-head logits (20 kernels), then log-softmax + beam score + top-k +
                # bookkeeping + the done flags to pinned memory (2 kernels, fed the LM head's tile maxima)
                _lib.check(lib.hirest_caption_beam_step(
                    C.byref(desc), B, num_beams, t - 1, ids.data_ptr(), parents.data_ptr(), ptrs[t & 1] if t > 1 else None,
                    ptrs[(t + 1) & 1], enc_ptrs, F, add.data_ptr(), logp.data_ptr(), max_words, EOS_ID, scores.data_ptr(),
                    tokens.data_ptr(), backptr.data_ptr(), n_steps.data_ptr(), done.data_ptr(), done_host[t - 1].data_ptr(),
                    ws.data_ptr(), ws.numel(), tail_ws.data_ptr(), tail_ws.numel(), st), "hirest_caption_beam_step")
            else:
                _lib.check(lib.hirest_caption_decode_step(
                    C.byref(desc), R, t - 1, ids.data_ptr(), parents.data_ptr() if t > 1 else None,
                    ptrs[t & 1] if t > 1 else None, ptrs[(t + 1) & 1], enc_ptrs, F, add.data_ptr(), logp.data_ptr(),
                    ws.data_ptr(), ws.numel(), st), "hirest_caption_decode_step")
                _lib.check(lib.hirest_topk_f32_ws(logp.data_ptr(), None, B, num_beams * Vp, num_beams, idx.data_ptr(), val.data_ptr(),
                                                  tk_ws.data_ptr(), tk_ws.numel(), st), "hirest_topk_f32_ws")
                _lib.check(lib.hirest_beam_advance(val.data_ptr(), idx.data_ptr(), B, num_beams, Vp, t - 1, max_words, EOS_ID,
                                                   scores.data_ptr(), tokens.data_ptr(), backptr.data_ptr(), n_steps.data_ptr(),
                                                   done.data_ptr(), ids.data_ptr(), parents.data_ptr(), add.data_ptr(), st),
                           "hirest_beam_advance")
                done_host[t - 1].copy_(done, non_blocking=True)
            if fused_tail:
                # the kernel stamps each sample's word with its step: read whatever has arrived of two steps ago, never wait
                if t >= 3 and all((v >> 1) == t - 2 and (v & 1) for v in done_host[t - 3].tolist()):
                    break
                continue
            copied[t - 1].record()
            if t >= 3:                                   # look at the flags of two steps ago: never waits for the GPU
                copied[t - 3].synchronize()
                if int(done_host[t - 3].min()) == 1:
                    break
        if self.caption_device_readout:
            return self._device_readout(scores, tokens, backptr, n_steps, B, num_beams, max_words, return_ids)
        ih = ibuf[:2 * nt + B].cpu()                       # tokens | backptr | n_steps in one copy (this is the batch's synchronisation)
        tok_h, bp_h = ih[:nt].view(B, max_words, num_beams).tolist(), ih[nt:2 * nt].view(B, max_words, num_beams).tolist()
        n_h, sc_h = ih[2 * nt:].tolist(), scores.view(B, -1).cpu().tolist()
        for b in range(B):                               # hand the recorded search to the host-side BeamState for the read-out
            beams[b].scores = sc_h[b]
            beams[b].backptr = [bp_h[b][j] for j in range(n_h[b])]
            beams[b].tokens = [[BOS_ID] * num_beams] + [tok_h[b][j] for j in range(n_h[b])]
        return self._caption_result(beams, return_ids)

    # -------------------------------------------------------------------------------------------------------------------
    # The same search replayed from hipGraphs (caption_batches): a word step is ~22 launches of ~3 us of host time each, and HIP
    # serialises launches across host threads, so three batches in flight were HOST-bound (532 -> 766 captions/s instead of the ~2x
    # the idle CUs allow).  Here all buffers of a search are static per (shape, slot), the word steps are captured once in chunks of
    # CAPTION_GRAPH_CHUNK words, and a batch costs max_words / chunk graph launches.  Same kernels, same arguments, same tokens.
    # -------------------------------------------------------------------------------------------------------------------
    CAPTION_GRAPH_CHUNK = 8

    def _caption_graph_ctx(self, B, num_beams, max_words, F, nl, slot):
        from .beam import BOS_ID
        c, lib = self._w(), _lib.load()
        ctxs = c.setdefault("caption_graphs", {})           # lives and dies with the weight cache: the graphs hold its pointers
        key = (B, num_beams, max_words, F, nl, slot, self.precision)      # (a captured graph holds the descriptor of its precision)
        ctx = ctxs.get(key)
        if ctx is not None:
            return ctx
        dev, desc = c["dev"], self._dec_desc()
        R, Dm, Vp = B * num_beams, 768, desc.vocab_padded
        nt = B * max_words * num_beams
        add0 = torch.full((B, num_beams), -3.0e38, dtype=torch.float32)
        add0[:, 0] = 0.0
        ctx = {
            "enc": [torch.empty((R, F, 2 * Dm), dtype=torch.float32, device=dev) for _ in range(nl)],
            "ws": torch.empty(lib.hirest_caption_step_workspace_bytes(C.byref(desc), R), dtype=torch.uint8, device=dev),
            "cache": [torch.empty((2 * nl, R, max_words, Dm), dtype=torch.float32, device=dev) for _ in range(2)],
            "logp": torch.empty((R, Vp), dtype=torch.float32, device=dev),
            "ibuf": torch.zeros((2 * nt + 2 * B + 2 * R,), dtype=torch.int32, device=dev),
            "fbuf": torch.zeros((2 * R,), dtype=torch.float32, device=dev),
            "ibuf0": torch.cat([torch.full((R,), BOS_ID, dtype=torch.int32), torch.arange(R, dtype=torch.int32)]).to(dev),
            "fbuf0": torch.cat([add0.reshape(-1), torch.zeros(R)]).to(dev),
            "tail_ws": torch.empty(max(int(lib.hirest_caption_beam_tail_workspace_bytes(B, num_beams, Vp)), 16), dtype=torch.uint8, device=dev),
            "done_rows": torch.zeros((max_words, B), dtype=torch.int32).pin_memory(),
            "graphs": [],
        }
        ctx["enc_ptrs"] = (C.c_void_p * nl)(*[e.data_ptr() for e in ctx["enc"]])
        ctx["ptrs"] = [(C.c_void_p * (2 * nl))(*[cb[i].data_ptr() for i in range(2 * nl)]) for cb in ctx["cache"]]
        ctxs[key] = ctx
        return ctx

    def _caption_issue_step(self, ctx, B, num_beams, max_words, F, t, st):
        """One word (1-based t) of the search on the static buffers of `ctx`: hirest_caption_beam_step on stream pointer `st`."""
        from .beam import EOS_ID
        c, lib = self._w(), _lib.load()
        R, nt = B * num_beams, B * max_words * num_beams
        ibuf, fbuf = ctx["ibuf"], ctx["fbuf"]
        tokens, backptr = ibuf[:nt], ibuf[nt:2 * nt]
        n_steps, done = ibuf[2 * nt:2 * nt + B], ibuf[2 * nt + B:2 * nt + 2 * B]
        ids, parents = ibuf[2 * nt + 2 * B:2 * nt + 2 * B + R], ibuf[2 * nt + 2 * B + R:]
        add, scores = fbuf[:R], fbuf[R:]
        _lib.check(lib.hirest_caption_beam_step(
            C.byref(self._dec_desc()), B, num_beams, t - 1, ids.data_ptr(), parents.data_ptr(), ctx["ptrs"][t & 1] if t > 1 else None,
            ctx["ptrs"][(t + 1) & 1], ctx["enc_ptrs"], F, add.data_ptr(), ctx["logp"].data_ptr(), max_words, EOS_ID, scores.data_ptr(),
            tokens.data_ptr(), backptr.data_ptr(), n_steps.data_ptr(), done.data_ptr(), ctx["done_rows"][t - 1].data_ptr(),
            ctx["ws"].data_ptr(), ctx["ws"].numel(), ctx["tail_ws"].data_ptr(), ctx["tail_ws"].numel(), st), "hirest_caption_beam_step")

    def _caption_capture(self, ctx, B, num_beams, max_words, F):
        """Capture the word steps of one search shape into hipGraphs (once per context; call from ONE thread while no other thread
        issues HIP work: stream capture is process-global)."""
        if ctx["graphs"]:
            return
        ctx["ibuf"][2 * B * max_words * num_beams + 2 * B:].copy_(ctx["ibuf0"])
        ctx["fbuf"].copy_(ctx["fbuf0"])
        for e in ctx["enc"]:
            e.zero_()
        for t in range(1, min(3, max_words) + 1):          # eager warm-up on these buffers (one-time kernel configuration must not be captured)
            self._caption_issue_step(ctx, B, num_beams, max_words, F, t, ops.stream_ptr())
        torch.cuda.synchronize()
        step = max(1, int(self.CAPTION_GRAPH_CHUNK))
        for lo in range(1, max_words + 1, step):
            hi = min(max_words, lo + step - 1)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st = ops.stream_ptr()
                for t in range(lo, hi + 1):
                    self._caption_issue_step(ctx, B, num_beams, max_words, F, t, st)
            ctx["graphs"].append((lo, hi, g))
        torch.cuda.synchronize()

    def _beam_search_graph(self, beams, enc_kv_all, num_beams, max_words, return_ids, slot):
        from .beam import BOS_ID
        B, F = enc_kv_all[0].shape[0], enc_kv_all[0].shape[1]
        nl = len(enc_kv_all)
        ctx = self._caption_graph_ctx(B, num_beams, max_words, F, nl, slot)
        if not ctx["graphs"]:
            raise RuntimeError("hirest_amd: caption graphs are captured by caption_batches before its threads start")
        R, nt = B * num_beams, B * max_words * num_beams
        for e, kv in zip(ctx["enc"], enc_kv_all):
            e.copy_(kv.repeat_interleave(num_beams, 0))
        ctx["ibuf"].zero_()
        ctx["ibuf"][2 * nt + 2 * B:].copy_(ctx["ibuf0"])
        ctx["fbuf"].copy_(ctx["fbuf0"])
        # (the previous search on this context ended with the blocking read-out below: nothing still writes the pinned table)
        done_rows = ctx["done_rows"]
        done_rows.zero_()
        events = []
        for k, (lo, hi, g) in enumerate(ctx["graphs"]):
            # at most two chunks ahead of the GPU, so that a search whose samples have all emitted [SEP] stops within two chunks: the
            # flags are those of the last word of chunk k - 2, which has completed (a finished sample's rows are inert meanwhile)
            if k >= 2:
                events[k - 2].synchronize()
                last = ctx["graphs"][k - 2][1]
                if all((v >> 1) == last and (v & 1) for v in done_rows[last - 1].tolist()):
                    break
            g.replay()
            ev = torch.cuda.Event()
            ev.record()
            events.append(ev)
        if self.caption_device_readout:
            ib = ctx["ibuf"]
            return self._device_readout(ctx["fbuf"][R:], ib[:nt], ib[nt:2 * nt], ib[2 * nt:2 * nt + B], B, num_beams, max_words, return_ids)
        ih = ctx["ibuf"][:2 * nt + B].cpu()
        tok_h, bp_h = ih[:nt].view(B, max_words, num_beams).tolist(), ih[nt:2 * nt].view(B, max_words, num_beams).tolist()
        n_h, sc_h = ih[2 * nt:].tolist(), ctx["fbuf"][R:].view(B, -1).cpu().tolist()
        for b in range(B):
            beams[b].scores = sc_h[b]
            beams[b].backptr = [bp_h[b][j] for j in range(n_h[b])]
            beams[b].tokens = [[BOS_ID] * num_beams] + [tok_h[b][j] for j in range(n_h[b])]
        return self._caption_result(beams, return_ids)

    @torch.no_grad()
    def test_step_captioning(self, batch, num_beams=5, return_ids=False, **kwargs):
        """modeling.py:556-632.  Returns {'prediction': [str]} (token strings joined like the reference; ids are
        printed as decimal strings when no BERT vocab is attached via ``tokenizer_vocab``)."""
        from .beam import BeamState
        dev = self._w()["dev"]
        c = self._w()
        max_frames = int(getattr(self.args, "max_frames_step_captioning", 20)) if self.args is not None else 20
        max_words = int(getattr(self.args, "max_words", 48)) if self.args is not None else 48
        vis = batch["vis_feats"].to(dev).float()
        mmask = batch["moment_mask"]
        B = vis.shape[0]
        rows = self._trim_rows(mmask, max_frames, dev)
        v = self._trim(vis, None, max_frames, idx=rows)
        a = self._trim(batch["asr_feats"].to(dev).float(), None, max_frames, idx=rows) if self.use_asr else None
        text = self._text_feat(batch, dev)
        ones = torch.ones((B, max_frames), dtype=torch.long, device=dev)
        base = self._fusion_base(v, text, a, ones)
        enc = self._features(base, ones.to(torch.int32).contiguous(), None, B, max_frames)          # [B*F, 768]
        # encoder-side K/V of the cross-attention are loop invariant: once per layer, [B, F, 1536]
        enc_kv_all = [self._gemm(enc, c[f"dec_kv_w.{i}"], c[f"dec_kv_b.{i}"]).reshape(B, max_frames, -1)
                      for i in range(len(self.clip4cap_model.decoder.decoder.layer))]
        beams = [BeamState(num_beams) for _ in range(B)]
        active = list(range(B))
        if bool(getattr(self, "caption_kv_cache", True)):
            if kwargs.get("graph_slot") is not None:         # caption_batches: replay the captured word steps of this slot's context
                ctx = self._caption_graph_ctx(B, num_beams, max_words, max_frames, len(enc_kv_all), kwargs["graph_slot"])
                if ctx["graphs"]:                            # (a batch of another size, e.g. the loader's last one, runs eagerly)
                    return self._beam_search_graph(beams, enc_kv_all, num_beams, max_words, return_ids, kwargs["graph_slot"])
            return self._beam_search_cached(beams, enc_kv_all, num_beams, max_words, return_ids)
        for t in range(1, max_words + 1):
            sel = torch.tensor([b for b in active for _ in range(num_beams)], dtype=torch.long, device=dev)
            enc_kv = [kv.index_select(0, sel).contiguous() for kv in enc_kv_all]
            # row_add = running beam scores (beam.py:76); on the first step only beam 0 competes (beam.py:78)
            add = torch.tensor([(x if (t > 1 or k == 0) else -3.0e38) for b in active
                                for k, x in enumerate(beams[b].scores)], dtype=torch.float32, device=dev)
            seqs = [s for b in active for s in beams[b].current_state()]                            # full-prefix recompute
            logp = self._decoder_last_logprob(torch.tensor(seqs, dtype=torch.long, device=dev), enc_kv, add)   # [n*beam, V]
            n, V = len(active), logp.shape[1]
            val, idx = ops.topk(logp.reshape(n, num_beams * V), num_beams)
            val_h, idx_h = val.cpu().tolist(), idx.cpu().tolist()
            active = [b for i, b in enumerate(active) if not beams[b].advance(val_h[i], idx_h[i], V)]
            if not active:
                break
        return self._caption_result(beams, return_ids)

    CAPTION_ROWS_IN_FLIGHT = 160      # beam rows of one merged search (caption_batches): 32 videos x 5 beams, the reference's default eval batch

    def _caption_merge(self, group, dev):
        """Several loader batches as ONE step-captioning batch: each batch is trimmed on its own (its T and its moment mask), the
        [B_i, max_frames, D] results are concatenated and the merged batch carries an all-ones moment mask — trim_feats of exactly
        max_frames selected rows is the identity, and rows a short moment left at zero stay zero.  Every kernel downstream is
        batch-invariant, so a video's caption does not depend on what it is merged with."""
        max_frames = int(getattr(self.args, "max_frames_step_captioning", 20)) if self.args is not None else 20
        vs, as_, ts = [], [], []
        for b in group:
            vis = b["vis_feats"].to(dev).float()
            rows = self._trim_rows(b["moment_mask"], max_frames, dev)
            vs.append(self._trim(vis, None, max_frames, idx=rows))
            if self.use_asr:
                as_.append(self._trim(b["asr_feats"].to(dev).float(), None, max_frames, idx=rows))
            ts.append(self._text_feat(b, dev))
        v = torch.cat(vs, 0)
        merged = {"tasks": ["step_captioning"], "vis_feats": v, "moment_mask": torch.ones((v.shape[0], max_frames), dtype=torch.long),
                  "text_feat": torch.cat(ts, 0)}
        if self.use_asr:
            merged["asr_feats"] = torch.cat(as_, 0)
        return merged

    @torch.no_grad()
    def caption_batches(self, batches, num_beams=5, streams=1, return_ids=False, graphs=True, merge=True, rows_in_flight=None):
        """Step captioning over a LIST of loader batches (the evaluation loop of run.py:328-336 / modeling.py:556-632 calls
        test_step once per batch).

        merge=True (default): consecutive batches are captioned by ONE beam search over the union of their beam rows, up to
        `rows_in_flight` rows (default CAPTION_ROWS_IN_FLIGHT = 160 = the reference's default --eval_batch_size 32 at beam 5,
        args.py:27) — a word's 162 MB of decoder weights are then streamed once for all of them instead of once per batch; each
        sample keeps its own done flag and the search ends when all have emitted [SEP].  The merged groups (if more than one) then go
        through the machinery below, up to `streams` in flight (default 1: a search of ~150 rows is matrix-pipe time, a second one beside
        it gains nothing — 1697 vs 1616 captions/s for twelve batches of 5 at beam 5).  merge=False: every loader batch is its own search (round 4).

        Up to `streams` searches in flight, each on its own HIP stream and host thread.

        Why: one batch of 5 videos x 5 beams is 25 rows — a word step is ~20 dependent kernels of a few microseconds each, bound by
        launch and memory latency, not by the 162 MB of weights it streams: the GPU is mostly idle.  Independent batches
        fill those gaps; only the LM head (HBM-bound, all CUs) serialises.  Every call owns its buffers (workspace, K / V cache, beam
        state, pinned done table), the weight cache is read-only, and every kernel is batch-invariant, so each batch's result is
        exactly what ``test_step`` returns for it alone.  Returns the per-batch result dicts in order."""
        import threading
        batches = list(batches)
        if merge and len(batches) > 1:
            dev0 = self._w()["dev"]
            cap = max(1, int(rows_in_flight or self.CAPTION_ROWS_IN_FLIGHT) // max(1, num_beams))      # videos per merged search
            groups, cur, cnt = [], [], 0
            for b in batches:
                nb = int(b["vis_feats"].shape[0])
                if cur and cnt + nb > cap:
                    groups.append(cur); cur, cnt = [], 0
                cur.append(b); cnt += nb
            if cur:
                groups.append(cur)
            if any(len(g) > 1 for g in groups):
                with torch.cuda.device(dev0):
                    merged = [self._caption_merge(g, dev0) if len(g) > 1 else g[0] for g in groups]
                res = self.caption_batches(merged, num_beams=num_beams, streams=streams, return_ids=return_ids, graphs=graphs, merge=False)
                out = []
                for g, r in zip(groups, res):           # hand each loader batch its own slice of the merged result
                    lo = 0
                    for b in g:
                        nb = int(b["vis_feats"].shape[0])
                        out.append({k: v[lo:lo + nb] for k, v in r.items()})
                        lo += nb
                return out
        n = max(1, min(int(streams), len(batches)))
        c = self._w()                                       # weight cache built (and the kernels' per-device setup done) before the threads
        dev = c["dev"]
        if n == 1 or len(batches) <= 1:
            return [self.test_step_captioning(b, num_beams=num_beams, return_ids=return_ids) for b in batches]
        results, errors = [None] * len(batches), []
        results[0] = self.test_step_captioning(batches[0], num_beams=num_beams, return_ids=return_ids)   # warm: one-time kernel configuration
        main = torch.cuda.current_stream(dev)
        side = [torch.cuda.Stream(device=dev) for _ in range(n)]
        if graphs and bool(getattr(self, "caption_kv_cache", True)) and bool(getattr(self, "caption_fused_tail", True)) and num_beams <= 16:
            # the word steps of this batch shape, captured once per slot (hipGraphs: a batch then costs max_words / 8 launches instead
            # of ~22 per word — three host threads issuing ~3-us launches through HIP's one launch lock were the bottleneck)
            max_frames = int(getattr(self.args, "max_frames_step_captioning", 20)) if self.args is not None else 20
            max_words = int(getattr(self.args, "max_words", 48)) if self.args is not None else 48
            B0 = batches[1]["vis_feats"].shape[0]
            nl = len(self.clip4cap_model.decoder.decoder.layer)
            with torch.cuda.device(dev):
                for w in range(n):
                    self._caption_capture(self._caption_graph_ctx(B0, num_beams, max_words, max_frames, nl, w), B0, num_beams, max_words, max_frames)
            slot_of = lambda w: w
        else:
            slot_of = lambda w: None

        def work(w):
            try:
                torch.cuda.set_device(dev)
                side[w].wait_stream(main)
                with torch.cuda.stream(side[w]):
                    for i in range(1 + w, len(batches), n):
                        results[i] = self.test_step_captioning(batches[i], num_beams=num_beams, return_ids=return_ids, graph_slot=slot_of(w))
            except BaseException as e:      # surfaced after the join
                errors.append(e)
        threads = [threading.Thread(target=work, args=(w,), daemon=True) for w in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for st in side:
            main.wait_stream(st)
        if errors:
            raise errors[0]
        return results

    caption_device_readout = True    # the best hypothesis of every sample walked back on the device (hirest_beam_backtrack); False: on the host (BeamState)

    def _device_readout(self, scores, tokens, backptr, n_steps, B, num_beams, max_words, return_ids):
        """The batch's synchronisation: one [B, max_words + 1] int32 copy (length | words of the best beam) instead of the whole token / parent
        tables and a Python walk per sample (0.3 ms of host time behind a B = 32 search, with the GPU idle)."""
        hyp = torch.empty((B, max_words + 1), dtype=torch.int32, device=scores.device)
        _lib.check(_lib.load().hirest_beam_backtrack(scores.data_ptr(), tokens.data_ptr(), backptr.data_ptr(), n_steps.data_ptr(), B, num_beams,
                                                     max_words, hyp.data_ptr(), ops.stream_ptr()), "hirest_beam_backtrack")
        return self._caption_texts([r[1:1 + r[0]] for r in hyp.cpu().tolist()], return_ids)

    def _caption_result(self, beams, return_ids):
        return self._caption_texts([bm.best_hypothesis() for bm in beams], return_ids)

    def _caption_texts(self, hyps, return_ids):
        vocab = self.tokenizer_vocab
        if vocab is None:                                    # ids printed as decimal strings: no '[SEP]' / '[PAD]' / '##' to handle
            texts = [" ".join(map(str, h)) for h in hyps]
        else:
            texts = []
            for h in hyps:
                toks = [vocab[i] for i in h]
                if "[SEP]" in toks:
                    toks = toks[:toks.index("[SEP]")]
                if "[PAD]" in toks:
                    toks = toks[:toks.index("[PAD]")]
                texts.append(" ".join(toks).replace(" ##", "").strip("##").strip())
        res = {"prediction": texts}
        if return_ids:
            res["token_ids"] = hyps
        return res
