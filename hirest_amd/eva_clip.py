"""Drop-in for the reference's ``EVA_clip/eva_clip.py`` + ``eva_model.py`` surface, running on
hand-written gfx950 kernels through the C ABI in include/hirest_hip.h.

Mirrored interface (names, argument meaning, error behaviour):
  build_eva_model_and_transforms   /root/reference/EVA_clip/eva_clip.py:155-172
  create_model / load_state_dict   eva_clip.py:68-120   (model|module|state_dict key search, 'module.' strip, strict)
  image_transform                  eva_clip.py:125-153  (bicubic short-side resize, center crop, RGB, /255, mean/std)
  EVA_CLIP.encode_image/encode_text/forward   eva_model.py:317-334
The module tree reproduces the reference's parameter names exactly (visual.blocks.N.attn.q_bias,
text.transformer.resblocks.N.attn.in_proj_weight, ...), so ``eva_clip_psz14.pt`` loads with
strict=True and ``state_dict()`` / ``parameters()`` / ``.to()`` / ``.float()`` / ``.eval()``
behave like the nn.Module callers expect (modeling.py:115-129, run.py:61-65).

Numerics: two kernel sets.  ``precision='bf16'`` (what ``EVA_CLIP(**cfg)`` starts in and what bench.py measures): GEMMs and
attention in bf16 on MFMA with fp32 accumulation; residual stream, LayerNorm statistics and softmax fp32; agrees with the
fp32 reference to the tolerance stated in tests/test_gpu_parity.py.  ``precision='fp32'`` (the default of
``create_model`` / ``build_eva_model_and_transforms``, as in the reference): exact-fp32 kernels end to end
(csrc/tower_f32.hip), embeddings within ~1e-6 of the reference and its retrieval ranks reproduced.
There is no CPU path: calling an encoder on CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from copy import deepcopy
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib, ops, synth

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # eva_clip.py:16
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # eva_clip.py:17

TOWER_PRECISIONS = ("fp32", "bf16x3", "bf16")
_NO_GUARD = (1 << 64) - 1           # hirest_vision_guard_offset: (size_t)-1 = this call does not fold

_MODEL_CONFIGS: Dict[str, dict] = {"EVA_CLIP_g_14": synth.EVA_CLIP_G_14, "EVA_CLIP_tiny_test": synth.EVA_CLIP_TINY,
                                   # the tiny towers with EVA-CLIP-g's 1024-d output: what MomentModel.clip_g_map_text takes
                                   "EVA_CLIP_tiny_e1024_test": dict(synth.EVA_CLIP_TINY, embed_dim=1024)}


def list_models():
    return list(_MODEL_CONFIGS.keys())


def add_model_config(path):
    """Register a config JSON (or a directory of them), eva_clip.py:52-58."""
    paths = [os.path.join(path, f) for f in os.listdir(path)] if os.path.isdir(path) else [path]
    for p in paths:
        if p.endswith(".json"):
            cfg = json.load(open(p))
            if all(k in cfg for k in ("embed_dim", "vision_cfg", "text_cfg")):
                _MODEL_CONFIGS[os.path.splitext(os.path.basename(p))[0]] = cfg


def get_model_config(model_name):
    return deepcopy(_MODEL_CONFIGS[model_name]) if model_name in _MODEL_CONFIGS else None


class _Leaf(nn.Module):
    """A module that only owns named parameters (shapes follow the reference schema)."""

    def __init__(self, **shapes):
        super().__init__()
        for name, shape in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.zeros(shape), requires_grad=True))


def _linear(out_f, in_f, bias=True):
    return _Leaf(weight=(out_f, in_f), bias=(out_f,)) if bias else _Leaf(weight=(out_f, in_f))


def _norm(d):
    return _Leaf(weight=(d,), bias=(d,))


class _Tower(nn.Module):
    """Shared machinery: lazily mirrors the fp32 master parameters into kernel-ready device
    buffers (bf16 GEMM weights, fused biases, ctypes descriptor structs) and re-does so whenever
    the parameters are moved / cast / reloaded."""

    def __init__(self):
        super().__init__()
        self._prepared = None
        self._workspace = None
        # 'bf16': GEMMs / attention on the bf16 MFMA kernels (fp32 accumulation, residual stream, statistics): the measured hot
        # path.  'fp32': every product in exact fp32 (csrc/tower_f32.hip): the reference's own precision (eva_clip.py:90), ~16x
        # the time; what ``create_model(..., precision='fp32')`` selects.
        self.precision = "bf16"

    def _apply(self, fn, *a, **k):
        self._prepared = None
        self._workspace = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._prepared = None
        return super()._load_from_state_dict(*a, **k)

    def invalidate(self):
        self._prepared = None

    def _ws(self, nbytes: int, device) -> torch.Tensor:
        if self._workspace is None or self._workspace.numel() < nbytes or self._workspace.device != device:
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._workspace

    @staticmethod
    def _f32(p: torch.Tensor) -> torch.Tensor:
        return p.detach().float().contiguous()

    @staticmethod
    def _bf16(p: torch.Tensor) -> torch.Tensor:
        return ops.to_bf16(p.detach().float().contiguous())


class VisionTower(_Tower):
    """EVA ViT (reference VisionTransformer, vit_model.py:248-351), parameter names identical."""

    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, embed_dim, quick_gelu=False):
        super().__init__()
        self.image_size, self.patch_size, self.width, self.layers, self.heads = image_size, patch_size, width, layers, heads
        self.mlp_dim = int(width * mlp_ratio)  # vit_model.py:166
        self.embed_dim = self.num_classes = embed_dim
        self.quick_gelu = quick_gelu
        self.num_tokens = (image_size // patch_size) ** 2 + 1
        D, Dm = width, self.mlp_dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_tokens, D))
        self.patch_embed = nn.Module()
        self.patch_embed.proj = _Leaf(weight=(D, 3, patch_size, patch_size), bias=(D,))
        blocks = []
        for _ in range(layers):
            b = nn.Module()
            b.norm1, b.norm2 = _norm(D), _norm(D)
            b.attn = _Leaf(q_bias=(D,), v_bias=(D,))
            b.attn.qkv = _linear(3 * D, D, bias=False)
            b.attn.proj = _linear(D, D)
            b.mlp = nn.Module()
            b.mlp.fc1, b.mlp.fc2 = _linear(Dm, D), _linear(D, Dm)
            blocks.append(b)
        self.blocks = nn.ModuleList(blocks)
        self.norm = _norm(D)
        self.head = _linear(embed_dim, D)
        self.image_mean, self.image_std = OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
        self.max_frames_per_call = 1024  # micro-batch per tower call (workspace ~5.4 GB at 1024 frames)
        self.max_frames_per_call_f32 = 256   # fp32 towers: 35.8 KB of activations per token -> 2.4 GB at 256 frames
        self.max_frames_per_call_x3 = 1024   # bf16x3: 66 KB per token -> 17.4 GB at 1024 frames (the persistent GEMMs want large M)
        self.fold_layernorm = True       # tower calls of >= 64 frames fold both LayerNorms of a block into its GEMMs
        self.fold_guard_ratio = 4.0      # a call whose worst token row sits > 4 sigma off zero is redone with LayerNorm passes
        self.last_fold_ratio = 0.0       # (None: never check).  Largest |mean| / sigma seen by the last forward()
        self.fold_fallbacks = 0          # calls redone so far
        self.prune_last_block = True     # the last block computes only what x[:, 0] needs (bit-identical CLS rows; False: A/B)
        # bf16 tower, folded calls: the residual stream between the blocks as bf16 hi + bf16 lo (16 significand bits) instead of an fp32
        # array: a fifth less epilogue traffic on proj / fc2.  True (or HIREST_F32_RESIDUAL=1) keeps the fp32 array of rounds 1-3.
        self.f32_residual = os.environ.get("HIREST_F32_RESIDUAL", "0") == "1"

    def _prepare(self, device):
        if self._prepared is not None and self._prepared["device"] == device and not self._prepared.get("f32"):
            return self._prepared
        if device.type != "cuda":
            raise RuntimeError("hirest_amd: the vision tower runs on MI355X only (no CPU fallback); move the model to a GPU")
        D, P = self.width, self.patch_size
        K = 3 * P * P
        kpad = (K + 63) // 64 * 64
        keep = []   # device tensors referenced by raw pointers below

        def hold(t):
            keep.append(t)
            return t.data_ptr()
        pw = torch.zeros((D, kpad), dtype=torch.float32, device=device)
        pw[:, :K] = self.patch_embed.proj.weight.detach().float().reshape(D, K)
        blocks = (_lib.BlockWeights * self.layers)()

        def fold(weight, bias, norm):
            """LayerNorm folded into the Linear that follows it (include/hirest_hip.h, HIREST_EPI_LNFOLD_*):
            LN(x) W^T + b = rstd (x W'^T - mean s) + b',  W' = W gamma (bf16), s = row sums of the bf16 W', b' = b + W beta:
            one kernel per Linear (hirest_fold_layernorm), once per checkpoint."""
            w32 = self._f32(weight)
            N, K = w32.shape
            wf = torch.empty((N, K), dtype=torch.bfloat16, device=device)
            bf = torch.empty((N,), dtype=torch.float32, device=device)
            cs = torch.empty((N,), dtype=torch.float32, device=device)
            _lib.check(_lib.load().hirest_fold_layernorm(w32.data_ptr(), self._f32(norm.weight).data_ptr(), self._f32(norm.bias).data_ptr(),
                                                         self._f32(bias).data_ptr(), wf.data_ptr(), bf.data_ptr(), cs.data_ptr(), N, K,
                                                         ops.stream_ptr()), "hirest_fold_layernorm")
            return hold(wf), hold(bf), hold(cs)
        for i, b in enumerate(self.blocks):
            qkv_b = torch.cat([b.attn.q_bias.detach().float(), torch.zeros(D, device=device), b.attn.v_bias.detach().float()])
            folded = (None,) * 6
            if self.fold_layernorm and not self.quick_gelu:
                folded = fold(b.attn.qkv.weight, qkv_b, b.norm1) + fold(b.mlp.fc1.weight, b.mlp.fc1.bias.detach().float(), b.norm2)
            blocks[i] = _lib.BlockWeights(
                hold(self._f32(b.norm1.weight)), hold(self._f32(b.norm1.bias)),
                hold(self._bf16(b.attn.qkv.weight)), hold(qkv_b.contiguous()),
                hold(self._bf16(b.attn.proj.weight)), hold(self._f32(b.attn.proj.bias)),
                hold(self._f32(b.norm2.weight)), hold(self._f32(b.norm2.bias)),
                hold(self._bf16(b.mlp.fc1.weight)), hold(self._f32(b.mlp.fc1.bias)),
                hold(self._bf16(b.mlp.fc2.weight)), hold(self._f32(b.mlp.fc2.bias)), *folded)
        mean = torch.tensor(self.image_mean, dtype=torch.float32, device=device)
        std = torch.tensor(self.image_std, dtype=torch.float32, device=device)
        desc = _lib.VisionTower(
            self.image_size, P, D, self.heads, D // self.heads, self.mlp_dim, self.layers, self.embed_dim, kpad,
            1 if self.quick_gelu else 0, 1e-6,   # norm_layer=partial(nn.LayerNorm, eps=1e-6), eva_model.py:304
            hold(ops.to_bf16(pw)), hold(self._f32(self.patch_embed.proj.bias)),
            hold(self._f32(self.cls_token).reshape(-1)), hold(self._f32(self.pos_embed).reshape(self.num_tokens, D)),
            blocks, hold(self._f32(self.norm.weight)), hold(self._f32(self.norm.bias)),
            hold(self._bf16(self.head.weight)), hold(self._f32(self.head.bias)), hold(mean), hold(std), None, None, 0)
        self._prepared = {"device": device, "desc": desc, "blocks": blocks, "keep": keep}
        return self._prepared

    def _prepare_f32(self, device):
        """fp32 master parameters as they are (contiguous fp32 views, no copies except the zero-padded patch weight)."""
        if self._prepared is not None and self._prepared["device"] == device and self._prepared.get("f32"):
            return self._prepared
        if device.type != "cuda":
            raise RuntimeError("hirest_amd: the vision tower runs on MI355X only (no CPU fallback); move the model to a GPU")
        D, P = self.width, self.patch_size
        K = 3 * P * P
        kpad = (K + 63) // 64 * 64
        keep = []

        def hold(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        pw = torch.zeros((D, kpad), dtype=torch.float32, device=device)
        pw[:, :K] = self.patch_embed.proj.weight.detach().float().reshape(D, K)
        blocks = (_lib.BlockWeightsF32 * self.layers)()
        for i, b in enumerate(self.blocks):
            qkv_b = torch.cat([b.attn.q_bias.detach().float(), torch.zeros(D, device=device), b.attn.v_bias.detach().float()])
            blocks[i] = _lib.BlockWeightsF32(
                hold(b.norm1.weight), hold(b.norm1.bias), hold(b.attn.qkv.weight), hold(qkv_b),
                hold(b.attn.proj.weight), hold(b.attn.proj.bias), hold(b.norm2.weight), hold(b.norm2.bias),
                hold(b.mlp.fc1.weight), hold(b.mlp.fc1.bias), hold(b.mlp.fc2.weight), hold(b.mlp.fc2.bias))
        mean = torch.tensor(self.image_mean, dtype=torch.float32, device=device)
        std = torch.tensor(self.image_std, dtype=torch.float32, device=device)
        desc = _lib.VisionTowerF32(
            self.image_size, P, D, self.heads, D // self.heads, self.mlp_dim, self.layers, self.embed_dim, kpad,
            1 if self.quick_gelu else 0, 1e-6, hold(pw), hold(self.patch_embed.proj.bias), hold(self.cls_token.reshape(-1)),
            hold(self.pos_embed.reshape(self.num_tokens, D)), blocks, hold(self.norm.weight), hold(self.norm.bias),
            hold(self.head.weight), hold(self.head.bias), hold(mean), hold(std), None, None, 0)
        self._prepared = {"device": device, "desc": desc, "blocks": blocks, "keep": keep, "f32": True}
        return self._prepared

    def _prepare_x3(self, device):
        """bf16x3: the fp32 descriptor plus, per block, the four linear weights split into bf16 hi | lo halves (ops.split2), once."""
        prep = self._prepare_f32(device)
        if "x3" in prep:
            return prep
        blocks = (_lib.BlockWeightsX3 * self.layers)()
        keep = prep["keep"]

        def split(w):
            t = ops.split2(w.detach().float().contiguous())
            keep.append(t)
            return t.data_ptr()
        for i, b in enumerate(self.blocks):
            blocks[i] = _lib.BlockWeightsX3(split(b.attn.qkv.weight), split(b.attn.proj.weight), split(b.mlp.fc1.weight), split(b.mlp.fc2.weight))
        prep["x3_blocks"] = blocks
        prep["x3"] = _lib.VisionTowerX3(C.pointer(prep["desc"]), blocks)
        return prep

    def _forward_x3(self, image: torch.Tensor) -> torch.Tensor:
        prep = self._prepare_x3(image.device)
        lib = _lib.load()
        B = image.shape[0]
        out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=image.device)
        if B == 0:
            return out
        calls = -(-B // max(1, int(self.max_frames_per_call_x3)))
        step = -(-B // calls)
        ws = self._ws(lib.hirest_vision_workspace_bytes_x3(C.byref(prep["x3"]), step), image.device)
        code = ops._IN_DTYPES[image.dtype]
        for s in range(0, B, step):
            n = min(step, B - s)
            _lib.check(lib.hirest_vision_forward_x3(C.byref(prep["x3"]), image[s:s + n].data_ptr(), code, n, out[s:s + n].data_ptr(),
                                                    ws.data_ptr(), ws.numel(), ops.stream_ptr()), "hirest_vision_forward_x3")
        return out

    def _forward_f32(self, image: torch.Tensor) -> torch.Tensor:
        prep = self._prepare_f32(image.device)
        lib = _lib.load()
        B = image.shape[0]
        out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=image.device)
        if B == 0:
            return out
        step = min(B, max(1, int(self.max_frames_per_call_f32)))
        ws = self._ws(lib.hirest_vision_workspace_bytes_f32(C.byref(prep["desc"]), step), image.device)
        code = ops._IN_DTYPES[image.dtype]
        for s in range(0, B, step):
            n = min(step, B - s)
            _lib.check(lib.hirest_vision_forward_f32(C.byref(prep["desc"]), image[s:s + n].data_ptr(), code, n, out[s:s + n].data_ptr(),
                                                     ws.data_ptr(), ws.numel(), ops.stream_ptr()), "hirest_vision_forward_f32")
        return out

    @torch.no_grad()
    @ops.on_tensor_device
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """image: [B,3,S,S] float (already normalised; NCHW) or uint8 [B,S,S,3] raw RGB (fused
        ToTensor+Normalize).  Returns [B, embed_dim] fp32, not normalised (vit_model.py:348-351)."""
        if image.dtype == torch.uint8:
            H, W = image.shape[1], image.shape[2]
        else:
            H, W = image.shape[-2], image.shape[-1]
            if image.dtype not in (torch.float32, torch.bfloat16):
                image = image.float()
        assert H == self.image_size and W == self.image_size, \
            f"Input image size ({H}*{W}) doesn't match model ({self.image_size}*{self.image_size})."  # vit_model.py:203
        image = image.contiguous()
        if self.precision == "fp32":
            return self._forward_f32(image)
        if self.precision == "bf16x3":
            return self._forward_x3(image)
        prep = self._prepare(image.device)
        lib = _lib.load()
        B = image.shape[0]
        out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=image.device)
        if B == 0:
            return out
        # Near-equal micro-batches instead of full ones + a remainder: with max_frames_per_call = 1024, 1030 frames run as
        # 515 + 515, not 1024 + 6, so a small tail never drops below the 64-frame boundary where the tower switches
        # kernels (folded LayerNorm / persistent attention) — every frame of a >= 64-frame call takes the same path.
        calls = -(-B // max(1, int(self.max_frames_per_call)))
        step = -(-B // calls)
        nbytes = lib.hirest_vision_workspace_bytes(C.byref(prep["desc"]), step)
        ws = self._ws(nbytes, image.device)
        code = ops._IN_DTYPES[image.dtype]
        self.last_fold_ratio = 0.0
        flags = (0 if self.prune_last_block else _lib.TOWER_NO_PRUNE) | (_lib.TOWER_F32_RESIDUAL if self.f32_residual else 0)
        goff = lib.hirest_vision_guard_offset(C.byref(prep["desc"]), step) if self.fold_guard_ratio is not None else _NO_GUARD
        starts = list(range(0, B, step))
        # Guard of the folded LayerNorm (include/hirest_hip.h): every call reports the largest |mean| / sigma any token row had
        # in any layer, as one float in its workspace.  The workspace is reused by the next micro-batch, so the float is copied
        # (device -> device, same stream) into a per-call slot and ALL slots are read with one 4 * calls-byte copy after the
        # last micro-batch has been enqueued: the host never waits for the GPU between micro-batches.
        guards = torch.zeros(len(starts), dtype=torch.float32, device=image.device) if goff != _NO_GUARD else None

        def call(s, extra=0):
            n = min(step, B - s)
            _lib.check(lib.hirest_vision_forward(C.byref(prep["desc"]), image[s:s + n].data_ptr(), code, n, out[s:s + n].data_ptr(),
                                                 ws.data_ptr(), ws.numel(), flags | extra, ops.stream_ptr()), "hirest_vision_forward")
            return n
        for i, s in enumerate(starts):
            n = call(s)
            if guards is not None:
                g = lib.hirest_vision_guard_offset(C.byref(prep["desc"]), n)
                if g != _NO_GUARD:
                    guards[i:i + 1].copy_(ws[g:g + 4].view(torch.float32))
        if guards is not None:
            ratios = guards.cpu().tolist()
            self.last_fold_ratio = max(ratios)
            for s, ratio in zip(starts, ratios):
                # Row offsets of more than `fold_guard_ratio` sigma would lose precision in the un-normalised bf16 operand: such
                # a micro-batch is repeated with the LayerNorm passes (non-finite rows report +inf, elementwise.hip).
                if not ratio <= self.fold_guard_ratio:
                    self.fold_fallbacks += 1
                    call(s, _lib.TOWER_NO_LNFOLD)
        return out


class TextTower(_Tower):
    """CLIP text transformer (reference TextTransformer, eva_model.py:177-250)."""

    def __init__(self, vocab_size, width, layers, heads, context_length, embed_dim, quick_gelu=False):
        super().__init__()
        self.vocab_size, self.width, self.layers, self.heads = vocab_size, width, layers, heads
        self.context_length, self.embed_dim, self.quick_gelu = context_length, embed_dim, quick_gelu
        D = width
        self.token_embedding = _Leaf(weight=(vocab_size, D))
        self.positional_embedding = nn.Parameter(torch.zeros(context_length, D))
        self.transformer = nn.Module()
        blocks = []
        for _ in range(layers):
            b = nn.Module()
            b.ln_1, b.ln_2 = _norm(D), _norm(D)
            b.attn = _Leaf(in_proj_weight=(3 * D, D), in_proj_bias=(3 * D,))
            b.attn.out_proj = _linear(D, D)
            b.mlp = nn.Module()
            b.mlp.c_fc, b.mlp.c_proj = _linear(4 * D, D), _linear(D, 4 * D)
            blocks.append(b)
        self.transformer.resblocks = nn.ModuleList(blocks)
        self.ln_final = _norm(D)
        self.text_projection = nn.Parameter(torch.zeros(D, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.max_rows_per_call = 1024

    def _prepare_f32(self, device):
        if self._prepared is not None and self._prepared["device"] == device and self._prepared.get("f32"):
            return self._prepared
        if device.type != "cuda":
            raise RuntimeError("hirest_amd: the text tower runs on MI355X only (no CPU fallback); move the model to a GPU")
        keep = []

        def hold(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        blocks = (_lib.BlockWeightsF32 * self.layers)()
        for i, b in enumerate(self.transformer.resblocks):
            blocks[i] = _lib.BlockWeightsF32(
                hold(b.ln_1.weight), hold(b.ln_1.bias), hold(b.attn.in_proj_weight), hold(b.attn.in_proj_bias),
                hold(b.attn.out_proj.weight), hold(b.attn.out_proj.bias), hold(b.ln_2.weight), hold(b.ln_2.bias),
                hold(b.mlp.c_fc.weight), hold(b.mlp.c_fc.bias), hold(b.mlp.c_proj.weight), hold(b.mlp.c_proj.bias))
        desc = _lib.TextTowerF32(
            self.context_length, self.vocab_size, self.width, self.heads, self.layers, self.embed_dim,
            1 if self.quick_gelu else 0, 1e-5, hold(self.token_embedding.weight), hold(self.positional_embedding), blocks,
            hold(self.ln_final.weight), hold(self.ln_final.bias), hold(self.text_projection.detach().float().t()))
        self._prepared = {"device": device, "desc": desc, "blocks": blocks, "keep": keep, "f32": True}
        return self._prepared

    def _prepare(self, device):
        if self._prepared is not None and self._prepared["device"] == device and not self._prepared.get("f32"):
            return self._prepared
        if device.type != "cuda":
            raise RuntimeError("hirest_amd: the text tower runs on MI355X only (no CPU fallback); move the model to a GPU")
        keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()
        blocks = (_lib.BlockWeights * self.layers)()
        for i, b in enumerate(self.transformer.resblocks):
            blocks[i] = _lib.BlockWeights(
                hold(self._f32(b.ln_1.weight)), hold(self._f32(b.ln_1.bias)),
                hold(self._bf16(b.attn.in_proj_weight)), hold(self._f32(b.attn.in_proj_bias)),
                hold(self._bf16(b.attn.out_proj.weight)), hold(self._f32(b.attn.out_proj.bias)),
                hold(self._f32(b.ln_2.weight)), hold(self._f32(b.ln_2.bias)),
                hold(self._bf16(b.mlp.c_fc.weight)), hold(self._f32(b.mlp.c_fc.bias)),
                hold(self._bf16(b.mlp.c_proj.weight)), hold(self._f32(b.mlp.c_proj.bias)))
        desc = _lib.TextTower(
            self.context_length, self.vocab_size, self.width, self.heads, self.layers, self.embed_dim,
            1 if self.quick_gelu else 0, 1e-5,   # LayerNorm default eps (eva_model.py:19-25)
            hold(self._f32(self.token_embedding.weight)), hold(self._f32(self.positional_embedding)), blocks,
            hold(self._f32(self.ln_final.weight)), hold(self._f32(self.ln_final.bias)),
            hold(self._bf16(self.text_projection.detach().float().t().contiguous())))
        self._prepared = {"device": device, "desc": desc, "blocks": blocks, "keep": keep}
        return self._prepared

    @torch.no_grad()
    @ops.on_tensor_device
    def forward(self, text: torch.Tensor) -> torch.Tensor:
        """text: [B, context_length] int64 token ids (EOT = row max). Returns [B, embed_dim] fp32."""
        if text.dim() != 2 or text.shape[1] != self.context_length:
            raise RuntimeError(f"encode_text expects [B,{self.context_length}] token ids, got {tuple(text.shape)}")
        f32 = self.precision == "fp32"
        prep = self._prepare_f32(text.device) if f32 else self._prepare(text.device)
        lib = _lib.load()
        text = text.to(torch.int64).contiguous()
        B = text.shape[0]
        out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=text.device)
        step = max(1, int(self.max_rows_per_call))
        ws_bytes, fwd = (lib.hirest_text_workspace_bytes_f32, lib.hirest_text_forward_f32) if f32 else \
            (lib.hirest_text_workspace_bytes, lib.hirest_text_forward)
        ws = self._ws(ws_bytes(C.byref(prep["desc"]), min(B, step)), text.device)
        for s in range(0, B, step):
            n = min(step, B - s)
            _lib.check(fwd(C.byref(prep["desc"]), text[s:s + n].data_ptr(), n, out[s:s + n].data_ptr(), ws.data_ptr(), ws.numel(),
                           ops.stream_ptr()), "hirest_text_forward_f32" if f32 else "hirest_text_forward")
        return out


class EVA_CLIP(nn.Module):
    """Same constructor arguments and methods as the reference class (eva_model.py:270-334)."""

    def __init__(self, embed_dim: int, vision_cfg: dict, text_cfg: dict, quick_gelu: bool = False):
        super().__init__()
        v, t = dict(vision_cfg), dict(text_cfg)
        self.visual = VisionTower(v.get("image_size", 224), v.get("patch_size", 16), v.get("width", 768),
                                  v.get("layers", 12), v.get("width", 768) // v.get("head_width", 64),
                                  v.get("mlp_ratio", 4.0), embed_dim)   # always nn.GELU: the reference passes
        # act_layer to the TextTransformer only (eva_model.py:283-312), also under force_quick_gelu
        self.text = TextTower(t.get("vocab_size", 49408), t.get("width", 512), t.get("layers", 12), t.get("heads", 8),
                              t.get("context_length", 77), embed_dim, quick_gelu)
        self.output_dtype = torch.float32

    def set_precision(self, precision: str):
        """'fp32' = the reference's own arithmetic (exact-fp32 kernels, eva_clip.py:90 default); 'bf16' = the bf16 MFMA towers;
        'bf16x3' = the fp32 forward with the vision tower's weight GEMMs on bf16 hi + lo splits of both operands (csrc/tower_x3.hip:
        ~16-bit products at 3/16 of the fp32 matrix cost; the text tower, 2 % of a retrieval run's work, stays exact fp32)."""
        if precision not in TOWER_PRECISIONS:
            raise ValueError(f"precision must be one of {TOWER_PRECISIONS}, got {precision!r}")
        self.visual.precision = precision
        self.text.precision = "fp32" if precision == "bf16x3" else precision
        return self

    @torch.no_grad()
    def init_random_(self, seed: int = 0):
        """Random-init every parameter IN PLACE on its current device with the same per-tensor
        scales as hirest_amd.synth (fast path for benchmarks: no pretrained weights exist offline)."""
        gen = torch.Generator(device=next(self.parameters()).device)
        gen.manual_seed(seed)
        for name, p in self.named_parameters():
            std, mean = synth._init_rule(name, tuple(p.shape))
            if std == 0.0:
                p.fill_(mean)
            else:
                p.normal_(mean, std, generator=gen)
        self.visual.invalidate()
        self.text.invalidate()
        return self

    def encode_image(self, image):
        return self.visual(image).to(self.output_dtype)

    def encode_text(self, text):
        return self.text(text).to(self.output_dtype)

    def forward(self, image, text):
        if image is None:
            return self.encode_text(text)
        elif text is None:
            return self.encode_image(image)
        img = self.visual(image)
        txt = self.text(text)
        # F.normalize(dim=-1) == the pooling kernel with one "frame" per row
        img_n = ops.pool_l2norm(img.unsqueeze(1)).to(self.output_dtype)
        txt_n = ops.pool_l2norm(txt.unsqueeze(1)).to(self.output_dtype)
        return img_n, txt_n, self.text.logit_scale.exp()


def load_state_dict(checkpoint_path: str, map_location: str = "cpu", model_key="model|module|state_dict"):
    """eva_clip.py:68-79."""
    if isinstance(checkpoint_path, str) and checkpoint_path.startswith("synth:"):
        raise RuntimeError("synthetic checkpoints are resolved in create_model")
    checkpoint = torch.load(checkpoint_path, map_location=map_location)
    state_dict = checkpoint
    for mk in model_key.split("|"):
        if isinstance(checkpoint, dict) and mk in checkpoint:
            state_dict = checkpoint[mk]
            break
    if next(iter(state_dict.items()))[0].startswith("module"):
        state_dict = {k[7:]: v for k, v in state_dict.items()}
    return state_dict


def load_checkpoint(model, checkpoint_path, model_key="model|module|state_dict", strict=True):
    """eva_clip.py:81-85."""
    sd = load_state_dict(checkpoint_path, model_key=model_key)
    return model.load_state_dict(sd, strict=strict)


def create_model(model_name: str, pretrained: str = "", precision: str = "fp32",
                 device: torch.device = torch.device("cpu"), force_quick_gelu: bool = False):
    """eva_clip.py:87-120.  ``pretrained`` is a checkpoint path as in the reference; additionally
    ``"synth:<seed>"`` fills the model with hirest_amd.synth's deterministic synthetic weights
    (no pretrained files exist offline)."""
    model_name = model_name.replace("/", "-")
    if model_name not in _MODEL_CONFIGS:
        raise RuntimeError(f"Model config for {model_name} not found.")
    cfg = deepcopy(_MODEL_CONFIGS[model_name])
    if force_quick_gelu:
        cfg["quick_gelu"] = True
    synthetic = isinstance(pretrained, str) and pretrained.startswith("synth:")
    if not synthetic and not os.path.isfile(pretrained):
        # what the reference's torch.load raises after it has built the 1.2 B-parameter model; raised before building here
        raise FileNotFoundError(f"[Errno 2] No such file or directory: '{pretrained}'")
    model = EVA_CLIP(**cfg)
    if isinstance(pretrained, str) and pretrained.startswith("synth:"):
        model.load_state_dict(synth.eva_clip_state_dict(cfg, int(pretrained.split(":", 1)[1])), strict=True)
    else:
        load_checkpoint(model, pretrained)
    device = torch.device(device)
    model.to(device=device)
    # precision: 'fp32' (the reference's default) runs the exact-fp32 towers, so ranks and indices downstream are the fp32
    # reference's; 'bf16' / 'amp' / 'amp_bf16' select the bf16 MFMA towers (the measured hot path, ~16x faster; embeddings
    # within cos 0.9999 of fp32); 'fp16' as the reference: half-precision outputs (computed on the bf16 towers).
    # HIREST_PRECISION=<fp32 | bf16x3 | bf16>: opt-in override for UNMODIFIED reference callers, none of which passes
    # ``precision`` (run.py / inference_video_retrieval.py / extract_features.py all get the fp32 default): the environment
    # selects the towers without an edit at the call site.  Unset: the argument decides, as in the reference.
    env = os.environ.get("HIREST_PRECISION", "").strip().lower()
    if env and env not in TOWER_PRECISIONS:
        raise ValueError(f"HIREST_PRECISION={env!r}: expected one of {TOWER_PRECISIONS}")
    if env and precision != "fp32" and precision != env:
        # an explicit non-default argument is the caller's decision; the environment only redirects the fp32 DEFAULT of unmodified callers
        import warnings
        warnings.warn(f"hirest_amd: HIREST_PRECISION={env} ignored: create_model was called with precision={precision!r}")
        env = ""
    if env:
        model.set_precision(env)
    elif precision == "fp32":
        model.set_precision("fp32")
    elif precision in TOWER_PRECISIONS:
        model.set_precision(precision)
    elif precision in ("bf16", "amp", "amp_bf16", "amp_bfloat16", "fp16"):
        model.set_precision("bf16")
    else:
        raise ValueError(f"unknown precision {precision!r} (fp32 | bf16 | amp | fp16)")
    if precision == "fp16":
        assert device.type != "cpu"
        model.output_dtype = torch.float16   # reference returns the model dtype (eva_model.py:337-358)
    model.visual.image_mean = OPENAI_DATASET_MEAN
    model.visual.image_std = OPENAI_DATASET_STD
    return model


class ImageTransform:
    """PIL image -> normalised [3,S,S] fp32 tensor: the reference's torchvision pipeline
    Resize(S, BICUBIC) -> CenterCrop(S) -> RGB -> ToTensor -> Normalize (eva_clip.py:125-153),
    restated on PIL + numpy (torchvision is not a dependency)."""

    def __init__(self, image_size: int, mean=None, std=None):
        if isinstance(image_size, (list, tuple)):
            image_size = image_size[0]
        self.size = int(image_size)
        mean = mean or OPENAI_DATASET_MEAN
        std = std or OPENAI_DATASET_STD
        mean = (mean,) * 3 if not isinstance(mean, (list, tuple)) else mean
        std = (std,) * 3 if not isinstance(std, (list, tuple)) else std
        self.mean = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
        self.std = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)

    def __call__(self, img):
        from PIL import Image
        w, h = img.size
        s = self.size
        if not ((w <= h and w == s) or (h <= w and h == s)):   # torchvision Resize(int): short side -> s
            if w < h:
                nw, nh = s, int(s * h / w)
            else:
                nh, nw = s, int(s * w / h)
            img = img.resize((nw, nh), Image.BICUBIC)
            w, h = nw, nh
        left, top = int(round((w - s) / 2.0)), int(round((h - s) / 2.0))
        img = img.crop((left, top, left + s, top + s)).convert("RGB")
        a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
        return torch.from_numpy((a - self.mean) / self.std)


def image_transform(image_size: int, mean: Optional[Tuple[float, ...]] = None, std: Optional[Tuple[float, ...]] = None):
    return ImageTransform(image_size, mean, std)


def build_eva_model_and_transforms(model_name: str, pretrained: str = "", precision: str = "fp32",
                                   device: torch.device = torch.device("cpu"), force_quick_gelu: bool = False,
                                   image_mean: Optional[Tuple[float, ...]] = None,
                                   image_std: Optional[Tuple[float, ...]] = None):
    """eva_clip.py:155-172: returns (model in train mode, preprocess)."""
    model = create_model(model_name, pretrained, precision, device, force_quick_gelu=force_quick_gelu)
    image_mean = image_mean or getattr(model.visual, "image_mean", None)
    image_std = image_std or getattr(model.visual, "image_std", None)
    preprocess = image_transform(model.visual.image_size, mean=image_mean, std=image_std)
    return model, preprocess
