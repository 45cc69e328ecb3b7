"""Drop-in for the reference's vendored OpenAI-CLIP module (/root/reference/EVA_clip/clip.py + model.py),
ViT variants only (the ResNet visual towers of model.py:10-163 are never instantiated by any HiREST caller).

    model, preprocess = clip.load("/path/to/ViT-B-32.pt", device="cuda")     # clip.py:94 (state-dict or JIT archive)
    model = clip.build_model(state_dict)                                       # model.py:434-471 (dims inferred)
    tokens = clip.tokenize(["a photo of a cat"])                               # clip.py:196

Behavioural notes that callers rely on (and that the parity tests pin against the reference):
* the vendored ``VisionTransformer.forward`` drops the CLS token and returns ``ln_post(patch tokens) @ proj``:
  ``encode_image`` -> ``[B, grid^2, embed_dim]`` (model.py:269-273; SURVEY hazard H4), not a CLS embedding;
* QuickGELU activations, ``ln_pre`` before the stack, conv1 without bias, LayerNorm eps 1e-5;
* ``encode_text`` = EOT-row gather @ text_projection (model.py:343-356).
Compute: the same gfx950 kernels as hirest_amd.eva_clip (bf16 MFMA GEMMs, fp32 statistics).  No download
support (no network): ``load`` needs a local file.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Union

import torch
from torch import nn

from . import _lib, ops
from .eva_clip import TextTower, _Leaf, _Tower, _linear, _norm, image_transform
from .tokenizer import tokenize  # noqa: F401  (clip.tokenize)

__all__ = ["available_models", "load", "tokenize", "build_model", "CLIP"]


def available_models() -> List[str]:
    """clip.py:89-91 lists downloadable names; offline there are none — pass a checkpoint path to ``load``."""
    return []


class OpenAIVisionTower(_Tower):
    """model.py:216-273, parameter names identical (conv1, class_embedding, positional_embedding, ln_pre,
    transformer.resblocks.N.*, ln_post, proj)."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.patch_size, self.width, self.layers, self.heads = input_resolution, patch_size, width, layers, heads
        self.output_dim = output_dim
        self.grid = input_resolution // patch_size
        self.num_tokens = self.grid ** 2 + 1
        D = width
        self.conv1 = _Leaf(weight=(D, 3, patch_size, patch_size))
        self.class_embedding = nn.Parameter(torch.zeros(D))
        self.positional_embedding = nn.Parameter(torch.zeros(self.num_tokens, D))
        self.ln_pre = _norm(D)
        self.transformer = nn.Module()
        blocks = []
        for _ in range(layers):
            b = nn.Module()
            b.attn = _Leaf(in_proj_weight=(3 * D, D), in_proj_bias=(3 * D,))
            b.attn.out_proj = _linear(D, D)
            b.ln_1, b.ln_2 = _norm(D), _norm(D)
            b.mlp = nn.Module()
            b.mlp.c_fc, b.mlp.c_proj = _linear(4 * D, D), _linear(D, 4 * D)
            blocks.append(b)
        self.transformer.resblocks = nn.ModuleList(blocks)
        # 'bf16' (default: the MFMA towers) | 'fp32' (exact-fp32 kernels of csrc/tower_f32.hip with its QuickGELU epilogue, ln_pre and
        # all-token head: the reference's own arithmetic, so cosines land within 1e-5 of it and its top-k ids are reproduced);
        # CLIP.set_precision / clip.load(..., precision=) / HIREST_PRECISION select it
        self.precision = "bf16"
        self.pip_head = False          # True: return the CLS embedding like the pip `clip` package (clip.load(..., pip_head=True))
        self.ln_post = _norm(D)
        self.proj = nn.Parameter(torch.zeros(D, output_dim))
        self.max_frames_per_call = 2048

    def _prepare(self, device):
        if self._prepared is not None and self._prepared["device"] == device:
            return self._prepared
        if device.type != "cuda":
            raise RuntimeError("hirest_amd: the CLIP vision tower runs on MI355X only (no CPU fallback)")
        D, P = self.width, self.patch_size
        K = 3 * P * P
        kpad = (K + 63) // 64 * 64
        keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()
        pw = torch.zeros((D, kpad), dtype=torch.float32, device=device)
        pw[:, :K] = self.conv1.weight.detach().float().reshape(D, K)
        blocks = (_lib.BlockWeights * self.layers)()
        for i, b in enumerate(self.transformer.resblocks):
            blocks[i] = _lib.BlockWeights(
                hold(self._f32(b.ln_1.weight)), hold(self._f32(b.ln_1.bias)),
                hold(self._bf16(b.attn.in_proj_weight)), hold(self._f32(b.attn.in_proj_bias)),
                hold(self._bf16(b.attn.out_proj.weight)), hold(self._f32(b.attn.out_proj.bias)),
                hold(self._f32(b.ln_2.weight)), hold(self._f32(b.ln_2.bias)),
                hold(self._bf16(b.mlp.c_fc.weight)), hold(self._f32(b.mlp.c_fc.bias)),
                hold(self._bf16(b.mlp.c_proj.weight)), hold(self._f32(b.mlp.c_proj.bias)))
        mean = torch.tensor((0.48145466, 0.4578275, 0.40821073), dtype=torch.float32, device=device)
        std = torch.tensor((0.26862954, 0.26130258, 0.27577711), dtype=torch.float32, device=device)
        desc = _lib.VisionTower(
            self.input_resolution, P, D, self.heads, D // self.heads, 4 * D, self.layers, self.output_dim, kpad,
            1, 1e-5,                                                  # QuickGELU (model.py:175), nn.LayerNorm default eps
            hold(ops.to_bf16(pw)), None,                              # conv1 has no bias (model.py:220)
            hold(self._f32(self.class_embedding)), hold(self._f32(self.positional_embedding)),
            blocks, hold(self._f32(self.ln_post.weight)), hold(self._f32(self.ln_post.bias)),
            hold(self._bf16(self.proj.detach().float().t().contiguous())), None, hold(mean), hold(std),
            hold(self._f32(self.ln_pre.weight)), hold(self._f32(self.ln_pre.bias)), 1)
        self._prepared = {"device": device, "desc": desc, "blocks": blocks, "keep": keep}
        return self._prepared

    def _prepare_f32(self, device):
        """The fp32 master parameters as they are (contiguous fp32 views; the zero-padded conv weight and proj^T are copies)."""
        if self._prepared is not None and self._prepared["device"] == device and self._prepared.get("f32"):
            return self._prepared
        if device.type != "cuda":
            raise RuntimeError("hirest_amd: the CLIP vision tower runs on MI355X only (no CPU fallback)")
        D, P = self.width, self.patch_size
        K = 3 * P * P
        kpad = (K + 63) // 64 * 64
        keep = []

        def hold(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        pw = torch.zeros((D, kpad), dtype=torch.float32, device=device)
        pw[:, :K] = self.conv1.weight.detach().float().reshape(D, K)
        blocks = (_lib.BlockWeightsF32 * self.layers)()
        for i, b in enumerate(self.transformer.resblocks):
            blocks[i] = _lib.BlockWeightsF32(
                hold(b.ln_1.weight), hold(b.ln_1.bias), hold(b.attn.in_proj_weight), hold(b.attn.in_proj_bias),
                hold(b.attn.out_proj.weight), hold(b.attn.out_proj.bias), hold(b.ln_2.weight), hold(b.ln_2.bias),
                hold(b.mlp.c_fc.weight), hold(b.mlp.c_fc.bias), hold(b.mlp.c_proj.weight), hold(b.mlp.c_proj.bias))
        mean = torch.tensor((0.48145466, 0.4578275, 0.40821073), dtype=torch.float32, device=device)
        std = torch.tensor((0.26862954, 0.26130258, 0.27577711), dtype=torch.float32, device=device)
        desc = _lib.VisionTowerF32(
            self.input_resolution, P, D, self.heads, D // self.heads, 4 * D, self.layers, self.output_dim, kpad,
            1, 1e-5,                                                  # QuickGELU (model.py:175), nn.LayerNorm default eps
            hold(pw), None,                                           # conv1 has no bias (model.py:220)
            hold(self.class_embedding), hold(self.positional_embedding), blocks, hold(self.ln_post.weight), hold(self.ln_post.bias),
            hold(self.proj.detach().float().t()), None, hold(mean), hold(std),
            hold(self.ln_pre.weight), hold(self.ln_pre.bias), 1)       # ln_pre; ln_post + proj on every token (model.py:229-273)
        self._prepared = {"device": device, "desc": desc, "blocks": blocks, "keep": keep, "f32": True}
        return self._prepared

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            x = x.float()
        f32 = self.precision != "bf16"
        if self._prepared is not None and bool(self._prepared.get("f32")) != f32:
            self._prepared = None                      # the other kernel set's descriptor
        prep = self._prepare_f32(x.device) if f32 else self._prepare(x.device)
        lib = _lib.load()
        x = x.contiguous()
        B, T, E = x.shape[0], self.num_tokens, self.output_dim
        out = torch.empty((B, T, E), dtype=torch.float32, device=x.device)
        step = max(1, int(self.max_frames_per_call))
        if B == 0:
            return out[:, 0, :] if self.pip_head else out[:, 1:, :]
        wsb = lib.hirest_vision_workspace_bytes_f32 if f32 else lib.hirest_vision_workspace_bytes
        ws = self._ws(wsb(C.byref(prep["desc"]), min(B, step)), x.device)
        for s in range(0, B, step):
            n = min(step, B - s)
            if f32:
                _lib.check(lib.hirest_vision_forward_f32(C.byref(prep["desc"]), x[s:s + n].data_ptr(), ops._IN_DTYPES[x.dtype], n,
                                                         out[s:s + n].data_ptr(), ws.data_ptr(), ws.numel(), ops.stream_ptr()),
                           "hirest_vision_forward_f32")
                continue
            _lib.check(lib.hirest_vision_forward(C.byref(prep["desc"]), x[s:s + n].data_ptr(), ops._IN_DTYPES[x.dtype], n,
                                                 out[s:s + n].data_ptr(), ws.data_ptr(), ws.numel(), 0, ops.stream_ptr()),
                       "hirest_vision_forward")
        if self.pip_head:
            # the pip `clip` package's head (openai/CLIP @ a9b1bf5, model.py VisionTransformer.forward: ln_post(x[:, 0, :]) @ proj):
            # LayerNorm and projection act per token, so it is row 0 of what the kernel computed for every token
            return out[:, 0, :]
        return out[:, 1:, :]            # w/o cls token (model.py:269)


class CLIP(nn.Module):
    """model.py:277-406 (ViT visual only)."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet visual towers are not used by HiREST and not implemented")
        self.context_length = context_length
        self.visual = OpenAIVisionTower(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        # the text tower's parameters sit at the top level in the reference; reuse TextTower and re-export them
        t = TextTower(vocab_size, transformer_width, transformer_layers, transformer_heads, context_length, embed_dim,
                      quick_gelu=True)
        object.__setattr__(self, "_text", t)          # not a registered submodule: names below are the real ones
        self.transformer = t.transformer
        self.token_embedding = t.token_embedding
        self.positional_embedding = t.positional_embedding
        self.ln_final = t.ln_final
        self.text_projection = t.text_projection
        self.logit_scale = t.logit_scale
        self.vocab_size = vocab_size

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._text.invalidate()
        self._text._workspace = None
        return r

    def _load_from_state_dict(self, *a, **k):
        self._text.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def set_precision(self, precision: str):
        """'bf16' = the bf16 MFMA towers (default); 'fp32' = both towers in exact fp32 (the reference's own arithmetic: model.py runs in
        fp32 on CPU, clip.py:136-138): cosines within 1e-5 of the reference, its top-k ids reproduced; 'bf16x3' is accepted and runs the
        fp32 kernels (ViT-B/32 is 4.4 GFLOP per frame: the exact path is cheap, no split-operand variant is built for it)."""
        if precision not in ("bf16", "fp32", "bf16x3"):
            raise ValueError(f"precision must be one of ('bf16', 'fp32', 'bf16x3'), got {precision!r}")
        p = "bf16" if precision == "bf16" else "fp32"
        self.visual.precision = p
        self._text.precision = p
        return self

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        return self._text(text)

    def forward(self, image, text):
        """model.py:358-372: cosine logits scaled by exp(logit_scale)."""
        img = self.encode_image(image)
        txt = self.encode_text(text)
        # F.normalize = the pooling kernel with one row per "video"; logits = exp(logit_scale) * img_n txt_n^T on the fp32 MFMA GEMM
        img = ops.pool_l2norm(img.float().unsqueeze(1).contiguous())
        txt = ops.pool_l2norm(txt.float().unsqueeze(1).contiguous())
        logits_per_image = ops.gemm_f32_strided(img, txt, float(torch.exp(self.logit_scale.detach().float().cpu())))
        return logits_per_image, logits_per_image.transpose(-1, -2)


def build_model(state_dict: Dict[str, torch.Tensor]) -> CLIP:
    """model.py:434-471: infer every dimension from the checkpoint, load strictly, return in eval mode."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("only ViT CLIP checkpoints are supported (ModifiedResNet is unused by HiREST)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
                 transformer_width, transformer_heads, transformer_layers)
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return model.eval()


def load(name: str, device: Union[str, torch.device] = "cuda", jit: bool = False, download_root: str = None, pip_head: bool = False,
         precision: str = "bf16"):
    """clip.py:94-193 for local checkpoints: a JIT archive or a plain state dict -> (model, preprocess).

    ``pip_head=True`` gives the model of the *pip* ``clip`` package (openai/CLIP @ a9b1bf5, requirements.txt:25) that the reference's
    ``'clip'`` retrieval branch imports (inference_video_retrieval.py:11,169; hirest_dataset.py:84): same weights, same tower, but
    ``encode_image`` returns the CLS embedding ``ln_post(x[:, 0]) @ proj`` -> [B, embed_dim] instead of the vendored copy's projected
    patch tokens.  That package is not part of the reference tree, so this head is checked against the oracle's restatement of its
    published forward only (parity unpinned)."""
    if not os.path.isfile(name):
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    try:
        state_dict = torch.jit.load(name, map_location="cpu").eval().state_dict()
    except RuntimeError:
        state_dict = torch.load(name, map_location="cpu")
    model = build_model(state_dict).to(device)
    # precision: this argument, unless HIREST_PRECISION (the opt-in for unmodified reference callers, as in eva_clip.create_model) is set
    env = os.environ.get("HIREST_PRECISION", "").strip().lower()
    model.set_precision(env or precision)
    model.visual.pip_head = bool(pip_head)
    return model, image_transform(model.visual.input_resolution)
