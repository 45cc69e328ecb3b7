"""Device-side frame preprocessing: the reference's ``image_transform`` (EVA_clip/eva_clip.py:125-153, identical to
EVA_clip/clip.py:79-86) applied to whole batches of decoded RGB frames on the GPU instead of one PIL image at a time
on one CPU thread (the loop at extraction/video_features/extract_features.py:46-50 that bounds the reference's
feature extraction).

    pre = FramePreprocessor(224)                       # or model.visual.image_size / mean / std
    u8  = pre(frames_u8)                               # [B,H,W,3] uint8 cuda -> [B,224,224,3] uint8 (resize + crop)
    emb = model.encode_image(u8)                       # normalisation fused into patch extraction
    x   = pre(frames_u8, normalized=True)              # [B,3,224,224] fp32 == torch.stack([preprocess(img) ...])

Bit-exact with Pillow's ``Image.resize(..., BICUBIC)`` + torchvision's Resize/CenterCrop size rules (pinned in
tests against Pillow itself).  RGB uint8 input only (``convert('RGB')`` is the identity for decoded video frames).
No CPU fallback: the per-image CPU transform stays available as ``hirest_amd.image_transform`` for PIL inputs.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib, ops

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # eva_clip.py:16
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # eva_clip.py:17


def host_plan(in_h: int, in_w: int, size: int) -> np.ndarray:
    """The int32 plan blob of hirest_preprocess_plan (header + bounds / fixed-point weight tables); host only."""
    lib = _lib.load()
    nbytes = lib.hirest_preprocess_plan_bytes(in_h, in_w, size)
    if nbytes < 0:
        raise ValueError(f"unsupported preprocess geometry {in_h}x{in_w} -> {size}")
    buf = np.zeros(nbytes // 4, dtype=np.int32)
    _lib.check(lib.hirest_preprocess_plan(in_h, in_w, size, buf.ctypes.data, nbytes), "hirest_preprocess_plan")
    return buf


class FramePreprocessor:
    def __init__(self, image_size: int = 224, mean=None, std=None):
        if isinstance(image_size, (list, tuple)):
            image_size = image_size[0]
        self.size = int(image_size)
        self.mean = tuple(mean or OPENAI_DATASET_MEAN)
        self.std = tuple(std or OPENAI_DATASET_STD)
        self._plans: Dict[Tuple[int, int, str], torch.Tensor] = {}
        self._consts: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._ws = None

    def _plan(self, h: int, w: int, device) -> torch.Tensor:
        key = (h, w, str(device))
        if key not in self._plans:
            self._plans[key] = torch.from_numpy(host_plan(h, w, self.size)).to(device)
        return self._plans[key]

    @torch.no_grad()
    def __call__(self, frames: torch.Tensor, normalized: bool = False) -> torch.Tensor:
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("FramePreprocessor expects uint8 RGB frames [B,H,W,3]")
        if frames.device.type != "cuda":
            raise RuntimeError("hirest_amd: FramePreprocessor runs on MI355X only (no CPU fallback; "
                               "use hirest_amd.image_transform for PIL images)")
        lib = _lib.load()
        frames = frames.contiguous()
        B, H, W, _ = frames.shape
        S = self.size
        dev = frames.device
        plan = self._plan(H, W, dev)
        if str(dev) not in self._consts:
            self._consts[str(dev)] = (torch.tensor(self.mean, dtype=torch.float32, device=dev),
                                      torch.tensor(self.std, dtype=torch.float32, device=dev))
        mean, std = self._consts[str(dev)]
        if normalized:
            out = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
        else:
            out = torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev)
        step = 4096                                                     # grid.y limit is 65535; keeps the workspace small
        need = lib.hirest_preprocess_workspace_bytes(H, W, S, min(B, step))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(max(int(need), 1), dtype=torch.uint8, device=dev)
        for s in range(0, B, step):
            n = min(step, B - s)
            _lib.check(lib.hirest_preprocess_u8(frames[s:s + n].data_ptr(), n, H, W, S, plan.data_ptr(), out[s:s + n].data_ptr(),
                                                1 if normalized else 0, mean.data_ptr(), std.data_ptr(), self._ws.data_ptr(),
                                                self._ws.numel(), ops.stream_ptr()), "hirest_preprocess_u8")
        return out
