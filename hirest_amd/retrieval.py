"""Video retrieval on top of the encoders: per-video frame batches -> mean-pool + L2 -> (RCCL
all-gather of embedding rows) -> text x video cosine matrix -> top-k with the reference's tie rule.

Reference path: /root/reference/inference_video_retrieval.py:203-215 (text loop), :257-288 (raw
frames -> encode_image -> view(B,F,1024) -> mean -> /= norm), :298-334 (feature files, linspace
subsample, scores = T @ V.T) and evaluate.py:33-81 (ranking, R@k).  The reference is single
process; its only sharding precedent is ``ids[process_id::num_process]`` with results written to
disk (:226-237).  Here videos are block-sharded over ranks (one process per GPU) and the pooled
``[V/R, E]`` fp32 rows are merged with ONE ``all_gather_into_tensor`` over RCCL/xGMI; scoring is
then replicated (546 x 4096 x 1024 is 4.6 GFLOP — microseconds).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import ops


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous block shard, equal padded size: returns (lo, hi, per_rank)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per), per


def subsample_ids(n_frames: int, n_model_frames: int) -> np.ndarray:
    """np.linspace(0, n-1, F).astype(int) (inference_video_retrieval.py:39,315)."""
    return np.linspace(0, n_frames - 1, n_model_frames).astype(int)


@torch.no_grad()
def encode_videos(model, frames: torch.Tensor, normalize_frames_first: bool = False,
                  return_frame_embeds: bool = False):
    """frames [V,F,3,S,S] (or uint8 [V,F,S,S,3]) -> pooled, L2-normalised [V,E] fp32
    (inference_video_retrieval.py:266-285).  Raw decoded uint8 frames [V,F,H,W,3] of any resolution are resized and
    cropped on the device first (the ``preprocess(Image.open(...))`` of :267-269 / extract_features.py:46-50)."""
    V, F = frames.shape[0], frames.shape[1]
    size = getattr(getattr(model, "visual", None), "image_size", None)
    if frames.dtype == torch.uint8 and size is not None and tuple(frames.shape[2:4]) != (size, size):
        from .preprocess import FramePreprocessor
        pre = getattr(model, "_frame_preprocessor", None)
        if pre is None:
            pre = FramePreprocessor(size, getattr(model.visual, "image_mean", None), getattr(model.visual, "image_std", None))
            object.__setattr__(model, "_frame_preprocessor", pre)
        frames = pre(frames.reshape((V * F,) + tuple(frames.shape[2:]))).reshape(V, F, size, size, 3)
    fe = model.encode_image(frames.reshape((V * F,) + tuple(frames.shape[2:]))).float().reshape(V, F, -1)
    pooled = ops.pool_l2norm(fe.contiguous(), normalize_frames_first)
    return (pooled, fe) if return_frame_embeds else pooled


@torch.no_grad()
def encode_texts(model, tokens: torch.Tensor) -> torch.Tensor:
    """[Q,77] token ids -> L2-normalised [Q,E] fp32 (inference_video_retrieval.py:207-212)."""
    te = model.encode_text(tokens).float()
    return ops.pool_l2norm(te.unsqueeze(1).contiguous())


class RowGather:
    """The exchange step of the sharded corpus: every rank contributes an equally padded block [per, E] of pooled
    rows, one ``all_gather_into_tensor`` (RCCL over xGMI on GPU tensors, gloo on CPU tensors) returns [n_total, E] in
    rank order.  Send and receive buffers are allocated ONCE per (shape, dtype, device) and reused by every call —
    a steady-state retrieval step allocates nothing."""

    def __init__(self, group=None):
        self.group = group
        self._key = None
        self._send = self._recv = None

    def _buffers(self, per, tail, dtype, device, world):
        key = (per, tail, dtype, device, world)
        if key != self._key:
            self._send = torch.zeros((per,) + tail, dtype=dtype, device=device)
            self._recv = torch.empty((world * per,) + tail, dtype=dtype, device=device)
            self._key = key
        return self._send, self._recv

    def __call__(self, local: torch.Tensor, n_total: int) -> torch.Tensor:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return local[:n_total]
        world = dist.get_world_size(self.group)
        per = (n_total + world - 1) // world
        if local.shape[0] > per:
            raise RuntimeError(f"gather_rows: {local.shape[0]} local rows > ceil({n_total}/{world}) = {per}")
        send, recv = self._buffers(per, tuple(local.shape[1:]), local.dtype, local.device, world)
        if local.shape[0] == per and local.is_contiguous():
            send = local                                   # full block: gather straight from the caller's tensor
        else:
            send[: local.shape[0]].copy_(local)            # the last rank's short block; the pad rows stay zero
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv[:n_total]


_default_gather: Dict[object, RowGather] = {}


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather equally padded row blocks [per, E] -> [n_total, E] in rank order.  Works on any backend (RCCL on GPU
    tensors, gloo on CPU tensors); a no-op without an initialised group.  Returns a tensor of its own (two consecutive
    gathers of the same shape — video rows, then text rows — do not alias); a steady-state loop that wants the
    allocation-free form holds a ``RowGather`` and calls that (bench.py does)."""
    g = _default_gather.get(group)
    if g is None:
        g = _default_gather[group] = RowGather(group)
    out = g(local, n_total)
    return out.clone() if out.data_ptr() != local.data_ptr() else out


def tie_rank_from_names(names: Sequence[str], device=None) -> torch.Tensor:
    """tie_rank[v] = rank of names[v] in ascending order, so that (score desc, tie_rank desc)
    equals evaluate.py:58-60's sorted(zip(scores, videos))[::-1]."""
    order = sorted(range(len(names)), key=lambda i: names[i])
    t = torch.empty(len(names), dtype=torch.int32)
    t[torch.tensor(order, dtype=torch.long)] = torch.arange(len(names), dtype=torch.int32)
    return t.to(device) if device is not None else t


@torch.no_grad()
def retrieve(text_n: torch.Tensor, video_n: torch.Tensor, k: int, tie_rank: Optional[torch.Tensor] = None):
    """scores = T @ V.T (inference_video_retrieval.py:334) + per-query top-k: (scores, values, indices)."""
    scores = ops.similarity(text_n.contiguous(), video_n.contiguous())
    val, idx = ops.topk(scores, k, tie_rank)
    return scores, val, idx


def recall_at_k(topk_idx: torch.Tensor, names: Sequence[str], gt: Sequence[Sequence[str]],
                ks=(1, 5, 10, 50)) -> Dict[str, float]:
    """evaluate_video_retrieval's 'all' bucket (evaluate.py:62-81) from ranked indices."""
    idx = topk_idx.cpu().tolist()
    out = {}
    for k in ks:
        if k > len(idx[0]):
            continue
        hit = sum(1 for q, row in enumerate(idx) if any(names[v] in set(gt[q]) for v in row[:k]))
        out[f"R@{k}"] = hit / len(idx) * 100
    return out
