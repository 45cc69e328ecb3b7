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
        # gloo has no all-gather on device tensors: stage through the host (functional runs of the multi-rank path on a box without
        # RCCL peers, e.g. two ranks sharing one GPU in tests/test_run_corpus.py; production is "nccl" = RCCL, device to device)
        staged = local.is_cuda and dist.get_backend(self.group) == "gloo"
        dev = torch.device("cpu") if staged else local.device
        send, recv = self._buffers(per, tuple(local.shape[1:]), local.dtype, dev, world)
        if local.shape[0] == per and local.is_contiguous() and not staged:
            send = local                                   # full block: gather straight from the caller's tensor
        else:
            send[: local.shape[0]].copy_(local)            # the last rank's short block; the pad rows stay zero
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv[:n_total].to(local.device) if staged else recv[:n_total]


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


# ------------------------------------------------------------------------------------------------------------------
# The whole corpus run (BASELINE configs[2]): sources -> shard -> encode / pool -> gather -> score -> the reference's dict
# ------------------------------------------------------------------------------------------------------------------

class FrameSource:
    """A corpus of decoded / preprocessed frames (the ``--raw_frame`` branch, inference_video_retrieval.py:216-288).

    ``video_ids``: names in corpus order.  ``frames``: either one tensor ``[V, F, 3, S, S]`` (float, normalised) /
    ``[V, F, H, W, 3]`` (uint8) or a callable ``frames(lo, hi) -> tensor`` of the videos ``[lo, hi)`` — the latter is how a
    corpus larger than memory (4096 x 32 frames = 39 GB in bf16) is streamed: ``videos_per_call`` videos are requested,
    encoded in one tower call and pooled before the next block is produced."""

    def __init__(self, video_ids: Sequence[str], frames, videos_per_call: int = 32):
        self.video_ids = list(video_ids)
        self.frames = frames
        self.videos_per_call = max(1, int(videos_per_call))

    def _block(self, lo, hi, device):
        blk = self.frames(lo, hi) if callable(self.frames) else self.frames[lo:hi]
        return blk.to(device, non_blocking=True)

    def pooled_rows(self, model, lo: int, hi: int, n_model_frames: Optional[int], device) -> torch.Tensor:
        rows = []
        for s in range(lo, hi, self.videos_per_call):
            blk = self._block(s, min(hi, s + self.videos_per_call), device)
            if n_model_frames is not None and n_model_frames > 0 and blk.shape[1] != n_model_frames:
                # VideoFramesDataset.__getitem__ subsamples the decoded frames to n_model_frames (:36-44)
                ids = torch.from_numpy(subsample_ids(blk.shape[1], n_model_frames)).to(device)
                blk = blk.index_select(1, ids)
            rows.append(encode_videos(model, blk))
        return torch.cat(rows) if len(rows) != 1 else rows[0]


class FeatureFileSource:
    """A corpus of per-video feature files ``<feature_dir>/<video_id>.pt`` (``[T, E]``; the default branch,
    inference_video_retrieval.py:290-329, and what ``features.FeatureWriter`` / ``extract_features.py`` write): each file is
    loaded on the host, fitted with ``np.linspace(0, T-1, F).astype(int)`` (:315) and pooled on the device — mean over the
    kept rows, then L2 (:322-328; no per-frame normalisation here: the files already hold whichever convention the
    extractor chose).  With ``n_model_frames <= 0`` the reference keeps all T rows (:311), so videos of a call
    are grouped by T before pooling."""

    def __init__(self, feature_dir, video_ids: Sequence[str], videos_per_call: int = 256):
        self.feature_dir = str(feature_dir)
        self.video_ids = list(video_ids)
        self.videos_per_call = max(1, int(videos_per_call))

    def _load(self, vid, n_model_frames):
        import os
        feats = torch.load(os.path.join(self.feature_dir, f"{vid}.pt"), map_location="cpu")
        if n_model_frames is not None and n_model_frames > 0:
            feats = feats[torch.from_numpy(subsample_ids(feats.shape[0], n_model_frames))]
        return feats.float()

    def pooled_rows(self, model, lo: int, hi: int, n_model_frames: Optional[int], device) -> torch.Tensor:
        out = None
        for s in range(lo, hi, self.videos_per_call):
            e = min(hi, s + self.videos_per_call)
            feats = [self._load(self.video_ids[v], n_model_frames) for v in range(s, e)]
            by_len: Dict[int, List[int]] = {}
            for i, f in enumerate(feats):
                by_len.setdefault(f.shape[0], []).append(i)
            for T, members in by_len.items():
                stack = torch.stack([feats[i] for i in members]).to(device)
                rows = ops.pool_l2norm(stack.contiguous())
                if out is None:
                    out = torch.empty((hi - lo, rows.shape[1]), dtype=torch.float32, device=rows.device)
                out[torch.tensor([s - lo + i for i in members], device=rows.device)] = rows
        return out


class RetrievalResult(dict):
    """``{prompt: {"videos": [...], "scores": [...]}}`` exactly as inference_video_retrieval.py:337-355 dumps it (every
    prompt lists the whole corpus in corpus order with its row of ``T @ V.T``), plus the tensors it was built from so a
    caller can rank or evaluate on the device without the JSON round trip: ``scores`` [Q, V] fp32, ``video_rows`` [V, E],
    ``text_rows`` [Q, E], ``video_ids``, ``prompts``."""

    scores: torch.Tensor
    video_rows: torch.Tensor
    text_rows: torch.Tensor
    video_ids: List[str]
    prompts: List[str]

    def topk(self, k: int = 10):
        """(values, indices) under evaluate.py:58-60's order: score descending, ties by file name descending."""
        k = min(k, len(self.video_ids))
        return ops.topk(self.scores, k, tie_rank_from_names(self.video_ids, self.scores.device))

    def save(self, run_name: str, save_dir: str = "VR_results") -> str:
        """``VR_results/<run_name>.json`` with indent 4 (inference_video_retrieval.py:348-355)."""
        import json, os
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, f"{run_name}.json")
        with open(path, "w") as f:
            json.dump(dict(self), f, indent=4)
        return path


def _rank_world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


@torch.no_grad()
def corpus_block_rows(model, source, rank: int, world: int, n_model_frames: Optional[int] = None, device=None) -> torch.Tensor:
    """Rank `rank` of `world`'s share of the corpus: pooled, L2-normalised rows of the videos ``shard_range`` assigns it
    (``[hi - lo, E]`` fp32 on the device; an empty block when the corpus is shorter than the rank count)."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    lo, hi, _ = shard_range(len(source.video_ids), rank, world)
    if hi <= lo:
        E = getattr(model, "embed_dim", None) or model.visual.embed_dim
        return torch.zeros((0, E), dtype=torch.float32, device=device)
    return source.pooled_rows(model, lo, hi, n_model_frames, device)


@torch.no_grad()
def encode_prompts(model, prompts: Sequence[str], device, batch_size: int = 1024, tokenizer=None) -> torch.Tensor:
    """inference_video_retrieval.py:203-214: tokenise + encode_text + L2 in batches -> [Q, E] fp32."""
    if tokenizer is None:
        from .tokenizer import tokenize as tokenizer
    if len(prompts) == 0:
        return torch.zeros((0, int(getattr(model, "embed_dim", 0) or 0)), dtype=torch.float32, device=device)
    rows = [encode_texts(model, tokenizer(list(prompts[s:s + batch_size])).to(device)) for s in range(0, len(prompts), batch_size)]
    return torch.cat(rows) if len(rows) != 1 else rows[0]


def score_corpus(text_rows: torch.Tensor, video_rows: torch.Tensor, video_ids: Sequence[str], prompts: Sequence[str]) -> RetrievalResult:
    """``text_to_video_scores = T @ V.T`` (:334) and the output dict (:337-346)."""
    if text_rows.shape[0] == 0 or video_rows.shape[0] == 0:           # no prompts / an empty corpus: the script's loops simply do not run
        scores = torch.zeros((text_rows.shape[0], video_rows.shape[0]), dtype=torch.float32, device=text_rows.device)
    else:
        scores = ops.similarity(text_rows.contiguous(), video_rows.contiguous())
    host = scores.cpu().tolist()                                  # ONE device -> host copy for the whole matrix
    ids = list(video_ids)
    res = RetrievalResult((p, {"videos": ids, "scores": host[i]}) for i, p in enumerate(prompts))
    res.scores, res.video_rows, res.text_rows, res.video_ids, res.prompts = scores, video_rows, text_rows, ids, list(prompts)
    return res


@torch.no_grad()
def run_corpus(model, source, prompts: Sequence[str], n_model_frames: Optional[int] = None, group=None,
               device=None, gather: Optional[RowGather] = None, tokenizer=None) -> RetrievalResult:
    """BASELINE configs[2] in one call: the body of inference_video_retrieval.py:203-355 for one rank of N.

    Every rank encodes the contiguous block of videos ``shard_range`` gives it (``source``: a ``FrameSource`` or a
    ``FeatureFileSource``; ``n_model_frames`` = ``args.n_model_frames``: the linspace subsample of :315 / :39), ONE
    ``all_gather_into_tensor`` of the padded ``[ceil(V/N), E]`` blocks (RCCL over xGMI on GPU tensors; gloo on CPU tensors)
    assembles the ``[V, E]`` matrix in corpus order on every rank, and text encoding + scoring are replicated (546 x 4096 x
    1024 is 4.6 GFLOP).  Without an initialised process group it is the single-process run.  Returns the reference's
    ``{prompt: {"videos", "scores"}}`` dict (``RetrievalResult``); identical on every rank.  Every video's row depends on that
    video alone, so the scores agree for every N to the kernels' tolerance; they are BIT-identical across N when every rank's
    block is a whole number of ``videos_per_call`` groups of >= 64 frames (4096 videos on 1 / 2 / 4 / 8 ranks at 32 per call):
    a remainder call of fewer than 64 frames takes the unfolded-LayerNorm / per-head attention kernels, whose bits differ from
    the folded path's (tools/c3_run.py compares digests only for such shard sizes)."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    rank, world = _rank_world(group)
    local = corpus_block_rows(model, source, rank, world, n_model_frames, device)
    V = len(source.video_ids)
    if world > 1:
        video_rows = (gather or RowGather(group))(local, V)
    else:
        video_rows = local
    text_rows = encode_prompts(model, prompts, device, tokenizer=tokenizer)
    return score_corpus(text_rows, video_rows, source.video_ids, prompts)


def corpus_digest(video_rows: torch.Tensor, topk_idx: torch.Tensor) -> Dict[str, str]:
    """SHA-256 of the pooled [V, E] fp32 rows and of the int32 top-k table: what an N-rank run is compared with (the
    committed 1-rank digests: tests/golden/c3_rank_blocks.json) to show 1 GPU == N GPUs bit for bit."""
    import hashlib
    return {"pooled_sha256": hashlib.sha256(video_rows.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest(),
            "top10_sha256": hashlib.sha256(topk_idx.detach().cpu().to(torch.int32).contiguous().numpy().tobytes()).hexdigest()}
