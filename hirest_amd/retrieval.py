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

    def __init__(self, video_ids: Sequence[str], frames, videos_per_call: int = 32, min_frames_per_call: int = 256):
        self.video_ids = list(video_ids)
        self.frames = frames
        self.videos_per_call = max(1, int(videos_per_call))
        # a tower call of a few dozen frames loses up to a fifth of its throughput to tile quantisation (bench `tower_small_calls`: 64 frames
        # 1780, 256 frames 1970 frames/s), so short videos are grouped until a call holds at least this many frames (0: exactly videos_per_call)
        self.min_frames_per_call = max(0, int(min_frames_per_call))
        self._frames_per_video = None

    def _per_call(self, n_model_frames: Optional[int], device) -> int:
        """Videos per tower call: videos_per_call, raised so that a call holds >= min_frames_per_call frames."""
        if self.min_frames_per_call <= 0 or not self.video_ids:
            return self.videos_per_call
        F = n_model_frames if (n_model_frames is not None and n_model_frames > 0) else self._frames_per_video
        if F is None:
            F = self._frames_per_video = int(self._block(0, 1, device).shape[1])
        return max(self.videos_per_call, -(-self.min_frames_per_call // max(1, int(F))))

    def _block(self, lo, hi, device):
        blk = self.frames(lo, hi) if callable(self.frames) else self.frames[lo:hi]
        return blk.to(device, non_blocking=True)

    def pooled_rows(self, model, lo: int, hi: int, n_model_frames: Optional[int], device) -> torch.Tensor:
        rows = []
        per = self._per_call(n_model_frames, device)
        for s in range(lo, hi, per):
            blk = self._block(s, min(hi, s + per), device)
            if n_model_frames is not None and n_model_frames > 0 and blk.shape[1] != n_model_frames:
                # VideoFramesDataset.__getitem__ subsamples the decoded frames to n_model_frames (:36-44)
                ids = torch.from_numpy(subsample_ids(blk.shape[1], n_model_frames)).to(device)
                blk = blk.index_select(1, ids)
            rows.append(encode_videos(model, blk))
        return torch.cat(rows) if len(rows) != 1 else rows[0]


def _id_runs(ids: Sequence[int]) -> List[Tuple[int, int]]:
    """Sorted video ids -> maximal runs [(lo, hi), ...] of consecutive ids (so a streaming source is asked for few ranges)."""
    runs: List[Tuple[int, int]] = []
    for v in ids:
        v = int(v)
        if runs and runs[-1][1] == v:
            runs[-1] = (runs[-1][0], v + 1)
        else:
            runs.append((v, v + 1))
    return runs


def _frame_source_rows_of(self, model, ids: Sequence[int], n_model_frames: Optional[int], device) -> torch.Tensor:
    """Pooled rows of an arbitrary (sorted) set of videos, ``videos_per_call`` of them per tower call whatever their positions in the
    corpus — the second pass of the margin-guarded re-rank (``run_corpus(rank_exact_k=...)``) re-encodes scattered videos and must not
    fall back to one small tower call per video."""
    rows = []
    ids = [int(v) for v in ids]
    per = self._per_call(n_model_frames, device)
    for s in range(0, len(ids), per):
        parts = [self._block(lo, hi, device) for lo, hi in _id_runs(ids[s:s + per])]
        blk = torch.cat(parts) if len(parts) != 1 else parts[0]
        if n_model_frames is not None and n_model_frames > 0 and blk.shape[1] != n_model_frames:
            sel = torch.from_numpy(subsample_ids(blk.shape[1], n_model_frames)).to(device)
            blk = blk.index_select(1, sel)
        rows.append(encode_videos(model, blk))
    if not rows:
        E = getattr(model, "embed_dim", None) or model.visual.embed_dim
        return torch.zeros((0, E), dtype=torch.float32, device=device)
    return torch.cat(rows) if len(rows) != 1 else rows[0]


FrameSource.pooled_rows_of = _frame_source_rows_of


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


# ------------------------------------------------------------------------------------------------------------------
# Margin-guarded re-rank (round 6): reference ranks at (nearly) the bf16 towers' speed
# ------------------------------------------------------------------------------------------------------------------
# The bf16 vision tower encodes 2.6x faster than the rank-exact one (bf16x3) but perturbs a score by up to eps ~ 1e-3, which flips the
# order of videos whose reference scores are closer than that.  Only THOSE videos need the precise tower: with every score known to
# within eps, the order of two videos is certain when their fast scores differ by more than 2 eps.  So: encode the corpus fast, find
# per query the videos whose place in the top k is not certain, re-encode the union precisely, and score again.  The top-k lists are
# then the precise tower's (evaluate.py:58-69 ranks by (score, name); R@k only looks at the top k).

def ambiguous_columns(top_val: np.ndarray, top_idx: np.ndarray, k: int, eps: float) -> np.ndarray:
    """Host logic on the per-query sorted score window ``top_val`` / ``top_idx`` [Q, W] (W > k, descending): the ids of the videos whose
    membership or position in some query's top k could change when every score moves by at most ``eps``.

    For a query, cand = window entries with score >= s_k - 2 eps (s_k = the k-th score): anything below cannot reach the top k.  Inside
    cand an entry is certain when both neighbouring gaps exceed 2 eps — its rank among the exact scores is its rank here — and
    ambiguous otherwise.  (A candidate beyond rank k is within 2 eps of the k-th entry by definition, so both are ambiguous.)
    The caller must pass a window that reaches below every query's threshold (``window_covers``)."""
    Q, W = top_val.shape
    if Q == 0 or W == 0:
        return np.zeros((0,), dtype=np.int64)
    k = min(k, W)
    thr = top_val[:, k - 1:k] - 2.0 * eps
    cand = top_val >= thr
    close_next = np.zeros((Q, W), dtype=bool)
    close_next[:, :-1] = (top_val[:, :-1] - top_val[:, 1:]) <= 2.0 * eps
    close_next[:, :-1] &= cand[:, 1:]
    close_prev = np.zeros((Q, W), dtype=bool)
    close_prev[:, 1:] = close_next[:, :-1]
    amb = cand & (close_next | close_prev)
    return np.unique(top_idx[amb].astype(np.int64))


def window_covers(top_val: np.ndarray, k: int, eps: float, n_videos: int) -> bool:
    """True when the sorted window [Q, W] reaches below s_k - 2 eps for every query (or holds the whole corpus)."""
    W = top_val.shape[1]
    if W >= n_videos or top_val.shape[0] == 0:
        return True
    return bool((top_val[:, -1] < top_val[:, min(k, W) - 1] - 2.0 * eps).all())


def _split_ids(ids: Sequence[int], rank: int, world: int) -> List[int]:
    per = (len(ids) + world - 1) // world
    return list(ids[rank * per:(rank + 1) * per])


def _rows_of(model, source, ids: Sequence[int], n_model_frames, device, group, gather) -> torch.Tensor:
    """Pooled rows of the videos ``ids`` (sorted), the work split over the ranks of ``group`` in contiguous chunks and merged with one
    all-gather: [len(ids), E] on every rank, in ``ids`` order."""
    rank, world = _rank_world(group)
    if not hasattr(source, "pooled_rows_of"):
        raise TypeError(f"{type(source).__name__} cannot re-encode single videos (rank_exact_k needs a FrameSource-like source)")
    local = source.pooled_rows_of(model, _split_ids(ids, rank, world), n_model_frames, device)
    if world == 1:
        return local
    return (gather or RowGather(group))(local, len(ids)).clone()


@torch.no_grad()
def rerank_exact(model, source, text_rows: torch.Tensor, video_rows: torch.Tensor, k: int, n_model_frames=None, group=None, device=None,
                 exact_precision: str = "bf16x3", eps: Optional[float] = None, sample: int = 64, safety: float = 2.0,
                 gather: Optional[RowGather] = None) -> Tuple[torch.Tensor, Dict[str, object]]:
    """Second pass of ``run_corpus(rank_exact_k=k)``: ``video_rows`` [V, E] from the fast tower, ``text_rows`` [Q, E] from the exact
    text tower -> (rows with every rank-ambiguous video re-encoded at ``exact_precision``, report).

    ``eps`` (bound of |fast score - exact score|): measured here when None — ``sample`` videos spread over the corpus are encoded at
    both precisions, eps = ``safety`` x the largest score difference over all queries x sampled videos.  The sampled videos are part
    of the re-encoded set, so measuring costs nothing extra.  Identical on every rank (all inputs are replicated)."""
    device = torch.device(device) if device is not None else video_rows.device
    V, Q = video_rows.shape[0], text_rows.shape[0]
    names = list(source.video_ids)
    report: Dict[str, object] = {"k": int(k), "videos": V, "queries": Q, "exact_precision": exact_precision}
    if V == 0 or Q == 0 or k <= 0:
        report.update({"reencoded": 0, "reencoded_fraction": 0.0, "eps": float(eps or 0.0)})
        return video_rows, report
    fast_precision = model.visual.precision
    out = video_rows.clone()
    done = np.zeros((0,), dtype=np.int64)
    try:
        model.visual.precision = exact_precision
        if eps is None:
            sample_ids = np.unique(np.linspace(0, V - 1, min(sample, V)).astype(np.int64))
            exact_s = _rows_of(model, source, sample_ids.tolist(), n_model_frames, device, group, gather)
            sel = torch.from_numpy(sample_ids).to(device)
            diff = ops.similarity(text_rows.contiguous(), video_rows.index_select(0, sel).contiguous()) - \
                ops.similarity(text_rows.contiguous(), exact_s.contiguous())
            measured = float(diff.abs().max().item())
            eps = safety * measured
            out.index_copy_(0, sel, exact_s)
            done = sample_ids
            report.update({"eps_measured_on": int(sample_ids.size), "max_abs_score_error_on_sample": measured, "safety": safety})
        tie = tie_rank_from_names(names, device)
        scores = ops.similarity(text_rows.contiguous(), video_rows.contiguous())
        W = min(V, max(2 * k, k + 32))
        while True:
            val, idx = ops.topk(scores, W, tie)
            val_h, idx_h = val.cpu().numpy(), idx.cpu().numpy()
            if window_covers(val_h, k, eps, V):
                break
            W = min(V, 2 * W)
        amb = ambiguous_columns(val_h, idx_h, k, eps)
        todo = np.setdiff1d(amb, done)
        if todo.size:
            rows = _rows_of(model, source, todo.tolist(), n_model_frames, device, group, gather)
            out.index_copy_(0, torch.from_numpy(todo).to(device), rows)
        n_re = int(np.union1d(amb, done).size)
        report.update({"eps": float(eps), "ambiguous": int(amb.size), "reencoded": n_re, "reencoded_fraction": n_re / V, "window": int(W)})
    finally:
        model.visual.precision = fast_precision
    return out, report


@torch.no_grad()
def run_corpus(model, source, prompts: Sequence[str], n_model_frames: Optional[int] = None, group=None,
               device=None, gather: Optional[RowGather] = None, tokenizer=None, rank_exact_k: int = 0,
               rank_exact_options: Optional[Dict[str, object]] = None) -> RetrievalResult:
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
    the folded path's (tools/c3_run.py compares digests only for such shard sizes).

    ``rank_exact_k`` > 0 (round 6) turns on the margin-guarded re-rank: see ``rerank_exact``; the result carries ``rank_exact`` (eps,
    how many videos were encoded twice)."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    rank, world = _rank_world(group)
    local = corpus_block_rows(model, source, rank, world, n_model_frames, device)
    V = len(source.video_ids)
    if world > 1:
        video_rows = (gather or RowGather(group))(local, V)
    else:
        video_rows = local
    if rank_exact_k <= 0:
        text_rows = encode_prompts(model, prompts, device, tokenizer=tokenizer)
        return score_corpus(text_rows, video_rows, source.video_ids, prompts)
    # rank_exact_k = k: the top-k lists of the PRECISE towers at close to the fast tower's speed — the text tower runs exact (2 % of the work),
    # the corpus fast, and only the videos whose place in some query's top k is within twice the measured score error are encoded again
    # precisely (rerank_exact).  Every rank computes the same set and re-encodes its share of it; one more all-gather merges the rows.
    text_precision = model.text.precision
    try:
        model.text.precision = "fp32"
        text_rows = encode_prompts(model, prompts, device, tokenizer=tokenizer)
    finally:
        model.text.precision = text_precision
    if world > 1:
        video_rows = video_rows.clone()                    # (the gather buffer is reused by the second pass)
    rows, report = rerank_exact(model, source, text_rows, video_rows, rank_exact_k, n_model_frames, group, device,
                                gather=gather, **(rank_exact_options or {}))
    res = score_corpus(text_rows, rows, source.video_ids, prompts)
    res.rank_exact = report
    return res


def corpus_digest(video_rows: torch.Tensor, topk_idx: torch.Tensor) -> Dict[str, str]:
    """SHA-256 of the pooled [V, E] fp32 rows and of the int32 top-k table: what an N-rank run is compared with (the
    committed 1-rank digests: tests/golden/c3_rank_blocks.json) to show 1 GPU == N GPUs bit for bit."""
    import hashlib
    return {"pooled_sha256": hashlib.sha256(video_rows.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest(),
            "top10_sha256": hashlib.sha256(topk_idx.detach().cpu().to(torch.int32).contiguous().numpy().tobytes()).hexdigest()}
