"""Moment-task metrics of the reference's evaluate.py with the interval arithmetic on the GPU, same function
names and dict layouts, so a driver holding predictions can score them without the JSON round trip the reference makes
(``run.py`` dumps predictions, ``evaluate.py`` re-reads them):

    evaluate_video_retrieval(gt, pred, prompt_to_cat)       # evaluate.py:33-81    R@1/5/10/50 with the (score, name) tie rule
    evaluate_moment_retrieval(gt, pred, prompt_to_cat)      # evaluate.py:83-121   R@0.5 / R@0.7 per prompt category
    compute_step_bound_scores(gt, pred, video_to_cat)       # evaluate.py:123-188  step recall / precision at tIoU
    preprocess_moment_bounds(gt, pred)                      # evaluate.py:322-412  filter + NMS + gap filling

``gt`` / ``pred`` are the reference's dicts (or JSON paths).  The category maps are arguments (the reference reads them
into module globals in ``__main__``, :444-466).  Intervals are flattened to float64 tensors, the per-pair / per-video
work runs in ``csrc/eval.hip`` in double precision with Python's operation order (identical decisions), and the final
means are taken on the host in the reference's summation order, so results are equal to the last bit.
Tensor-level entry points (`interval_iou`, `step_bound_pr`, `preprocess_bounds`) take device tensors directly.
No CPU fallback.  Caption metrics (CLIPScore / BERTScore / entailment / COCO, :190-320) are out of scope.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops


def _load(x):
    if isinstance(x, str):
        with open(x, "r") as f:
            return json.load(f)
    assert isinstance(x, dict), "data should be a str path or a dict"          # evaluate.py:7-22
    return x


def _dev(device):
    device = torch.device(device if device is not None else "cuda:0")
    if device.type != "cuda":
        raise RuntimeError("hirest_amd.evaluation runs on MI355X only (no CPU fallback)")
    return device


def _categories(cat_map: Dict[str, str]) -> List[str]:
    return sorted(set(cat_map.values())) + ["all"]


# ------------------------------------------------------------------------------------------------ tensor level

def interval_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """compute_iou(a[i], b[i]) for [n,2] float64 CUDA tensors."""
    a, b = a.contiguous(), b.contiguous()
    assert a.dtype == b.dtype == torch.float64 and a.shape == b.shape and a.shape[-1] == 2 and a.is_cuda
    out = torch.empty(a.shape[0], dtype=torch.float64, device=a.device)
    _lib.check(_lib.load().hirest_interval_iou_f64(a.data_ptr(), b.data_ptr(), a.shape[0], out.data_ptr(), ops.stream_ptr()),
               "hirest_interval_iou_f64")
    return out


def step_bound_pr(refs: torch.Tensor, ref_off: torch.Tensor, preds: torch.Tensor, pred_off: torch.Tensor, tiou: float):
    """Ragged per-video step recall / precision; returns (recall[V], precision[V], best_iou[sum preds])."""
    V = ref_off.numel() - 1
    dev = refs.device
    rec = torch.empty(V, dtype=torch.float64, device=dev)
    prc = torch.empty(V, dtype=torch.float64, device=dev)
    best = torch.empty(preds.shape[0], dtype=torch.float64, device=dev)
    _lib.check(_lib.load().hirest_step_bound_pr(refs.data_ptr(), ref_off.data_ptr(), preds.data_ptr(), pred_off.data_ptr(), V,
                                                float(tiou), rec.data_ptr(), prc.data_ptr(), best.data_ptr(), ops.stream_ptr()),
               "hirest_step_bound_pr")
    return rec, prc, best


def preprocess_bounds(preds: torch.Tensor, pred_off: torch.Tensor, gt_minmax: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (bounds [V,max_out,2] float64, count [V] int32)."""
    V = pred_off.numel() - 1
    dev = preds.device
    counts_in = (pred_off[1:] - pred_off[:-1])
    max_in = int(counts_in.max().item()) if V else 0
    if max_in > 128:
        raise ValueError("at most 128 predicted bounds per video")
    max_out = 2 * max_in + 1
    out = torch.zeros((V, max_out, 2), dtype=torch.float64, device=dev)
    cnt = torch.zeros(V, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().hirest_preprocess_moment_bounds(preds.data_ptr(), pred_off.data_ptr(), gt_minmax.data_ptr(), V,
                                                           out.data_ptr(), cnt.data_ptr(), max_out, ops.stream_ptr()),
               "hirest_preprocess_moment_bounds")
    return out, cnt


def _ragged(lists: Sequence[Sequence[Sequence[float]]], device):
    flat, off = [], [0]
    for l in lists:
        flat.extend([float(b[0]), float(b[1])] for b in l)
        off.append(len(flat))
    t = torch.tensor(flat if flat else [[0.0, 0.0]], dtype=torch.float64).reshape(-1, 2)[: len(flat)]
    return t.to(device).contiguous(), torch.tensor(off, dtype=torch.int32, device=device)


# ------------------------------------------------------------------------------------------------ evaluate.py level

def evaluate_video_retrieval(gt_data, pred_data, prompt_to_cat: Optional[Dict[str, str]] = None, device=None,
                             ks: Sequence[int] = (1, 5, 10, 50)) -> dict:
    """evaluate.py:33-81.  ``pred`` is the retrieval script's dict ``{prompt: {"videos": [...], "scores": [...]}}`` (a path, a
    plain dict, or the ``RetrievalResult`` of ``retrieval.run_corpus``, whose device-resident score matrix is then ranked
    in place).  The reference sorts ``zip(scores, videos)`` ascending and reverses: score descending, exact ties broken by
    file name descending; here that order is one top-``max(ks)`` selection per prompt on the device (``hirest_topk_f32`` with
    the name ranks as tie key), and only the ``[Q, max(ks)]`` index table comes back for the membership test.  Prompts whose
    ``videos`` lists differ are ranked in groups sharing a list.  ``prompt_to_cat`` None: only the 'all' bucket."""
    from . import retrieval
    gt, pred = _load(gt_data), _load(pred_data)
    device = _dev(device)
    cat_of = prompt_to_cat or {}
    cats = _categories(cat_of)
    ks = [int(k) for k in ks]
    count = {c: {str(k): 0 for k in ks} for c in cats}
    total = {c: 0 for c in cats}
    prompts = list(gt)
    # group prompts by the identity / content of their video list (the script writes one shared list)
    by_id: Dict[int, Tuple[List[str], List[str]]] = {}
    groups: Dict[tuple, Tuple[List[str], List[str]]] = {}
    for p in prompts:
        vids = pred[p]["videos"]
        g = by_id.get(id(vids))
        if g is None:
            g = by_id[id(vids)] = groups.setdefault(tuple(vids), (list(vids), []))
        g[1].append(p)
    whole = getattr(pred, "scores", None)
    for vids, members in groups.values():
        kmax = min(max(ks), len(vids))
        if whole is not None and vids == getattr(pred, "video_ids", None) and members == getattr(pred, "prompts", None):
            scores = whole.to(device)
        else:
            scores = torch.tensor([pred[p]["scores"] for p in members], dtype=torch.float32, device=device)
        _, idx = ops.topk(scores.contiguous(), kmax, retrieval.tie_rank_from_names(vids, device))
        idx = idx.cpu().tolist()
        for p, row in zip(members, idx):
            gt_videos = set(gt[p].keys()) if isinstance(gt[p], dict) else set(gt[p])     # (the split files hold {video: annotation} dicts)
            buckets = ["all"] + ([cat_of[p]] if prompt_to_cat is not None else [])
            for c in buckets:
                total[c] += 1
            for k in ks:
                if any(vids[v] in gt_videos for v in row[:k]):
                    for c in buckets:
                        count[c][str(k)] += 1
    results = {}
    for c in cats:
        if total[c] > 0:
            results[c] = {"total_prompt_count": total[c]}
            for k in ks:
                results[c][f"R@{k}"] = (count[c][str(k)] / total[c]) * 100
    return results


def evaluate_moment_retrieval(gt_data, pred_data, prompt_to_cat: Dict[str, str], device=None) -> dict:
    gt, pred = _load(gt_data), _load(pred_data)
    device = _dev(device)
    cats = _categories(prompt_to_cat)
    keys = [(p, v) for p in gt for v in gt[p] if gt[p][v]["clip"]]
    score_dict = {c: {} for c in cats}
    if not keys:
        return score_dict
    g = torch.tensor([[float(x) for x in gt[p][v]["bounds"][:2]] for p, v in keys], dtype=torch.float64, device=device)
    q = torch.tensor([[float(x) for x in pred[p][v]["bounds"][:2]] for p, v in keys], dtype=torch.float64, device=device)
    iou = interval_iou(g, q)
    for tiou in (0.5, 0.7):
        hit = (~(iou < tiou)).cpu().tolist()                        # score = 0 if iou < tIoU else 1
        scores = {c: [] for c in cats}
        for (p, _), h in zip(keys, hit):
            scores["all"].append(int(h))
            scores[prompt_to_cat[p]].append(int(h))
        for c in cats:
            if len(scores[c]) > 0:
                score_dict[c]["total_videos"] = len(scores[c])
                score_dict[c][f"R@{tiou}"] = sum(scores[c]) / len(scores[c]) * 100
    return score_dict


def compute_step_bound_scores(gt_data, pred_data, video_to_cat: Dict[str, str], device=None) -> dict:
    gt, pred = _load(gt_data), _load(pred_data)
    device = _dev(device)
    cats = _categories(video_to_cat)
    videos = list(gt)
    results = {c: {"recall": {}, "precision": {}} for c in cats}
    for v in videos:
        if len(pred[v]["bounds"]) == 0 or len(gt[v]["bounds"]) == 0:
            raise ValueError(f"{v}: empty bounds list (the reference divides by the list length)")
    refs, ref_off = _ragged([gt[v]["bounds"] for v in videos], device)
    preds, pred_off = _ragged([pred[v]["bounds"] for v in videos], device)
    for tiou in (0.5, 0.7):
        rec, prc, _ = step_bound_pr(refs, ref_off, preds, pred_off, tiou)
        rec, prc = rec.cpu().tolist(), prc.cpu().tolist()
        recall = {c: [] for c in cats}
        precision = {c: [] for c in cats}
        for v, r, p in zip(videos, rec, prc):
            for c in (video_to_cat[v], "all"):
                recall[c].append(r)
                precision[c].append(p)
        for c in cats:
            if len(recall[c]) > 0:
                results[c]["recall"][f"{tiou}"] = sum(recall[c]) / len(recall[c]) * 100
                results[c]["precision"][f"{tiou}"] = sum(precision[c]) / len(precision[c]) * 100
                results[c]["total"] = len(recall[c])
    return results


def preprocess_moment_bounds(gt_data, pred_data, device=None) -> dict:
    gt, pred = _load(gt_data), _load(pred_data)
    device = _dev(device)
    videos = list(pred)
    preds, pred_off = _ragged([pred[v]["bounds"] for v in videos], device)
    mm = torch.tensor([[float(gt[v]["bounds"][0][0]), float(gt[v]["bounds"][-1][1])] for v in videos], dtype=torch.float64,
                      device=device)
    out, cnt = preprocess_bounds(preds, pred_off, mm)
    out, cnt = out.cpu(), cnt.cpu().tolist()
    res = {}
    for i, v in enumerate(videos):
        res[v] = dict(pred[v])
        res[v]["bounds"] = out[i, : cnt[i]].tolist()
    return res
