"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Plain-Python restatement of the moment-task metrics of the reference's ``evaluate.py`` (paths relative to
/root/reference): ``compute_iou`` (:24-31), ``evaluate_video_retrieval`` (:33-81), ``evaluate_moment_retrieval`` (:83-121), ``compute_step_bound_scores``
(:123-188), ``NMS`` (:322-356) and ``preprocess_moment_bounds`` (:358-412).  Category maps are passed in instead of
read from module globals (the reference sets PROMPT_TO_CAT / VIDEOS_TO_CAT in ``__main__``, :444-466).

PARITY PIN: ``tests/test_evaluation.py`` checks every function against ``tests/golden/moment_eval.json``, produced by
importing and running the real ``evaluate.py`` functions on the same seeded inputs (``tests/golden/make_golden.py
moment_eval``); ``evaluate_video_retrieval`` against ``tests/golden/retrieval_run.json`` — the real function's result on
the JSON the real ``inference_video_retrieval.py`` wrote (``make_golden.py retrieval_run``) — in ``tests/test_run_corpus.py``.
"""
from __future__ import annotations

from typing import Dict, List


def compute_iou(interval_1, interval_2) -> float:
    start_i, end_i = interval_1[0], interval_1[1]
    start, end = interval_2[0], interval_2[1]
    intersection = max(0, min(end, end_i) - max(start, start_i))
    union = min(max(end, end_i) - min(start, start_i), end - start + end_i - start_i)
    return float(intersection) / (union + 1e-8)


def _categories(cat_map: Dict[str, str]) -> List[str]:
    return sorted(set(cat_map.values())) + ["all"]


def evaluate_video_retrieval(gt: dict, pred: dict, prompt_to_cat: Dict[str, str], ks=(1, 5, 10, 50)) -> dict:
    """evaluate.py:33-81: per prompt, sorted(zip(scores, videos)) reversed (score descending, ties by name descending);
    a prompt counts for R@k when any of its ground-truth videos is among the first k."""
    cats = _categories(prompt_to_cat)
    count = {c: {f"{k}": 0 for k in ks} for c in cats}
    total = {c: 0 for c in cats}
    for prompt in gt:
        prompt_cat = prompt_to_cat[prompt]
        gt_videos = list(gt[prompt].keys()) if isinstance(gt[prompt], dict) else list(gt[prompt])
        total["all"] += 1
        total[prompt_cat] += 1
        scores, videos = zip(*sorted(zip(pred[prompt]["scores"], pred[prompt]["videos"])))
        videos = videos[::-1]
        for k in ks:
            for v in videos[:k]:
                if v in gt_videos:
                    count["all"][f"{k}"] += 1
                    count[prompt_cat][f"{k}"] += 1
                    break
    results = {}
    for c in cats:
        if total[c] > 0:
            results[c] = {"total_prompt_count": total[c]}
            for k in ks:
                results[c][f"R@{k}"] = (count[c][f"{k}"] / total[c]) * 100
    return results


def evaluate_moment_retrieval(gt: dict, pred: dict, prompt_to_cat: Dict[str, str]) -> dict:
    cats = _categories(prompt_to_cat)
    score_dict = {c: {} for c in cats}
    for tiou in (0.5, 0.7):
        scores = {c: [] for c in cats}
        for prompt in gt:
            for video in gt[prompt]:
                if gt[prompt][video]["clip"]:
                    iou = compute_iou(gt[prompt][video]["bounds"], pred[prompt][video]["bounds"])
                    s = 0 if iou < tiou else 1
                    scores["all"].append(s)
                    scores[prompt_to_cat[prompt]].append(s)
        for c in cats:
            if len(scores[c]) > 0:
                score_dict[c]["total_videos"] = len(scores[c])
                score_dict[c][f"R@{tiou}"] = sum(scores[c]) / len(scores[c]) * 100
    return score_dict


def compute_step_bound_scores(gt: dict, pred: dict, video_to_cat: Dict[str, str]) -> dict:
    cats = _categories(video_to_cat)
    results = {c: {"recall": {}, "precision": {}} for c in cats}
    for tiou in (0.5, 0.7):
        recall = {c: [] for c in cats}
        precision = {c: [] for c in cats}
        for video in gt:
            refs, preds = gt[video]["bounds"], pred[video]["bounds"]
            ref_cov, pred_cov = set(), set()
            for pi, p in enumerate(preds):
                for ri, r in enumerate(refs):
                    if compute_iou(p, r) > tiou:
                        ref_cov.add(ri)
                        pred_cov.add(pi)
            pr = float(len(pred_cov)) / len(preds)
            rc = float(len(ref_cov)) / len(refs)
            for c in (video_to_cat[video], "all"):
                recall[c].append(rc)
                precision[c].append(pr)
        for c in cats:
            if len(recall[c]) > 0:
                results[c]["recall"][f"{tiou}"] = sum(recall[c]) / len(recall[c]) * 100
                results[c]["precision"][f"{tiou}"] = sum(precision[c]) / len(precision[c]) * 100
                results[c]["total"] = len(recall[c])
    return results


def nms_intervals(bounds: List[List[float]]) -> List[List[float]]:
    """NMS(boxes = [x1, 0, x2, 1], overlapThresh = 0): candidates are visited from the last index down (np.argsort of
    the constant y2 column is the identity); a box is suppressed when it overlaps the picked one by more than 0."""
    idxs = list(range(len(bounds)))
    pick = []
    while idxs:
        i = idxs[-1]
        pick.append(i)
        keep = []
        for j in idxs[:-1]:
            xx1 = max(float(bounds[i][0]), float(bounds[j][0]))
            xx2 = min(float(bounds[i][1]), float(bounds[j][1]))
            w = max(0.0, xx2 - xx1 + 1)
            h = max(0.0, 1.0 - 0.0 + 1)
            area = (float(bounds[j][1]) - float(bounds[j][0]) + 1) * (1.0 - 0.0 + 1)
            try:
                overlap = (w * h) / area
            except ZeroDivisionError:           # numpy: inf (suppressed) when w > 0, nan (kept) when w == 0
                overlap = float("inf") if w * h > 0 else float("nan")
            if not overlap > 0:
                keep.append(j)
        idxs = keep
    return [[float(bounds[i][0]), float(bounds[i][1])] for i in pick]


def preprocess_moment_bounds(gt: dict, pred: dict) -> dict:
    out = {}
    for video in pred:
        gt_bounds = gt[video]["bounds"]
        min_x, max_x = gt_bounds[0][0], gt_bounds[-1][1]
        bounds = [b for b in pred[video]["bounds"] if (b[0] > min_x and b[1] < max_x)]
        boxes = nms_intervals(bounds)
        if len(boxes) > 0:
            boxes.sort(key=lambda x: x[0])
            new_bounds = []
            if boxes[0][0] > min_x:
                new_bounds.append([min_x, boxes[0][0]])
            for i in range(len(boxes)):
                new_bounds.append(boxes[i])
                if i + 1 < len(boxes):
                    new_bounds.append([boxes[i][1], boxes[i + 1][0]])
            if new_bounds[-1][1] < max_x:
                new_bounds.append([new_bounds[-1][1], max_x])
        else:
            new_bounds = [[min_x, max_x]]
        out[video] = dict(pred[video])
        out[video]["bounds"] = new_bounds
    return out
