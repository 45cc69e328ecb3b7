"""Moment-task evaluation: oracle/eval_cpu.py and the GPU path hirest_amd/evaluation.py against golden
results of the real evaluate.py (tests/golden/moment_eval.json; inputs are regenerated from the same seeded generator)."""
import copy
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(__file__)
from hirest_amd.synth import moment_eval_inputs
from oracle import eval_cpu as E  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "moment_eval.json")))
INP = moment_eval_inputs()


def _same(a, b):
    """Nested dicts / lists of numbers equal to the last bit."""
    if isinstance(a, dict):
        assert set(a) == set(b), (set(a), set(b))
        for k in a:
            _same(a[k], b[k])
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    else:
        assert float(a) == float(b), (a, b)


def test_oracle_iou_samples():
    for s in GOLD["iou_samples"]:
        assert E.compute_iou(s["a"], s["b"]) == s["iou"]


def test_oracle_moment_retrieval_and_step_bounds():
    _same(E.evaluate_moment_retrieval(INP["mr_gt"], INP["mr_pred"], INP["prompt_to_cat"]), GOLD["moment_retrieval"])
    _same(E.compute_step_bound_scores(INP["sb_gt"], INP["sb_pred"], INP["video_to_cat"]), GOLD["step_bounds_raw"])
    pre = E.preprocess_moment_bounds(INP["sb_gt"], copy.deepcopy(INP["sb_pred"]))
    _same({v: pre[v]["bounds"] for v in pre}, GOLD["preprocessed"])
    _same(E.compute_step_bound_scores(INP["sb_gt"], pre, INP["video_to_cat"]), GOLD["step_bounds_preprocessed"])


def test_no_cpu_fallback():
    from hirest_amd import evaluation
    with pytest.raises(RuntimeError):
        evaluation.evaluate_moment_retrieval(INP["mr_gt"], INP["mr_pred"], INP["prompt_to_cat"], device="cpu")


@pytest.mark.gpu
def test_gpu_evaluation_matches_reference_bit_for_bit():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from hirest_amd import evaluation
    dev = torch.device("cuda:0")
    a = torch.tensor([s["a"] for s in GOLD["iou_samples"]], dtype=torch.float64, device=dev)
    b = torch.tensor([s["b"] for s in GOLD["iou_samples"]], dtype=torch.float64, device=dev)
    assert evaluation.interval_iou(a, b).cpu().tolist() == [s["iou"] for s in GOLD["iou_samples"]]
    _same(evaluation.evaluate_moment_retrieval(INP["mr_gt"], INP["mr_pred"], INP["prompt_to_cat"]), GOLD["moment_retrieval"])
    _same(evaluation.compute_step_bound_scores(INP["sb_gt"], INP["sb_pred"], INP["video_to_cat"]), GOLD["step_bounds_raw"])
    pre = evaluation.preprocess_moment_bounds(INP["sb_gt"], copy.deepcopy(INP["sb_pred"]))
    _same({v: pre[v]["bounds"] for v in pre}, GOLD["preprocessed"])
    _same(evaluation.compute_step_bound_scores(INP["sb_gt"], pre, INP["video_to_cat"]), GOLD["step_bounds_preprocessed"])


@pytest.mark.gpu
def test_gpu_evaluation_edge_cases_vs_oracle():
    """No surviving prediction, a single prediction, nested / identical boxes, float bounds, zero-length intervals."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from hirest_amd import evaluation
    gt = {"a": {"bounds": [[10, 20], [20, 35], [35, 50]]}, "b": {"bounds": [[0, 8], [8, 9]]}, "c": {"bounds": [[5, 6]]},
          "d": {"bounds": [[0, 100]]}}
    pred = {"a": {"bounds": [[5, 12], [12, 18], [12, 18], [13, 17], [30, 49.5], [49, 50]]},      # first/last touch the limits
            "b": {"bounds": [[1, 2]]}, "c": {"bounds": [[5, 6], [5.5, 5.75]]},
            "d": {"bounds": [[10, 10], [20, 30], [30, 40], [39, 41], [50.25, 60.5], [1, 99]]}}
    cat = {k: "x" for k in gt}
    pre = evaluation.preprocess_moment_bounds(gt, copy.deepcopy(pred))
    ref = E.preprocess_moment_bounds(gt, copy.deepcopy(pred))
    _same({v: pre[v]["bounds"] for v in pre}, {v: ref[v]["bounds"] for v in ref})
    _same(evaluation.compute_step_bound_scores(gt, pre, cat), E.compute_step_bound_scores(gt, ref, cat))
    _same(evaluation.compute_step_bound_scores(gt, pred, cat), E.compute_step_bound_scores(gt, pred, cat))
