"""Frame index <-> timestamp conversion of the reference (``hirest_dataset.py:12-68``) and the result
dicts ``run.py`` builds from the joint model's predictions (``run.py:719-790``), so predictions flow
``MomentModel.test_step`` -> timestamps -> ``hirest_amd.evaluation`` without leaving the process.

    timestamp_to_frame_index(timestamp, video_duration, n_frames=32)     # hirest_dataset.py:12-40, same name / arguments
    frame_index_to_timestamp(frame_index, video_duration, n_frames=32)   # hirest_dataset.py:42-68
    frames_to_timestamps(frames, durations, n_frames)                    # batched, on device (csrc/eval.hip)
    timestamps_to_frames(timestamps, durations, n_frames)
    moment_retrieval_results(...) / moment_segmentation_results(...)     # run.py:719-745 / :757-785 dict layouts

The reference materialises ``np.linspace(0, int(duration) - 1, n)`` for every single conversion; here a bin value is one
multiply (numpy builds linspace as ``arange(n) * step`` with the last element overwritten by ``stop``, which is restated
exactly, in double), on the host for scalars and in a HIP kernel for batches.  Integer results: bit-exact.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import _lib, ops

_BAD = -(1 << 63)


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("hirest_amd.timeline batch conversion needs an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _bins(video_duration, n_frames):
    d = int(video_duration)
    n = d if n_frames < 0 else int(n_frames)
    stop = float(d - 1)
    step = stop / (n - 1) if n > 1 else 0.0
    return d, n, stop, step


def _bin_value(i: int, n: int, stop: float, step: float) -> float:
    return stop if (i == n - 1 and n > 1) else (i * step if n > 1 else 0.0)


def frame_index_to_timestamp(frame_index, video_duration, n_frames=32) -> int:
    """``int(np.linspace(0, int(video_duration) - 1, n_frames)[frame_index])`` (hirest_dataset.py:42-68); ``n_frames < 0``
    means one frame per second.  Raises IndexError where numpy's indexing does."""
    d, n, stop, step = _bins(video_duration, n_frames)
    i = int(frame_index)
    if i < 0:
        i += n
    if i < 0 or i >= n:
        raise IndexError(f"index {frame_index} is out of bounds for axis 0 with size {n}")
    return int(_bin_value(i, n, stop, step))


def timestamp_to_frame_index(timestamp, video_duration, n_frames=32) -> int:
    """``min(np.digitize(timestamp, bins, right=True), n_frames - 1)`` (hirest_dataset.py:12-40)."""
    d, n, stop, step = _bins(video_duration, n_frames)
    if d < 1:
        raise ValueError("videos shorter than one second have no usable bins (the reference's linspace is empty or decreasing)")
    x = float(timestamp)
    if x != x:
        return n - 1
    k = 0
    if step > 0.0 and x > 0.0:
        g = math.ceil(x / step)
        k = n if g >= n else int(g)
    while k > 0 and not (_bin_value(k - 1, n, stop, step) < x):
        k -= 1
    while k < n and _bin_value(k, n, stop, step) < x:
        k += 1
    return min(k, n - 1)


# ------------------------------------------------------------------------------------------------ batched, on device

def _prep(values: torch.Tensor, durations, n_frames, dtype):
    if not values.is_cuda:
        raise RuntimeError("hirest_amd.timeline batch conversion runs on MI355X only (no CPU fallback); "
                           "use the scalar functions on the host")
    dev = values.device
    v = values.to(dtype).contiguous()
    V = v.shape[0] if v.dim() > 0 else 1
    per_video = max(v.numel() // max(V, 1), 1)
    dur = torch.as_tensor(durations, dtype=torch.float64, device=dev).reshape(-1).contiguous()
    if dur.numel() != V:
        raise ValueError(f"{V} rows of values but {dur.numel()} durations")
    if isinstance(n_frames, int):
        nf, nf_all = None, n_frames
    else:
        nf = torch.as_tensor(n_frames, dtype=torch.int32, device=dev).reshape(-1).contiguous()
        nf_all = 0
        if nf.numel() != V:
            raise ValueError(f"{V} rows of values but {nf.numel()} n_frames")
    return v, dur, nf, nf_all, per_video


def frames_to_timestamps(frames: torch.Tensor, durations, n_frames: Union[int, Sequence[int], torch.Tensor] = 32,
                         strict: bool = True) -> torch.Tensor:
    """frames int64 [V] or [V,k] (CUDA) -> int64 timestamps of the same shape; row v uses durations[v] (and n_frames[v]).
    ``strict``: raise IndexError if any index is outside its video's bins (the reference's behaviour); otherwise those
    entries are ``-2**63``."""
    v, dur, nf, nf_all, per_video = _prep(frames, durations, n_frames, torch.int64)
    out = torch.empty_like(v)
    _lib.check(_lib.load().hirest_frame_to_timestamp(v.data_ptr(), dur.data_ptr(), nf.data_ptr() if nf is not None else None,
                                                     nf_all, per_video, v.numel(), out.data_ptr(), ops.stream_ptr()),
               "hirest_frame_to_timestamp")
    if strict and v.numel() and bool((out == _BAD).any()):
        raise IndexError("frame index outside its video's bins")
    return out


def timestamps_to_frames(timestamps: torch.Tensor, durations, n_frames: Union[int, Sequence[int], torch.Tensor] = 32) -> torch.Tensor:
    """timestamps (seconds, any float/int dtype) [V] or [V,k] (CUDA) -> int64 frame indices of the same shape."""
    v, dur, nf, nf_all, per_video = _prep(timestamps, durations, n_frames, torch.float64)
    out = torch.empty(v.shape, dtype=torch.int64, device=v.device)
    _lib.check(_lib.load().hirest_timestamp_to_frame(v.data_ptr(), dur.data_ptr(), nf.data_ptr() if nf is not None else None,
                                                     nf_all, per_video, v.numel(), out.data_ptr(), ops.stream_ptr()),
               "hirest_timestamp_to_frame")
    if v.numel() and bool((out == _BAD).any()):
        raise ValueError("a video shorter than one second has no usable bins")
    return out


# ------------------------------------------------------------------------------------------------ run.py result dicts

def moment_retrieval_results(predictions, prompts: Sequence[str], video_fnames: Sequence[str], video_durations: Sequence[float],
                             n_model_frames: int, targets: Optional[Sequence] = None) -> Dict:
    """run.py:719-745: ``{prompt: {video: {'bounds': [start_s, end_s], 'video_duration': d[, 'target_bounds': t]}}}`` from
    ``test_step``'s ``[B,2]`` frame-index predictions (tensor on the GPU, or a list)."""
    pred = torch.as_tensor(predictions)
    if pred.dim() != 2 or pred.shape[1] != 2:
        raise AssertionError("moment retrieval predictions are [start_frame, end_frame] pairs")      # run.py:730
    if not pred.is_cuda:
        pred = pred.to(_default_device())
    ts = frames_to_timestamps(pred, list(video_durations), n_model_frames).tolist()
    out: Dict = {}
    for i, (prompt, video) in enumerate(zip(prompts, video_fnames)):
        entry = out.setdefault(prompt, {}).setdefault(video, {})
        entry["bounds"] = ts[i]
        entry["video_duration"] = video_durations[i]
        if targets is not None:
            entry["target_bounds"] = targets[i]
    return out


def moment_segmentation_results(predictions: Sequence[Sequence[int]], video_fnames: Sequence[str],
                                video_durations: Sequence[float], n_model_frames: int, targets: Optional[Sequence] = None,
                                device=None) -> Dict:
    """run.py:757-785: ``{video: {'bounds': [[s0,s1],[s1,s2],...], 'video_duration', 'pred_bounds', 'target_bounds'}}``
    from ``test_step``'s ragged boundary lists.  A boundary outside its video's bins leaves that pair incomplete, as the
    reference's ``try/except`` does (:767-772)."""
    dev = torch.device(device) if device is not None else _default_device()
    flat = [int(b) for p in predictions for b in p]
    durs = [float(video_durations[i]) for i, p in enumerate(predictions) for _ in p]
    ts: List[int] = []
    if flat:
        ts = frames_to_timestamps(torch.tensor(flat, dtype=torch.int64, device=dev), durs, n_model_frames, strict=False).tolist()
    out: Dict = {}
    pos = 0
    for i, p in enumerate(predictions):
        mine = ts[pos:pos + len(p)]
        pos += len(p)
        bounds = []
        for j in range(len(p) - 1):
            bound = []
            if mine[j] != _BAD:
                bound.append(mine[j])
                if mine[j + 1] != _BAD:
                    bound.append(mine[j + 1])
            bounds.append(bound)
        entry = out.setdefault(video_fnames[i], {})
        entry["bounds"] = bounds
        entry["video_duration"] = video_durations[i]
        entry["pred_bounds"] = list(p)
        entry["target_bounds"] = targets[i] if targets is not None else None
    return out
