"""Frame-feature files and the frame-count rules around them: what lets the MI355X encoder regenerate
``data/eva_clip_features*/<video>.pt`` and feed the reference's unmodified ``run.py`` / retrieval driver.

Reference conventions (both exist, SURVEY hazard H2):
  * extraction/video_features/extract_features.py:56-69 — one file per video, ``[T,1024]`` fp32, every frame embedding
    **L2-normalised** (``video_features /= video_features.norm(dim=-1, keepdim=True)``), ``torch.save``.
  * inference_video_retrieval.py:275-280 (``--save_feats``) — ``[F,1024]`` fp32 **raw** ``encode_image`` output.
Frame-count rules:
  * retrieval (inference_video_retrieval.py:312-317): always ``np.linspace(0, n-1, F).astype(int)`` (repeats rows when
    F > n);
  * joint-model dataset (hirest_dataset.py:333-356): the same subsample when n > F, else the *bucket up-sample*: row k
    is repeated ``((k+1)*F)//n - (k*F)//n`` times, result fp32;
  * check_feature_size.py:31-37: a file longer than ``round(v_duration)`` frames is cut to that length.

Compute (encoding, per-frame L2 normalisation) runs on the GPU through the C ABI; the file I/O is host work and is taken
off the critical path by ``FeatureWriter`` (pinned staging + a writer thread), so the encoder never waits for the disk.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Optional

import numpy as np
import torch

from . import ops


def subsample_ids(n_frames: int, n_model_frames: int) -> np.ndarray:
    """np.linspace(0, n-1, F).astype(int) (inference_video_retrieval.py:315, hirest_dataset.py:338-339)."""
    return np.linspace(0, n_frames - 1, n_model_frames).astype(int)


def upsample_ids(n_frames: int, n_model_frames: int) -> np.ndarray:
    """hirest_dataset.py:342-354: F slots are cut into n buckets [(j*F)//n, ((j+1)*F)//n); frame j fills bucket j."""
    F, N = int(n_model_frames), int(n_frames)
    counts = [((j + 1) * F) // N - (j * F) // N for j in range(N)]
    return np.repeat(np.arange(N), counts)


def fit_frame_ids(n_frames: int, n_model_frames: int, rule: str = "dataset") -> np.ndarray:
    """Row indices that turn an [n,E] feature file into the [F,E] tensor the consumer wants."""
    if rule == "retrieval" or n_frames > n_model_frames:
        return subsample_ids(n_frames, n_model_frames)
    if rule != "dataset":
        raise ValueError(f"unknown rule {rule!r}")
    return upsample_ids(n_frames, n_model_frames)


def fit_frames(features: torch.Tensor, n_model_frames: int, rule: str = "dataset") -> torch.Tensor:
    """hirest_dataset.py:333-356 / inference_video_retrieval.py:312-317 on a CPU or CUDA tensor.  n_model_frames <= 0
    returns the input (the reference skips the step).  The dataset's up-sample branch yields fp32."""
    if n_model_frames <= 0:
        return features
    n = features.shape[0]
    ids = torch.from_numpy(fit_frame_ids(n, n_model_frames, rule)).to(features.device)
    out = features.index_select(0, ids)
    if rule == "dataset" and n <= n_model_frames:
        out = out.float()
    return out


def warp_asr(asr_features: torch.Tensor, sub_spans, len_vid: int) -> torch.Tensor:
    """hirest_dataset.py:369-380: one row per second of video; subtitle i's embedding fills seconds [start_i, end_i)
    (``sub.start.seconds`` / ``sub.end.seconds``: whole seconds).  Later subtitles overwrite earlier ones.
    NOTE (reference behaviour, kept): ``len_vid`` is the length of the video features *after* they were fitted to
    n_model_frames, so the subtitle seconds index the resampled frame axis."""
    out = torch.zeros(len_vid, asr_features.shape[1]).float()
    for i, (start, end) in enumerate(sub_spans):
        out[int(start):int(end)] = asr_features[i]
    return out


def fit_asr(asr_features: torch.Tensor, sub_spans, fitted_video_features: torch.Tensor, n_model_frames: int) -> torch.Tensor:
    """The whole ASR branch of MomentDataset.__getitem__ (hirest_dataset.py:358-402) -> ``asr_feats``."""
    warped = warp_asr(asr_features, sub_spans, fitted_video_features.shape[0])
    return fit_frames(warped, n_model_frames, "dataset")


def load_video_features(path, n_model_frames: int = 0, rule: str = "dataset", device=None) -> torch.Tensor:
    """torch.load(<video>.pt, map_location='cpu') + the frame-count rule of the chosen consumer."""
    feats = torch.load(path, map_location="cpu")
    feats = fit_frames(feats, n_model_frames, rule)
    if rule == "retrieval":
        feats = feats.float()                               # inference_video_retrieval.py:319
    return feats.to(device) if device is not None else feats


def trim_to_duration(features: torch.Tensor, v_duration: float) -> torch.Tensor:
    """check_feature_size.py:24,35-36: at most round(v_duration) rows (Python round = half-to-even)."""
    n = int(round(v_duration))
    return features[:n] if features.shape[0] != n else features


def _model_input_size(model) -> int:
    """Input resolution of the frame tower: EVA-CLIP keeps it as ``visual.image_size``, the OpenAI CLIP port as ``input_resolution``."""
    vis = getattr(model, "visual", None)
    size = getattr(vis, "image_size", None) or getattr(vis, "input_resolution", None) or getattr(model, "input_resolution", None)
    if size is None:
        raise AttributeError("model exposes neither visual.image_size nor input_resolution: cannot preprocess uint8 frames")
    return int(size[0] if isinstance(size, (tuple, list)) else size)


def _prepare_frames(model, frames: torch.Tensor) -> torch.Tensor:
    """uint8 [T,H,W,3] decoded frames -> the model's input (resize + crop + normalise on the device); anything else passes through.
    The input size is looked up only when uint8 frames actually arrive (preprocessed float frames need nothing from the model)."""
    if frames.dtype != torch.uint8:
        return frames
    size = _model_input_size(model)
    if tuple(frames.shape[1:3]) == (size, size):
        return frames
    from .preprocess import FramePreprocessor
    pre = getattr(model, "_frame_preprocessor", None)
    if pre is None:
        vis = getattr(model, "visual", None)
        pre = FramePreprocessor(size, getattr(vis, "image_mean", None), getattr(vis, "image_std", None))
        object.__setattr__(model, "_frame_preprocessor", pre)
    return pre(frames)


def _embed_dim(model) -> int:
    for owner, name in ((model, "embed_dim"), (getattr(model, "visual", None), "output_dim"), (getattr(model, "visual", None), "num_classes")):
        v = getattr(owner, name, None)
        if isinstance(v, int) and v > 0:
            return v
    raise AttributeError("cannot tell the embedding width of this model for an empty video")


@torch.no_grad()
def frame_features(model, frames: torch.Tensor, batch_size: int = 1024, normalize: bool = True) -> torch.Tensor:
    """extract_features.py:52-65 for one video: frames [T,3,S,S] (preprocessed) or uint8 [T,H,W,3] (decoded; resized and
    cropped on the device) -> [T,E] fp32 on the GPU; ``normalize`` selects the L2-normalised file convention.  A video of zero
    frames gives an empty [0,E] tensor."""
    if frames.shape[0] == 0:
        return torch.zeros((0, _embed_dim(model)), dtype=torch.float32, device=frames.device)
    frames = _prepare_frames(model, frames)
    outs = [model.encode_image(frames[s:s + batch_size]).float() for s in range(0, frames.shape[0], batch_size)]
    feats = torch.cat(outs) if len(outs) > 1 else outs[0]
    if normalize:
        feats = ops.pool_l2norm(feats.unsqueeze(1).contiguous())       # F = 1: per-frame L2 normalisation
    return feats


@torch.no_grad()
def frame_features_many(model, videos, min_call: int = 256, max_call: int = 1024, normalize: bool = True):
    """``frame_features`` for a LIST of videos (each [T_i,3,S,S] preprocessed, or uint8 [T_i,H,W,3] decoded frames of one size per
    video): consecutive videos are encoded TOGETHER until a call holds at least ``min_call`` frames (at most ``max_call``), because a
    tower call of a few dozen frames loses a fifth of its throughput to tile quantisation (a 64-frame call is 6.09 rounds of 256 x 256
    tiles run as 7; below 64 frames the LayerNorm fold and the persistent attention are off as well).  extract_features.py:46-69
    encodes one video at a time; this is the same per-video output — a list of [T_i,E] fp32 GPU tensors — with short videos riding
    along with their neighbours.  Every frame's embedding depends on that frame alone; its bits are those of ``frame_features`` on the
    same video whenever both calls take the same kernels (both >= 64 frames), and within the bf16 towers' tolerance otherwise."""
    videos = list(videos)
    out = [None] * len(videos)
    group, count = [], 0

    def flush():
        nonlocal group, count
        if not group:
            return
        prepared = [_prepare_frames(model, videos[i]) for i in group]
        same = len({(p.dtype, tuple(p.shape[1:])) for p in prepared}) == 1
        if same:
            feats = frame_features(model, torch.cat(prepared) if len(prepared) > 1 else prepared[0], batch_size=max_call, normalize=normalize)
            lo = 0
            for i, p in zip(group, prepared):
                out[i] = feats[lo:lo + p.shape[0]]
                lo += p.shape[0]
        else:                                   # mixed input formats in one group: nothing to concatenate, one call each
            for i, p in zip(group, prepared):
                out[i] = frame_features(model, p, batch_size=max_call, normalize=normalize)
        group, count = [], 0
    for i, f in enumerate(videos):
        n = int(f.shape[0])
        if n == 0:                               # nothing to encode: an empty [0,E] result, and it must not pad a group
            out[i] = frame_features(model, f, normalize=normalize)
            continue
        if group and count + n > max_call:
            flush()
        group.append(i)
        count += n
        if count >= min_call:
            flush()
    flush()
    return out


class FeatureWriter:
    """Streaming ``torch.save`` of per-video feature tensors: ``submit`` enqueues an asynchronous device->pinned-host
    copy on a side stream and returns; a writer thread waits for the copy and writes ``<save_dir>/<name>.pt`` (a plain
    fp32 CPU tensor, exactly what the reference writes).  ``close`` drains the queue and re-raises writer errors."""

    def __init__(self, save_dir: str, max_pending: int = 16):
        self.save_dir = str(save_dir)
        os.makedirs(self.save_dir, exist_ok=True)
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)
        self._err: Optional[BaseException] = None
        self._stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()
        self.written = 0

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            name, host, event = item
            try:
                if event is not None:
                    event.synchronize()
                torch.save(host.clone() if host.is_pinned() else host, os.path.join(self.save_dir, f"{name}.pt"))
                self.written += 1
            except BaseException as e:      # surfaced by close()
                self._err = e

    def submit(self, name: str, features: torch.Tensor):
        if self._err is not None:
            raise self._err
        features = features.detach()
        if features.device.type == "cuda":
            host = torch.empty(features.shape, dtype=features.dtype, pin_memory=True)
            done = torch.cuda.Event()
            self._stream.wait_stream(torch.cuda.current_stream(features.device))
            with torch.cuda.stream(self._stream):
                host.copy_(features, non_blocking=True)
                features.record_stream(self._stream)
                done.record(self._stream)
            self._q.put((name, host, done))
        else:
            self._q.put((name, features.clone(), None))

    def close(self):
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
