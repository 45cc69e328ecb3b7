#!/usr/bin/env python3
"""extract_features.py end to end on the device (rows a1/f1 -> a2-a8 -> f2): decoded uint8 frames (default 720p) -> Pillow-exact
resize + crop -> EVA-CLIP-g/14 -> per-frame L2 -> <video>.pt files written by the streaming writer.  Frames start on the HOST (as a
video decoder would hand them over) unless --resident.  Prints frames/s and the stage split.
    python tools/pipeline_bench.py [--videos 16] [--frames 256] [--height 720 --width 1280]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirest_amd  # noqa: E402
from hirest_amd import features, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=16)
    ap.add_argument("--frames", type=int, default=256, help="frames per video")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--resident", action="store_true", help="frames already in HBM (no PCIe leg)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
    model.init_random_(seed=1234)
    model.visual.max_frames_per_call = 1024
    g = torch.Generator(); g.manual_seed(0)
    host = torch.randint(0, 256, (a.frames, a.height, a.width, 3), dtype=torch.uint8, generator=g).pin_memory()
    resident = host.to(dev) if a.resident else None
    out_dir = tempfile.mkdtemp(prefix="hirest_feats_")
    try:
        features.frame_features(model, host[:64].to(dev))            # warm-up (kernel configuration, preprocess plan)
        torch.cuda.synchronize()
        t_total = 0.0
        copy_stream = torch.cuda.Stream()
        bufs = [torch.empty_like(host, device=dev) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        used = [torch.cuda.Event() for _ in range(2)]

        def prefetch(v):                                            # H2D of video v on the copy stream, under video v - 1's compute
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(used[v & 1])
                bufs[v & 1].copy_(host, non_blocking=True)
                ready[v & 1].record(copy_stream)
        t0 = time.perf_counter()
        for e in used:
            e.record()
        if not a.resident:
            prefetch(0)
        with features.FeatureWriter(out_dir) as w:
            for v in range(a.videos):
                if a.resident:
                    frames = resident
                else:
                    torch.cuda.current_stream().wait_event(ready[v & 1])
                    frames = bufs[v & 1]
                    if v + 1 < a.videos:
                        prefetch(v + 1)
                w.submit(f"video_{v:04d}", features.frame_features(model, frames))
                used[v & 1].record()
            torch.cuda.synchronize()
            t_gpu = time.perf_counter() - t0
        t_total = time.perf_counter() - t0
        n = a.videos * a.frames
        files = len(os.listdir(out_dir))
        sample = torch.load(os.path.join(out_dir, "video_0000.pt"))
        print(f"{a.videos} videos x {a.frames} frames of {a.height}x{a.width} uint8 ({'resident' if a.resident else 'from pinned host memory'}): "
              f"{n / t_total:.0f} frames/s end to end ({t_total:.2f} s; GPU work done after {t_gpu:.2f} s; H2D copies double-buffered on a copy stream); "
              f"{files} files, sample {tuple(sample.shape)} {sample.dtype}, row norm {sample[0].norm().item():.6f}")
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


if __name__ == "__main__":
    main()
