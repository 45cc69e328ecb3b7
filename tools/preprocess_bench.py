#!/usr/bin/env python3
"""Throughput of the device preprocess (resize + crop) at video geometries; HBM roofline on the input read.
Also times Pillow on the host for the same frames (the reference's path, one thread)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import synth  # noqa: E402
from hirest_amd.preprocess import FramePreprocessor, host_plan  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
pre = FramePreprocessor(224)
for (h, w) in [(360, 640), (720, 1280), (1080, 1920)]:
    one = synth.rgb_frames("ppbench", (4, h, w, 3), 1)
    x = torch.from_numpy(one).to(dev).repeat((a.frames // 4, 1, 1, 1))
    for normalized in (False, True):
        for _ in range(2):
            pre(x, normalized=normalized)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            pre(x, normalized=normalized)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        pl = host_plan(h, w, 224)
        nrows, ncols = int(pl[10]), int(pl[12])
        alg = x.shape[0] * (nrows * ncols * 3 + 224 * 224 * 3 * (4 if normalized else 1))     # bytes that must move
        print(f"{h}x{w} B={x.shape[0]} out={'f32 NCHW' if normalized else 'u8 NHWC'}: {ms:.3f} ms  {x.shape[0] / ms * 1e3:.0f} frames/s  "
              f"{alg / ms / 1e6:.0f} GB/s algorithmic", flush=True)
    try:
        from PIL import Image
        from oracle import preprocess_cpu as P
        nw, nh = P.resized_size(w, h, 224)
        imgs = [Image.fromarray(f) for f in one]
        t = time.time()
        for _ in range(5):
            for im in imgs:
                im.resize((nw, nh), Image.BICUBIC)
        dt = (time.time() - t) / 20
        print(f"    Pillow resize on 1 host thread: {1 / dt:.0f} frames/s")
    except ImportError:
        pass
