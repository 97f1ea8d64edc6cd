#!/usr/bin/env python3
"""Does the chip have room beside one handle's launch chain?  One handle working through batches of 1024 VGA frames against TWO handles
on the same GPU whose asynchronous batches interleave (the pyramid+FAST launches of one beside blur / quadtree / orientation of the other):
if the pair is not faster, the stages already add up to the chip's vector-issue capacity and a double-buffered handle cannot win either.
Usage on the GPU box: python tools/experiments/two_handles.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

import torch  # noqa: E402

pkg = graft.load_package()
cfg = bench.CONFIGS["c2"]
B, W, H = 1024, 640, 480
frames, _ = bench.make_frames(pkg, cfg, 0, B)
d = [torch.from_numpy(frames).cuda(), torch.from_numpy(frames[::-1].copy()).cuda()]
ex = [pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B) for _ in range(2)]


def run(handles, steps=20):
    for e, x in zip(handles, d):
        e.extract_batch_device(x.data_ptr(), B, W, H, W, W * H, cfg["lap"])
    for e in handles:
        e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, x in zip(handles, d):
            e.extract_batch_device(x.data_ptr(), B, W, H, W, W * H, cfg["lap"])
    for e in handles:
        e.sync()
    dt = time.perf_counter() - t0
    return steps * len(handles) * B / dt


one = run(ex[:1])
two = run(ex)
print(f"one handle: {one:.0f} frames/s ({B / one * 1e3:.3f} ms per 1024-frame batch)")
print(f"two handles interleaved on one GPU: {two:.0f} frames/s ({B / two * 1e3:.3f} ms per 1024-frame batch) = {two / one:.3f} x")
