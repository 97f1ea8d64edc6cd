"""Device-resident batches dealt over TWO handles in turn (batch k on handle k mod 2) against one handle: does the back end of one batch
(blur || quadtree, orientation + descriptors: latency- / HBM-bound) run under the pyramid + FAST kernel of the next (vector-issue bound)?
Usage on the GPU box: python3 tools/experiments/two_handles_device.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
B, W, H, NB = 1024, 640, 480, 6
frames = [torch.from_numpy(np.stack([pkg.synth_frame(b * B + i, W, H) for i in range(B)])).cuda() for b in range(NB)]
exs = [pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B) for _ in range(2)]
def run(nh, reps=20):
    for e in exs[:nh]:
        e.extract_batch_device(frames[0].data_ptr(), B, W, H, W, W * H, (0, 1000))
    for e in exs[:nh]:
        e.sync()
    t = time.perf_counter()
    for r in range(reps):
        for b in range(NB):
            exs[(r * NB + b) % nh].extract_batch_device(frames[b].data_ptr(), B, W, H, W, W * H, (0, 1000))
    for e in exs[:nh]:
        e.sync()
    dt = time.perf_counter() - t
    return reps * NB * B / dt
for nh in (1, 2, 1, 2):
    print(f"{nh} handle(s): {run(nh):.0f} frames/s", flush=True)
