"""FAST candidates per (frame, level) on the benchmark frames: which quadtree instances hold their keys in registers
(M <= QT3_KCH * 64).  Usage on the GPU box: python tools/experiments/cand_counts.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
B = 64
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
frames = np.stack([pkg.synth_frame(i, 640, 480) for i in range(B)])
ex.extract_batch(frames)
M = np.array([[len(ex.debug_candidates(f, l)) for l in range(8)] for f in range(B)])
print("level:   " + " ".join("%6d" % l for l in range(8)))
print("mean M:  " + " ".join("%6.0f" % v for v in M.mean(0)))
print("max M:   " + " ".join("%6d" % v for v in M.max(0)))
print("> 2048:  " + " ".join("%6d" % v for v in (M > 2048).sum(0)), "of", B)
print("quota:   " + " ".join("%6d" % v for v in ex.features_per_level()))
