#!/usr/bin/env python3
"""Batched, device-resident ORBmatcher::SearchForInitialization (orbx_search_for_initialization_frames: UndistortKeyPoints of both frames,
then one wave per frame pair) against single-threaded runs of the oracle restatement - the monocular initializer's setting: 5 x nFeatures
extractor, window 100, ratio 0.9, orientation check.
Usage on the GPU box: python tests/tools/time_search_init_batch.py > profiles/r02_search_init_batch.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch

pkg = g.load_package(); orc = g.load_oracle()
w, h, nf = 752, 480, 5000
k4 = np.float32([458.654, 457.296, 367.215, 248.375]); dist = np.float32([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
out = {"workload": f"{w}x{h}, nFeatures={nf}: extractor outputs -> UndistortKeyPoints x 2 -> SearchForInitialization(window 100, 0.9, orientation)",
       "results": {}}
for P in (1, 16, 64, 256):
    base = np.stack([pkg.synth_frame(500 + i, w, h) for i in range(P)])
    nxt = np.stack([np.roll(np.roll(base[i], -5, axis=0), 8, axis=1) for i in range(P)])
    ex1 = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    ex2 = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    n1, _, k1, d1 = ex1.extract_batch(base, (0, 0))
    n2, _, k2, d2 = ex2.extract_batch(nxt, (0, 0))
    m = pkg.ORBmatcher(8192, 8192)
    bounds = m.ComputeImageBounds(w, h, k4, dist)
    nm, m12, prev = m.SearchForInitializationFrames(ex1, ex2, P, k4, dist, bounds, 100, 0.9, True, download=False)  # warm-up
    m.sync()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        nm, m12, prev = m.SearchForInitializationFrames(ex1, ex2, P, k4, dist, bounds, 100, 0.9, True, download=False)
    m.sync()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    nm_h = nm.cpu().numpy()
    ns = min(P, 8)
    t0 = time.perf_counter()
    for p in range(ns):
        a = orc.undistort_keypoints(k1[p, : n1[p]], k4, dist); b = orc.undistort_keypoints(k2[p, : n2[p]], k4, dist)
        rn, _, _ = orc.search_for_initialization(a, d1[p, : n1[p]], b, d2[p, : n2[p]], np.stack([a["x"], a["y"]], axis=1), bounds, 100, 0.9, True)
        assert rn == nm_h[p]
    cpu_ms = (time.perf_counter() - t0) / ns * P * 1e3
    out["results"][str(P)] = {"gpu_ms_per_call": round(gpu_ms, 3), "gpu_us_per_pair": round(gpu_ms / P * 1e3, 1),
                              "oracle_1thread_ms_for_P_pairs": round(cpu_ms, 1), "ratio": round(cpu_ms / gpu_ms, 1),
                              "mean_keypoints": float(n1[:P].mean()), "mean_matches": float(nm_h.mean())}
    ex1.close(); ex2.close(); m.close()
print(json.dumps(out, indent=1))
