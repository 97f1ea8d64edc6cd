#!/usr/bin/env python3
"""Batched, device-resident tracking matcher (orbx_search_by_projection_frames: UndistortKeyPoints -> AssignFeaturesToGrid ->
SearchByProjection(Frame, Frame) for P frames in six launches) against P single-threaded runs of the oracle restatement.
Usage on the GPU box: python tests/tools/time_tracking_batch.py > profiles/r02_tracking_batch.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch

pkg = g.load_package(); orc = g.load_oracle()
w, h, nf = 752, 480, 1200
k4 = (458.654, 457.296, 367.215, 248.375); dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
rng = np.random.default_rng(1)
out = {"workload": f"{w}x{h}, nFeatures={nf}: extractor outputs -> UndistortKeyPoints -> grid -> SearchByProjection(Frame, Frame), th=15", "results": {}}
for P in (1, 16, 64, 256):
    last = np.stack([pkg.synth_frame(900 + i, w, h) for i in range(P)])
    cur = np.stack([np.roll(np.roll(last[i], -4, axis=0), 6, axis=1) for i in range(P)])
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    nl, _, kl, dl = ex.extract_batch(last, (0, 0)); kl, dl = kl.copy(), dl.copy()
    nc, _, kc, dc = ex.extract_batch(cur, (0, 0))
    sf = ex.GetScaleFactors().copy()
    m = pkg.ORBmatcher(4096, 4096)
    bounds = m.ComputeImageBounds(w, h, k4, dist)
    proj, pdesc = [], []
    for p in range(P):
        k1 = orc.undistort_keypoints(kl[p, : nl[p]], k4, dist)
        pr = np.zeros(len(k1), pkg.PROJECTION_DTYPE)
        pr["u"] = k1["x"] + 6 + rng.normal(0, 3, len(k1)); pr["v"] = k1["y"] - 4 + rng.normal(0, 3, len(k1))
        pr["invz"] = 0.2; pr["angle"] = k1["angle"]; pr["octave"] = k1["octave"]; pr["flags"] = 3
        proj.append(pr); pdesc.append(dl[p, : nl[p]])
    # device-resident inputs once; time the call itself
    lib = pkg.load_library()
    nm, assign, kun = m.SearchByProjectionFrames(ex, P, k4, dist, bounds, 0.0, proj, pdesc, 15.0, download=False)  # warm-up, also builds device inputs
    # re-create the device inputs outside the timed region
    import ctypes
    cap = ex.capacity; max_np = max(len(p) for p in proj)
    hp = np.zeros((P, max_np), pkg.PROJECTION_DTYPE); hd = np.zeros((P, max_np, 32), np.uint8); hn = np.zeros(P, np.int32)
    for p in range(P):
        hn[p] = len(proj[p]); hp[p, : hn[p]] = proj[p]; hd[p, : hn[p]] = pdesc[p]
    d_p = torch.from_numpy(hp.view(np.uint8)).cuda(); d_d = torch.from_numpy(hd).cuda(); d_n = torch.from_numpy(hn).cuda()
    d_a = torch.empty((P, cap), dtype=torch.int32, device="cuda"); d_nm = torch.empty((P,), dtype=torch.int32, device="cuda")
    k4a = np.asarray(k4, np.float32); da = np.asarray(dist, np.float32)
    def call():
        rc = lib.orbx_search_by_projection_frames(m._h, ex._h, P, k4a.ctypes.data, da.ctypes.data, len(da), None, bounds[0], bounds[1], bounds[2], bounds[3],
                                                  0.0, None, None, ctypes.c_void_p(d_p.data_ptr()), ctypes.c_void_p(d_d.data_ptr()), ctypes.c_void_p(d_n.data_ptr()),
                                                  max_np, 15.0, 0, 0, 1, None, ctypes.c_void_p(d_a.data_ptr()), ctypes.c_void_p(d_nm.data_ptr()))
        assert rc == 0, lib.orbx_last_error()
    torch.cuda.synchronize()
    call(); m.sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    m.sync()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    # oracle, one thread: the same per-frame work
    ns = min(P, 16)
    t0 = time.perf_counter()
    tot = 0
    for p in range(ns):
        rk = orc.undistort_keypoints(kc[p, : nc[p]], k4, dist)
        rn, _ = orc.search_by_projection(rk, dc[p, : nc[p]], np.full(nc[p], -1, np.int8), bounds, sf, proj[p], pdesc[p], 15.0, None, 0.0, False, False, True)
        tot += rn
    cpu_ms = (time.perf_counter() - t0) / ns * 1e3 * P
    out["results"][f"P{P}"] = {"gpu_ms_per_batch": round(gpu_ms, 4), "gpu_us_per_frame": round(gpu_ms / P * 1e3, 2),
                               "oracle_1thread_ms_per_batch": round(cpu_ms, 3), "speedup": round(cpu_ms / gpu_ms, 1), "matches_per_frame": round(tot / ns, 1)}
    m.close(); ex.close()
print(json.dumps(out, indent=1))
