"""Randomised parity sweep of the tracking batch forms on device-resident extractor batches (development tool, not part of the test
suite): orbx_search_by_projection_frames - UndistortKeyPoints -> grid -> SearchByProjection(Frame, Frame), Tracking::TrackWithMotionModel's
matcher - and its map-point form (Tracking::SearchLocalPoints) over P frames, P next to the schedules' thresholds (1 .. 260), random frame
sizes, feature counts, distortion, stereo columns, windows; the oracle checks the first, a middle and the last frame.
Usage on the GPU box: python tests/tools/stress_tracking_frames.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
from oracle import oracle as orc
from test_search_projection_gpu import _map_points, _projections

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for c in range(cases):
    P = int(rng.choice([1, 2, 5, 16, 17, 60, 260]))
    w, h = int(rng.integers(320, 800 if P < 100 else 480)), int(rng.integers(240, 560 if P < 100 else 360))
    h = min(h, int(1.4 * w))
    nf = int(rng.integers(200, 2500 if P < 100 else 1200))
    distorted, stereo, points = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
    seeds = rng.integers(0, 1 << 20, P)
    shifts = rng.integers(-10, 11, (P, 2))
    last = np.stack([pkg.synth_frame(int(s), w, h) for s in seeds])
    cur = np.stack([np.clip(np.roll(np.roll(last[i], int(shifts[i][1]), axis=0), int(shifts[i][0]), axis=1).astype(np.int16)
                            + rng.integers(-3, 4, (h, w), dtype=np.int16), 0, 255).astype(np.uint8) for i in range(P)])
    print(f"case {c}: P={P} {w}x{h} nf={nf} distorted={distorted} stereo={stereo} points={points}", end=" ", flush=True)
    try:
        ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
        nl, _, kl, dl = ex.extract_batch(last, (0, 0))
        kl, dl = kl.copy(), dl.copy()
        nc, _, kc, dc = ex.extract_batch(cur, (0, 0))
        sf = ex.GetScaleFactors().copy()
        k4 = (0.72 * w, 0.95 * h, w / 2 + 3.2, h / 2 - 1.6)
        dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05) if distorted and not points else ()
        m = pkg.ORBmatcher(ex.capacity, ex.capacity)
        bounds = m.ComputeImageBounds(w, h, k4, dist)
        check = sorted({0, P // 2, P - 1})
        ok, tot = True, 0
        if points:
            th, ratio = float(rng.choice([1.0, 3.0, 5.0])), float(rng.choice([0.8, 0.9]))
            mps = [_map_points(pkg, kl[p, : nl[p]], rng, int(shifts[p][0]), int(shifts[p][1])) for p in range(P)]
            descs = [dl[p, : nl[p]] for p in range(P)]
            nm, assign, kun = m.SearchByProjectionFrames(ex, P, k4, dist, bounds, 0.0, mps, descs, th, map_points_nnratio=ratio)
            for p in check:
                state = np.full(nc[p], -1, np.int8)
                rn, rassign = orc.search_by_projection_points(kc[p, : nc[p]], dc[p, : nc[p]], state, bounds, sf, mps[p], descs[p], th, ratio)
                ok = ok and kun[p, : nc[p]].tobytes() == kc[p, : nc[p]].tobytes() and nm[p] == rn and np.array_equal(assign[p, : nc[p]], rassign)
                tot += int(rn)
        else:
            th, mbf = (float(rng.choice([7.0, 10.0])), float(rng.uniform(20, 80))) if stereo else (float(rng.choice([15.0, 30.0, 4.0])), 0.0)
            proj, pdesc, ur_all = [], [], np.full((P, ex.capacity), -1, np.float32)
            for p in range(P):
                k1 = orc.undistort_keypoints(kl[p, : nl[p]], k4, dist)
                proj.append(_projections(pkg, k1, rng, int(shifts[p][0]), int(shifts[p][1]))); pdesc.append(dl[p, : nl[p]])
                if stereo:
                    ur_all[p, : nc[p]] = np.where(rng.random(nc[p]) < 0.6, kc[p, : nc[p]]["x"] - rng.uniform(0.5, 40, nc[p]), -1)
            nm, assign, kun = m.SearchByProjectionFrames(ex, P, k4, dist, bounds, mbf, proj, pdesc, th, u_right=ur_all if stereo else None)
            for p in check:
                rk = orc.undistort_keypoints(kc[p, : nc[p]], k4, dist)
                state = np.full(nc[p], -1, np.int8)
                rn, rassign = orc.search_by_projection(rk, dc[p, : nc[p]], state, bounds, sf, proj[p], pdesc[p], th, ur_all[p, : nc[p]] if stereo else None,
                                                       mbf, False, False, True)
                ok = ok and kun[p, : nc[p]].tobytes() == rk.tobytes() and nm[p] == rn and np.array_equal(assign[p, : nc[p]], rassign)
                tot += int(rn)
        m.close(); ex.close()
        print(f"matches={tot}", end=" ")
    except Exception as e:  # noqa: BLE001
        ok = False
        print("   exception:", e)
    print("ok" if ok else "MISMATCH", flush=True)
    bad += not ok
print("ALL OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
