"""Randomised parity sweep of the device-resident batch forms behind a stereo rig (development tool, not part of the test suite):
orbx_match_knn2_frames (Frame::ComputeStereoFishEyeMatches' BFMatcher call, with and without the lapping-area row ranges) and
orbx_compute_stereo_matches_frames (Frame::ComputeStereoMatches) over P pairs of two batch handles, P next to the schedules' thresholds,
random frame sizes, feature counts and thresholds; the oracle checks the first, a middle and the last pair.
Usage on the GPU box: python tests/tools/stress_stereo_frames.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for c in range(cases):
    P = int(rng.choice([1, 2, 5, 16, 17, 40, 129]))
    w, h = int(rng.integers(320, 900 if P < 100 else 520)), int(rng.integers(240, 600 if P < 100 else 400))
    h = min(h, int(1.4 * w))
    nf = int(rng.integers(200, 3000))
    ini = int(rng.integers(8, 30)); mn = int(rng.integers(2, ini + 1))
    lapping = bool(rng.integers(0, 2))
    lap = (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w))) if lapping else (0, 0)
    seeds = rng.integers(0, 1 << 20, P)
    left = np.stack([pkg.synth_frame(int(s), w, h) for s in seeds])
    shifts = rng.integers(2, 50, P)
    right = np.stack([np.clip(np.roll(left[i], -int(shifts[i]), axis=1).astype(np.int16) + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8) for i in range(P)])
    print(f"case {c}: P={P} {w}x{h} nf={nf} th={ini}/{mn} lap={lap}", end=" ", flush=True)
    try:
        exl = pkg.ORBextractor(nf, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=P)
        exr = pkg.ORBextractor(nf, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=P)
        nl, ml, kl, dl = exl.extract_batch(left, lap)
        nr, mr, kr, dr = exr.extract_batch(right, lap)
        m = pkg.ORBmatcher(exl.capacity, exr.capacity)
        idx, dist = m.knnMatchFrames(exl, exr, P, lapping_only=lapping)
        mb, mbf = float(rng.uniform(0.05, 0.6)), float(rng.uniform(20, 400))
        ur, dp = m.ComputeStereoMatchesFrames(exl, exr, P, mb, mbf)
        ok = True
        for p in sorted({0, P // 2, P - 1}):
            oL, oR = orc.Oracle(nf, 1.2, 8, ini, mn), orc.Oracle(nf, 1.2, 8, ini, mn)
            rml, rkl, rdl = oL.extract(left[p], lap); rmr, rkr, rdr = oR.extract(right[p], lap)
            ok = ok and nl[p] == len(rkl) and nr[p] == len(rkr) and kl[p][: nl[p]].tobytes() == rkl.tobytes() and dr[p][: nr[p]].tobytes() == rdr.tobytes()
            qb, tb = (int(rml), int(rmr)) if lapping else (0, 0)
            if lapping and (qb < 0 or tb < 0):
                qb = tb = 0
            q, t = rdl[qb:], rdr[tb:]
            if len(q) and len(t) >= 0:
                ridx, rdist = orc.knn2(q, t)
                ok = ok and np.array_equal(idx[p, : len(q)], ridx) and np.array_equal(dist[p, : len(q)], rdist)
            rur, rdp, _ = orc.stereo_matches(oL, oR, rkl, rdl, rkr, rdr, mb, mbf)
            ok = ok and np.array_equal(ur[p, : nl[p]].view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp[p, : nl[p]].view(np.uint32), rdp.view(np.uint32))
        m.close(); exl.close(); exr.close()
    except Exception as e:  # noqa: BLE001
        ok = False
        print("   exception:", e)
    print("ok" if ok else "MISMATCH", flush=True)
    bad += not ok
print("ALL OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
