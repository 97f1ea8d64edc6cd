"""Randomised parity sweep of the matcher kernels against the CPU oracle: kNN-2 (random sizes, duplicated rows, random split counts -
the matrix-core scan or, with ORBX_KNN_MFMA=0, the vector-ALU one) and Frame::ComputeStereoMatches (random frame sizes, feature
counts, shifts, noise and baselines).  Usage on the GPU box: python tests/tools/stress_matcher.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
m = pkg.ORBmatcher(8192, 8192)
for c in range(cases):
    nq, nt = int(rng.integers(1, 3000)), int(rng.integers(0, 3000))
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nt > 4:  # duplicated train rows and exact copies of queries: ties in distance, decided by the index
        for _ in range(int(rng.integers(0, 40))):
            t[rng.integers(0, nt)] = t[rng.integers(0, nt)]
        for _ in range(int(rng.integers(0, 40))):
            t[rng.integers(0, nt)] = q[rng.integers(0, nq)]
        few = rng.integers(0, 2)
        if few:  # descriptors that differ in a few bits only: many equal distances
            t[:] = t[0]
            flips = rng.integers(0, 256, (nt, 3))
            for i in range(nt):
                for b in flips[i][: rng.integers(0, 4)]:
                    t[i, b >> 3] ^= 1 << (b & 7)
    split = int(rng.choice([0, 0, 1, 2, 3, 5, 16, 61]))
    if split:
        os.environ["ORBX_KNN_SPLIT"] = str(split)
    else:
        os.environ.pop("ORBX_KNN_SPLIT", None)
    idx, dist = m.knnMatch(q, t)
    ridx, rdist = orc.knn2(q, t)
    ok = np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    bad += not ok
    print(f"knn2 case {c}: {nq} x {nt} split={split} {'ok' if ok else 'MISMATCH'}", flush=True)
os.environ.pop("ORBX_KNN_SPLIT", None)
for c in range(max(cases // 6, 4)):
    w, h = int(rng.integers(320, 1000)), int(rng.integers(240, 700))
    h = min(h, int(1.4 * w))  # (W / H < 0.5 on any level - the 16-px borders count - leaves DistributeOctTree without a root node, src/ORBextractor.cc:589: orbx_create rejects it)
    nf = int(rng.integers(300, 2500))
    shift = int(rng.integers(2, 60))
    left = pkg.synth_frame(int(rng.integers(0, 10000)), w, h)
    right = np.roll(left, -shift, axis=1).copy()
    right = np.clip(right.astype(np.int16) + rng.integers(-3, 4, right.shape), 0, 255).astype(np.uint8)
    exL = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    exR = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    _, kl, dl = exL(left, None, (0, 0)); _, kr, dr = exR(right, None, (0, 0))
    oL, oR = orc.Oracle(nf, 1.2, 8, 20, 7), orc.Oracle(nf, 1.2, 8, 20, 7)
    _, rkl, rdl = oL.extract(left, (0, 0)); _, rkr, rdr = oR.extract(right, (0, 0))
    mb, mbf = float(rng.uniform(0.05, 0.6)), float(rng.uniform(20, 400))
    mm = pkg.ORBmatcher(4096, 4096)
    ur, dp = mm.ComputeStereoMatches(exL, exR, kl, dl, kr, dr, mb, mbf)
    rur, rdp, _ = orc.stereo_matches(oL, oR, rkl, rdl, rkr, rdr, mb, mbf)
    ok = kl.tobytes() == rkl.tobytes() and np.array_equal(ur.view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))
    bad += not ok
    print(f"stereo case {c}: {w}x{h} nf={nf} shift={shift} matches={(ur >= 0).sum()} {'ok' if ok else 'MISMATCH'}", flush=True)
    exL.close(); exR.close(); mm.close()
print("ALL OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
