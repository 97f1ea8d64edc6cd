"""Determinism soak: the same inputs through the batched entry points many times - every run must give the same bytes (a race between
workgroups, streams or LDS buffers would show up as a difference).  No oracle needed.
Usage on the GPU box: python tests/tools/soak_determinism.py [runs]"""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def soak(name, fn):
    global bad
    first = fn()
    diff = sum(fn() != first for _ in range(runs - 1))
    bad += diff
    print(f"{name}: {runs} runs, {diff} differ ({first})", flush=True)


for (w, h, nf, P) in ((640, 480, 1000, 192), (512, 512, 1500, 700), (1920, 1080, 4000, 8)):
    imgs = np.stack([pkg.synth_frame(100 + i, w, h) for i in range(min(P, 24))])
    imgs = np.concatenate([imgs] * ((P + len(imgs) - 1) // len(imgs)))[:P]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)

    def extract():
        n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
        return digest(n, *[kps[f, : n[f]] for f in range(P)], *[desc[f, : n[f]] for f in range(P)])
    soak(f"extract {P} x {w}x{h}/{nf}", extract)
    ex.close()

P, w, h, nf = 48, 752, 480, 1200
left = np.stack([pkg.synth_frame(300 + i, w, h) for i in range(P)])
right = np.stack([np.roll(left[i], -(6 + i % 30), axis=1) for i in range(P)])
exl = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
exr = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
exl.extract_batch(left, (0, 0)); exr.extract_batch(right, (0, 0))
m = pkg.ORBmatcher(8192, 8192)


def stereo():
    ur, dp = m.ComputeStereoMatchesFrames(exl, exr, P, 0.11, 47.9)
    return digest(ur, dp)


def knn():
    idx, dist = m.knnMatchFrames(exl, exr, P)
    return digest(idx, dist)


k4 = np.float32([458.654, 457.296, 367.215, 248.375]); dist4 = np.float32([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
bounds = m.ComputeImageBounds(w, h, k4, dist4)


def search_init():
    nm, m12, prev = m.SearchForInitializationFrames(exl, exr, P, k4, dist4, bounds, 100, 0.9, True)
    return digest(nm, m12, prev)


soak(f"ComputeStereoMatches {P} pairs", stereo)
soak(f"knnMatch {P} pairs", knn)
soak(f"SearchForInitialization {P} pairs", search_init)
print("ALL OK" if bad == 0 else f"{bad} DIFFERENCES")
sys.exit(1 if bad else 0)
