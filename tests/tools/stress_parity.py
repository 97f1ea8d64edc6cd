"""Randomised end-to-end parity sweep (development tool, not part of the test suite): random image sizes, feature counts, level
counts, scale factors, thresholds and image content against the CPU oracle.  Usage: python tests/tools/stress_parity.py [n] [seed]
STRESS_BATCH=N: every case runs N frames on a BATCH handle (max_batch > 16: the 360-px tiles, margin bands and the capped group queue of
k_pyr_fast; the oracle checks the first, one in the middle and the last frame), image sizes capped at 1300 x 800; STRESS_BATCH=-1: a
batch size next to one of the schedule's thresholds per case (17 .. 700 frames)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # skip the cases before this one (same random stream)
bad = 0
for case in range(n_cases):
    big = int(os.environ.get("STRESS_BATCH", "0"))
    if big < 0:  # a batch size next to one of the schedule's thresholds (graph / batch schedule, multi-wave / one-wave quadtree, LDS key buffer, launch groups)
        big = int(rng.choice([17, 31, 64, 127, 128, 129, 200, 640, 700] if big == -1 else [4, 7, 8, 15, 16]))   # (-2: the graph schedule's batch sizes)
    w = int(rng.integers(120, (560 if big >= 600 else 1300) if big else 2000))
    h = int(rng.integers(100, (420 if big >= 600 else 800) if big else 1300))
    nlev = int(rng.integers(1, 9))
    scale = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
    while min(w, h) / scale ** (nlev - 1) < 70:  # every level must hold one 35-px FAST cell + borders
        nlev -= 1
    if (w - 32) / (h - 32) < 0.55 :  # nIni = round(W / H) = 0: undefined behaviour in the reference, rejected by the library
        h = int(w * 1.5)
    nf = int(rng.integers(50, 6000))
    ini = int(rng.integers(5, 40))
    mn = int(rng.integers(0, ini + 1))
    kind = rng.integers(0, 4)
    if kind == 0:
        img = pkg.synth_frame(int(rng.integers(0, 1 << 30)), w, h)
    elif kind == 1:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 2:
        img = (128 + rng.integers(-9, 10, (h, w))).astype(np.uint8)
    else:
        img = np.kron(rng.integers(0, 256, (h // 16 + 1, w // 16 + 1), dtype=np.uint8), np.ones((16, 16), np.uint8))[:h, :w].copy()
    lap = (int(rng.integers(-10, w)), int(rng.integers(0, 2 * w)))
    if case < first:
        continue
    print(f"case {case}: {'B=%d ' % big if big else ''}{w}x{h} nf={nf} levels={nlev} scale={scale} th={ini}/{mn} kind={kind} lap={lap}", end=" ", flush=True)
    bv, am = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    nb = int(rng.integers(1, 4))
    pad_w, pad_h = int(rng.integers(0, 40)), int(rng.integers(0, 40))
    # which of the extra dimensions to exercise ("am": trig_mode 1 = glibc cosf / sinf in computeOrbDescriptor)
    dims = os.environ.get("STRESS_DIMS", "bv,am,nb,pad").split(",")
    bv = bv if "bv" in dims else 0
    am = am if "am" in dims else 0
    nb = nb if "nb" in dims else 1
    if big:
        nb = big
    pad_w, pad_h = (pad_w, pad_h) if "pad" in dims else (0, 0)
    try:
        ex = pkg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w + pad_w, max_height=h + pad_h,
                              max_batch=nb, blur_variant=bv, trig_mode=am)
        imgs = np.stack([img] + [np.roll(img, 17 * (i + 1), axis=1) for i in range(nb - 1)])
        n, mono, kps, desc = ex.extract_batch(imgs, lap)
        ok = True
        ref = orc.Oracle(nf, scale, nlev, ini, mn, blur_variant=bv, trig_mode=am)
        for i in (sorted({0, nb // 2, nb - 1}) if big else range(nb)):
            rmono, rk, rd = ref.extract(imgs[i], lap)
            ok = ok and mono[i] == rmono and n[i] == len(rk) and np.array_equal(kps[i][: n[i]].view(np.uint8), rk.view(np.uint8)) \
                and np.array_equal(desc[i][: n[i]], rd)
        ex.close()
    except Exception as e:  # noqa: BLE001
        if "aspect ratio below 0.5" in str(e):  # a small level became too narrow: undefined behaviour in the reference, rejected here
            print("skipped (portrait level)", flush=True)
            continue
        ok = False
        print("   exception:", e)
    print(f"n={len(rk) if ok else '?'} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("ALL OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
