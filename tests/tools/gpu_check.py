"""Stage-by-stage GPU-vs-oracle comparison for development (prints the first divergence per stage).
Usage: python tests/tools/gpu_check.py [w h nfeatures nframes [content]]   (content: synth (default) | noise | saturated)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()

w = int(sys.argv[1]) if len(sys.argv) > 1 else 640
h = int(sys.argv[2]) if len(sys.argv) > 2 else 480
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
nframes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
content = sys.argv[5] if len(sys.argv) > 5 else "synth"
lap = (0, 1000)

if content == "synth":
    imgs = np.stack([pkg.synth_frame(i, w, h) for i in range(nframes)])
else:  # corners everywhere: the queue-overflow paths of k_pyr_fast
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, (nframes, h, w), dtype=np.uint8)
    if content == "saturated":
        imgs = np.where(imgs < 128, 0, 255).astype(np.uint8)
ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nframes)
t = time.time()
n, mono, kps, desc = ex.extract_batch(imgs, lap)
print("gpu batch", time.time() - t, "n", n, "mono", mono)
ok = True
for f in range(nframes):
    ref = orc.Oracle(nf, 1.2, 8, 20, 7)
    rmono, rk, rd = ref.extract(imgs[f], lap)
    for l in range(8):
        a, b = ex.pyramid_level(f, l), ref.level(l)
        if not np.array_equal(a, b):
            d = np.argwhere(a != b)
            print(f"frame {f} level {l} PYRAMID differs at {len(d)} px, first {d[0]} gpu {a[tuple(d[0])]} ref {b[tuple(d[0])]}")
            ok = False
        c, rc = ex.debug_candidates(f, l), ref.candidates(l)
        if not np.array_equal(c, rc):
            print(f"frame {f} level {l} CANDIDATES differ: gpu {len(c)} ref {len(rc)}")
            m = min(len(c), len(rc))
            bad = np.nonzero(c[:m] != rc[:m])[0]
            if len(bad):
                i = bad[0]
                print("   first diff idx", i, "gpu", c[i] & 0xFFF, (c[i] >> 12) & 0xFFF, c[i] >> 24, "ref", rc[i] & 0xFFF, (rc[i] >> 12) & 0xFFF, rc[i] >> 24)
            sc, src_ = set(c.tolist()), set(rc.tolist())
            print("   only gpu", len(sc - src_), "only ref", len(src_ - sc))
            ok = False
        s = ex.debug_selected(f, l)
        rkl = ref.level_keypoints(l)
        rs = ((rkl["x"].astype(np.int64) - 16) | ((rkl["y"].astype(np.int64) - 16) << 12) | (rkl["response"].astype(np.int64) << 24)).astype(np.uint32)
        if not np.array_equal(s, rs):
            print(f"frame {f} level {l} SELECTED differ: gpu {len(s)} ref {len(rs)}")
            m = min(len(s), len(rs))
            bad = np.nonzero(s[:m] != rs[:m])[0]
            if len(bad):
                print("   first diff idx", bad[0], "same set:", set(s.tolist()) == set(rs.tolist()))
            ok = False
        a, b = ex.pyramid_level(f, l, True), ref.level(l, True)
        if not np.array_equal(a, b):
            d = np.argwhere(a != b)
            print(f"frame {f} level {l} BLUR differs at {len(d)} px, first {d[0]} gpu {a[tuple(d[0])]} ref {b[tuple(d[0])]}")
            ok = False
    k = kps[f][: n[f]]
    dd = desc[f][: n[f]]
    if n[f] != len(rk) or mono[f] != rmono:
        print(f"frame {f}: n {n[f]} vs {len(rk)}, mono {mono[f]} vs {rmono}")
        ok = False
    else:
        for name in k.dtype.names:
            if not np.array_equal(k[name].view(np.uint32), rk[name].view(np.uint32)):
                bad = np.nonzero(k[name].view(np.uint32) != rk[name].view(np.uint32))[0]
                print(f"frame {f}: field {name} differs at {len(bad)} kps, first {bad[0]}: gpu {k[name][bad[0]]!r} ref {rk[name][bad[0]]!r}")
                ok = False
        if not np.array_equal(dd, rd):
            bad = np.nonzero((dd != rd).any(axis=1))[0]
            print(f"frame {f}: descriptors differ at {len(bad)} kps, first {bad[0]}")
            ok = False
print("ALL STAGES BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
