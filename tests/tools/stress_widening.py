"""Randomised parity sweep of the widened entry points against the CPU oracle: both tracking
SearchByProjection variants, UndistortKeyPoints and the vocabulary transform, with random sizes / parameters per case.
Usage on the GPU box: python tests/tools/stress_widening.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
from oracle import oracle as orc
from voc_synth import features_near_words, make_vocabulary
import scene3d

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def frames(seed, w, h, nfeat):
    ex = pkg.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
    img1 = pkg.synth_frame(seed, w, h)
    dx, dy = int(rng.integers(-8, 9)), int(rng.integers(-8, 9))
    img2 = np.clip(np.roll(np.roll(img1, dy, 0), dx, 1).astype(np.int16) + rng.integers(-4, 5, img1.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    _, k2, d2 = ex(img2, None, (0, 0)); k2, d2 = k2.copy(), d2.copy()
    sf = ex.GetScaleFactors().copy(); ex.close()
    return k1, d1, k2, d2, sf, dx, dy


m = pkg.ORBmatcher(8192, 8192)
for c in range(cases):
    w, h = [(640, 480), (752, 480), (512, 512), (1241, 376)][int(rng.integers(0, 4))]
    k1, d1, k2, d2, sf, dx, dy = frames(int(rng.integers(0, 1 << 20)), w, h, int(rng.integers(300, 2500)))
    bounds = (0.0, float(w), 0.0, float(h))
    n1 = len(k1)
    pr = np.zeros(n1, pkg.PROJECTION_DTYPE)
    noise = float(rng.uniform(0.2, 6.0))
    pr["u"] = k1["x"] + dx + rng.normal(0, noise, n1); pr["v"] = k1["y"] + dy + rng.normal(0, noise, n1)
    pr["octave"] = np.clip(k1["octave"] + rng.integers(-1, 2, n1), 0, 7)
    pr["flags"] = (rng.random(n1) < rng.uniform(0.3, 1.0)).astype(np.int32) | ((rng.random(n1) < rng.uniform(0.0, 1.0)).astype(np.int32) << 1)
    state = rng.choice(np.int8([-1, -1, 0, 1]), len(k2))
    ur = np.where(rng.random(len(k2)) < 0.5, k2["x"] - rng.uniform(0.5, 40, len(k2)), -1).astype(np.float32) if rng.random() < 0.5 else None
    # SearchByProjection(Frame, Frame)
    pr["invz"] = 1.0 / rng.uniform(0.5, 20.0, n1); pr["angle"] = k1["angle"]
    th = float(rng.choice([2, 7, 15, 30])); fwd, bwd = bool(rng.random() < 0.2), False
    mbf = float(rng.uniform(10, 60)) if ur is not None else 0.0
    chk = bool(rng.random() < 0.7)
    a = m.SearchByProjection(k2, d2, state, bounds, sf, pr, d1, th, u_right=ur, mbf=mbf, forward=fwd, backward=bwd, checkOrientation=chk)
    ra = orc.search_by_projection(k2, d2, state, bounds, sf, pr, d1, th, ur, mbf, fwd, bwd, chk)
    ok0 = a[0] == ra[0] and np.array_equal(a[1], ra[1])
    # SearchByProjection(Frame, MapPoints)
    pm = pr.copy(); pm["angle"] = rng.choice(np.float32([0.3, 0.99, 0.998, 0.9985, 1.0]), n1); pm["invz"] = pm["u"] - rng.uniform(0.5, 40, n1)
    thm, ratio = float(rng.choice([1, 3, 5, 15])), float(rng.choice([0.6, 0.8, 0.9, 1.0]))
    g = m.SearchByProjectionPoints(k2, d2, state, bounds, sf, pm, d1, thm, ratio, u_right=ur)
    r = orc.search_by_projection_points(k2, d2, state, bounds, sf, pm, d1, thm, ratio, ur)
    ok1 = g[0] == r[0] and np.array_equal(g[1], r[1])
    # UndistortKeyPoints
    k4 = (float(rng.uniform(300, 900)), float(rng.uniform(300, 900)), w / 2 + float(rng.uniform(-30, 30)), h / 2 + float(rng.uniform(-30, 30)))
    nd = int(rng.choice([4, 5, 8, 12]))
    scale = np.float32([1, 0.3, 0.01, 0.01, 0.1, 0.05, 0.05, 0.05, 0.005, 0.005, 0.005, 0.005])[:nd]
    dist = (rng.uniform(-0.4, 0.4, nd) * scale).astype(np.float32)
    ok2 = m.UndistortKeyPoints(k2, k4, dist).tobytes() == orc.undistort_keypoints(k2, k4, dist).tobytes() and \
        m.ComputeImageBounds(w, h, k4, dist) == orc.compute_image_bounds(w, h, k4, dist)
    # vocabulary
    kk, LL = int(rng.integers(2, 13)), int(rng.integers(1, 5))
    sc, wt = int(rng.integers(0, 6)), int(rng.integers(0, 4))
    par, leaf, vd, vw = make_vocabulary(kk, LL, seed=int(rng.integers(0, 1 << 20)), ragged=bool(rng.random() < 0.5), p_stop=float(rng.uniform(0, 0.3)), min_leaf_depth=LL)
    voc = pkg.ORBVocabulary(kk, LL, wt, sc, parent=par, is_leaf=leaf, desc=vd, weight=vw)
    ref = orc.Vocabulary(kk, LL, sc, wt, par, leaf, vd, vw)
    f = np.concatenate([features_near_words(vd, leaf, int(rng.integers(1, 3000)), int(rng.integers(0, 1000))), d2[: int(rng.integers(0, len(d2)))]])
    lu = int(rng.integers(0, LL + 2))
    o, e = voc.transform(f, lu), ref.transform(f, lu)
    ok3 = all(o[key].tobytes() == e[key].tobytes() for key in e)
    voc.close()
    # projection head (round 4): both modes, both camera models, random poses / depths / distance ranges, records bit for bit
    cam = (scene3d.PINHOLE, scene3d.KB8)[int(rng.integers(0, 2))]
    cam = (cam[0], tuple(float(v) * float(rng.uniform(0.8, 1.25)) if i < 4 else float(v) for i, v in enumerate(cam[1])))
    pts, q, t, R, Ow = scene3d.make_scene(rng, k1, cam, pkg.WORLD_POINT_DTYPE, motion=float(rng.uniform(0, 0.5)), rot=float(rng.uniform(0, 0.3)),
                                          behind=float(rng.uniform(0, 0.2)))
    a4 = m.ProjectLastFramePoints(q, t, cam[0], cam[1], pts); r4 = orc.project_last_frame_points(q, t, cam[0], cam[1], pts)
    lsf = float(np.log(np.float32(rng.uniform(1.1, 2.0)))); nlev = int(rng.integers(1, 9)); farp = bool(rng.random() < 0.5)
    args = (R, t, Ow, cam[0], cam[1], bounds, float(rng.uniform(0, 60)), float(rng.uniform(0.2, 0.8)), lsf, nlev, pts)
    thf = float(rng.uniform(2, 12))
    a5 = m.ProjectInFrustum(*args, far_points=farp, th_far_points=thf); r5 = orc.project_in_frustum(*args, far_points=farp, th_far_points=thf)
    ok4 = a4.tobytes() == r4.tobytes() and a5.tobytes() == r5.tobytes()
    # round 5: the keyframe-side head (orbx_window_point records; any combination of its three tests) and the fisheye-rig head (both cameras)
    sfl = (np.float32(rng.uniform(1.1, 2.0)) ** np.arange(nlev, dtype=np.float32)).astype(np.float32); tests = int(rng.integers(0, 8))
    wargs = (q, t, Ow, cam[0], cam[1], bounds, float(rng.uniform(0, 60)), float(rng.uniform(1, 8)), sfl, lsf, tests, pts)
    a6 = m.ProjectWindowPoints(*wargs); r6 = orc.project_window_points(*wargs)
    Rrl = scene3d.rotation_matrix(scene3d.quat_from_axis_angle(rng.normal(size=3), float(rng.uniform(0, 0.6)))); trl = rng.normal(0, 0.1, 3).astype(np.float32)
    tlr = (-(Rrl.astype(np.float64).T @ trl.astype(np.float64))).astype(np.float32)
    kb1 = tuple(float(v) * float(rng.uniform(0.9, 1.1)) for v in scene3d.KB8[1]); kb2 = tuple(float(v) * float(rng.uniform(0.9, 1.1)) for v in scene3d.KB8[1])
    fargs = (R, t, Ow, Rrl, trl, tlr, np.ascontiguousarray(R.T), kb1, kb2, bounds, float(rng.uniform(0.2, 0.8)), lsf, nlev, pts)
    a7 = m.ProjectInFrustumFisheye(*fargs, far_points=farp, th_far_points=thf); r7 = orc.project_in_frustum_fisheye(*fargs, far_points=farp, th_far_points=thf)
    ok4 = ok4 and a6.tobytes() == r6.tobytes() and a7[0].tobytes() == r7[0].tobytes() and a7[1].tobytes() == r7[1].tobytes()
    if not (ok0 and ok1 and ok2 and ok3 and ok4):
        bad += 1
        print("case %d MISMATCH: frame search %s, map-point search %s, undistort %s, vocabulary %s, projection head %s (k=%d L=%d scoring=%d weighting=%d n=%d levelsup=%d)" % (c, ok0, ok1, ok2, ok3, ok4, kk, LL, sc, wt, len(f), lu), flush=True)
    elif c % 10 == 0:
        print("case %d ok (%d x %d, %d / %d keypoints, %d matches; vocabulary k=%d L=%d, %d features -> %d words)" % (c, w, h, n1, len(k2), g[0], kk, LL, len(f), len(o["bow_word"])), flush=True)
print("cases: %d, mismatches: %d" % (cases, bad))
sys.exit(1 if bad else 0)
