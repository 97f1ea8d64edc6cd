"""GPU parity of ORBmatcher::SearchByProjection(Frame, Frame) against the CPU oracle: bit-exact nmatches and
the resulting CurrentFrame.mvpMapPoints assignment.  Inputs mimic Tracking::TrackWithMotionModel (src/Tracking4.cc:190-218): the
last frame's keypoints carry map points, their projections into the current frame are the keypoint positions moved by the image
motion plus a prediction error; th = 15 (monocular) or 7 (stereo)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOUNDS = (0.0, 640.0, 0.0, 480.0)


def _frames(pkg, seed, dx, dy):
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7)
    img1 = pkg.synth_frame(seed, 640, 480)
    rng = np.random.default_rng(seed)
    img2 = np.clip(np.roll(np.roll(img1, dy, axis=0), dx, axis=1).astype(np.int16) + rng.integers(-3, 4, img1.shape, dtype=np.int16), 0, 255)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    _, k2, d2 = ex(img2.astype(np.uint8), None, (0, 0)); k2, d2 = k2.copy(), d2.copy()
    sf = ex.GetScaleFactors().copy()
    ex.close()
    return k1, d1, k2, d2, sf, rng


def _projections(pkg, k1, rng, dx, dy, p_mp=0.7, p_obs=0.8, noise=3.0):
    pr = np.zeros(len(k1), pkg.PROJECTION_DTYPE)
    pr["u"] = k1["x"] + dx + rng.normal(0, noise, len(k1)).astype(np.float32)
    pr["v"] = k1["y"] + dy + rng.normal(0, noise, len(k1)).astype(np.float32)
    pr["invz"] = (1.0 / rng.uniform(0.5, 20.0, len(k1))).astype(np.float32)
    pr["invz"][rng.random(len(k1)) < 0.02] *= -1  # behind the camera
    pr["angle"] = k1["angle"]
    pr["octave"] = k1["octave"]
    has = rng.random(len(k1)) < p_mp
    obs = rng.random(len(k1)) < p_obs
    pr["flags"] = has.astype(np.int32) | (obs.astype(np.int32) << 1)
    return pr


@pytest.mark.parametrize("th,forward,backward,check,stereo", [(15, False, False, True, False), (7, False, False, True, True),
                                                             (7, True, False, True, True), (7, False, True, False, True),
                                                             (30, False, False, True, False), (2, False, False, True, False)])
def test_search_by_projection(pkg, orc, th, forward, backward, check, stereo):
    dx, dy = 6, -4
    k1, d1, k2, d2, sf, rng = _frames(pkg, 17, dx, dy)
    pr = _projections(pkg, k1, rng, dx, dy)
    state = rng.choice(np.int8([-1, -1, -1, 0, 1]), len(k2))  # some current keypoints already hold map points
    ur = None
    mbf = 0.0
    if stereo:
        ur = np.where(rng.random(len(k2)) < 0.6, k2["x"] - rng.uniform(0.5, 40, len(k2)), -1).astype(np.float32)
        mbf = 40.0
    m = pkg.ORBmatcher(8192, 8192)
    n, assign = m.SearchByProjection(k2, d2, state, BOUNDS, sf, pr, d1, th, u_right=ur, mbf=mbf, forward=forward, backward=backward,
                                     checkOrientation=check)
    rn, rassign = orc.search_by_projection(k2, d2, state, BOUNDS, sf, pr, d1, th, ur, mbf, forward, backward, check)
    assert n == rn and np.array_equal(assign, rassign)
    if th >= 7 and not stereo:
        assert n > 50
    m.close()


def test_search_by_projection_conflicts_and_edges(pkg, orc):
    """Many projections compete for few keypoints (every projection lands on one of 40 spots): long chains of "an earlier point with
    observations took it"; points without observations do not block and are overwritten by later takers."""
    k1, d1, k2, d2, sf, rng = _frames(pkg, 23, 0, 0)
    pr = _projections(pkg, k1, rng, 0, 0, p_mp=1.0, p_obs=0.5, noise=0.5)
    spots = rng.integers(0, len(k2), 40)
    pick = spots[rng.integers(0, 40, len(k1))]
    pr["u"], pr["v"] = k2["x"][pick], k2["y"][pick]
    pr["octave"] = k2["octave"][pick]
    d1b = d2[pick].copy()
    d1b[:, :2] ^= rng.integers(0, 256, (len(k1), 2), dtype=np.uint8)  # near-identical descriptors: distance ties and near ties
    state = np.full(len(k2), -1, np.int8)
    m = pkg.ORBmatcher(8192, 8192)
    for th in (15, 40):
        n, assign = m.SearchByProjection(k2, d2, state, BOUNDS, sf, pr, d1b, th)
        rn, rassign = orc.search_by_projection(k2, d2, state, BOUNDS, sf, pr, d1b, th)
        assert n == rn and np.array_equal(assign, rassign)
    # no projections / no valid flags / empty current frame
    n, assign = m.SearchByProjection(k2, d2, state, BOUNDS, sf, pr[:0], d1b[:0], 15)
    assert n == 0 and (assign == -1).all()
    pr2 = pr.copy(); pr2["flags"] = 0
    n, assign = m.SearchByProjection(k2, d2, state, BOUNDS, sf, pr2, d1b, 15)
    assert n == 0 and (assign == -1).all()
    n, assign = m.SearchByProjection(k2[:0], d2[:0], state[:0], BOUNDS, sf, pr, d1b, 15)
    assert n == 0 and len(assign) == 0
    with pytest.raises(pkg.OrbxError, match="octave"):
        pr3 = pr.copy(); pr3["octave"][0] = 9
        m.SearchByProjection(k2, d2, state, BOUNDS, sf, pr3, d1b, 15)
    m.close()


def _map_points(pkg, k1, rng, dx, dy, noise=1.5):
    """Frame::isInFrustum outputs for the local map points: projections near the keypoints of another view, predicted level near the
    keypoint's octave, viewing cosines on both sides of RadiusByViewingCos' 0.998 (and exactly float(0.998))."""
    pr = np.zeros(len(k1), pkg.PROJECTION_DTYPE)
    pr["u"] = k1["x"] + dx + rng.normal(0, noise, len(k1)).astype(np.float32)
    pr["v"] = k1["y"] + dy + rng.normal(0, noise, len(k1)).astype(np.float32)
    pr["angle"] = rng.choice(np.float32([0.5, 0.9, 0.99, 0.998, 0.9981, 0.9999, 1.0]), len(k1))
    pr["octave"] = np.clip(k1["octave"] + rng.integers(-1, 2, len(k1)), 0, 7)
    pr["invz"] = pr["u"] - rng.uniform(0.5, 40, len(k1)).astype(np.float32)  # mTrackProjXR
    inview = rng.random(len(k1)) < 0.85
    obs = rng.random(len(k1)) < 0.8
    pr["flags"] = inview.astype(np.int32) | (obs.astype(np.int32) << 1)
    return pr


@pytest.mark.parametrize("th,nnratio,stereo", [(1.0, 0.8, False), (3.0, 0.8, False), (5.0, 0.8, True), (15.0, 0.6, False), (1.0, 0.9, True)])
def test_search_by_projection_map_points(pkg, orc, th, nnratio, stereo):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) as Tracking::SearchLocalPoints calls it (matcher(0.8), th 1 / 3 / 5 / 15)."""
    dx, dy = 3, 2
    k1, d1, k2, d2, sf, rng = _frames(pkg, 31, dx, dy)
    pr = _map_points(pkg, k1, rng, dx, dy)
    state = rng.choice(np.int8([-1, -1, -1, 0, 1]), len(k2))
    ur = np.where(rng.random(len(k2)) < 0.6, k2["x"] - rng.uniform(0.5, 40, len(k2)), -1).astype(np.float32) if stereo else None
    m = pkg.ORBmatcher(8192, 8192)
    n, assign = m.SearchByProjectionPoints(k2, d2, state, BOUNDS, sf, pr, d1, th, nnratio, u_right=ur)
    rn, rassign = orc.search_by_projection_points(k2, d2, state, BOUNDS, sf, pr, d1, th, nnratio, ur)
    assert n == rn and np.array_equal(assign, rassign)
    if not stereo:
        assert n > 50
    m.close()


def test_search_by_projection_map_points_ratio_and_conflicts(pkg, orc):
    """Dense competition: several map points per keypoint with near-identical descriptors, so the same-level ratio test and the
    "already has a point with observations" rule both fire; plus empty inputs."""
    k1, d1, k2, d2, sf, rng = _frames(pkg, 37, 0, 0)
    n1 = len(k1)
    pr = _map_points(pkg, k1, rng, 0, 0, noise=0.3)
    spots = rng.integers(0, len(k2), 60)
    pick = spots[rng.integers(0, 60, n1)]
    pr["u"], pr["v"] = k2["x"][pick], k2["y"][pick]
    pr["octave"] = np.clip(k2["octave"][pick] + rng.integers(0, 2, n1), 0, 7)
    dmp = d2[pick].copy()
    dmp[:, :3] ^= rng.integers(0, 256, (n1, 3), dtype=np.uint8)
    state = np.full(len(k2), -1, np.int8)
    m = pkg.ORBmatcher(8192, 8192)
    tot = 0
    for th, ratio in ((1.0, 0.8), (4.0, 0.8), (4.0, 0.5), (15.0, 1.0)):
        n, assign = m.SearchByProjectionPoints(k2, d2, state, BOUNDS, sf, pr, dmp, th, ratio)
        rn, rassign = orc.search_by_projection_points(k2, d2, state, BOUNDS, sf, pr, dmp, th, ratio)
        assert n == rn and np.array_equal(assign, rassign)
        tot += n
    assert tot > 0
    n, assign = m.SearchByProjectionPoints(k2, d2, state, BOUNDS, sf, pr[:0], dmp[:0], 1.0, 0.8)
    assert n == 0 and (assign == -1).all()
    n, assign = m.SearchByProjectionPoints(k2[:0], d2[:0], state[:0], BOUNDS, sf, pr, dmp, 1.0, 0.8)
    assert n == 0 and len(assign) == 0
    m.close()


@pytest.mark.parametrize("distorted,stereo", [(True, False), (False, True)])
def test_search_by_projection_frames_batch(pkg, orc, distorted, stereo):
    """orbx_search_by_projection_frames: extractor outputs -> UndistortKeyPoints -> grid -> SearchByProjection(Frame, Frame) for a batch of
    frames without leaving HBM == the oracle's undistort_keypoints + search_by_projection per frame (bit-exact keypoints and assignment)."""
    P, w, h = 5, 640, 480
    rng = np.random.default_rng(23)
    shifts = [(6, -4), (3, 5), (-7, 2), (0, 0), (10, 9)]
    last = np.stack([pkg.synth_frame(600 + i, w, h) for i in range(P)])
    cur = np.stack([np.clip(np.roll(np.roll(last[i], shifts[i][1], axis=0), shifts[i][0], axis=1).astype(np.int16)
                            + rng.integers(-3, 4, last[i].shape, dtype=np.int16), 0, 255).astype(np.uint8) for i in range(P)])
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    nl, _, kl, dl = ex.extract_batch(last, (0, 0))
    kl, dl = kl.copy(), dl.copy()
    nc, _, kc, dc = ex.extract_batch(cur, (0, 0))          # the extractor's device buffers now hold the CURRENT frames
    sf = ex.GetScaleFactors().copy()
    k4 = (458.654, 457.296, 367.215, 248.375)
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05) if distorted else ()
    m = pkg.ORBmatcher(4096, 4096)
    bounds = m.ComputeImageBounds(w, h, k4, dist) if hasattr(m, "ComputeImageBounds") else orc.compute_image_bounds(w, h, k4, dist)
    th, mbf = (7.0, 40.0) if stereo else (15.0, 0.0)
    proj, pdesc, ur_all = [], [], np.full((P, ex.capacity), -1, np.float32)
    for p in range(P):
        k1 = orc.undistort_keypoints(kl[p, : nl[p]], k4, dist)
        pr = _projections(pkg, k1, rng, shifts[p][0], shifts[p][1])
        proj.append(pr); pdesc.append(dl[p, : nl[p]])
        if stereo:
            ur_all[p, : nc[p]] = np.where(rng.random(nc[p]) < 0.6, kc[p, : nc[p]]["x"] - rng.uniform(0.5, 40, nc[p]), -1)
    nm, assign, kun = m.SearchByProjectionFrames(ex, P, k4, dist, bounds, mbf, proj, pdesc, th, u_right=ur_all if stereo else None)
    for p in range(P):
        rk = orc.undistort_keypoints(kc[p, : nc[p]], k4, dist)
        assert kun[p, : nc[p]].tobytes() == rk.tobytes(), p
        state = np.full(nc[p], -1, np.int8)
        rn, rassign = orc.search_by_projection(rk, dc[p, : nc[p]], state, bounds, sf, proj[p], pdesc[p], th, ur_all[p, : nc[p]] if stereo else None,
                                               mbf, False, False, True)
        assert nm[p] == rn and np.array_equal(assign[p, : nc[p]], rassign), p
        assert rn > 50
    m.close(); ex.close()


def test_search_by_projection_points_frames_batch(pkg, orc):
    """orbx_search_by_projection_points_frames: Tracking::SearchLocalPoints' matcher for a batch of frames on the extractor's device outputs ==
    the oracle's search_by_projection_points per frame."""
    P, w, h = 4, 640, 480
    rng = np.random.default_rng(31)
    imgs = np.stack([pkg.synth_frame(700 + i, w, h) for i in range(P)])
    other = np.stack([np.roll(imgs[i], 3 + i, axis=1) for i in range(P)])
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    no, _, ko, do = ex.extract_batch(other, (0, 0)); ko, do = ko.copy(), do.copy()
    nc, _, kc, dc = ex.extract_batch(imgs, (0, 0))
    sf = ex.GetScaleFactors().copy()
    k4, dist = (500.0, 500.0, 320.0, 240.0), ()
    bounds = (0.0, float(w), 0.0, float(h))
    mps = [_map_points(pkg, ko[p, : no[p]], rng, -(3 + p), 0) for p in range(P)]
    descs = [do[p, : no[p]] for p in range(P)]
    m = pkg.ORBmatcher(4096, 4096)
    nm, assign, kun = m.SearchByProjectionFrames(ex, P, k4, dist, bounds, 0.0, mps, descs, 3.0, map_points_nnratio=0.8)
    for p in range(P):
        assert kun[p, : nc[p]].tobytes() == kc[p, : nc[p]].tobytes()       # no distortion: mvKeysUn = mvKeys
        state = np.full(nc[p], -1, np.int8)
        rn, rassign = orc.search_by_projection_points(kc[p, : nc[p]], dc[p, : nc[p]], state, bounds, sf, mps[p], descs[p], 3.0, 0.8)
        assert nm[p] == rn and np.array_equal(assign[p, : nc[p]], rassign), p
        assert rn > 30
    m.close(); ex.close()


def test_global_resolve_kernel_in_subprocess(pkg):
    """k_sbp_resolve keeps the owner arrays and the per-point state of frames up to 12288 keypoints in LDS; larger frames go through
    k_sbp_resolve_walk (global memory).  ORBX_SBP_RESOLVE_GLOBAL (read once per process) forces it: this file's parity tests once more."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_SBP_RESOLVE_GLOBAL"):
        pytest.skip("already the forced run")
    env = dict(os.environ, ORBX_SBP_RESOLVE_GLOBAL="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not subprocess"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.parametrize("entries", ["0", "4"])
def test_resolve_list_entries_in_lds_in_subprocess(pkg, entries):
    """k_sbp_resolve keeps the first 8 entries of a point's sorted candidate list in registers, the next (up to 24) in LDS, and walks on
    in global memory behind them.  ORBX_SBP_LIST_LDS (read once per process) shortens the LDS part to 0 or 4 entries, so that the walk
    crosses both borders for many points: this file's parity tests once more under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_SBP_LIST_LDS") or os.environ.get("ORBX_SBP_RESOLVE_GLOBAL"):
        pytest.skip("already a forced run")
    env = dict(os.environ, ORBX_SBP_LIST_LDS=entries)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not subprocess"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


def test_search_by_projection_large_frame(pkg, orc):
    """12000 keypoints and 12000 projections on a fresh matcher: the call's packed input (1.5 MB) outgrows the page-locked staging
    while it is being filled (the pieces packed so far move with it), the owner arrays take 94 KB of the resolve kernel's LDS and no
    list entries fit behind them; then the same matcher takes a small frame again."""
    rng = np.random.default_rng(4242)
    n = 12000
    k2 = np.zeros(n, pkg.KEYPOINT_DTYPE)
    k2["x"] = rng.uniform(20, 1220, n).astype(np.float32); k2["y"] = rng.uniform(20, 350, n).astype(np.float32)
    k2["octave"] = rng.integers(0, 8, n); k2["angle"] = rng.uniform(0, 360, n).astype(np.float32); k2["size"] = 31
    d2 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    src = rng.permutation(n)
    pr = np.zeros(n, pkg.PROJECTION_DTYPE)
    pr["u"] = k2["x"][src] + rng.normal(0, 2, n).astype(np.float32); pr["v"] = k2["y"][src] + rng.normal(0, 2, n).astype(np.float32)
    pr["invz"] = 0.1; pr["angle"] = k2["angle"][src]; pr["octave"] = k2["octave"][src]
    pr["flags"] = 1 | ((rng.random(n) < 0.7).astype(np.int32) << 1)
    d1 = d2[src].copy()
    d1[:, :3] ^= rng.integers(0, 256, (n, 3), dtype=np.uint8)  # ~12 bits away from the true partner
    state = rng.choice(np.int8([-1, -1, -1, 0, 1]), n)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    m = pkg.ORBmatcher(16384, 16384)
    nm, assign = m.SearchByProjection(k2, d2, state, bounds, sf, pr, d1, 15)
    rn, rassign = orc.search_by_projection(k2, d2, state, bounds, sf, pr, d1, 15)
    assert nm == rn and np.array_equal(assign, rassign) and nm > 4000
    nm, assign = m.SearchByProjection(k2[:900], d2[:900], state[:900], bounds, sf, pr[:700], d1[:700], 15)
    rn, rassign = orc.search_by_projection(k2[:900], d2[:900], state[:900], bounds, sf, pr[:700], d1[:700], 15)
    assert nm == rn and np.array_equal(assign, rassign)
    m.close()
