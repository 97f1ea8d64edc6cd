"""ORBmatcher::Fuse (reference src/ORBmatcher2.cc:420-610 and the Sim3 overload :612-770) and ORBmatcher::SearchBySim3
(src/ORBmatcher3.cc:37-254): the per-map-point window search over KeyFrame::GetFeaturesInArea.  Hand-traced cases for the oracle (no GPU),
then the HIP path through the C ABI against the oracle."""
import numpy as np
import pytest

LEVELS = 8
SF = np.float32(1.2) ** np.arange(LEVELS, dtype=np.float32)
INV_SIGMA2 = (1.0 / (SF * SF)).astype(np.float32)
BOUNDS = (0.0, 752.0, 0.0, 480.0)


def _keys(pkg, pts, octaves):
    k = np.zeros(len(pts), pkg.KEYPOINT_DTYPE)
    p = np.asarray(pts, np.float32).reshape(-1, 2)
    k["x"], k["y"], k["octave"], k["size"] = p[:, 0], p[:, 1], octaves, 31
    return k


def _flip(d, bits):
    d = d.copy()
    for b in bits:
        d[b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def _points(orc, rows):
    p = np.zeros(len(rows), orc.WINDOW_POINT_DTYPE)
    for i, r in enumerate(rows):
        p[i] = r
    return p


def test_oracle_hand_traced_window_search(pkg, orc):
    """One map point at (100, 100), level 2, radius 10: candidates by grid order (cells are 11.75 x 10 px: ix outer, iy inner).
    Worked out by hand from src/ORBmatcher2.cc:522-579 and src/KeyFrame.cc:704-748."""
    rng = np.random.default_rng(3)
    D = rng.integers(0, 256, 32, dtype=np.uint8)
    #   idx 0: exact twin but level 0 (outside [1, 2]);  1: distance 4, level 2, e2 = 5, cell (9, 10);  2: distance 4, level 1, e2 = 2, cell
    #   (8, 10) - the EARLIER cell, so it wins the tie;  3: distance 2 but 10.5 px away in x (|dx| < r fails);  4: distance 3, level 2, fails
    #   the chi2 test only (4 px off: 16 / 2.0736 = 7.7 > 5.99);  5: distance 60 (never the best)
    k = _keys(pkg, [(100, 100), (102, 101), (99, 99), (110.5, 100), (104, 100), (99.5, 100)], [0, 2, 1, 2, 2, 2])
    d = np.stack([D, _flip(D, [1, 2, 3, 4]), _flip(D, [9, 10, 11, 12]), _flip(D, [5, 6]), _flip(D, [20, 21, 22]), _flip(D, range(60))])
    pts = _points(orc, [(100, 100, 0, 10, 2, 1)])
    bi, bd = orc.search_keyframe_windows(k, d, BOUNDS, pts, D[None], inv_level_sigma2=INV_SIGMA2, chi2_test=True)
    assert (bi[0], bd[0]) == (2, 4)
    bi, bd = orc.search_keyframe_windows(k, d, BOUNDS, pts, D[None], chi2_test=False)   # the Sim3 overload has no reprojection test
    assert (bi[0], bd[0]) == (4, 3)
    # stereo keypoint: er enters e2 and the bound becomes 7.8: keypoint 2 (1 px, 1 px off, level 1) with ur 5 px off: (1 + 1 + 25) / 1.44 = 18.75
    ur = np.full(6, -1, np.float32); ur[2] = 90
    pts = _points(orc, [(100, 100, 95, 10, 2, 1)])
    bi, bd = orc.search_keyframe_windows(k, d, BOUNDS, pts, D[None], u_right=ur, inv_level_sigma2=INV_SIGMA2, chi2_test=True)
    assert (bi[0], bd[0]) == (1, 4)
    pts = _points(orc, [(100, 100, 0, 10, 2, 0), (400, 400, 0, 10, 2, 1)])              # not searched / an empty window
    bi, bd = orc.search_keyframe_windows(k, d, BOUNDS, pts, np.stack([D, D]))
    assert bi.tolist() == [-1, -1] and (bd == np.iinfo(np.int32).max).all()


def _scene(pkg, orc, n, np_, seed, dense=False):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.uniform(5, 747, n), rng.uniform(5, 475, n)], 1)
    if dense:
        pos[: n // 2] = np.stack([rng.uniform(300, 340, n // 2), rng.uniform(200, 240, n // 2)], 1)  # hundreds of keypoints in a few cells
    octs = rng.integers(0, LEVELS, n)
    k = _keys(pkg, pos, octs)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d[rng.integers(0, n, n // 10)] = d[rng.integers(0, n, n // 10)]   # twins: equal distances inside a window
    src = rng.integers(0, n, np_)
    pts = np.zeros(np_, orc.WINDOW_POINT_DTYPE)
    pts["u"] = pos[src, 0] + rng.normal(0, 2.0, np_); pts["v"] = pos[src, 1] + rng.normal(0, 2.0, np_)
    pts["level"] = np.clip(octs[src] + rng.integers(0, 2, np_), 0, LEVELS - 1)
    pts["radius"] = (rng.choice([3.0, 4.0, 7.5, 30.0], np_) * SF[pts["level"]]).astype(np.float32)
    pts["ur"] = pts["u"] - rng.uniform(1, 30, np_)
    pts["search"] = rng.random(np_) < 0.9
    far = rng.random(np_) < 0.05
    pts["u"][far] += 2000   # outside the grid
    pd = np.stack([_flip(d[s], rng.integers(0, 256, int(rng.integers(0, 70)))) for s in src])
    ur = np.where(rng.random(n) < 0.5, pos[:, 0] - rng.uniform(1, 30, n), -1).astype(np.float32)
    return k, d, ur, pts, pd


@pytest.mark.gpu
@pytest.mark.parametrize("n,np_,dense", [(2000, 1500, False), (3000, 800, True), (1, 5, False), (300, 1, False)])
def test_window_search_and_fuse_match_oracle(pkg, orc, n, np_, dense):
    k, d, ur, pts, pd = _scene(pkg, orc, n, np_, seed=n + np_, dense=dense)
    m = pkg.ORBmatcher(8192, 8192)
    for chi2, u in ((False, None), (True, None), (True, ur)):
        got = m.SearchKeyFrameWindows(k, d, BOUNDS, pts, pd, LEVELS, u_right=u, inv_level_sigma2=INV_SIGMA2, chi2_test=chi2)
        ref = orc.search_keyframe_windows(k, d, BOUNDS, pts, pd, u_right=u, inv_level_sigma2=INV_SIGMA2, chi2_test=chi2)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        nf, fi = m.Fuse(k, d, BOUNDS, pts, pd, LEVELS, u_right=u, inv_level_sigma2=INV_SIGMA2, chi2_test=chi2)
        exp = np.where(ref[1] <= 50, ref[0], -1)
        assert np.array_equal(fi, exp) and nf == int((exp >= 0).sum())
    if n >= 2000:
        assert (got[0] >= 0).sum() > np_ // 4
    m.close()


@pytest.mark.gpu
def test_search_by_sim3_matches_oracle(pkg, orc):
    """Two keyframes that see the same points: every feature carries a map point whose descriptor is near the feature's; projections of
    KF1's points into KF2 land near their partner keypoint (and vice versa), some partners are missing or at the wrong level."""
    rng = np.random.default_rng(21)
    n = 1800
    pos1 = np.stack([rng.uniform(5, 747, n), rng.uniform(5, 475, n)], 1)
    perm = rng.permutation(n)
    pos2 = pos1[perm] + rng.normal(0, 1.5, (n, 2))
    oct1 = rng.integers(0, LEVELS, n); oct2 = np.clip(oct1[perm] + (rng.random(n) < 0.15), 0, LEVELS - 1)
    k1, k2 = _keys(pkg, pos1, oct1), _keys(pkg, pos2, oct2)
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    fl = lambda D, mx: np.stack([_flip(x, rng.integers(0, 256, int(rng.integers(0, mx)))) for x in D])
    d1, d2 = fl(base, 30), fl(base[perm], 30)
    mp1, mp2 = fl(base, 20), fl(base[perm], 20)
    inv = np.argsort(perm)
    p12 = np.zeros(n, orc.WINDOW_POINT_DTYPE); p21 = np.zeros(n, orc.WINDOW_POINT_DTYPE)
    p12["u"], p12["v"] = (pos2[inv] + rng.normal(0, 2, (n, 2))).T   # where KF1's map point i falls in KF2
    p21["u"], p21["v"] = (pos1[perm] + rng.normal(0, 2, (n, 2))).T
    p12["level"] = oct1; p21["level"] = oct1[perm]
    p12["radius"] = 7.5 * SF[p12["level"]]; p21["radius"] = 7.5 * SF[p21["level"]]
    p12["search"] = rng.random(n) < 0.85; p21["search"] = rng.random(n) < 0.85
    m = pkg.ORBmatcher(8192, 8192)
    got = m.SearchBySim3(k1, d1, BOUNDS, k2, d2, BOUNDS, p12, mp1, p21, mp2, LEVELS)
    ref = orc.search_by_sim3(k1, d1, BOUNDS, k2, d2, BOUNDS, p12, mp1, p21, mp2)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[0] > n // 4
    ok = got[1] >= 0
    assert np.array_equal(perm[got[1][ok]], np.flatnonzero(ok)) or (perm[got[1][ok]] == np.flatnonzero(ok)).mean() > 0.95
    m.close()


def test_oracle_distinctive_descriptor_hand_traced(orc):
    """MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:364-398).  Four descriptors A, A^1bit, A^2bits (disjoint from the
    first), B far away: distance rows sorted = A: [0 1 2 128+], A1: [0 1 3 ..], A2: [0 2 3 ..], B: [0 x x x]; the median is element
    (int)(0.5 * 3) = 1 -> 1, 1, 2, big: the FIRST row with median 1 wins."""
    A = np.zeros(32, np.uint8)
    rows = np.stack([A, _flip(A, [0]), _flip(A, [8, 9]), np.full(32, 0xFF, np.uint8)])
    assert orc.distinctive_descriptors(rows, [0, 4]).tolist() == [0]
    assert orc.distinctive_descriptors(rows[[3, 2, 1, 0]], [0, 4]).tolist() == [2]      # order reversed: A1 (median 1) comes before A
    assert orc.distinctive_descriptors(rows, [0, 0, 1, 4]).tolist() == [-1, 0, 0]       # no observation / one / three ([A1 A2 B]: medians 3 3 254)
    assert orc.distinctive_descriptors(rows, [0, 2]).tolist() == [0]                    # N = 2: median = element 0 = 0 for both rows


@pytest.mark.gpu
def test_distinctive_descriptors_match_oracle(pkg, orc):
    rng = np.random.default_rng(31)
    counts = np.concatenate([rng.integers(0, 12, 400), [64, 65, 130, 1, 0, 300]])
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    base = rng.integers(0, 256, (len(counts), 32), dtype=np.uint8)
    desc = np.concatenate([np.stack([_flip(base[p], rng.integers(0, 256, int(rng.integers(0, 40)))) for _ in range(c)]) if c else
                           np.zeros((0, 32), np.uint8) for p, c in enumerate(counts)])
    desc[start[3]:start[4]] = desc[start[3]]   # a point whose observations are all equal: every median 0 -> row 0
    m = pkg.ORBmatcher(64, 64)
    got = m.ComputeDistinctiveDescriptors(desc, start)
    assert np.array_equal(got, orc.distinctive_descriptors(desc, start))
    assert got[3] <= 0 and (got[counts == 0] == -1).all()
    m.close()


def test_oracle_projection_sim3_hand_traced(pkg, orc):
    """src/ORBmatcher1.cc:494-523: three points want keypoint 1 (the twin of their descriptor).  Point 0 takes it; point 1 then takes its
    second choice (keypoint 2, distance 5); point 2 finds both taken and its third choice is at distance 60 > TH_LOW -> no match.  With
    keypoint 1 matched on entry, point 0 takes keypoint 2 and the others nothing; ratioHamming 0.09 (bound 4.5) accepts distance <= 4."""
    rng = np.random.default_rng(5)
    D = rng.integers(0, 256, 32, dtype=np.uint8)
    k = _keys(pkg, [(200, 200), (201, 200), (202, 201), (203, 199)], [1, 1, 1, 1])
    d = np.stack([_flip(D, range(60)), D, _flip(D, [1, 2, 3, 4, 5]), _flip(D, range(100, 170))])
    pts = _points(orc, [(201, 200, 0, 8, 1, 1)] * 3)
    pd = np.stack([D, D, D])
    n, nm = orc.search_by_projection_sim3(k, d, BOUNDS, pts, pd, [0, 0, 0, 0])
    assert n == 2 and nm.tolist() == [-1, 0, 1, -1]
    n, nm = orc.search_by_projection_sim3(k, d, BOUNDS, pts, pd, [0, 1, 0, 0])
    assert n == 1 and nm.tolist() == [-1, -1, 0, -1]
    n, nm = orc.search_by_projection_sim3(k, d, BOUNDS, pts, pd, [0, 0, 0, 0], ratio_hamming=0.09)
    assert n == 1 and nm.tolist() == [-1, 0, -1, -1]


@pytest.mark.gpu
@pytest.mark.parametrize("n,np_,dense,ratio", [(2000, 1500, False, 1.0), (1500, 3000, True, 1.0), (800, 800, False, 0.6)])
def test_projection_sim3_matches_oracle(pkg, orc, n, np_, dense, ratio):
    """Many points per keypoint (np > n, dense clusters): chains of "an earlier point took it" that the fixed-point iteration must resolve
    exactly like the reference's in-order walk."""
    k, d, ur, pts, pd = _scene(pkg, orc, n, np_, seed=7 * n + np_, dense=dense)
    rng = np.random.default_rng(n)
    matched = (rng.random(n) < 0.2).astype(np.uint8)
    m = pkg.ORBmatcher(8192, 8192)
    got = m.SearchByProjectionSim3(k, d, BOUNDS, pts, pd, matched, ratio, LEVELS)
    ref = orc.search_by_projection_sim3(k, d, BOUNDS, pts, pd, matched, ratio)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    assert got[0] > 50 and not (got[1][matched == 1] >= 0).any()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,np_,dense,dist,check", [(2000, 1500, False, 100, True), (1500, 2500, True, 64, True), (900, 900, False, 100, False)])
def test_projection_keyframe_matches_oracle(pkg, orc, n, np_, dense, dist, check):
    """Tracking::Relocalization's SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): levels nPredictedLevel - 1 .. + 1, taken
    keypoints skipped in walk order, rotation histogram."""
    k, d, ur, pts, pd = _scene(pkg, orc, n, np_, seed=11 * n + np_, dense=dense)
    rng = np.random.default_rng(n + 1)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    pa = ((rng.normal(10, 4, np_) + np.where(rng.random(np_) < 0.2, rng.uniform(0, 360, np_), 0)) % 360).astype(np.float32)
    has = (rng.random(n) < 0.25).astype(np.uint8)
    m = pkg.ORBmatcher(8192, 8192)
    got = m.SearchByProjectionKeyFrame(k, d, BOUNDS, pts, pd, pa, has, dist, check, LEVELS)
    ref = orc.search_by_projection_keyframe(k, d, BOUNDS, pts, pd, pa, has, dist, check)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[0] == int((got[1] >= 0).sum()) and got[0] > 30
    m.close()


@pytest.mark.gpu
def test_projection_rounds_on_the_host_in_subprocess(pkg):
    """Both projection searches run their rounds inside one launch (sorted candidate lists + k_sbp_resolve); a window with more than 512
    candidates falls back to rounds of the window search with a round trip each.  ORBX_PROJECTION_HOST_ROUNDS (read once per process)
    forces that fallback: the two parity tests above once more under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_PROJECTION_HOST_ROUNDS"):
        pytest.skip("already the forced run")
    env = dict(os.environ, ORBX_PROJECTION_HOST_ROUNDS="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "projection and matches_oracle"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.gpu
def test_projection_sim3_window_beyond_the_list_capacity(pkg, orc):
    """600 keypoints inside one window (more than a candidate list holds): the call falls back to the rounds of the window search by
    itself and still equals the oracle."""
    rng = np.random.default_rng(77)
    n = 900
    k = np.zeros(n, pkg.KEYPOINT_DTYPE)
    k["x"] = 300 + rng.uniform(-12, 12, n).astype(np.float32); k["y"] = 200 + rng.uniform(-12, 12, n).astype(np.float32)
    k["x"][600:] = rng.uniform(20, 620, n - 600).astype(np.float32); k["y"][600:] = rng.uniform(20, 460, n - 600).astype(np.float32)
    k["octave"] = 1; k["size"] = 31
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pts = _points(orc, [(300, 200, 0, 20, 1, 1)] * 40 + [(float(k["x"][i]), float(k["y"][i]), 0, 6, 1, 1) for i in range(600, 700)])
    src = np.concatenate([rng.integers(0, 600, 40), np.arange(600, 700)])
    pd = d[src].copy(); pd[:, 0] ^= rng.integers(0, 8, len(src), dtype=np.uint8)
    matched = np.zeros(n, np.uint8)
    m = pkg.ORBmatcher(8192, 8192)
    got = m.SearchByProjectionSim3(k, d, BOUNDS, pts, pd, matched, 1.0, LEVELS)
    ref = orc.search_by_projection_sim3(k, d, BOUNDS, pts, pd, matched, 1.0)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[0] > 100
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("distorted", [False, True])
def test_fuse_and_window_search_over_a_batch_of_keyframes(pkg, orc, distorted):
    """orbx_fuse_frames / orbx_search_keyframe_windows_frames: K keyframes = the frames of an extractor batch, map points projected into each of
    them, everything device-resident; per keyframe the result equals the single-keyframe oracle on that frame's (undistorted) keypoints -
    LocalMapping::SearchInNeighbors' shape (reference src/LocalMapping.cc:772-803 around src/ORBmatcher2.cc:420-610)."""
    import torch
    K = 5
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=K)
    imgs = np.stack([pkg.synth_frame(50 + f, 640, 480) for f in range(K)])
    d_imgs = torch.from_numpy(imgs).to("cuda:0")
    ex.extract_batch_device(d_imgs.data_ptr(), K, 640, 480, 640, 640 * 480, (0, 0)); ex.sync()
    n, _, kk, dd = ex.download_outputs(K)
    inv_sigma2 = ex.GetInverseScaleSigmaSquares().copy()
    sf = ex.GetScaleFactors().copy()
    k4 = np.float32([458.654, 457.296, 367.215, 248.375])
    dist = np.float32([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]) if distorted else np.zeros(0, np.float32)
    m = pkg.ORBmatcher(8192, 8192)
    bounds = m.ComputeImageBounds(640, 480, k4, dist) if distorted else (0.0, 640.0, 0.0, 480.0)
    rng = np.random.default_rng(9)
    pts_all, pd_all, ur_all, kun_all = [], [], np.full((K, ex.capacity), -1, np.float32), []
    for f in range(K):
        nc = int(n[f])
        kun = orc.undistort_keypoints(kk[f, :nc], k4, dist) if distorted else kk[f, :nc]
        kun_all.append(kun)
        np_ = int(rng.integers(1, 1500)) if f else 0         # keyframe 0: no points at all
        src = rng.integers(0, nc, np_)
        pts = np.zeros(np_, orc.WINDOW_POINT_DTYPE)
        pts["u"] = kun["x"][src] + rng.normal(0, 2.0, np_); pts["v"] = kun["y"][src] + rng.normal(0, 2.0, np_)
        pts["level"] = np.clip(kun["octave"][src] + rng.integers(0, 2, np_), 0, 7)
        pts["radius"] = (rng.choice([3.0, 4.0, 7.5], np_) * sf[pts["level"]]).astype(np.float32)
        pts["ur"] = pts["u"] - rng.uniform(1, 30, np_)
        pts["search"] = rng.random(np_) < 0.9
        pd = np.stack([_flip(dd[f, s], rng.integers(0, 256, int(rng.integers(0, 70)))) for s in src]) if np_ else np.zeros((0, 32), np.uint8)
        ur_all[f, :nc] = np.where(rng.random(nc) < 0.5, kun["x"] - rng.uniform(1, 30, nc), -1)
        pts_all.append(pts); pd_all.append(pd)
    for chi2, ur in ((False, None), (True, None), (True, ur_all)):
        bi, bd = m.FuseFrames(ex, K, k4, dist, bounds, pts_all, pd_all, chi2_test=chi2, u_right=ur, fuse=False)
        fi, nf = m.FuseFrames(ex, K, k4, dist, bounds, pts_all, pd_all, chi2_test=chi2, u_right=ur, fuse=True)
        for f in range(K):
            nc, np_ = int(n[f]), len(pts_all[f])
            ref = orc.search_keyframe_windows(kun_all[f], dd[f, :nc], bounds, pts_all[f], pd_all[f], u_right=None if ur is None else ur[f, :nc],
                                              inv_level_sigma2=inv_sigma2, chi2_test=chi2)
            assert np.array_equal(bi[f, :np_], ref[0]) and np.array_equal(bd[f, :np_], ref[1]), (chi2, f)
            exp = np.where(ref[1] <= 50, ref[0], -1)
            assert np.array_equal(fi[f, :np_], exp) and nf[f] == int((exp >= 0).sum()), (chi2, f)
        assert nf[0] == 0 and nf[1:].sum() > 100
    m.close(); ex.close()
