"""GPU parity of the frame grid and ORBmatcher::SearchForInitialization against the CPU oracle:
bit-exact grid, vnMatches12, nmatches and updated vbPrevMatched.  The inputs are what the reference's monocular initializer
hands over (src/Tracking3.cc:700-741): the keypoints / descriptors of two frames from the 5 x nFeatures extractor,
vbPrevMatched = the first frame's keypoint positions, window 100, ORBmatcher(0.9, true)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOUNDS = (0.0, 640.0, 0.0, 480.0)  # Frame::mnMinX, mnMaxX, mnMinY, mnMaxY without distortion (src/Frame.cc:832-837)


def _second_view(img, dx, dy, seed):
    """The same scene shifted by (dx, dy) px with fresh sensor noise: matches exist but are not trivial."""
    rng = np.random.default_rng(seed)
    out = np.roll(np.roll(img, dy, axis=0), dx, axis=1).astype(np.int16)
    out += rng.integers(-3, 4, out.shape, dtype=np.int16)
    return np.clip(out, 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def two_frames(pkg):
    ex = pkg.ORBextractor(5000, 1.2, 8, 20, 7)  # mpIniORBextractor: 5 * nFeatures (src/Tracking1.cc:601)
    img1 = pkg.synth_frame(7, 640, 480)
    img2 = _second_view(img1, 9, -6, 1)
    _, k1, d1 = ex(img1, None, (0, 0))
    _, k2, d2 = ex(img2, None, (0, 0))
    k1, d1, k2, d2 = k1.copy(), d1.copy(), k2.copy(), d2.copy()
    ex.close()
    return k1, d1, k2, d2


def test_assign_features_to_grid(pkg, orc, two_frames):
    k1 = two_frames[0]
    m = pkg.ORBmatcher(8192, 8192)
    for bounds in (BOUNDS, (-7.25, 651.5, -3.5, 470.75)):  # undistorted bounds need not be the image rectangle (src/Frame.cc:812-831)
        cs, items = m.AssignFeaturesToGrid(k1, bounds)
        rcs, ritems = orc.assign_grid(k1, bounds)
        assert np.array_equal(cs, rcs) and np.array_equal(items, ritems)
    cs, items = m.AssignFeaturesToGrid(k1[:0], BOUNDS)
    assert cs[-1] == 0 and len(items) == 0
    m.close()


@pytest.mark.parametrize("window,ratio,check", [(100, 0.9, True), (10, 0.6, True), (100, 0.9, False), (30, 0.75, True), (400, 1.0, True)])
def test_search_for_initialization(pkg, orc, two_frames, window, ratio, check):
    k1, d1, k2, d2 = two_frames
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    m = pkg.ORBmatcher(8192, 8192)
    n, m12, pv = m.SearchForInitialization(k1, d1, k2, d2, prev, BOUNDS, window, ratio, check)
    rn, rm12, rpv = orc.search_for_initialization(k1, d1, k2, d2, prev, BOUNDS, window, ratio, check)
    assert n == rn and np.array_equal(m12, rm12) and np.array_equal(pv.view(np.uint32), rpv.view(np.uint32))
    if window == 100 and check:
        assert n > 100  # the shifted view is matchable
        lvl0 = np.nonzero(m12 >= 0)[0]
        assert (k1["octave"][lvl0] == 0).all() and (k2["octave"][m12[lvl0]] == 0).all()
        assert len(set(m12[lvl0].tolist())) == len(lvl0)  # one-to-one (vnMatches21)
    # the tracker calls it again with the updated vbPrevMatched while it waits for enough parallax
    n2, m12b, pv2 = m.SearchForInitialization(k1, d1, k2, d2, pv, BOUNDS, window, ratio, check)
    rn2, rm12b, rpv2 = orc.search_for_initialization(k1, d1, k2, d2, rpv, BOUNDS, window, ratio, check)
    assert n2 == rn2 and np.array_equal(m12b, rm12b) and np.array_equal(pv2.view(np.uint32), rpv2.view(np.uint32))
    m.close()


def test_search_for_initialization_edge_cases(pkg, orc, two_frames):
    k1, d1, k2, d2 = two_frames
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    m = pkg.ORBmatcher(8192, 8192)
    # empty second frame, no level-0 keypoint in the first frame, windows that leave the grid, duplicated descriptors (ties)
    n, m12, pv = m.SearchForInitialization(k1, d1, k2[:0], d2[:0], prev, BOUNDS, 100, 0.9, True)
    assert n == 0 and (m12 == -1).all() and np.array_equal(pv, prev)
    hi = k1["octave"] > 0
    n, m12, _ = m.SearchForInitialization(k1[hi], d1[hi], k2, d2, prev[hi], BOUNDS, 100, 0.9, True)
    assert n == 0 and (m12 == -1).all()
    far = prev + np.float32([5000, -5000])
    n, m12, _ = m.SearchForInitialization(k1, d1, k2, d2, far, BOUNDS, 100, 0.9, True)
    rn, rm12, _ = orc.search_for_initialization(k1, d1, k2, d2, far, BOUNDS, 100, 0.9, True)
    assert n == rn == 0 and np.array_equal(m12, rm12)
    d2t = d2.copy()
    lvl0 = np.nonzero(k2["octave"] == 0)[0]
    d2t[lvl0[1::2]] = d2t[lvl0[0:len(lvl0[1::2]) * 2:2]]  # pairs of identical descriptors: best == second best, order decides
    for ratio in (0.9, 1.5):
        n, m12, pv = m.SearchForInitialization(k1, d1, k2, d2t, prev, BOUNDS, 100, ratio, True)
        rn, rm12, rpv = orc.search_for_initialization(k1, d1, k2, d2t, prev, BOUNDS, 100, ratio, True)
        assert n == rn and np.array_equal(m12, rm12) and np.array_equal(pv.view(np.uint32), rpv.view(np.uint32))
    with pytest.raises(pkg.OrbxError, match="capacity"):
        pkg.ORBmatcher(16, 16).SearchForInitialization(k1, d1, k2, d2, prev, BOUNDS, 100, 0.9, True)
    with pytest.raises(pkg.OrbxError, match="bounds"):
        m.SearchForInitialization(k1, d1, k2, d2, prev, (0, 0, 0, 480), 100, 0.9, True)
    m.close()


@pytest.mark.parametrize("var,val", [("ORBX_SEARCH_INIT_GENERIC", "1"), ("ORBX_SI_TOPK", "0"), ("ORBX_SI_TOPK", "1"), ("ORBX_SI_TOPK", "2"),
                                     ("ORBX_SI_BLOCK", "0"), ("ORBX_SI_ROUNDS", "0"), ("ORBX_SI_ROUNDS_CAP", "300")])
def test_other_kernels_in_subprocess(pkg, var, val):
    """orbx_search_for_initialization takes an LDS-resident kernel when the octave-0 subset of F2 fits (<= 2048 keypoints) and a
    generic one otherwise; ORBX_SEARCH_INIT_GENERIC (read once per process) forces the generic kernel.  The LDS-resident walk decides
    nearly every keypoint on the 8 shortest precomputed candidates and falls back to its full scan when they do not decide:
    ORBX_SI_TOPK=0 switches the short lists off, 1 and 2 shorten them so that the full scan runs behind many undecided lists (with them
    the block-wise walk - eight keypoints per step where they do not depend on each other - meets undecided keypoints in nearly every
    block and falls back to single steps); ORBX_SI_BLOCK=0 walks step by step throughout.  Round 4: a single pair runs as fixed-point rounds
    over all keypoints (k_search_init_rounds; with ORBX_SI_TOPK = 1 / 2 most of its keypoints take the per-round scan of all of F2);
    ORBX_SI_ROUNDS=0 takes the single-wave walk again; batches of pairs take the rounds too, a workgroup per pair, and
    ORBX_SI_ROUNDS_CAP=300 leaves the pairs with more than 300 octave-0 keypoints in F1 to the walk kernel launched behind the rounds
    (the path of a pair beyond the rounds' 2048).  Run this file's parity tests once more under each."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_SEARCH_INIT_GENERIC") or os.environ.get("ORBX_SI_TOPK") or os.environ.get("ORBX_SI_BLOCK") or os.environ.get("ORBX_SI_ROUNDS") or os.environ.get("ORBX_SI_ROUNDS_CAP"):
        pytest.skip("already a forced run")
    env = dict(os.environ)
    env[var] = val
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not subprocess"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


def test_search_for_initialization_frames_batch(pkg, orc):
    """A batch of frame pairs in one launch, device-resident: consecutive frames of one extractor batch (frame p with frame p + 1), with
    a distorted camera so that Frame::UndistortKeyPoints runs in front, against the oracle pair by pair (undistort + search); then a second
    round on the updated vbPrevMatched, and two extractors as the two sides."""
    P = 6
    base = pkg.synth_frame(11, 640, 480)
    imgs = np.stack([_second_view(base, 4 * i, -3 * i, 20 + i) for i in range(P + 1)])
    ex = pkg.ORBextractor(3000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=P + 1)
    n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
    cap = ex.capacity
    k4 = np.float32([458.654, 457.296, 367.215, 248.375]); dist = np.float32([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    m = pkg.ORBmatcher(8192, 8192)
    bounds = m.ComputeImageBounds(640, 480, k4, dist)
    for dcoef in (dist, np.float32([])):
        b = bounds if len(dcoef) else BOUNDS
        nm, m12, prev = m.SearchForInitializationFrames(ex, ex, P, k4, dcoef, b, 100, 0.9, True, frame2_offset=1)
        nm2, m12b, prev2 = m.SearchForInitializationFrames(ex, ex, P, k4, dcoef, b, 100, 0.9, True, frame2_offset=1, prev_matched=prev)
        for p in range(P):
            k1, d1 = kps[p, : n[p]], desc[p, : n[p]]
            k2, d2 = kps[p + 1, : n[p + 1]], desc[p + 1, : n[p + 1]]
            if len(dcoef):
                k1, k2 = orc.undistort_keypoints(k1, k4, dcoef), orc.undistort_keypoints(k2, k4, dcoef)
            pv0 = np.stack([k1["x"], k1["y"]], axis=1)
            rn, rm12, rpv = orc.search_for_initialization(k1, d1, k2, d2, pv0, b, 100, 0.9, True)
            assert nm[p] == rn and np.array_equal(m12[p, : n[p]], rm12), p
            assert np.array_equal(prev[p, : n[p]].view(np.uint32), rpv.view(np.uint32)), p
            rn2, rm12b, rpv2 = orc.search_for_initialization(k1, d1, k2, d2, rpv, b, 100, 0.9, True)
            assert nm2[p] == rn2 and np.array_equal(m12b[p, : n[p]], rm12b) and np.array_equal(prev2[p, : n[p]].view(np.uint32), rpv2.view(np.uint32)), p
        assert nm.min() > 50
    # two extractors as the two sides (frame p of each), no orientation check, a tighter ratio
    ex2 = pkg.ORBextractor(3000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=P + 1)
    n2, _, kps2, desc2 = ex2.extract_batch(imgs[::-1].copy(), (0, 0))
    nm, m12, prev = m.SearchForInitializationFrames(ex, ex2, 3, k4, np.float32([]), BOUNDS, 60, 0.7, False)
    for p in range(3):
        k1, d1, k2, d2 = kps[p, : n[p]], desc[p, : n[p]], kps2[p, : n2[p]], desc2[p, : n2[p]]
        rn, rm12, rpv = orc.search_for_initialization(k1, d1, k2, d2, np.stack([k1["x"], k1["y"]], axis=1), BOUNDS, 60, 0.7, False)
        assert nm[p] == rn and np.array_equal(m12[p, : n[p]], rm12) and np.array_equal(prev[p, : n[p]].view(np.uint32), rpv.view(np.uint32)), p
    with pytest.raises(pkg.OrbxError):
        m.SearchForInitializationFrames(ex, ex, P + 1, k4, np.float32([]), BOUNDS, 100, 0.9, True, frame2_offset=1)  # frame P + 1 does not exist
    ex.close(); ex2.close(); m.close()
