"""ORBmatcher::SearchForTriangulation (reference src/ORBmatcher2.cc:179-418; LocalMapping::CreateNewMapPoints): hand-traced cases for the
CPU oracle (no GPU), then the HIP path through the C ABI against the oracle on synthetic two-view scenes - pinhole (monocular, stereo
keypoints, bOnlyStereo, bCoarse), KannalaBrandt8 with one camera and as a two-camera rig (Tll / Tlr / Trl / Trr)."""
import numpy as np
import pytest

from fisheye_synth import CAM1, CAM2, rig, scene
from voc_synth import make_vocabulary

LEVELS = 8
SF = np.float32(1.2) ** np.arange(LEVELS, dtype=np.float32)
SIGMA2 = (SF * SF).astype(np.float32)


def _fv(groups):
    """FeatureVector dict from {node: [feature indices]}."""
    nodes = sorted(groups)
    start = np.cumsum([0] + [len(groups[n]) for n in nodes]).astype(np.int32)
    feat = np.array([i for n in nodes for i in groups[n]], np.int32)
    return dict(fv_node=np.array(nodes, np.int32), fv_start=start, fv_feature=feat)


def _keys(pkg, pts, octaves=None, angles=None):
    k = np.zeros(len(pts), pkg.KEYPOINT_DTYPE)
    k["x"], k["y"] = np.asarray(pts, np.float32)[:, 0], np.asarray(pts, np.float32)[:, 1]
    k["octave"] = 0 if octaves is None else octaves
    k["angle"] = 0 if angles is None else angles
    k["size"] = 31
    return k


def _flip(d, bits):
    d = d.copy()
    for b in bits:
        d[b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def test_oracle_hand_traced_cases(pkg, orc):
    """x-translation between two normalised cameras: F12 = [t]x with t = (1, 0, 0), so the epipolar line of (x1, y1) is y = y1 and a pair
    passes iff (y2 - y1)^2 < 3.84 (level 0).  Worked out by hand from :246-353."""
    rng = np.random.default_rng(1)
    A, B = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
    k1 = _keys(pkg, [(0, 0), (10, 10), (50, 20)])
    d1 = np.stack([A, A, B])
    k2 = _keys(pkg, [(12, 10.5), (14, 11), (13, 10), (11, 14), (52, 20)])
    d2 = np.stack([_flip(A, [1, 2, 3]), _flip(A, [100, 101, 102]), A, _flip(A, [7]), _flip(B, range(60))])
    fv1, fv2 = _fv({7: [0, 1, 2], 9: []}), _fv({3: [], 7: [3, 0, 2, 1, 4]})
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)

    def run(ep, mp2=(0, 0, 1, 0, 0), ur1=None, ur2=None, **kw):
        a = pkg.keyframe_view(k1, d1, [1, 0, 0], fv1, SF, SIGMA2, u_right=ur1)
        b = pkg.keyframe_view(k2, d2, list(mp2), fv2, SF, SIGMA2, u_right=ur2)
        return orc.search_for_triangulation(a, b, pkg.triangulation_geometry(ep, F), **kw)

    n, pairs, m12 = run((1000, 1000))
    assert n == 1 and m12.tolist() == [-1, 1, -1] and pairs.tolist() == [[1, 1]]  # distance 3 twice: the later one in the node wins
    n, pairs, m12 = run((22.5, 11))                     # feature 1 of KF2 lies 8.5 px from the epipole (< 10 sqrt(scale)) -> skipped; feature 0
    assert n == 1 and m12.tolist() == [-1, 0, -1]       # (10.5 px away, the other distance-3 one) wins
    n, pairs, m12 = run((14, 11))                       # every candidate within 10 px of the epipole
    assert n == 0
    n, pairs, m12 = run((1000, 1000), coarse=True)      # no epipolar test: the closest descriptor (distance 1, 4 px off the line)
    assert n == 1 and m12.tolist() == [-1, 3, -1]
    n, pairs, m12 = run((1000, 1000), mp2=(0, 0, 0, 0, 0))  # the exact twin no longer has a map point
    assert m12.tolist() == [-1, 2, -1]
    n, pairs, m12 = run((14, 11), ur1=[-1, 5, -1], ur2=[-1, 3, -1, 2, -1], only_stereo=True)  # stereo keypoints: no epipole test, 0 is not stereo
    assert n == 1 and m12.tolist() == [-1, 1, -1]
    n, pairs, m12 = run((14, 11), ur1=[-1, -1, 4], ur2=[1, 1, 1, 1, 1], only_stereo=True)  # feature 1 of KF1 is not stereo -> nothing
    assert n == 0 and (m12 == -1).all() and len(pairs) == 0


def _pinhole_scene(pkg, n, seed, outliers=0.15):
    """Two pinhole views of n points: keypoints, descriptors (view 2 = view 1 with up to 45 flipped bits, some exact ties by duplicating
    features), F12 = K^-T [t12]x R12 K^-1 and the epipole, in float32 numpy (the same numbers go to the oracle and to the GPU)."""
    rng = np.random.default_rng(seed)
    K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]], np.float64)
    X1 = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(2, 12, n)], 1)
    a = 0.05
    R12 = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t12 = np.array([0.4, 0.02, 0.05])            # x1 = R12 x2 + t12
    X2 = (X1 - t12) @ R12
    p1 = (X1 / X1[:, 2:]) @ K.T
    p2 = (X2 / X2[:, 2:]) @ K.T
    p2[:, :2] += rng.normal(0, 0.6, (n, 2))
    bad = rng.random(n) < outliers
    p2[bad, 1] += rng.uniform(8, 40, bad.sum())  # off the epipolar line
    oct1 = rng.integers(0, LEVELS, n); oct2 = np.clip(oct1 + rng.integers(-1, 2, n), 0, LEVELS - 1)
    ang1 = rng.uniform(0, 360, n).astype(np.float32)
    ang2 = ((ang1 + rng.normal(4, 3, n) + np.where(rng.random(n) < 0.1, rng.uniform(0, 360, n), 0)) % 360).astype(np.float32)
    d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d2 = np.stack([_flip(d1[i], rng.integers(0, 256, int(rng.integers(0, 46)))) for i in range(n)])
    # duplicates in view 2: same descriptor distance twice inside a node (ties -> the later feature)
    dup = rng.integers(0, n, n // 8)
    p2 = np.concatenate([p2[:, :2], p2[dup, :2] + rng.normal(0, 0.5, (len(dup), 2))]); d2 = np.concatenate([d2, d2[dup]])
    oct2 = np.concatenate([oct2, oct2[dup]]); ang2 = np.concatenate([ang2, ang2[dup]])
    perm = rng.permutation(len(p2))
    k1 = _keys(pkg, p1[:, :2], oct1, ang1); k2 = _keys(pkg, p2[perm], oct2[perm], ang2[perm])
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    c2 = -R12.T @ t12                              # camera centre 1 in frame 2
    ep = (K @ (c2 / c2[2]))[:2] if abs(c2[2]) > 1e-9 else np.array([1e6, 1e6])
    return k1, d1, k2, d2[perm], F, np.asarray(ep, np.float32), rng


@pytest.mark.gpu
@pytest.mark.parametrize("n,only_stereo,coarse,check", [(1500, False, False, True), (1500, True, False, True), (700, False, True, True),
                                                         (2500, False, False, False), (40, False, False, True)])
def test_pinhole_pairs_match_oracle(pkg, orc, n, only_stereo, coarse, check):
    k1, d1, k2, d2, F, ep, rng = _pinhole_scene(pkg, n, seed=n + only_stereo)
    parent, leaf, desc, weight = make_vocabulary(6, 3, seed=3)
    voc = pkg.ORBVocabulary(6, 3, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    fv1, fv2 = voc.transform(d1, 2), voc.transform(d2, 2)   # 6 nodes one level below the root: hundreds of features per node
    mp1 = (rng.random(len(k1)) < 0.3).astype(np.uint8); mp2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(len(k2)) < 0.5, k2["x"] - 5, -1).astype(np.float32)
    if n == 40:
        ep = np.float32([k2["x"][0], k2["y"][0]])            # a keypoint on the epipole
    m = pkg.ORBmatcher(8192, 8192)
    for stereo in (False, True):
        a = pkg.keyframe_view(k1, d1, mp1, fv1, SF, SIGMA2, u_right=ur1 if stereo or only_stereo else None)
        b = pkg.keyframe_view(k2, d2, mp2, fv2, SF, SIGMA2, u_right=ur2 if stereo or only_stereo else None)
        g = pkg.triangulation_geometry(ep, F)
        got = m.SearchForTriangulation(a, b, g, only_stereo, coarse, check)
        ref = orc.search_for_triangulation(a, b, g, only_stereo, coarse, check)
        assert got[0] == ref[0] and np.array_equal(got[2], ref[2]) and np.array_equal(got[1], ref[1])
        assert got[0] == int((got[2] >= 0).sum())
        if n >= 700:
            assert got[0] > n // 20
    # empty / degenerate inputs and the checks of the entry point
    a = pkg.keyframe_view(k1, d1, np.ones(len(k1), np.uint8), fv1, SF, SIGMA2)
    assert m.SearchForTriangulation(a, b, g)[0] == 0
    empty = dict(fv_node=np.zeros(0, np.int32), fv_start=np.zeros(1, np.int32), fv_feature=np.zeros(0, np.int32))
    assert m.SearchForTriangulation(pkg.keyframe_view(k1, d1, mp1, empty, SF, SIGMA2), b, g)[0] == 0
    kbad = k1.copy(); kbad["octave"][0] = 9
    with pytest.raises(pkg.OrbxError, match="octave"):
        m.SearchForTriangulation(pkg.keyframe_view(kbad, d1, mp1, fv1, SF, SIGMA2), b, g)
    with pytest.raises(pkg.OrbxError, match="rig"):
        m.SearchForTriangulation(a, b, pkg.triangulation_geometry(ep, F, model=0, rig=True))
    voc.close(); m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("as_rig", [False, True])
def test_kannala_brandt_pairs_match_oracle(pkg, orc, as_rig):
    """KannalaBrandt8::epipolarConstrain = TriangulateMatches(...) > 0.0001 per candidate pair.  One camera: both keyframes see the scene
    through cam 1, T12 = the rig's (R12, t12).  Rig: every keyframe holds left ++ right keypoints (NLeft = n), the four transforms are
    picked per pair by (bRight1, bRight2)."""
    n = 600
    X1, pl, pr, R12, t12 = scene(n, seed=5)
    rng = np.random.default_rng(17)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    dn = np.stack([_flip(d[i], rng.integers(0, 256, int(rng.integers(0, 40)))) for i in range(n)])
    octs = rng.integers(0, LEVELS, n); ang = rng.uniform(0, 360, n).astype(np.float32)
    parent, leaf, desc, weight = make_vocabulary(5, 3, seed=8)
    voc = pkg.ORBVocabulary(5, 3, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    m = pkg.ORBmatcher(8192, 8192)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    if not as_rig:
        # "keyframe 2" = the right camera's view re-projected through cam 1's model is not available in closed form: use the rig's second
        # camera as the second keyframe's (single) camera - the model only asks for cam_left on both sides, so give it equal intrinsics
        k1 = _keys(pkg, pl + rng.normal(0, 0.3, pl.shape), octs, ang); k2 = _keys(pkg, pr + rng.normal(0, 0.3, pr.shape), octs, ang)
        fv1, fv2 = voc.transform(d, 2), voc.transform(dn, 2)
        a = pkg.keyframe_view(k1, d, rng.random(n) < 0.2, fv1, SF, SIGMA2)
        b = pkg.keyframe_view(k2, dn, rng.random(n) < 0.2, fv2, SF, SIGMA2)
        g = pkg.triangulation_geometry((250, 255), model=1, cam_left=CAM1, cam_right=CAM2, R=R12, t=t12)
    else:
        # both keyframes at the same pose for simplicity of the scene: Tll = I, Tlr = (R12, t12), Trl = its inverse, Trr = I
        Rinv = R12.T.copy(); tinv = (-R12.T @ t12).astype(np.float32)
        k1 = _keys(pkg, np.concatenate([pl, pr]) + rng.normal(0, 0.3, (2 * n, 2)), np.concatenate([octs, octs]), np.concatenate([ang, ang]))
        k2 = _keys(pkg, np.concatenate([pl, pr]) + rng.normal(0, 0.3, (2 * n, 2)), np.concatenate([octs, octs]), np.concatenate([ang, ang]))
        dd1, dd2 = np.concatenate([d, dn]), np.concatenate([dn, d])
        fv1, fv2 = voc.transform(dd1, 2), voc.transform(dd2, 2)
        a = pkg.keyframe_view(k1, dd1, rng.random(2 * n) < 0.2, fv1, SF, SIGMA2, n_left=n)
        b = pkg.keyframe_view(k2, dd2, rng.random(2 * n) < 0.2, fv2, SF, SIGMA2, n_left=n)
        g = pkg.triangulation_geometry((250, 255), model=1, rig=True, cam_left=CAM1, cam_right=CAM2,
                                       R=np.stack([I3, R12, Rinv, I3]), t=np.stack([z3, t12, tinv, z3]))
    for coarse in (False, True):
        got = m.SearchForTriangulation(a, b, g, False, coarse, True)
        ref = orc.search_for_triangulation(a, b, g, False, coarse, True)
        assert got[0] == ref[0] and np.array_equal(got[2], ref[2]) and np.array_equal(got[1], ref[1])
    assert got[0] > 20
    voc.close(); m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("stereo", [False, True])
def test_search_for_triangulation_over_the_neighbours_of_a_keyframe(pkg, orc, stereo):
    """orbx_search_for_triangulation_frames: one keyframe against K neighbours with keypoints, descriptors and FeatureVectors where the
    extractor and the vocabulary left them on the device (LocalMapping::CreateNewMapPoints, reference src/LocalMapping.cc:466) - per
    neighbour the same matches as the single-pair oracle on host copies.  Keyframes: shifted views of one synthetic frame (so that the views
    share words), a small vocabulary around the reference view's descriptors, pinhole epipolar geometry of a pure x / y translation."""
    import torch
    from golden_scenarios import vocabulary
    P = 5
    shifts = [(0, 0), (9, -6), (-7, 4), (14, 3), (3, 11)]
    base = pkg.synth_frame(77, 640, 480)
    rng = np.random.default_rng(21)
    imgs = np.stack([np.clip(np.roll(np.roll(base, dy, 0), dx, 1).astype(np.int16) + rng.integers(-3, 4, base.shape, dtype=np.int16), 0, 255).astype(np.uint8)
                     for dx, dy in shifts])
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=P)
    d_imgs = torch.from_numpy(imgs).to("cuda:0")
    ex.extract_batch_device(d_imgs.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); ex.sync()
    n, _, kk, dd = ex.download_outputs(P)
    parent, leaf, vdesc, vweight = vocabulary(dd[0, : n[0]])
    voc = pkg.ORBVocabulary(8, 3, 0, 0, parent=parent, is_leaf=leaf, desc=vdesc, weight=vweight)
    fvs = voc.transform_batch(ex, P, levelsup=2)
    sf = ex.GetScaleFactors().copy(); sigma2 = ex.GetScaleSigmaSquares().copy()
    cap = ex.capacity
    has = np.zeros((P, cap), np.uint8); ur = np.full((P, cap), -1, np.float32)
    for f in range(P):
        has[f, : n[f]] = rng.random(n[f]) < 0.3
        if stereo:
            ur[f, : n[f]] = np.where(rng.random(n[f]) < 0.6, kk[f, : n[f]]["x"] - rng.uniform(1, 30, n[f]), -1)
    k4 = np.float32([458.654, 457.296, 367.215, 248.375])
    K = np.array([[k4[0], 0, k4[2]], [0, k4[1], k4[3]], [0, 0, 1]], np.float64)
    neighbours, geoms = [], []
    for f in range(1, P):
        dx, dy = shifts[f]
        # views related by a translation parallel to the image plane: t12 ~ (dx, dy, 0); F12 = K^-T [t]x K^-1; the epipole lies at infinity:
        # far outside the image, as for a sideways motion
        t = np.array([dx, dy, 0.0]) / np.hypot(dx, dy)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Kinv = np.linalg.inv(K)
        F12 = (Kinv.T @ tx @ Kinv).astype(np.float32)
        geoms.append(pkg.triangulation_geometry((1e6 * t[0], 1e6 * t[1]), F12))
        neighbours.append(f)
    m = pkg.ORBmatcher(8192, 8192)
    for only_stereo, coarse in ((False, False), (False, True)) + (((True, False),) if stereo else ()):
        nm, m12 = m.SearchForTriangulationFrames(voc, ex, 0, neighbours, k4, np.zeros(0, np.float32), geoms, has, u_right=ur if stereo else None,
                                                 bOnlyStereo=only_stereo, bCoarse=coarse)
        total = 0
        for k, f in enumerate(neighbours):
            a = pkg.keyframe_view(kk[0, : n[0]], dd[0, : n[0]], has[0, : n[0]], fvs[0], sf, sigma2, u_right=ur[0, : n[0]] if stereo else None)
            b = pkg.keyframe_view(kk[f, : n[f]], dd[f, : n[f]], has[f, : n[f]], fvs[f], sf, sigma2, u_right=ur[f, : n[f]] if stereo else None)
            rn, _, rm = orc.search_for_triangulation(a, b, geoms[k], only_stereo, coarse, True)
            assert nm[k] == rn and np.array_equal(m12[k, : n[0]], rm), (only_stereo, coarse, f)
            assert (m12[k, n[0]:] == -1).all()
            total += rn
        if not only_stereo:
            assert total > 40
    # a neighbour list that repeats a keyframe and includes the reference itself; and the guard against a stale transform
    nm, m12 = m.SearchForTriangulationFrames(voc, ex, 2, [2, 1, 1], k4, np.zeros(0, np.float32), [geoms[0]] * 3, has)
    assert nm[1] == nm[2] and np.array_equal(m12[1], m12[2])
    ex2 = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=P)
    ex2.extract_batch_device(d_imgs.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); ex2.sync()
    with pytest.raises(pkg.OrbxError, match="transform_batch"):
        m.SearchForTriangulationFrames(voc, ex2, 0, [1], k4, np.zeros(0, np.float32), geoms[:1], has)
    m.close(); voc.close(); ex.close(); ex2.close()
