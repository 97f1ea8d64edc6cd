"""GPU parity of the CurrentFrame.Nleft != -1 (KB8 fisheye rig) branches of both ORBmatcher::SearchByProjection functions (reference
src/ORBmatcher3.cc:300-436, src/ORBmatcher1.cc:45-215) against the CPU oracle: bit-exact nmatches and mvpMapPoints assignment over the
combined left + right keypoint range.  The right image is the left one moved by a disparity; the projections into the right camera are
the left ones moved by the same disparity plus a prediction error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOUNDS = (0.0, 640.0, 0.0, 480.0)


def _rig(pkg, seed, dx, dy, disp):
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    img1 = pkg.synth_frame(seed, 640, 480)
    rng = np.random.default_rng(seed)
    noise = lambda im: np.clip(im.astype(np.int16) + rng.integers(-3, 4, im.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    img2 = noise(np.roll(np.roll(img1, dy, axis=0), dx, axis=1))
    img2r = noise(np.roll(img2, -disp, axis=1))
    out = []
    for im in (img1, img2, img2r):
        _, k, d = ex(im, None, (0, 0))
        out.append((k.copy(), d.copy()))
    sf = ex.GetScaleFactors().copy()
    ex.close()
    (k1, d1), (kl, dl), (kr, dr) = out
    return k1, d1, np.concatenate([kl, kr]), np.concatenate([dl, dr]), len(kl), sf, rng


def _records(pkg, k1, rng, dx, dy, disp, p_mp=0.7, p_obs=0.8, noise=3.0):
    n = len(k1)
    pl = np.zeros(n, pkg.PROJECTION_DTYPE)
    pl["u"] = k1["x"] + dx + rng.normal(0, noise, n).astype(np.float32)
    pl["v"] = k1["y"] + dy + rng.normal(0, noise, n).astype(np.float32)
    pl["invz"] = (1.0 / rng.uniform(0.5, 20.0, n)).astype(np.float32)
    pl["invz"][rng.random(n) < 0.02] *= -1
    pl["angle"] = k1["angle"]
    pl["octave"] = k1["octave"]
    pl["flags"] = (rng.random(n) < p_mp).astype(np.int32) | ((rng.random(n) < p_obs).astype(np.int32) << 1)
    pr = pl.copy()
    pr["u"] = pl["u"] - disp + rng.normal(0, noise, n).astype(np.float32)
    pr["v"] = pl["v"] + rng.normal(0, noise, n).astype(np.float32)
    return pl, pr


def _stereo_partners(rng, n_left, n_right, frac=0.4):
    """mvLeftToRightMatch / mvRightToLeftMatch: a random one-to-one pairing of a fraction of the keypoints."""
    l2r = np.full(n_left, -1, np.int32)
    r2l = np.full(n_right, -1, np.int32)
    npair = int(min(n_left, n_right) * frac)
    ls = rng.permutation(n_left)[:npair]
    rs = rng.permutation(n_right)[:npair]
    l2r[ls] = rs
    r2l[rs] = ls
    return l2r, r2l


@pytest.mark.parametrize("th,forward,backward,check", [(15, False, False, True), (7, True, False, True), (7, False, True, False),
                                                       (30, False, False, True), (2, False, False, True)])
def test_search_by_projection_fisheye(pkg, orc, th, forward, backward, check):
    dx, dy, disp = 6, -4, 12
    k1, d1, kc, dc, nl, sf, rng = _rig(pkg, 31, dx, dy, disp)
    pl, pr = _records(pkg, k1, rng, dx, dy, disp)
    state = rng.choice(np.int8([-1, -1, -1, 0, 1]), len(kc))
    m = pkg.ORBmatcher(8192, 8192)
    n, assign = m.SearchByProjectionFisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, d1, th, forward=forward, backward=backward,
                                            checkOrientation=check)
    rn, rassign = orc.search_by_projection_fisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, d1, th, forward, backward, check)
    assert n == rn and np.array_equal(assign, rassign)
    if th >= 7:
        assert n > 60 and (assign[nl:] >= 0).sum() > 20 and (assign[:nl] >= 0).sum() > 20  # both cameras take matches
    m.close()


@pytest.mark.parametrize("th,nnratio", [(1.0, 0.8), (3.0, 0.8), (1.0, 0.6), (5.0, 0.95)])
def test_search_by_projection_points_fisheye(pkg, orc, th, nnratio):
    dx, dy, disp = 3, 2, 9
    k1, d1, kc, dc, nl, sf, rng = _rig(pkg, 37, dx, dy, disp)
    pl, pr = _records(pkg, k1, rng, dx, dy, disp, p_mp=0.8, p_obs=0.9, noise=1.5)
    pl["angle"] = rng.choice(np.float32([0.9, 0.9985, 0.999, 0.5]), len(k1))  # mTrackViewCos
    pr["angle"] = rng.choice(np.float32([0.9, 0.9985, 0.999, 0.5]), len(k1))
    pr["flags"] = (pl["flags"] & 2) | (rng.random(len(k1)) < 0.7)              # mbTrackInViewR on its own
    pr["octave"] = np.where(rng.random(len(k1)) < 0.05, -1, pl["octave"])       # mnTrackScaleLevelR == -1
    l2r, r2l = _stereo_partners(rng, nl, len(kc) - nl)
    state = rng.choice(np.int8([-1, -1, -1, 0, 1]), len(kc))
    m = pkg.ORBmatcher(8192, 8192)
    n, assign = m.SearchByProjectionPointsFisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, d1, l2r, r2l, th, nnratio)
    rn, rassign = orc.search_by_projection_points_fisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, d1, l2r, r2l, th, nnratio)
    assert n == rn and np.array_equal(assign, rassign)
    if th >= 3:
        assert n > 60
    m.close()


def test_search_by_projection_fisheye_conflicts(pkg, orc):
    """Every projection lands on one of a few spots on both sides, half of the points have no observations and many keypoints have stereo
    partners: chains of blocked keypoints, and holders replaced through the partner write by points that do not block."""
    k1, d1, kc, dc, nl, sf, rng = _rig(pkg, 41, 0, 0, 10)
    nr = len(kc) - nl
    n = len(k1)
    pl, pr = _records(pkg, k1, rng, 0, 0, 10, p_mp=1.0, p_obs=0.5, noise=0.5)
    spots_l = rng.integers(0, nl, 30)
    spots_r = rng.integers(0, nr, 30)
    s = rng.integers(0, 30, n)
    pl["u"], pl["v"], pl["octave"] = kc["x"][spots_l[s]], kc["y"][spots_l[s]], kc["octave"][spots_l[s]]
    pr["u"], pr["v"], pr["octave"] = kc["x"][nl + spots_r[s]], kc["y"][nl + spots_r[s]], kc["octave"][nl + spots_r[s]]
    pr["flags"] = pl["flags"]
    pl["angle"] = 0.9
    pr["angle"] = 0.9
    dq = np.where((rng.random(n) < 0.5)[:, None], dc[spots_l[s]], dc[nl + spots_r[s]]).copy()
    dq[:, :3] ^= rng.integers(0, 256, (n, 3), dtype=np.uint8)
    # partners inside the spot sets, so partner writes hit contested keypoints
    l2r = np.full(nl, -1, np.int32)
    r2l = np.full(nr, -1, np.int32)
    ul, ur = np.unique(spots_l), np.unique(spots_r)
    k = min(len(ul), len(ur))
    l2r[ul[:k]] = ur[:k]
    r2l[ur[:k]] = ul[:k]
    state = np.full(len(kc), -1, np.int8)
    m = pkg.ORBmatcher(8192, 8192)
    for th in (1.0, 4.0):
        got = m.SearchByProjectionPointsFisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, dq, l2r, r2l, th, 0.9)
        ref = orc.search_by_projection_points_fisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, dq, l2r, r2l, th, 0.9)
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
        assert got[0] > 10
    pl2 = pl.copy()
    pl2["angle"] = k1["angle"]
    pl2["invz"] = 0.5
    for th in (15, 40):
        got = m.SearchByProjectionFisheye(kc, dc, nl, state, BOUNDS, sf, pl2, pr, dq, th)
        ref = orc.search_by_projection_fisheye(kc, dc, nl, state, BOUNDS, sf, pl2, pr, dq, th)
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    # edges: no points, one empty side
    got = m.SearchByProjectionFisheye(kc, dc, nl, state, BOUNDS, sf, pl2[:0], pr[:0], dq[:0], 15)
    assert got[0] == 0 and (got[1] == -1).all()
    got = m.SearchByProjectionFisheye(kc[:nl], dc[:nl], nl, state[:nl], BOUNDS, sf, pl2, pr, dq, 15)
    ref = orc.search_by_projection_fisheye(kc[:nl], dc[:nl], nl, state[:nl], BOUNDS, sf, pl2, pr, dq, 15)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    got = m.SearchByProjectionPointsFisheye(kc[nl:], dc[nl:], 0, state[nl:], BOUNDS, sf, pl, pr, dq, l2r[:0], np.full(nr, -1, np.int32), 1.0, 0.9)
    ref = orc.search_by_projection_points_fisheye(kc[nl:], dc[nl:], 0, state[nl:], BOUNDS, sf, pl, pr, dq, l2r[:0], np.full(nr, -1, np.int32), 1.0, 0.9)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    with pytest.raises(pkg.OrbxError):
        bad = l2r.copy(); bad[0] = nr
        m.SearchByProjectionPointsFisheye(kc, dc, nl, state, BOUNDS, sf, pl, pr, dq, bad, r2l, 1.0, 0.9)
    m.close()


def test_fisheye_known_answers(pkg):
    """The hand-traced cases of tests/fisheye_sbp_kat.py through the C-ABI: expected values written down, not taken from the oracle."""
    import fisheye_sbp_kat as kat
    m = pkg.ORBmatcher(64, 64)
    for name, mode, kw, n_exp, a_exp in kat.cases(pkg.KEYPOINT_DTYPE, pkg.PROJECTION_DTYPE):
        kw = dict(kw)
        if mode == "points":
            n, a = m.SearchByProjectionPointsFisheye(kw.pop("kc"), kw.pop("dc"), kw.pop("n_left"), kw.pop("cur_state"), kat.BOUNDS, kat.SCALE,
                                                     kw.pop("mp_l"), kw.pop("mp_r"), kw.pop("mp_desc"), kw.pop("left_to_right"),
                                                     kw.pop("right_to_left"), **kw)
        else:
            n, a = m.SearchByProjectionFisheye(kw.pop("kc"), kw.pop("dc"), kw.pop("n_left"), kw.pop("cur_state"), kat.BOUNDS, kat.SCALE,
                                               kw.pop("proj_l"), kw.pop("proj_r"), kw.pop("proj_desc"), kw.pop("th"),
                                               checkOrientation=kw.pop("check_orientation"), **kw)
        assert n == n_exp and np.array_equal(a, a_exp), (name, n, a)
    m.close()


def test_fisheye_frames_batch_equals_single_calls(pkg):
    """orbx_search_by_projection_fisheye_frames / _points_fisheye_frames: a batch of rig frames, device-resident on two extractors' outputs,
    against the single-frame entry points (which the tests above hold against the oracle) - combined indices map i -> i, capL + j -> nl + j."""
    P, w, h = 5, 640, 480
    rng = np.random.default_rng(77)
    left = np.stack([pkg.synth_frame(900 + p, w, h) for p in range(P)])
    right = np.stack([np.roll(left[p], -(8 + p), axis=1) for p in range(P)])
    exl = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=P); exr = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=P)
    nl, _, kl, dl = exl.extract_batch(left, (0, 0)); nr, _, kr, dr = exr.extract_batch(right, (0, 0))
    capL, capR = exl.capacity, exr.capacity
    sf = exl.GetScaleFactors().copy()
    # "last frame" of every frame: its own left keypoints moved by a small motion
    pls, prs, descs, states, l2rs, r2ls = [], [], [], [], [], []
    for p in range(P):
        k1, d1 = kl[p][: nl[p]], dl[p][: nl[p]]
        n_pts = len(k1) - 37 * p  # ragged point counts
        pl, pr = _records(pkg, k1[:n_pts], rng, 2, -1, 8 + p)
        if p == 3:
            pl, pr = pl[:0], pr[:0]  # a frame without points
        pls.append(pl); prs.append(pr); descs.append(d1[: len(pl)])
        st = np.full(capL + capR, -1, np.int8)
        st[: nl[p]] = rng.choice(np.int8([-1, -1, 0, 1]), nl[p]); st[capL: capL + nr[p]] = rng.choice(np.int8([-1, -1, 0, 1]), nr[p])
        states.append(st)
        a, b = _stereo_partners(rng, nl[p], nr[p])
        l2r = np.full(capL, -1, np.int32); r2l = np.full(capR, -1, np.int32); l2r[: nl[p]] = a; r2l[: nr[p]] = b
        l2rs.append(l2r); r2ls.append(r2l)
    m = pkg.ORBmatcher(8192, 8192)
    for mode in ("frames", "points"):
        if mode == "frames":
            nm, assign = m.SearchByProjectionFisheyeFrames(exl, exr, P, BOUNDS, np.stack(states), pls, prs, descs, 15.0)
        else:
            pls2 = [x.copy() for x in pls]; prs2 = [x.copy() for x in prs]
            for x, y in zip(pls2, prs2):
                x["angle"] = rng.choice(np.float32([0.9, 0.999]), len(x)); y["angle"] = x["angle"]
            nm, assign = m.SearchByProjectionFisheyeFrames(exl, exr, P, BOUNDS, np.stack(states), pls2, prs2, descs, 3.0, left_to_right=np.stack(l2rs),
                                                           right_to_left=np.stack(r2ls), map_points_nnratio=0.8)
        for p in range(P):
            kc = np.concatenate([kl[p][: nl[p]], kr[p][: nr[p]]]); dc = np.concatenate([dl[p][: nl[p]], dr[p][: nr[p]]])
            st = np.concatenate([states[p][: nl[p]], states[p][capL: capL + nr[p]]])
            if mode == "frames":
                rn, ra = m.SearchByProjectionFisheye(kc, dc, int(nl[p]), st, BOUNDS, sf, pls[p], prs[p], descs[p], 15.0)
            else:
                rn, ra = m.SearchByProjectionPointsFisheye(kc, dc, int(nl[p]), st, BOUNDS, sf, pls2[p], prs2[p], descs[p], l2rs[p][: nl[p]],
                                                           r2ls[p][: nr[p]], 3.0, 0.8)
            got = np.concatenate([assign[p][: nl[p]], assign[p][capL: capL + nr[p]]])
            assert nm[p] == rn and np.array_equal(got, ra), (mode, p, nm[p], rn)
        assert nm[3] == 0 and nm.sum() > 100
    m.close(); exl.close(); exr.close()
