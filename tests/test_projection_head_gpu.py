"""GPU parity of the projection head of the two tracking searches (round 4; include/orbx.h "The projection head") against the CPU oracle:
records bit for bit (float fields compared as bits), both camera models, both modes; and "project-then-match" end to end - the records of
the device head fed to the batched searches on the device give what the oracle's projection + the oracle's search give."""
import numpy as np
import pytest

import scene3d

pytestmark = pytest.mark.gpu

BOUNDS = (0.0, 640.0, 0.0, 480.0)
LOGSF = float(scene3d.libm_logf(np.float32(1.2)))   # Frame::mfLogScaleFactor = log(mfScaleFactor) (src/Frame.cc:114): the platform's logf


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _frames(pkg, seed, w=640, h=480, nfeat=1200):
    ex = pkg.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
    img1 = pkg.synth_frame(seed, w, h)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    ex.close()
    return k1, d1


@pytest.mark.parametrize("cam", [scene3d.PINHOLE, scene3d.KB8], ids=["pinhole", "kb8"])
def test_motion_model_head_matches_the_oracle(pkg, orc, cam):
    k1, _ = _frames(pkg, 31, *((640, 480) if cam[0] == 0 else (512, 512)))
    m = pkg.ORBmatcher(8192, 8192)
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        pts, q, t, _, _ = scene3d.make_scene(rng, k1, cam, pkg.WORLD_POINT_DTYPE, motion=0.02 * (seed + 1), rot=0.01 * (seed + 1))
        got = m.ProjectLastFramePoints(q, t, cam[0], cam[1], pts)
        ref = orc.project_last_frame_points(q, t, cam[0], cam[1], pts)
        assert np.array_equal(_bits(got), _bits(ref)), f"seed {seed}: {np.flatnonzero((_bits(got).reshape(len(pts), -1) != _bits(ref).reshape(len(pts), -1)).any(1))[:5]}"
        assert (got["invz"] < 0).any() and (got["invz"] > 0).sum() > 0.9 * len(pts)   # points behind the camera are in the set
    # identity pose: x3Dc == x3Dw exactly; a point on the optical axis projects to the principal point
    one = np.zeros(2, pkg.WORLD_POINT_DTYPE); one["pos"] = [[0, 0, 2.5], [0.5, -0.25, 4.0]]; one["flags"] = 3
    got = m.ProjectLastFramePoints([0, 0, 0, 1], [0, 0, 0], cam[0], cam[1], one)
    assert got["u"][0] == np.float32(cam[1][2]) and got["v"][0] == np.float32(cam[1][3]) and got["invz"][0] == np.float32(0.4)
    assert m.ProjectLastFramePoints([0, 0, 0, 1], [0, 0, 0], cam[0], cam[1], one[:0]).shape == (0,)
    m.close()


@pytest.mark.parametrize("cam", [scene3d.PINHOLE, scene3d.KB8], ids=["pinhole", "kb8"])
def test_in_frustum_head_matches_the_oracle(pkg, orc, cam):
    w, h = (640, 480) if cam[0] == 0 else (512, 512)
    k1, _ = _frames(pkg, 33, w, h)
    bounds = (0.0, float(w), 0.0, float(h))
    m = pkg.ORBmatcher(8192, 8192)
    causes = 0
    for seed in range(6):
        rng = np.random.default_rng(200 + seed)
        pts, q, t, R, Ow = scene3d.make_scene(rng, k1, cam, pkg.WORLD_POINT_DTYPE, motion=0.1 * (seed + 1), rot=0.03 * (seed + 1), behind=0.05)
        far = bool(seed & 1)
        got = m.ProjectInFrustum(R, t, Ow, cam[0], cam[1], bounds, 47.9, 0.5, LOGSF, 8, pts, far_points=far, th_far_points=9.0)
        ref = orc.project_in_frustum(R, t, Ow, cam[0], cam[1], bounds, 47.9, 0.5, LOGSF, 8, pts, far_points=far, th_far_points=9.0)
        assert np.array_equal(_bits(got), _bits(ref)), f"seed {seed}"
        kept = (got["flags"] & 1).sum(); had = (pts["flags"] & 1).sum()
        assert 0 < kept < had                                    # the geometric tests reject some and keep some
        assert set(np.unique(got["octave"])) <= set(range(8)) and len(np.unique(got["octave"])) >= 3
        causes += had - kept
    assert causes > 100
    m.close()


def test_predict_scale_at_level_boundaries(pkg, orc):
    """MapPoint::PredictScale (src/MapPoint.cc:531-546) in the reference's FLOAT arithmetic (logf / float divide / ceilf): map points seen from
    the distance of their reference observation, mfMaxDistance = fl(dist * 1.2^l), so that log(ratio) / mfLogScaleFactor sits on an integer
    up to rounding - the float and the double evaluation (rounds 1-4) give different levels on a tenth of these.  The device level equals
    the oracle's AND a restatement from the definition over the platform's logf."""
    cam = scene3d.PINHOLE
    rng = np.random.default_rng(77)
    n = 20000
    pts = np.zeros(n, pkg.WORLD_POINT_DTYPE)
    z = rng.uniform(1.0, 20.0, n)
    pts["pos"] = np.stack([rng.uniform(-0.5, 0.5, n) * z, rng.uniform(-0.4, 0.4, n) * z, z], -1).astype(np.float32)
    pts["normal"] = pts["pos"] / np.linalg.norm(pts["pos"], axis=1, keepdims=True)
    pts["flags"] = 1
    # dist exactly as the head forms it (Ow = 0): sqrtf(x0 + (x1 + x2)) of float squares
    P = pts["pos"]
    sq = (P * P).astype(np.float32)
    dist = np.sqrt((sq[:, 0] + (sq[:, 1] + sq[:, 2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    lev = rng.integers(0, 10, n)
    pts["max_distance"] = (dist * (np.float32(1.2) ** lev.astype(np.float32)).astype(np.float32)).astype(np.float32)
    pts["min_distance"] = 0.0
    m = pkg.ORBmatcher(32768, 64)
    got = m.ProjectInFrustum(np.eye(3), [0, 0, 0], [0, 0, 0], cam[0], cam[1], (-1e9, 1e9, -1e9, 1e9), 0.0, 0.5, LOGSF, 8, pts)
    ref = orc.project_in_frustum(np.eye(3), [0, 0, 0], [0, 0, 0], cam[0], cam[1], (-1e9, 1e9, -1e9, 1e9), 0.0, 0.5, LOGSF, 8, pts)
    assert np.array_equal(_bits(got), _bits(ref))
    ratio = (pts["max_distance"] / dist).astype(np.float32)
    want = np.array([scene3d.predict_scale_float(r, LOGSF, 8) for r in ratio])
    wrong = np.array([scene3d.predict_scale_double(r, LOGSF, 8) for r in ratio])
    assert np.array_equal(got["octave"], want)
    assert (want != wrong).sum() > 500           # these inputs tell the two evaluations apart
    m.close()


def _two_views(pkg, seed, dx, dy):
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7)
    img1 = pkg.synth_frame(seed, 640, 480)
    rng = np.random.default_rng(seed)
    img2 = np.clip(np.roll(np.roll(img1, dy, axis=0), dx, axis=1).astype(np.int16) + rng.integers(-3, 4, img1.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    return ex, img1, img2, rng


def test_project_then_match_on_the_device(pkg, orc):
    """Tracking::TrackWithMotionModel's matcher for a batch of frames with NOTHING of the projections crossing the link: the head writes the
    records on the device, orbx_search_by_projection_frames consumes them there.  Reference: src/ORBmatcher3.cc:256-467 from :281 on.
    The scene: the last frame's keypoints at random depths; the current view is the last image shifted by (dx, dy) pixels, which a pure
    translation reproduces only on average - the window search (th = 15) absorbs the parallax."""
    cam = scene3d.PINHOLE
    dx, dy = 6, -4
    ex, img1, img2, rng = _two_views(pkg, 41, dx, dy)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    P = 3
    exb = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=P)
    import torch
    d_imgs = torch.from_numpy(np.stack([img2] * P)).to("cuda:0")
    exb.extract_batch_device(d_imgs.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); exb.sync()
    n2, _, kk, dd = exb.download_outputs(P)
    sf = ex.GetScaleFactors().copy()
    m = pkg.ORBmatcher(8192, 8192)
    pts_all, poses, ref = [], [], []
    for p in range(P):
        pts, q, t, _, _ = scene3d.make_scene(rng, k1, cam, pkg.WORLD_POINT_DTYPE, motion=0.0, rot=0.0)
        # a translation that moves a point at the mean depth by (dx, dy) pixels
        zbar = float(np.median(pts["pos"][:, 2][pts["pos"][:, 2] > 0]))
        t = np.float32([dx * zbar / cam[1][0], dy * zbar / cam[1][1], 0.0]) * np.float32(1.0 + 0.1 * p)
        pts["pos"][:, 2] = np.where(pts["pos"][:, 2] > 0, zbar * rng.uniform(0.9, 1.1, len(pts)), pts["pos"][:, 2]).astype(np.float32)
        pts["pos"][:, 0] = ((k1["x"] - cam[1][2]) / cam[1][0] * pts["pos"][:, 2]).astype(np.float32)
        pts["pos"][:, 1] = ((k1["y"] - cam[1][3]) / cam[1][1] * pts["pos"][:, 2]).astype(np.float32)
        pts_all.append(pts); poses.append(np.concatenate([q, t]))
        ref.append(orc.project_last_frame_points(q, t, cam[0], cam[1], pts))
    d_proj, d_np, max_np = m.ProjectPointsFrames(0, np.stack(poses), cam[0], cam[1], pts_all, device=0)
    k4 = np.float32(cam[1]); nod = np.zeros(0, np.float32)
    nm, assign, _ = m.SearchByProjectionFrames(exb, P, k4, nod, BOUNDS, 0.0, d_proj, [d1] * P, 15.0, proj_counts=d_np)
    got_proj = d_proj.cpu().numpy().view(pkg.PROJECTION_DTYPE).reshape(P, max_np)
    total = 0
    for p in range(P):
        assert np.array_equal(_bits(got_proj[p, : len(ref[p])]), _bits(ref[p]))
        nc = int(n2[p])
        state = np.full(nc, -1, np.int8)
        rn, rassign = orc.search_by_projection(kk[p, :nc], dd[p, :nc], state, BOUNDS, sf, ref[p], d1, 15.0)
        assert nm[p] == rn and np.array_equal(assign[p, :nc], rassign)
        total += rn
    assert total > 150 * P
    m.close(); ex.close(); exb.close()


def test_in_frustum_then_match_on_the_device(pkg, orc):
    """Tracking::SearchLocalPoints for a batch: Frame::isInFrustum on the device (mode 1) feeding orbx_search_by_projection_points_frames
    (src/Tracking4.cc:700-737, src/Frame.cc:542-600, src/ORBmatcher1.cc:45-215)."""
    cam = scene3d.PINHOLE
    ex, img1, img2, rng = _two_views(pkg, 43, 0, 0)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    P = 2
    import torch
    exb = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=P)
    d_imgs = torch.from_numpy(np.stack([img2] * P)).to("cuda:0")
    exb.extract_batch_device(d_imgs.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); exb.sync()
    n2, _, kk, dd = exb.download_outputs(P)
    sf = ex.GetScaleFactors().copy()
    m = pkg.ORBmatcher(8192, 8192)
    pts_all, poses, ref = [], [], []
    for p in range(P):
        pts, q, t, R, Ow = scene3d.make_scene(rng, k1, cam, pkg.WORLD_POINT_DTYPE, motion=0.002, rot=0.0005, behind=0.01)
        # scale level of the observing keypoint: mfMaxDistance = dist * 1.2^octave (MapPoint::UpdateNormalAndDepth)
        d = np.linalg.norm(pts["pos"] - Ow, axis=1)
        pts["max_distance"] = (d * np.float32(1.2) ** k1["octave"] * rng.uniform(0.97, 1.03, len(pts))).astype(np.float32)
        pts["min_distance"] = (pts["max_distance"] / np.float32(1.2) ** 7).astype(np.float32)
        pts_all.append(pts); poses.append(np.concatenate([R.reshape(9), t, Ow]))
        ref.append(orc.project_in_frustum(R, t, Ow, cam[0], cam[1], BOUNDS, 0.0, 0.5, LOGSF, 8, pts))
    d_proj, d_np, max_np = m.ProjectPointsFrames(1, np.stack(poses), cam[0], cam[1], pts_all, device=0, bounds=BOUNDS, mbf=0.0,
                                                 viewing_cos_limit=0.5, log_scale_factor=LOGSF, nlevels=8)
    k4 = np.float32(cam[1]); nod = np.zeros(0, np.float32)
    nm, assign, _ = m.SearchByProjectionFrames(exb, P, k4, nod, BOUNDS, 0.0, d_proj, [d1] * P, 3.0, proj_counts=d_np, map_points_nnratio=0.8)
    got_proj = d_proj.cpu().numpy().view(pkg.PROJECTION_DTYPE).reshape(P, max_np)
    total = 0
    for p in range(P):
        assert np.array_equal(_bits(got_proj[p, : len(ref[p])]), _bits(ref[p]))
        nc = int(n2[p])
        state = np.full(nc, -1, np.int8)
        rn, rassign = orc.search_by_projection_points(kk[p, :nc], dd[p, :nc], state, BOUNDS, sf, ref[p], d1, 3.0, 0.8)
        assert nm[p] == rn and np.array_equal(assign[p, :nc], rassign)
        total += rn
    assert total > 100 * P
    m.close(); ex.close(); exb.close()


def _rig_pose(R, t, Ow, angle=0.4):
    Rrl = scene3d.rotation_matrix(scene3d.quat_from_axis_angle([0.1, 1.0, 0.05], angle)); trl = np.float32([-0.101, 0.002, 0.001])
    tlr = (-(Rrl.astype(np.float64).T @ trl.astype(np.float64))).astype(np.float32)
    return Rrl, trl, tlr, np.ascontiguousarray(R.T)


def test_fisheye_frustum_head_matches_the_oracle(pkg, orc):
    """Mode 2: Frame::isInFrustum of a fisheye rig (reference src/Frame.cc:604-614, isInFrustumChecks :1198-1272) - both cameras' records bit for
    bit, single call and a batch of frames on the device; the far-point test reads the left depth for both sides."""
    cam1 = scene3d.KB8[1]; cam2 = tuple(np.float32(cam1) * np.float32([1.01, 0.99, 1.0, 1.0, 1.1, 0.9, 1.0, 1.0]))
    k1, _ = _frames(pkg, 35, 512, 512)
    bounds = (0.0, 512.0, 0.0, 512.0)
    m = pkg.ORBmatcher(8192, 8192)
    poses, pts_all, refs = [], [], []
    for seed in range(5):
        rng = np.random.default_rng(300 + seed)
        pts, q, t, R, Ow = scene3d.make_scene(rng, k1, scene3d.KB8, pkg.WORLD_POINT_DTYPE, motion=0.1 * (seed + 1), rot=0.03 * (seed + 1), behind=0.05)
        if seed == 4:
            pts = pts[:0]                                    # a frame without points
        Rrl, trl, tlr, Rwc = _rig_pose(R, t, Ow)
        far = bool(seed & 1)
        gl, gr = m.ProjectInFrustumFisheye(R, t, Ow, Rrl, trl, tlr, Rwc, cam1, cam2, bounds, 0.5, LOGSF, 8, pts, far_points=far, th_far_points=9.0)
        rl, rr = orc.project_in_frustum_fisheye(R, t, Ow, Rrl, trl, tlr, Rwc, cam1, cam2, bounds, 0.5, LOGSF, 8, pts, far_points=far, th_far_points=9.0)
        assert np.array_equal(_bits(gl), _bits(rl)) and np.array_equal(_bits(gr), _bits(rr)), f"seed {seed}"
        if len(pts):
            assert 0 < (gl["flags"] & 1).sum() < (pts["flags"] & 1).sum() and ((gl["flags"] ^ gr["flags"]) & 1).sum() > 5
        poses.append(np.concatenate([R.reshape(9), t, Ow, Rrl.reshape(9), trl, tlr, Rwc.reshape(9)])); pts_all.append(pts)
        refs.append(orc.project_in_frustum_fisheye(R, t, Ow, Rrl, trl, tlr, Rwc, cam1, cam2, bounds, 0.5, LOGSF, 8, pts))
    d_l, d_r, d_n, max_np = m.ProjectInFrustumFisheyeFrames(np.stack(poses), cam1, cam2, bounds, 0.5, LOGSF, 8, pts_all, device=0)
    m.sync()                                                 # (asynchronous on the matcher's stream)
    gl = d_l.cpu().numpy().view(pkg.PROJECTION_DTYPE).reshape(len(pts_all), max_np)
    gr = d_r.cpu().numpy().view(pkg.PROJECTION_DTYPE).reshape(len(pts_all), max_np)
    for f, (rl, rr) in enumerate(refs):
        assert np.array_equal(_bits(gl[f, : len(rl)]), _bits(rl)) and np.array_equal(_bits(gr[f, : len(rr)]), _bits(rr)), f"frame {f}"
    m.close()


@pytest.mark.parametrize("cam", [scene3d.PINHOLE, scene3d.KB8], ids=["pinhole", "kb8"])
def test_window_point_head_matches_the_oracle(pkg, orc, cam):
    """Mode 3: the head of Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame) (reference src/ORBmatcher2.cc:446-520,
    src/ORBmatcher1.cc:451-494, src/ORBmatcher3.cc:486-519) - orbx_window_point records bit for bit, every combination of the three test bits."""
    w, h = (640, 480) if cam[0] == 0 else (512, 512)
    k1, _ = _frames(pkg, 37, w, h)
    bounds = (0.0, float(w), 0.0, float(h))
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    m = pkg.ORBmatcher(8192, 8192)
    seen = 0
    for seed in range(4):
        rng = np.random.default_rng(400 + seed)
        pts, q, t, R, Ow = scene3d.make_scene(rng, k1, cam, pkg.WORLD_POINT_DTYPE, motion=0.1 * (seed + 1), rot=0.03 * (seed + 1), behind=0.05)
        d = np.linalg.norm(pts["pos"] - Ow, axis=1)
        pts["max_distance"] = (d * np.float32(1.2) ** rng.integers(0, 9, len(pts)) * rng.uniform(0.9, 1.1, len(pts))).astype(np.float32)
        pts["min_distance"] = (pts["max_distance"] / np.float32(1.2) ** 7).astype(np.float32)
        for tests in range(8):
            got = m.ProjectWindowPoints(q, t, Ow, cam[0], cam[1], bounds, 47.9, 3.0 + tests, sf, LOGSF, tests, pts)
            ref = orc.project_window_points(q, t, Ow, cam[0], cam[1], bounds, 47.9, 3.0 + tests, sf, LOGSF, tests, pts)
            assert np.array_equal(_bits(got), _bits(ref)), (seed, tests)
            seen += int(got["search"].sum())
        assert 0 < got["search"].sum() < (pts["flags"] & 1).sum() and len(np.unique(got["level"])) >= 4
    assert seen > 1000
    with pytest.raises(pkg.OrbxError):
        m.ProjectWindowPoints(q, t, Ow, cam[0], cam[1], bounds, 0.0, 3.0, sf, LOGSF, 8, pts)        # unknown test bit
    m.close()


def test_window_point_head_then_fuse_on_the_device(pkg, orc):
    """LocalMapping::SearchInNeighbors' shape with NOTHING of the projections crossing the link: the head writes the orbx_window_point records
    of K keyframes on the device (poses, map points in), orbx_fuse_frames consumes them there; per keyframe the same fused indices as the
    oracle's head + the oracle's window search (reference src/ORBmatcher2.cc:420-610 from :446 on)."""
    import torch
    cam = scene3d.PINHOLE
    K = 4
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=K)
    imgs = np.stack([pkg.synth_frame(60 + f, 640, 480) for f in range(K)])
    d_imgs = torch.from_numpy(imgs).to("cuda:0")
    ex.extract_batch_device(d_imgs.data_ptr(), K, 640, 480, 640, 640 * 480, (0, 0)); ex.sync()
    n, _, kk, dd = ex.download_outputs(K)
    inv_sigma2 = ex.GetInverseScaleSigmaSquares().copy(); sf = ex.GetScaleFactors().copy()
    k4 = np.float32(cam[1]); nod = np.zeros(0, np.float32)
    bounds = (0.0, 640.0, 0.0, 480.0)
    rng = np.random.default_rng(21)
    m = pkg.ORBmatcher(8192, 8192)
    poses, pts_all, pd_all, ref_w = [], [], [], []
    for f in range(K):
        nc = int(n[f])
        keys = kk[f, :nc]
        pts, q, t, R, Ow = scene3d.make_scene(rng, keys, cam, pkg.WORLD_POINT_DTYPE, motion=0.0, rot=0.0, behind=0.02)
        # the keyframe's pose = identity + a small step: the points re-project next to the keypoints they were lifted from
        t = rng.normal(0, 0.002, 3).astype(np.float32); Ow = (-t).astype(np.float32)
        dist = np.linalg.norm(pts["pos"] - Ow, axis=1)
        pts["max_distance"] = (dist * np.float32(1.2) ** keys["octave"] * rng.uniform(0.97, 1.03, nc)).astype(np.float32)
        pts["min_distance"] = (pts["max_distance"] / np.float32(1.2) ** 7).astype(np.float32)
        pts["normal"] = (pts["pos"] - Ow) / np.maximum(np.linalg.norm(pts["pos"] - Ow, axis=1, keepdims=True), 1e-6)
        sel = rng.permutation(nc)[: int(rng.integers(200, 900))]
        pts = pts[sel]
        pd = np.stack([dd[f, s] ^ (rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 2, 32, dtype=np.uint8) * np.uint8(3)) for s in sel])
        poses.append(np.concatenate([[0, 0, 0, 1], t, Ow])); pts_all.append(pts); pd_all.append(pd)
        ref_w.append(orc.project_window_points([0, 0, 0, 1], t, Ow, cam[0], cam[1], bounds, 40.0, 3.0, sf, LOGSF, 3, pts))
    dev_pts = m.ProjectWindowPointsFrames(np.stack(poses), cam[0], cam[1], bounds, 40.0, 3.0, sf, LOGSF, 3, pts_all, device=0)
    fi, nf = m.FuseFrames(ex, K, k4, nod, bounds, dev_pts, pd_all, chi2_test=True, fuse=True)   # (downloads: the matcher's stream is drained)
    got_w = dev_pts[0].cpu().numpy().view(pkg.WINDOW_POINT_DTYPE).reshape(K, dev_pts[2])
    total = 0
    for f in range(K):
        nc, np_ = int(n[f]), len(pts_all[f])
        assert np.array_equal(_bits(got_w[f, :np_]), _bits(ref_w[f]))
        ref = orc.search_keyframe_windows(kk[f, :nc], dd[f, :nc], bounds, ref_w[f], pd_all[f], inv_level_sigma2=inv_sigma2, chi2_test=True)
        exp = np.where(ref[1] <= 50, ref[0], -1)
        assert np.array_equal(fi[f, :np_], exp) and nf[f] == int((exp >= 0).sum()), f
        total += int(nf[f])
    assert total > 100 * K
    m.close(); ex.close()


def test_projection_head_argument_errors(pkg):
    m = pkg.ORBmatcher(64, 64)
    pts = np.zeros(4, pkg.WORLD_POINT_DTYPE); pts["pos"][:, 2] = 1
    with pytest.raises(pkg.OrbxError):
        m.ProjectLastFramePoints([0, 0, 0, 1], [0, 0, 0], 2, scene3d.PINHOLE[1], pts)       # unknown camera model
    with pytest.raises(pkg.OrbxError):
        m.ProjectInFrustum(np.eye(3), [0, 0, 0], [0, 0, 0], 0, scene3d.PINHOLE[1], BOUNDS, 0.0, 0.5, 0.0, 8, pts)   # log_scale_factor <= 0
    with pytest.raises(pkg.OrbxError):
        m.ProjectLastFramePoints([0, 0, 0, 1], [0, 0, 0], 0, scene3d.PINHOLE[1], np.zeros(100, pkg.WORLD_POINT_DTYPE))  # capacity
    m.close()
