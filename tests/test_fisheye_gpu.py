"""Frame::ComputeStereoFishEyeMatches + KannalaBrandt8::TriangulateMatches (reference src/Frame.cc:1156-1196,
src/CameraModels/KannalaBrandt8.cpp:306-375): HIP path against the oracle restatement, bit for bit."""
import numpy as np
import pytest

import fisheye_synth as fs

pytestmark = pytest.mark.gpu


def _scenario(pkg, n_pts, n_mono_l, n_mono_r, seed, noise_px=0.0, dup=False):
    """Keypoints / descriptors of a synthetic fisheye stereo frame: the first mono_* keypoints of each image lie outside the lapping
    area (no partner), the rest are projections of common 3-D points (right image shuffled) plus unmatched distractors."""
    rng = np.random.default_rng(seed)
    X1, uv1, uv2, R12, t12 = fs.scene(n_pts, seed=seed)
    uv2 = uv2 + rng.normal(0, noise_px, uv2.shape)
    desc = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)
    dl_match = desc.copy()
    dr_match = desc.copy()
    flip = rng.integers(0, 256, (n_pts, 3))                      # a few differing bits per pair
    for i in range(n_pts):
        for b in flip[i]:
            dr_match[i, b // 8] ^= np.uint8(1 << (b % 8))
    perm = rng.permutation(n_pts)
    n_dis = n_pts // 3
    kl = np.zeros(n_mono_l + n_pts + n_dis, pkg.KEYPOINT_DTYPE); kr = np.zeros(n_mono_r + n_pts + n_dis, pkg.KEYPOINT_DTYPE)
    dl = rng.integers(0, 256, (len(kl), 32), dtype=np.uint8); dr = rng.integers(0, 256, (len(kr), 32), dtype=np.uint8)
    for k in (kl, kr):
        k["x"] = rng.uniform(20, 490, len(k)); k["y"] = rng.uniform(20, 490, len(k)); k["octave"] = rng.integers(0, 8, len(k)); k["size"] = 31
    kl["x"][n_mono_l:n_mono_l + n_pts] = uv1[:, 0]; kl["y"][n_mono_l:n_mono_l + n_pts] = uv1[:, 1]
    dl[n_mono_l:n_mono_l + n_pts] = dl_match
    kr["x"][n_mono_r:n_mono_r + n_pts] = uv2[perm, 0]; kr["y"][n_mono_r:n_mono_r + n_pts] = uv2[perm, 1]
    dr[n_mono_r:n_mono_r + n_pts] = dr_match[perm]
    if dup:  # two left keypoints with the same best right partner: the later one must own mvRightToLeftMatch
        kl[n_mono_l + 1] = kl[n_mono_l]; dl[n_mono_l + 1] = dl[n_mono_l]
        dr[n_mono_r + n_pts:] = dr[n_mono_r + n_pts:] | 0x55     # keep distractors far from everything
    return kl, dl, kr, dr, R12, t12


@pytest.mark.parametrize("n_pts,mono_l,mono_r,noise,dup", [(400, 150, 90, 0.0, False), (1200, 0, 0, 0.4, False), (300, 40, 700, 1.5, True), (5, 0, 3, 0.0, False)])
def test_compute_stereo_fisheye_matches(pkg, orc, n_pts, mono_l, mono_r, noise, dup):
    kl, dl, kr, dr, R12, t12 = _scenario(pkg, n_pts, mono_l, mono_r, seed=n_pts + mono_l, noise_px=noise, dup=dup)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    sigma2 = ex.GetScaleSigmaSquares()
    ex.close()
    want = orc.fisheye_stereo_matches(kl, dl, mono_l, kr, dr, mono_r, fs.CAM1, fs.CAM2, R12, t12, sigma2)
    m = pkg.ORBmatcher(4096, 4096)
    got = m.ComputeStereoFishEyeMatches(kl, dl, mono_l, kr, dr, mono_r, fs.CAM1, fs.CAM2, R12, t12, sigma2)
    assert got[0] == want[0]
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    assert got[3].tobytes() == want[3].tobytes()                       # mvDepth, bit for bit
    ok = want[3] > 0
    assert got[4][ok].tobytes() == want[4][ok].tobytes()               # mvStereo3Dpoints where a match exists
    if n_pts >= 300 and noise < 1.0:
        assert want[0] > 0.5 * n_pts                                   # most true pairs survive (the rest: parallax / ratio)
    assert np.all(want[1][:mono_l] == -1)
    m.close()


def test_fisheye_matches_frames_batch(pkg, orc):
    """The batched, device-resident form on two extractors' outputs (lapping area -> monoIndex offsets) equals the oracle per pair."""
    P, w, h = 3, 512, 512
    left = np.stack([pkg.synth_frame(500 + i, w, h) for i in range(P)])
    right = np.stack([np.roll(left[i], -(9 + 4 * i), axis=1) for i in range(P)])
    exl = pkg.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    exr = pkg.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    nl, ml, kl, dl = exl.extract_batch(left, (150, 511))
    nr, mr, kr, dr = exr.extract_batch(right, (0, 360))
    R12, t12 = fs.rig()
    m = pkg.ORBmatcher(4096, 4096)
    nm, l2r, r2l, depth, p3d = m.ComputeStereoFishEyeMatchesFrames(exl, exr, P, fs.CAM1, fs.CAM2, R12, t12)
    sigma2 = exl.GetScaleSigmaSquares()
    for p in range(P):
        assert 0 < ml[p] < nl[p] and 0 < mr[p] < nr[p]
        want = orc.fisheye_stereo_matches(kl[p, : nl[p]], dl[p, : nl[p]], ml[p], kr[p, : nr[p]], dr[p, : nr[p]], mr[p], fs.CAM1, fs.CAM2, R12, t12, sigma2)
        assert nm[p] == want[0]
        assert np.array_equal(l2r[p, : nl[p]], want[1]) and np.array_equal(r2l[p, : nr[p]], want[2])
        assert depth[p, : nl[p]].tobytes() == want[3].tobytes()
        ok = want[3] > 0
        assert p3d[p, : nl[p]][ok].tobytes() == want[4][ok].tobytes()
    m.close(); exl.close(); exr.close()
