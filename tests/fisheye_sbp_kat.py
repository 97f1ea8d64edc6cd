"""Hand-traced known answers for the CurrentFrame.Nleft != -1 branches of ORBmatcher::SearchByProjection (reference src/ORBmatcher1.cc:45-215,
src/ORBmatcher3.cc:300-436), shared by the oracle's CPU test and the GPU parity test.  Each case: (name, mode, kwargs, nmatches, assign);
mode "points" = the map-point overload, "frames" = the frame-to-frame one.  All keypoints sit in level 0 of a 640 x 480 image."""
import numpy as np

BOUNDS = (0.0, 640.0, 0.0, 480.0)
SCALE = np.float32(1.2) ** np.arange(8, dtype=np.float32)


def _kps(dtype, pts):
    k = np.zeros(len(pts), dtype)
    for i, (x, y) in enumerate(pts):
        k["x"][i], k["y"][i] = x, y
    k["size"] = 31.0
    return k


def _rec(dtype, n):
    r = np.zeros(n, dtype)
    r["angle"] = 0.5  # viewing cosine below 0.998: radius 4 (map points); keypoint angle 0.5 deg (frames)
    return r


def _flip(d, nbits):
    d = d.copy()
    for b in range(nbits):
        d[b // 8] ^= np.uint8(1 << (b % 8))
    return d


def cases(kp_dtype, proj_dtype):
    out = []
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, 32, dtype=np.uint8)

    # A. a point WITHOUT observations taken on the right is copied to its left stereo partner, whose previous holder had observations:
    #    the left keypoint is free again for the next point (which has observations and then blocks the third).
    kc = _kps(kp_dtype, [(100, 100), (100, 100)])  # L0 | R0
    dc = np.stack([base, base])
    pl, pr = _rec(proj_dtype, 3), _rec(proj_dtype, 3)
    pl["u"] = pr["u"] = 100; pl["v"] = pr["v"] = 100
    pl["flags"] = [0, 3, 3]  # P0: left not in view, no observations; P1, P2: in view with observations
    pr["flags"] = [1, 0, 1]
    out.append(("partner_write_unblocks", "points",
                dict(kc=kc, dc=dc, n_left=1, cur_state=np.int8([1, -1]), mp_l=pl, mp_r=pr, mp_desc=np.stack([base] * 3),
                     left_to_right=np.int32([0]), right_to_left=np.int32([0]), th=1.0, nnratio=0.8), 4, np.int32([1, 1])))

    # B. the left ratio test fails (10 vs 11 bits in one level, 10 > 0.8 * 11): `continue` to the next map point, the perfect right
    #    match is never searched; with ratio 0.95 the left match stands and the right one is taken too.
    kc = _kps(kp_dtype, [(200, 150), (200, 150), (190, 150)])  # L0 L1 | R0
    mpd = base
    dc = np.stack([_flip(base, 10), _flip(base, 10), base])
    dc[1][31] ^= np.uint8(0x80)  # 11 bits
    pl, pr = _rec(proj_dtype, 1), _rec(proj_dtype, 1)
    pl["u"], pl["v"], pr["u"], pr["v"] = 200, 150, 190, 150
    pl["flags"] = 3; pr["flags"] = 1
    kw = dict(kc=kc, dc=dc, n_left=2, cur_state=np.int8([-1, -1, -1]), mp_l=pl, mp_r=pr, mp_desc=mpd[None], left_to_right=np.int32([-1, -1]),
              right_to_left=np.int32([-1]), th=1.0)
    out.append(("left_ratio_failure_skips_right", "points", dict(kw, nnratio=0.8), 0, np.int32([-1, -1, -1])))
    out.append(("left_ratio_pass_then_right", "points", dict(kw, nnratio=0.95), 2, np.int32([0, -1, 0])))

    # C. the right search radius takes no th factor: 3 px off with cosine 0.999 (radius 2.5) and th = 2 is inside on the left (5), outside
    #    on the right (2.5).
    kc = _kps(kp_dtype, [(300, 200), (290, 200)])
    dc = np.stack([base, base])
    pl, pr = _rec(proj_dtype, 1), _rec(proj_dtype, 1)
    pl["angle"] = pr["angle"] = 0.999
    pl["u"], pl["v"], pr["u"], pr["v"] = 303, 200, 293, 200
    pl["flags"] = 3; pr["flags"] = 1
    out.append(("right_radius_without_th", "points",
                dict(kc=kc, dc=dc, n_left=1, cur_state=np.int8([-1, -1]), mp_l=pl, mp_r=pr, mp_desc=base[None], left_to_right=np.int32([-1]),
                     right_to_left=np.int32([-1]), th=2.0, nnratio=0.8), 1, np.int32([0, -1])))

    # D. frame-to-frame: an empty left window `continue`s past the right search; a non-empty one whose best distance is above TH_HIGH does not.
    kc = _kps(kp_dtype, [(500, 400), (120, 100)])  # L0 far away | R0 at the right projection
    dc = np.stack([base, base])
    pl, pr = _rec(proj_dtype, 1), _rec(proj_dtype, 1)
    pl["u"], pl["v"], pl["invz"], pr["u"], pr["v"] = 130, 100, 0.5, 120, 100
    pl["flags"] = 3
    kw = dict(dc=dc, n_left=1, cur_state=np.int8([-1, -1]), proj_l=pl, proj_r=pr, proj_desc=base[None], th=7.0, check_orientation=False)
    out.append(("empty_left_window_skips_right", "frames", dict(kw, kc=kc), 0, np.int32([-1, -1])))
    kc2 = kc.copy(); kc2["x"][0], kc2["y"][0] = 131, 100
    dc2 = np.stack([_flip(base, 120), base])
    out.append(("left_above_th_high_then_right", "frames", dict(kw, kc=kc2, dc=dc2), 1, np.int32([-1, 0])))
    return out
