"""Known-answer tests that pin the CPU oracle against the reference TEXT: constructor
tables, umax literal, sampling-pattern hash, Hamming identities, packing rule, OpenCV-semantics spot checks."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# SURVEY.md App. C (derived from reference src/ORBextractor.cc:418-447)
SCALE = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
QUOTAS = {1000: [217, 181, 151, 126, 105, 87, 73, 60], 1200: [261, 217, 181, 151, 126, 105, 87, 72],
          1500: [326, 271, 226, 189, 157, 131, 109, 91], 4000: [869, 724, 603, 503, 419, 349, 291, 242]}
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]  # literal at reference src/OpenCL/Kernel/Angle.cl:3
PATTERN_SHA256 = "3f202c09967ef499081baca510075490fe60c9626ed8e748ff8e94378229a598"
LEVEL_SIZES = {
    (640, 480): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
    (752, 480): [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)],
    (1920, 1080): [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)],
}


@pytest.mark.parametrize("nf", sorted(QUOTAS))
def test_constructor_tables(orc, nf):
    t = orc.Oracle(nf, 1.2, 8, 20, 7).tables()
    assert list(t["features_per_level"]) == QUOTAS[nf]
    assert np.allclose(t["scale"], np.array(SCALE, np.float32), rtol=0, atol=1e-9 * 4)
    assert list(t["umax"]) == UMAX
    assert np.array_equal(t["inv_scale"], (np.float32(1) / t["scale"]).astype(np.float32))
    assert np.array_equal(t["sigma2"], (t["scale"] * t["scale"]).astype(np.float32))


def _pattern_values(path):
    body = open(path).read()
    body = body[body.index("{") + 1: body.rindex("}")]
    return [int(v) for v in re.findall(r"-?\d+", body)]


@pytest.mark.parametrize("rel", ["oracle/pattern_table.inc", "wut-cuda-orb-slam3_amd/csrc/orbx_pattern.inc"])
def test_pattern_hash(rel):
    vals = _pattern_values(os.path.join(ROOT, rel))
    assert len(vals) == 1024 and min(vals) == -13 and max(vals) == 12
    assert hashlib.sha256(bytes((v + 128) & 255 for v in vals)).hexdigest() == PATTERN_SHA256


def test_hamming_identities(orc):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, 32, dtype=np.uint8)
    b = rng.integers(0, 256, 32, dtype=np.uint8)
    assert orc.hamming256(a, a) == 0
    assert orc.hamming256(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256
    assert orc.hamming256(a, b) == orc.hamming256(b, a) == int(np.unpackbits(a ^ b).sum())


def test_level_sizes_and_mono_packing_rule(orc):
    img = orc.synth_frame(3, 640, 480)
    o = orc.Oracle(1000, 1.2, 8, 20, 7)
    mono, kps, desc = o.extract(img, (0, 1000))  # Frame.cc:313 mono call: every keypoint is "lapping"
    assert mono == 0 and len(kps) >= 1000
    assert [o.level_size(l) for l in range(8)] == LEVEL_SIZES[(640, 480)]
    # fully reversed level-major order: octaves non-increasing
    assert np.all(np.diff(kps["octave"]) <= 0)
    mono2, kps2, desc2 = o.extract(img, (2000, 3000))  # no keypoint is lapping -> plain level-major order
    assert mono2 == len(kps2) == len(kps)
    assert kps2[::-1].tobytes() == kps.tobytes() and desc2[::-1].tobytes() == desc.tobytes()
    # split: lapping ones at the tail in reverse order
    mono3, kps3, _ = o.extract(img, (300, 400))
    inlap = (kps2["x"] >= 300) & (kps2["x"] <= 400)
    assert mono3 == int((~inlap).sum())
    assert kps3[:mono3].tobytes() == kps2[~inlap].tobytes()
    assert kps3[mono3:].tobytes() == kps2[inlap][::-1].tobytes()


def test_empty_image(orc):
    o = orc.Oracle()
    mono, kps, desc = o.extract(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(kps) == 0


@pytest.mark.parametrize("size", sorted(LEVEL_SIZES))
def test_level_sizes(orc, size):
    w, h = size
    o = orc.Oracle(1200, 1.2, 8, 20, 7)
    o.extract(orc.synth_frame(0, w, h), (0, 0))
    assert [o.level_size(l) for l in range(8)] == LEVEL_SIZES[size]


def test_fast_atan2_known_values(orc):
    # exact axis values of cv::fastAtan2: polynomial gives 0 at c=0; 90-0, 180-0, 360-90
    assert orc.fast_atan2(0.0, 1.0) == 0.0
    assert orc.fast_atan2(1.0, 0.0) == 90.0
    assert orc.fast_atan2(0.0, -1.0) == 180.0
    assert orc.fast_atan2(-1.0, 0.0) == 270.0
    for y, x in [(1, 1), (3, -2), (-5, -7), (-1, 9)]:
        want = np.degrees(np.arctan2(y, x)) % 360
        assert abs(orc.fast_atan2(float(y), float(x)) - want) < 0.02  # documented accuracy ~0.01 deg


def test_resize_identity_and_constant(orc):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear(img, 60, 40), img)  # scale 1: weights (2048, 0)
    flat = np.full((48, 64), 137, np.uint8)
    assert np.all(orc.resize_linear(flat, 53, 40) == 137)


def test_blur_constant_and_taps(orc):
    flat = np.full((30, 30), 200, np.uint8)
    # OpenCV 4.5.0 taps sum to 257/256: 200 * (257/256)^2 = 201.57 -> 202 ; >= 4.5.1 taps sum to 1
    assert np.all(orc.gaussian_blur7(flat, 0) == 202)
    assert np.all(orc.gaussian_blur7(flat, 1) == 200)
    imp = np.zeros((15, 15), np.uint8)
    imp[7, 7] = 255
    out = orc.gaussian_blur7(imp, 1)
    taps = np.array([18, 34, 48, 56, 48, 34, 18])
    want = ((np.outer(taps, taps) * 255 + 32768) >> 16).astype(np.uint8)
    assert np.array_equal(out[4:11, 4:11], want)


def test_fast_roi_synthetic_corner(orc):
    # a bright 90-degree wedge, brightest at its tip, on a dark background: the tip is the unique NMS survivor
    # (a flat wedge gives a plateau of equal scores, which strict `>` NMS suppresses entirely)
    img = np.full((21, 21), 10, np.uint8)
    yy, xx = np.mgrid[10:21, 10:21]
    img[10:, 10:] = (200 - (xx - 10) - (yy - 10)).astype(np.uint8)
    kp = orc.fast_roi(img, 20)
    xs, ys, sc = kp & 0xFFF, (kp >> 12) & 0xFFF, kp >> 24
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == [(10, 10, 189)]  # |200 - 10| - 1
    flat = np.full((21, 21), 10, np.uint8)
    flat[10:, 10:] = 200
    assert len(orc.fast_roi(flat, 20)) == 0
    assert len(orc.fast_roi(np.full((21, 21), 77, np.uint8), 7)) == 0


def test_distribute_keeps_best_and_quota(orc):
    rng = np.random.default_rng(9)
    W, H = 608, 448
    pts = set()
    while len(pts) < 3000:
        pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
    pts = sorted(pts, key=lambda p: (p[1], p[0]))
    cand = np.array([x | (y << 12) | (int(rng.integers(7, 255)) << 24) for x, y in pts], np.uint32)
    sel = orc.distribute(cand, W, H, 217)
    assert 217 <= len(sel) <= 220 and len(set(sel.tolist())) == len(sel)
    one = orc.distribute(cand[:1], W, H, 217)
    assert list(one) == [0]


def test_frame_grid_and_area_query_against_brute_force(orc):
    """Frame::AssignFeaturesToGrid / GetFeaturesInArea restated in the oracle vs a direct numpy evaluation of
    the same formulas (src/Frame.cc:687-766)."""
    o = orc.Oracle(2000, 1.2, 8, 20, 7)
    _, k, d = o.extract(orc.synth_frame(0x4F524230 + 3, 640, 480), (0, 0))
    bounds = (-4.5, 648.25, 2.0, 470.0)
    inv_w = np.float32(64) / (np.float32(bounds[1]) - np.float32(bounds[0]))
    inv_h = np.float32(48) / (np.float32(bounds[3]) - np.float32(bounds[2]))
    fx = (k["x"] - np.float32(bounds[0])) * inv_w
    fy = (k["y"] - np.float32(bounds[2])) * inv_h
    px = (np.sign(fx) * np.floor(np.abs(fx) + np.float32(0.5))).astype(int)  # std::round: half away from zero
    py = (np.sign(fy) * np.floor(np.abs(fy) + np.float32(0.5))).astype(int)
    inside = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    cell = np.where(inside, px * 48 + py, -1)
    cs, items = orc.assign_grid(k, bounds)
    assert cs[-1] == inside.sum() == len(items)
    for c in range(64 * 48):
        assert list(items[cs[c]:cs[c + 1]]) == list(np.nonzero(cell == c)[0])
    for (x, y, r, lo, hi) in [(320, 240, 100, 0, 0), (10, 470, 50, -1, -1), (600, 20, 25.5, 1, 3), (-300, 240, 100, 0, 0), (320, 240, 2000, 0, 7)]:
        got = orc.features_in_area(k, bounds, x, y, r, lo, hi)
        x32, y32, r32 = np.float32(x), np.float32(y), np.float32(r)
        cx0 = max(0, int(np.floor((x32 - np.float32(bounds[0]) - r32) * inv_w)))
        cx1 = min(63, int(np.ceil((x32 - np.float32(bounds[0]) + r32) * inv_w)))
        cy0 = max(0, int(np.floor((y32 - np.float32(bounds[2]) - r32) * inv_h)))
        cy1 = min(47, int(np.ceil((y32 - np.float32(bounds[2]) + r32) * inv_h)))
        want = []
        if cx0 < 64 and cx1 >= 0 and cy0 < 48 and cy1 >= 0:
            check = lo > 0 or hi >= 0
            for ix in range(cx0, cx1 + 1):
                for iy in range(cy0, cy1 + 1):
                    for i in np.nonzero(cell == ix * 48 + iy)[0]:
                        if check and (k["octave"][i] < lo or (hi >= 0 and k["octave"][i] > hi)):
                            continue
                        if abs(k["x"][i] - x32) < r32 and abs(k["y"][i] - y32) < r32:
                            want.append(i)
        assert list(got) == want, (x, y, r, lo, hi)


def test_search_for_initialization_properties(orc):
    """ORBmatcher::SearchForInitialization restated in the oracle: a frame matched against itself pairs every level-0 keypoint
    with itself; matches are one-to-one, only between level-0 keypoints, and vbPrevMatched follows the matches."""
    o = orc.Oracle(3000, 1.2, 8, 20, 7)
    _, k, d = o.extract(orc.synth_frame(0x4F524230 + 5, 640, 480), (0, 0))
    bounds = (0.0, 640.0, 0.0, 480.0)
    prev = np.stack([k["x"], k["y"]], axis=1)
    n, m12, pv = orc.search_for_initialization(k, d, k, d, prev, bounds, 100, 0.9, True)
    lvl0 = k["octave"] == 0
    assert n == lvl0.sum() and np.array_equal(m12[lvl0], np.nonzero(lvl0)[0]) and (m12[~lvl0] == -1).all()
    assert np.array_equal(pv, prev)
    # against a different frame: plausibility only (no reference vectors exist for this path: parity unpinned)
    _, k2, d2 = o.extract(orc.synth_frame(0x4F524230 + 6, 640, 480), (0, 0))
    n, m12, pv = orc.search_for_initialization(k, d, k2, d2, prev, bounds, 100, 0.9, True)
    hit = np.nonzero(m12 >= 0)[0]
    assert n == len(hit) and len(set(m12[hit].tolist())) == len(hit)
    assert (k["octave"][hit] == 0).all() and (k2["octave"][m12[hit]] == 0).all()
    assert np.array_equal(pv[hit], np.stack([k2["x"][m12[hit]], k2["y"][m12[hit]]], axis=1))
    assert (np.abs(k2["x"][m12[hit]] - prev[hit, 0]) < 100).all() and (np.abs(k2["y"][m12[hit]] - prev[hit, 1]) < 100).all()


def test_search_by_projection_properties(orc):
    """ORBmatcher::SearchByProjection(Frame, Frame) restated in the oracle (reference src/ORBmatcher3.cc:256-467): projecting a
    frame's own keypoints with their own descriptors matches every valid point to itself; a second copy of each projection is
    blocked exactly when the first taker has observations; pre-occupied keypoints with observations are never taken."""
    o = orc.Oracle(800, 1.2, 8, 20, 7)
    _, k, d = o.extract(orc.synth_frame(0x4F524230 + 9, 640, 480), (0, 0))
    sf = o.tables()["scale"]
    bounds = (0.0, 640.0, 0.0, 480.0)
    n = len(k)
    pr = np.zeros(n, orc.PROJECTION_DTYPE)
    pr["u"], pr["v"], pr["invz"], pr["angle"], pr["octave"], pr["flags"] = k["x"], k["y"], 0.5, k["angle"], k["octave"], 3
    free = np.full(n, -1, np.int8)
    nm, assign = orc.search_by_projection(k, d, free, bounds, sf, pr, d, 2.0, check_orientation=False)
    assert nm == n and np.array_equal(assign, np.arange(n))
    # invalid / behind the camera / outside the image are skipped
    pr2 = pr.copy()
    pr2["flags"][0::4] = 0
    pr2["invz"][1::4] = -0.1
    pr2["u"][2::4] = 700.0
    nm, assign = orc.search_by_projection(k, d, free, bounds, sf, pr2, d, 2.0, check_orientation=False)
    assert nm == len(range(3, n, 4)) and (assign[3::4] == np.arange(n)[3::4]).all() and (np.delete(assign, np.s_[3::4]) == -1).all()
    # every projection twice: the copy is blocked iff the first taker has observations (bit 1); without, it overwrites (and counts)
    pr3 = np.concatenate([pr, pr])
    pr3["flags"][:n] = np.where(np.arange(n) % 2 == 0, 3, 1)
    nm, assign = orc.search_by_projection(k, d, free, bounds, sf, pr3, np.concatenate([d, d]), 0.5, check_orientation=False)
    even = np.arange(n) % 2 == 0
    lone = np.array([np.sum((np.abs(k["x"] - k["x"][i]) < 0.5 * sf[k["octave"][i]]) & (np.abs(k["y"] - k["y"][i]) < 0.5 * sf[k["octave"][i]])) == 1
                     for i in range(n)])  # keypoints whose window holds no other keypoint
    assert (assign[lone & even] == np.arange(n)[lone & even]).all()          # blocked copy: the first taker stays
    assert (assign[lone & ~even] == (np.arange(n) + n)[lone & ~even]).all()  # unobserved first taker: the copy overwrites
    # keypoints that already hold an observed map point are never taken
    busy = np.full(n, 1, np.int8)
    nm, assign = orc.search_by_projection(k, d, busy, bounds, sf, pr, d, 15.0)
    assert nm == 0 and (assign == -1).all()


def test_search_by_projection_map_points_properties(orc):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) restated in the oracle (reference src/ORBmatcher1.cc:45-215): a frame's own
    keypoints as map points at their own level match themselves; the window follows RadiusByViewingCos (2.5 above cos 0.998, else 4,
    times th unless th == 1, times the level's scale) and only levels level-1 .. level are searched; a twin in the same level within
    the ratio is rejected, a twin in another level is not."""
    o = orc.Oracle(800, 1.2, 8, 20, 7)
    _, k, d = o.extract(orc.synth_frame(0x4F524230 + 11, 640, 480), (0, 0))
    sf = o.tables()["scale"]
    bounds = (0.0, 640.0, 0.0, 480.0)
    n = len(k)
    pr = np.zeros(n, orc.PROJECTION_DTYPE)
    pr["u"], pr["v"], pr["invz"], pr["angle"], pr["octave"], pr["flags"] = k["x"], k["y"], -1.0, 0.9999, k["octave"], 3
    free = np.full(n, -1, np.int8)
    nm, assign = orc.search_by_projection_points(k, d, free, bounds, sf, pr, d, 1.0, 0.8)
    took = assign >= 0
    assert nm == took.sum() and (assign[took] == np.arange(n)[took]).all() and nm > 0.9 * n  # losers: a same-level twin in the window
    # predicted level two above the keypoint's: outside [level-1, level], nothing found for isolated points
    one = np.zeros(1, orc.PROJECTION_DTYPE)
    i0 = int(np.flatnonzero(k["octave"] == 0)[0])
    one["u"], one["v"], one["angle"], one["flags"] = k["x"][i0], k["y"][i0], 0.5, 3
    kk, dd = k[i0:i0 + 1], d[i0:i0 + 1]
    for lvl, expect in ((0, 1), (1, 1), (2, 0)):
        one["octave"] = lvl
        nm, assign = orc.search_by_projection_points(kk, dd, free[:1], bounds, sf, one, dd, 1.0, 0.8)
        assert nm == expect
    # radius: offset 3 px at level 0 is inside 4 (cos 0.5) but outside 2.5 (cos 0.999); th = 2 doubles both; the compare is in
    # double, so float(0.998) = 0.99800003 counts as above 0.998 and the float just below it does not
    one["octave"] = 0
    one["u"] = k["x"][i0] + 3.0
    for cosv, th, expect in ((0.5, 1.0, 1), (0.999, 1.0, 0), (0.999, 2.0, 1), (np.float32(0.998), 1.0, 0),
                             (np.nextafter(np.float32(0.998), np.float32(0)), 1.0, 1)):
        one["angle"] = cosv
        nm, _ = orc.search_by_projection_points(kk, dd, free[:1], bounds, sf, one, dd, th, 0.8)
        assert nm == expect, (cosv, th)
    # a same-level twin at nearly the same distance (10 vs 11 bits): 10 > 0.8 * 11, rejected by the ratio test; the same twin in the
    # next level is not compared, accepted
    two_k = np.concatenate([kk, kk]); two_d = np.concatenate([dd, dd])
    two_d[1, 31] ^= 0x01
    mpd = dd.copy(); mpd[0, 0] ^= 0xFF; mpd[0, 1] ^= 0x03  # 10 bits from keypoint 0, 11 from keypoint 1
    assert orc.hamming256(mpd[0], two_d[0]) == 10 and orc.hamming256(mpd[0], two_d[1]) == 11
    one["u"] = k["x"][i0]; one["angle"] = 0.5; one["octave"] = 1
    nm, _ = orc.search_by_projection_points(two_k, two_d, free[:2], bounds, sf, one, mpd, 1.0, 0.8)
    assert nm == 0
    nm, assign = orc.search_by_projection_points(two_k, two_d, free[:2], bounds, sf, one, mpd, 1.0, 0.95)
    assert nm == 1 and assign[0] == 0
    two_k["octave"][1] = 1
    nm, assign = orc.search_by_projection_points(two_k, two_d, free[:2], bounds, sf, one, mpd, 1.0, 0.8)
    assert nm == 1 and assign[0] == 0 and assign[1] == -1


def test_undistort_keypoints_inverts_the_distortion_model(orc):
    """Frame::UndistortKeyPoints restated in the oracle (reference src/Frame.cc:777-811 over cv::undistortPoints): pushing the
    undistorted points forward through the Brown-Conrady model (k1 k2 p1 p2 k3) must land on the input pixels (five iterations
    converge to under 0.01 px away from the corners for EuRoC's distortion); without distortion the call copies; the image bounds are the
    undistorted corners."""
    k4 = (458.654, 457.296, 367.215, 248.375)
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    rng = np.random.default_rng(0)
    kps = np.zeros(2000, orc.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 752, len(kps)); kps["y"] = rng.uniform(0, 480, len(kps)); kps["octave"] = 3
    un = orc.undistort_keypoints(kps, k4, dist)
    fx, fy, cx, cy = k4
    k1, k2, p1, p2, k3 = dist
    x = (un["x"].astype(np.float64) - cx) / fx; y = (un["y"].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    err = np.hypot(xd * fx + cx - kps["x"], yd * fy + cy - kps["y"])
    # five fixed iterations: converged in the middle of the image, a few tenths of a pixel short in the far corners (as OpenCV is)
    assert err[r2 < 0.5 ** 2].max() < 0.01 and np.median(err) < 0.01 and err.max() < 0.5
    assert (un["octave"] == 3).all()
    assert orc.undistort_keypoints(kps, k4, (0.0, 0.5, 0, 0)).tobytes() == kps.tobytes()
    b = orc.compute_image_bounds(752, 480, k4, dist)
    c = orc.undistort_keypoints(np.array([(0, 0, 0, 0, 0, 0, 0), (752, 0, 0, 0, 0, 0, 0), (0, 480, 0, 0, 0, 0, 0), (752, 480, 0, 0, 0, 0, 0)],
                                         orc.KEYPOINT_DTYPE), k4, dist)
    assert b == (min(c["x"][0], c["x"][2]), max(c["x"][1], c["x"][3]), min(c["y"][0], c["y"][1]), max(c["y"][2], c["y"][3]))
    assert b[0] < 0 and b[1] > 752 and b[2] < 0 and b[3] > 480  # barrel distortion: the undistorted image is larger
    assert orc.compute_image_bounds(752, 480, k4, (0.0,) * 4) == (0.0, 752.0, 0.0, 480.0)


def test_vocabulary_transform_against_a_plain_python_tree_walk(orc):
    """DBoW2 transform restated in the oracle (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1261) against a literal
    Python walk with dicts: first child with the strictly smallest Hamming distance at every level, tf-idf accumulation in feature
    order, L1 normalisation in ascending word order, node `levelsup` levels above the word."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from voc_synth import features_near_words, make_vocabulary
    k, L, levelsup = 4, 3, 1
    parent, leaf, desc, weight = make_vocabulary(k, L, seed=2, p_stop=0.1)
    V = orc.Vocabulary(k, L, 0, 0, parent, leaf, desc, weight)
    children = {}
    for i in range(1, len(parent)):
        children.setdefault(int(parent[i]), []).append(i)
    word_of = {}
    for i in range(1, len(parent)):
        if leaf[i]:
            word_of[i] = len(word_of)
    f = features_near_words(desc, leaf, 300, 3)
    bow, fv = {}, {}
    for i, q in enumerate(f):
        node, level, nid = 0, 0, None
        while node in children:
            level += 1
            ds = [orc.hamming256(q, desc[c]) for c in children[node]]
            node = children[node][int(np.argmin(ds))]  # argmin: the first minimum
            if level == L - levelsup:
                nid = node
        if weight[node] > 0:
            bow[word_of[node]] = bow[word_of[node]] + weight[node] if word_of[node] in bow else weight[node]
            fv.setdefault(nid, []).append(i)
    norm = 0.0
    for wid in sorted(bow):
        norm += abs(bow[wid])
    out = V.transform(f, levelsup)
    assert out["bow_word"].tolist() == sorted(bow) and out["bow_value"].tolist() == [bow[wid] / norm for wid in sorted(bow)]
    assert out["fv_node"].tolist() == sorted(fv)
    for j, nid in enumerate(sorted(fv)):
        assert out["fv_feature"][out["fv_start"][j]:out["fv_start"][j + 1]].tolist() == fv[nid]
    assert abs(out["bow_value"].sum() - 1.0) < 1e-12 and V.size() == len(word_of)


def test_search_by_bow_properties(orc):
    """ORBmatcher::SearchByBoW restated in the oracle (reference src/ORBmatcher1.cc:225-427): a frame against itself matches
    every feature of a non-stopped word to itself; features only meet inside a shared node; a frame feature is given away once;
    keyframe features without a (good) map point never match."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from voc_synth import make_vocabulary
    o = orc.Oracle(600, 1.2, 8, 20, 7)
    _, k, d = o.extract(orc.synth_frame(0x4F524230 + 21, 640, 480), (0, 0))
    parent, leaf, desc, weight = make_vocabulary(5, 3, seed=4, p_stop=0.1)
    V = orc.Vocabulary(5, 3, 0, 0, parent, leaf, desc, weight)
    fv = V.transform(d, 2)
    n = len(k)
    in_fv = np.zeros(n, bool); in_fv[fv["fv_feature"]] = True
    nm, matches = orc.search_by_bow(fv, k, d, np.ones(n, np.uint8), fv, k, d, 0.7, True)
    assert nm == in_fv.sum() and (matches[in_fv] == np.arange(n)[in_fv]).all() and (matches[~in_fv] == -1).all()
    # only the even keyframe features carry map points
    valid = (np.arange(n) % 2 == 0).astype(np.uint8)
    nm, matches = orc.search_by_bow(fv, k, d, valid, fv, k, d, 0.7, False)
    got = matches >= 0
    assert (valid[matches[got]] == 1).all() and len(set(matches[got].tolist())) == got.sum() == nm
    node_of = np.full(n, -1); node_of[fv["fv_feature"]] = np.repeat(fv["fv_node"], np.diff(fv["fv_start"]))
    assert (node_of[matches[got]] == node_of[np.flatnonzero(got)]).all()
    # a second, disjoint node set on the frame side: nothing in common, nothing matched
    other = dict(fv_node=fv["fv_node"] + 100000, fv_start=fv["fv_start"], fv_feature=fv["fv_feature"])
    nm, matches = orc.search_by_bow(fv, k, d, np.ones(n, np.uint8), other, k, d, 0.7, True)
    assert nm == 0 and (matches == -1).all()


def test_glibc_sincosf_restatement(tmp_path):
    """include/orbx_trig.h (what the GPU evaluates for trig_mode 1) against the REAL libm cosf / sinf of this host, on every
    32nd float in [0, 6.5] and every float in [3.0, 3.3] - the angle in radians lies in [0, 2 pi].  (All 1.09e9 floats of the
    range: tests/tools/validate_against_opencv.py --exhaustive-trig; checked clean on glibc 2.35, fused and un-fused.)"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <math.h>
#include <stdio.h>
#include "orbx_trig.h"
int main(void) {
    long n = 0, bad = 0;
    for (uint32_t b = 0; b <= 0x40d00000u; b += (b >= 0x40400000u && b < 0x40533333u) ? 1 : 32) {
        float y; memcpy(&y, &b, 4);
        ++n;
        if (orbx_glibc_sincosf(y, 1) != cosf(y) || orbx_glibc_sincosf(y, 0) != sinf(y)) ++bad;
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
''')
    exe = tmp_path / "t"
    for flags in (["-ffp-contract=off"], ["-ffp-contract=fast", "-mfma"]):
        subprocess.run(["gcc", "-O2", "-std=c99", *flags, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src), "-lm"], check=True)
        r = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
        n, bad = map(int, r.stdout.split())
        assert r.returncode == 0 and bad == 0 and n > 30_000_000, (flags, r.stdout)


def test_oracle_trig_modes_differ_only_in_descriptors(orc):
    """trig_mode 1 (libm cosf / sinf) changes nothing but - rarely - descriptor bits: same keypoints, and almost all descriptors equal."""
    img = orc.synth_frame(3, 640, 480)
    m0, k0, d0 = orc.Oracle(1000, 1.2, 8, 20, 7, trig_mode=0).extract(img, (0, 0))
    m1, k1, d1 = orc.Oracle(1000, 1.2, 8, 20, 7, trig_mode=1).extract(img, (0, 0))
    assert m0 == m1 and k0.tobytes() == k1.tobytes()
    assert (d0 != d1).any(axis=1).mean() < 0.02


def test_kb8_triangulation_recovers_synthetic_points(orc):
    """oracle_kb8_triangulate_matches against an independent double-precision numpy model of the KannalaBrandt8 rig: exact projections
    of known 3-D points triangulate back to them (depth = z in the left camera), noisy / swapped observations are rejected with the
    reference's codes (src/CameraModels/KannalaBrandt8.cpp:316-371)."""
    import fisheye_synth as fs
    X1, uv1, uv2, R12, t12 = fs.scene(300, seed=4)
    ok = 0
    for i in range(len(X1)):
        d, p = orc.kb8_triangulate_matches(fs.CAM1, fs.CAM2, uv1[i], uv2[i], R12, t12, 1.0, 1.0)
        if d > 0:
            ok += 1
            assert abs(d - X1[i, 2]) < 2e-3 * X1[i, 2] + 1e-3, (i, d, X1[i])
            assert np.allclose(p, X1[i], rtol=3e-3, atol=2e-3)
        else:
            assert d == -1 and np.linalg.norm(X1[i]) > 3.5        # only far points fail, on the parallax limit (cos > 0.9998 at 10 cm baseline)
    assert ok > 200
    # same ray in both cameras -> no parallax
    far = np.array([[0.3, 0.2, 900.0]])
    uvf1 = fs.project(fs.CAM1, far); uvf2 = fs.project(fs.CAM2, (far - t12) @ R12.astype(np.float64))
    assert orc.kb8_triangulate_matches(fs.CAM1, fs.CAM2, uvf1[0], uvf2[0], R12, t12, 1.0, 1.0)[0] == -1
    # a wrong partner: large reprojection error or a point behind a camera
    codes = {orc.kb8_triangulate_matches(fs.CAM1, fs.CAM2, uv1[i], uv2[(i + 7) % len(uv2)], R12, t12, 1.0, 1.0)[0] for i in range(60)}
    assert codes & {-2.0, -3.0, -4.0, -5.0}


def test_kb8_smallest_singular_vector_matches_numpy(orc):
    """The fixed one-sided Jacobi restatement of JacobiSVD(A).matrixV().col(3) agrees with numpy's SVD on the triangulated point."""
    import fisheye_synth as fs
    X1, uv1, uv2, R12, t12 = fs.scene(40, seed=9)
    for i in range(len(X1)):
        d, p = orc.kb8_triangulate_matches(fs.CAM1, fs.CAM2, uv1[i], uv2[i], R12, t12, 1e6, 1e6)
        if d <= 0:
            continue
        # rebuild A in double from the same observations
        def unproject(cam, uv):
            c = cam.astype(np.float64)
            pw = np.array([(uv[0] - c[2]) / c[0], (uv[1] - c[3]) / c[1]])
            td = np.hypot(*pw); th = td
            for _ in range(30):
                th -= (th * (1 + c[4] * th**2 + c[5] * th**4 + c[6] * th**6 + c[7] * th**8) - td) / (1 + 3 * c[4] * th**2 + 5 * c[5] * th**4 + 7 * c[6] * th**6 + 9 * c[7] * th**8)
            return pw * np.tan(th) / td
        r1, r2 = unproject(fs.CAM1, uv1[i]), unproject(fs.CAM2, uv2[i])
        R21 = R12.astype(np.float64).T
        T1 = np.hstack([np.eye(3), np.zeros((3, 1))]); T2 = np.hstack([R21, (-R21 @ t12.astype(np.float64))[:, None]])
        A = np.stack([r1[0] * T1[2] - T1[0], r1[1] * T1[2] - T1[1], r2[0] * T2[2] - T2[0], r2[1] * T2[2] - T2[1]])
        v = np.linalg.svd(A)[2][3]
        assert np.allclose(p, v[:3] / v[3], rtol=2e-3, atol=1e-3), i


def test_fisheye_search_by_projection_known_answers(orc):
    """The Nleft != -1 branches restated in the oracle, against hand-traced cases (tests/fisheye_sbp_kat.py): the stereo-partner write that
    frees a keypoint, the `continue`s that skip the right search, the right radius without the th factor."""
    import fisheye_sbp_kat as kat
    for name, mode, kw, n_exp, a_exp in kat.cases(orc.KEYPOINT_DTYPE, orc.PROJECTION_DTYPE):
        kw = dict(kw)
        if mode == "points":
            n, a = orc.search_by_projection_points_fisheye(kw.pop("kc"), kw.pop("dc"), kw.pop("n_left"), kw.pop("cur_state"), kat.BOUNDS, kat.SCALE,
                                                           kw.pop("mp_l"), kw.pop("mp_r"), kw.pop("mp_desc"), kw.pop("left_to_right"),
                                                           kw.pop("right_to_left"), **kw)
        else:
            n, a = orc.search_by_projection_fisheye(kw.pop("kc"), kw.pop("dc"), kw.pop("n_left"), kw.pop("cur_state"), kat.BOUNDS, kat.SCALE,
                                                    kw.pop("proj_l"), kw.pop("proj_r"), kw.pop("proj_desc"), kw.pop("th"), **kw)
        assert n == n_exp and np.array_equal(a, a_exp), (name, n, a)


def _fast9_16_from_the_definition(img, threshold):
    """FAST-9/16 with non-maximum suppression written from its published definition (Rosten & Drummond; cv::FAST TYPE_9_16 docs), not from
    OpenCV's code: a pixel p is a corner at threshold t iff 9 CONTIGUOUS pixels of the 16-pixel Bresenham circle of radius 3 are all
    brighter than I(p) + t or all darker than I(p) - t; its score is the largest t for which that still holds (= the best arc's smallest
    absolute difference, minus 1); a corner survives iff its score is STRICTLY greater than the scores of its 8 neighbours (non-corners
    score 0).  Pure numpy loops - independent of oracle/orb_oracle.cpp's SWAR-free but OpenCV-shaped implementation."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    I = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([I[y, x] - I[y + dy, x + dx] for dx, dy in ring])  # > 0: ring pixel darker
            best = 0
            for s in range(16):
                arc = d[[(s + k) % 16 for k in range(9)]]
                best = max(best, arc.min(), (-arc).min())  # all darker by at least .., all brighter by at least ..
            if best > threshold:
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if s > nb.max():
                    out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,threshold", [(1, 20), (2, 7), (3, 40), (4, 20), (5, 12)])
def test_fast_against_its_definition(orc, seed, threshold):
    """The oracle's cv::FAST restatement (fast_roi: one cell ROI, NMS on) against a brute-force implementation of the detector's definition:
    same corners, same scores, same (raster) order - on noisy blobs, checkerboards and ramps, at the thresholds the extractor uses."""
    rng = np.random.default_rng(seed)
    h, w = 41, 47
    img = rng.integers(60, 90, (h, w)).astype(np.int32)
    for _ in range(10):  # bright / dark rectangles and a few single pixels
        x0, y0 = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
        img[y0:y0 + int(rng.integers(2, 12)), x0:x0 + int(rng.integers(2, 12))] = int(rng.integers(0, 256))
    if seed == 4:
        yy, xx = np.mgrid[0:h, 0:w]
        img = (((xx // 6 + yy // 6) % 2) * 120 + 40 + rng.integers(-5, 6, (h, w)))  # checkerboard: equal scores next to each other
    if seed == 5:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.clip(xx * 5 + yy * 3 + rng.integers(-8, 9, (h, w)), 0, 255)         # ramps: both polarities pass the quick test
    img = np.clip(img, 0, 255).astype(np.uint8)
    kp = orc.fast_roi(img, threshold)
    got = list(zip((kp & 0xFFF).tolist(), ((kp >> 12) & 0xFFF).tolist(), (kp >> 24).tolist()))
    want = _fast9_16_from_the_definition(img, threshold)
    assert got == [(x, y, s) for x, y, s in want]
    if seed in (1, 2):
        assert len(got) >= 3


def test_resize_and_blur_stay_within_one_level_of_their_real_valued_definitions(orc):
    """cv::resize(INTER_LINEAR) and cv::GaussianBlur(7x7, sigma 2, REFLECT_101) are fixed-point implementations of real-valued operators;
    whatever the fixed-point details (which only a real OpenCV can pin), the oracle's restatement must stay within one grey level of the
    operator's definition: bilinear interpolation at source coordinate (d + 0.5) * src / dst - 0.5 with edge replication, and the
    convolution with the normalised 7-tap Gaussian (4.5.1 taps / 256; the 4.5.0 taps sum to 257 / 256: + 0.8 % gain)."""
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:97, 0:131]
    img = np.clip(100 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.integers(-20, 21, xx.shape), 0, 255).astype(np.uint8)
    for dw, dh in ((109, 81), (131, 97), (66, 49)):
        out = orc.resize_linear(img, dw, dh).astype(np.float64)
        sx = np.clip((np.arange(dw) + 0.5) * (img.shape[1] / dw) - 0.5, 0, img.shape[1] - 1)
        sy = np.clip((np.arange(dh) + 0.5) * (img.shape[0] / dh) - 0.5, 0, img.shape[0] - 1)
        x0 = np.floor(sx).astype(int); x1 = np.minimum(x0 + 1, img.shape[1] - 1); fx = sx - x0
        y0 = np.floor(sy).astype(int); y1 = np.minimum(y0 + 1, img.shape[0] - 1); fy = sy - y0
        I = img.astype(np.float64)
        want = ((I[np.ix_(y0, x0)] * (1 - fx) + I[np.ix_(y0, x1)] * fx) * (1 - fy)[:, None] + (I[np.ix_(y1, x0)] * (1 - fx) + I[np.ix_(y1, x1)] * fx) * fy[:, None])
        assert np.abs(out - want).max() <= 1.0, (dw, dh, np.abs(out - want).max())
    for variant, taps in ((1, [18, 34, 48, 56, 48, 34, 18]), (0, [18, 34, 49, 55, 49, 34, 18])):
        out = orc.gaussian_blur7(img, variant).astype(np.float64)
        k = np.array(taps, np.float64) / 256.0
        pad = np.pad(img.astype(np.float64), 3, mode="reflect")  # numpy "reflect" = BORDER_REFLECT_101
        tmp = sum(k[i] * pad[:, i:i + img.shape[1]] for i in range(7))
        want = sum(k[i] * tmp[i:i + img.shape[0], :] for i in range(7))
        assert np.abs(out - np.minimum(want, 255.0)).max() <= 1.0, (variant, np.abs(out - want).max())


# ---- projection head of the tracking searches (round 4): oracle_project_last_frame_points / oracle_project_in_frustum ----------------
def _world_points(orc, pos, **fields):
    pts = np.zeros(len(pos), orc.WORLD_POINT_DTYPE)
    pts["pos"] = np.float32(pos)
    pts["flags"] = 3
    pts["max_distance"] = 10.0
    pts["min_distance"] = 0.1
    pts["normal"] = [0, 0, -1]
    for k, v in fields.items():
        pts[k] = v
    return pts


def test_projection_head_motion_model_known_answers(orc):
    """Sophus::SE3f * point + Pinhole::project (reference src/ORBmatcher3.cc:283-293): identity, pure translation, a quarter turn about the
    optical axis (exact in float: q = (0, 0, s, s) with s = sqrt(0.5) gives 2 s^2 = 1 up to rounding), and agreement with a double-precision
    rotation matrix to float accuracy for random poses."""
    cam = (0, (400.0, 300.0, 320.0, 240.0))
    pts = _world_points(orc, [[0, 0, 2], [1, 0, 4], [0, -1, 5], [0.5, 0.25, -2]], angle=[1, 2, 3, 4], octave=[0, 1, 2, 3])
    out = orc.project_last_frame_points([0, 0, 0, 1], [0, 0, 0], *cam, pts)
    assert list(out["u"]) == [320.0, 420.0, 320.0, 220.0] and list(out["v"]) == [240.0, 240.0, 180.0, 202.5]
    assert list(out["invz"]) == [0.5, 0.25, np.float32(0.2), -0.5]              # behind the camera: invzc < 0, the search skips it (:290)
    assert list(out["angle"]) == [1, 2, 3, 4] and list(out["octave"]) == [0, 1, 2, 3] and list(out["flags"]) == [3] * 4
    out = orc.project_last_frame_points([0, 0, 0, 1], [0.5, 0, 2], *cam, pts[:2])
    assert list(out["u"]) == [370.0, 420.0] and list(out["invz"]) == [0.25, np.float32(1.0 / 6.0)]
    rng = np.random.default_rng(5)
    for _ in range(20):
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis); ang = rng.uniform(-1, 1)
        q = np.float32(np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])); t = np.float32(rng.normal(0, 0.3, 3))
        P = rng.uniform(-2, 2, (50, 3)); P[:, 2] += 6
        out = orc.project_last_frame_points(q, t, *cam, _world_points(orc, P))
        qd = q.astype(np.float64); qd /= np.linalg.norm(qd)
        x, y, z, w = qd
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        Xc = np.float32(P).astype(np.float64) @ R.T + t
        assert np.allclose(out["u"], 400 * Xc[:, 0] / Xc[:, 2] + 320, atol=2e-3) and np.allclose(out["invz"], 1 / Xc[:, 2], rtol=1e-5)


def test_projection_head_in_frustum_each_test_rejects_on_its_own(orc):
    """Frame::isInFrustum (reference src/Frame.cc:542-600): every rejection cause alone clears bit 0 and nothing else; the track data of an
    accepted point are the reference's formulas (mTrackProjXR = u - mbf / z, viewCos, PredictScale)."""
    cam = (0, (400.0, 300.0, 320.0, 240.0))
    bounds = (0.0, 640.0, 0.0, 480.0)
    R, t, Ow = np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    logsf = float(np.log(np.float32(1.2)))
    base = dict(mbf=40.0, viewing_cos_limit=0.5, log_scale_factor=logsf, nlevels=8)
    ok = _world_points(orc, [[0, 0, 4.0]], normal=[[0, 0, 1]], max_distance=4.0 * 1.2 ** 3 * 0.999, min_distance=0.5)
    out = orc.project_in_frustum(R, t, Ow, *cam, bounds, points=ok, **base)
    assert out["flags"][0] == 3 and out["u"][0] == 320.0 and out["v"][0] == 240.0
    assert out["invz"][0] == np.float32(320.0) - np.float32(40.0) * np.float32(0.25)         # mTrackProjXR
    assert out["angle"][0] == 1.0 and out["octave"][0] == 3                                   # viewCos; ceil(log(1.2^3 * 0.999) / log 1.2) = 3
    cases = {
        "behind": _world_points(orc, [[0, 0, -4.0]], normal=[[0, 0, -1]]),
        "left of the image": _world_points(orc, [[-4.0, 0, 4.0]], normal=[[-0.7, 0, 0.7]]),
        "below the image": _world_points(orc, [[0, 4.0, 4.0]], normal=[[0, 0.7, 0.7]]),
        "too far (> 1.2 max)": _world_points(orc, [[0, 0, 4.0]], normal=[[0, 0, 1]], max_distance=3.3),
        "too close (< 0.8 min)": _world_points(orc, [[0, 0, 4.0]], normal=[[0, 0, 1]], min_distance=5.1),
        "viewing angle > 60 deg": _world_points(orc, [[0, 0, 4.0]], normal=[[0.9, 0, 0.4358899]]),
        "not a candidate on entry": _world_points(orc, [[0, 0, 4.0]], normal=[[0, 0, 1]], flags=2),
    }
    for name, pts in cases.items():
        out = orc.project_in_frustum(R, t, Ow, *cam, bounds, points=pts, **base)
        assert out["flags"][0] == 2, name
    far = orc.project_in_frustum(R, t, Ow, *cam, bounds, points=ok, far_points=True, th_far_points=3.9, **base)
    near = orc.project_in_frustum(R, t, Ow, *cam, bounds, points=ok, far_points=True, th_far_points=4.1, **base)
    assert far["flags"][0] == 2 and near["flags"][0] == 3                                      # src/ORBmatcher1.cc:57-58 on mTrackDepth
    # PredictScale clamps (src/MapPoint.cc:540-543)
    lo = _world_points(orc, [[0, 0, 4.0]], normal=[[0, 0, 1]], max_distance=3.5, min_distance=0.1)     # ratio < 1: level 0
    hi = _world_points(orc, [[0, 0, 0.5]], normal=[[0, 0, 1]], max_distance=40.0, min_distance=0.1)    # ratio 80: level 7
    assert orc.project_in_frustum(R, t, Ow, *cam, bounds, points=lo, **base)["octave"][0] == 0
    assert orc.project_in_frustum(R, t, Ow, *cam, bounds, points=hi, **base)["octave"][0] == 7


def test_projection_head_kb8_agrees_with_the_fisheye_model(orc):
    """KannalaBrandt8::project (reference src/CameraModels/KannalaBrandt8.cpp:67-84) through the head: r(theta) polynomial along the ray's azimuth."""
    k = (190.0, 191.0, 255.0, 256.0, 0.0035, 0.0007, -0.002, 0.0002)
    rng = np.random.default_rng(6)
    P = rng.uniform(-3, 3, (200, 3)); P[:, 2] = rng.uniform(0.5, 6, 200)
    out = orc.project_last_frame_points([0, 0, 0, 1], [0, 0, 0], 1, k, _world_points(orc, P))
    Pf = np.float32(P).astype(np.float64)
    th = np.arctan2(np.hypot(Pf[:, 0], Pf[:, 1]), Pf[:, 2]); psi = np.arctan2(Pf[:, 1], Pf[:, 0])
    r = th + k[4] * th ** 3 + k[5] * th ** 5 + k[6] * th ** 7 + k[7] * th ** 9
    assert np.allclose(out["u"], k[0] * r * np.cos(psi) + k[2], atol=1e-3) and np.allclose(out["v"], k[1] * r * np.sin(psi) + k[3], atol=1e-3)


def test_glibc_libm_restatements(tmp_path):
    """include/orbx_libm.h (what the GPU evaluates for logf / atanf / atan2f / tanf: MapPoint::PredictScale src/MapPoint.cc:538,
    KannalaBrandt8::project / unproject src/CameraModels/KannalaBrandt8.cpp:69-70,139) against the REAL libm of this host: every 509th float
    for the one-argument functions, 5 x 4 M argument pairs for atan2f.  The exhaustive run (stride 1, 5 x 10^9 pairs: a minute on 8 cores)
    is recorded in profiles/r05_libm_pin.txt."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "libm_pin"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "tools", "libm_pin.cpp"), "-o", str(exe), "-lm"])
    r = subprocess.run([str(exe), "509", "4", "4"], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "0 mismatches" in r.stdout and "OK:" in r.stdout, r.stdout


def test_predict_scale_is_the_reference_float_arithmetic(orc):
    """MapPoint::PredictScale (src/MapPoint.cc:531-546): `ceil(log(ratio) / mfLogScaleFactor)` is std::log(float) / float / std::ceil(float)
    (using namespace std reaches MapPoint.cc through TemplatedVocabulary.h:36).  Inputs: ratio = fl(fl(d * 1.2^l) / d), a map point seen from
    the distance of its reference observation - where the float and the double evaluation (rounds 1-4) disagree on a tenth of the cases."""
    import scene3d
    lsf = scene3d.libm_logf(np.float32(1.2))
    assert float(lsf).hex() == "0x1.7565060000000p-3"               # the correctly rounded log(1.2f); numpy's float32 log returns ...08p-3
    rng = np.random.default_rng(5)
    differ = 0
    for _ in range(4000):
        d = np.float32(rng.uniform(0.5, 30.0)); l = int(rng.integers(0, 10))
        ratio = np.float32(np.float32(d * np.float32(1.2) ** np.float32(l)) / d)
        want = scene3d.predict_scale_float(ratio, lsf, 8)
        assert orc.predict_scale(ratio, lsf, 8) == want, (d, l, ratio)
        differ += want != scene3d.predict_scale_double(ratio, lsf, 8)
    assert differ > 100                                            # the inputs tell the two evaluations apart
    for _ in range(2000):                                          # generic ratios
        ratio = np.float32(rng.uniform(0.2, 6.0))
        assert orc.predict_scale(ratio, lsf, 8) == scene3d.predict_scale_float(ratio, lsf, 8)
    # clamps and the degenerate ratios (dist = 0 -> inf, 0 / 0 -> NaN: int conversion as x86-64 does it, INT_MIN -> level 0)
    assert orc.predict_scale(1e-3, lsf, 8) == 0 and orc.predict_scale(1e6, lsf, 8) == 7 and orc.predict_scale(1.0, lsf, 8) == 0
    assert orc.predict_scale(float("inf"), lsf, 8) == 0 and orc.predict_scale(float("nan"), lsf, 8) == 0


def _wp(orc, rows):
    pts = np.zeros(len(rows), orc.WORLD_POINT_DTYPE)
    for i, (pos, normal, mn, mx, flags) in enumerate(rows):
        pts[i]["pos"] = pos; pts[i]["normal"] = normal; pts[i]["min_distance"] = mn; pts[i]["max_distance"] = mx; pts[i]["flags"] = flags
    return pts


def test_window_point_head_known_answers(orc):
    """The projection head of Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame) (reference src/ORBmatcher2.cc:446-520,
    src/ORBmatcher1.cc:451-494, src/ORBmatcher3.cc:486-519): identity pose, pinhole - every value by hand, every test rejecting on its own."""
    import scene3d
    cam = (400.0, 400.0, 320.0, 240.0)
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    lsf = scene3d.libm_logf(np.float32(1.2))
    I = [0, 0, 0, 1]
    Z = [0, 0, 0]
    pts = _wp(orc, [((0, 0, 2), (0, 0, 1), 0.5, 4.0, 1),          # on the axis, in range: ratio 4 / 2 -> level ceil(log 2 / log 1.2) = 4
                    ((0, 0, -2), (0, 0, -1), 0.5, 4.0, 1),        # behind the camera
                    ((1.6, 0, 2), (0, 0, 1), 0.5, 4.0, 1),        # u = 320 + 400 * 0.8 = 640: KeyFrame::IsInImage (x < mnMaxX) rejects, the Frame form accepts
                    ((0, 0, 2), (0, 0, 1), 3.0, 4.0, 1),          # dist 2 < 0.8 * 3
                    ((0, 0, 2), (0, 0, 1), 0.5, 1.5, 1),          # dist 2 > 1.2 * 1.5
                    ((0, 0, 2), (1, 0, 0), 0.5, 4.0, 1),          # viewing direction 90 degrees off the normal
                    ((0, 0, 2), (0, 0, 1), 0.5, 4.0, 0)])         # the caller's own skip
    bounds = (0.0, 640.0, 0.0, 480.0)
    w = orc.project_window_points(I, Z, Z, 0, cam, bounds, 40.0, 3.0, sf, lsf, 3, pts)
    assert w["search"].tolist() == [1, 0, 0, 0, 0, 0, 0]
    assert (w["u"][0], w["v"][0], w["ur"][0], w["level"][0]) == (320.0, 240.0, 300.0, 4) and w["radius"][0] == np.float32(3.0) * sf[4]
    assert w["u"][2] == 640.0
    wf = orc.project_window_points(I, Z, Z, 0, cam, bounds, 40.0, 3.0, sf, lsf, 4, pts)     # Frame form: inclusive bounds, no depth / angle test
    assert wf["search"].tolist() == [1, 1, 1, 0, 0, 1, 0]
    w1 = orc.project_window_points(I, Z, Z, 0, cam, bounds, 40.0, 3.0, sf, lsf, 1, pts)     # depth test only
    assert w1["search"].tolist() == [1, 0, 0, 0, 0, 1, 0]
    # a translated / rotated pose: Ow = -R^T t; a quarter turn about z maps (x, y, z) -> (-y, x, z)
    q = [0.0, 0.0, np.sqrt(0.5), np.sqrt(0.5)]
    p2 = _wp(orc, [((0.5, 0.25, 2.0), (0, 0, 1), 0.1, 50.0, 1)])
    w2 = orc.project_window_points(q, [0.1, -0.2, 0.3], [0.2, 0.1, -0.3], 0, cam, bounds, 0.0, 1.0, sf, lsf, 3, p2)
    xc = np.float32([-0.25 + 0.1, 0.5 - 0.2, 2.3])
    assert abs(w2["u"][0] - (400 * xc[0] / xc[2] + 320)) < 1e-3 and abs(w2["v"][0] - (400 * xc[1] / xc[2] + 240)) < 1e-3 and w2["search"][0] == 1


def test_fisheye_frustum_head_equals_the_single_camera_head_per_side(orc):
    """Frame::isInFrustum of a fisheye rig (reference src/Frame.cc:604-614 + isInFrustumChecks :1198-1272): the left records are the single-camera
    head's with the left pose, the right ones the single-camera head's with the pose the reference forms per call (mR = Rrl * mRcw, mt = Rrl *
    mtcw + trl, twc = mRwc * tlr + mOw; 3-term sums x0 + (x1 + x2)), and the far-point test of SearchByProjection reads the LEFT depth for both."""
    import scene3d
    rng = np.random.default_rng(12)
    lsf = scene3d.libm_logf(np.float32(1.2))
    cam1 = scene3d.KB8[1]; cam2 = tuple(np.float32(cam1) * np.float32([1.01, 0.99, 1.0, 1.0, 1.1, 0.9, 1.0, 1.0]))
    keys = np.zeros(3000, [("x", "<f4"), ("y", "<f4"), ("angle", "<f4"), ("octave", "<i4")])
    keys["x"] = rng.uniform(20, 490, 3000); keys["y"] = rng.uniform(20, 490, 3000)
    pts, q, t, R, Ow = scene3d.make_scene(rng, keys, scene3d.KB8, orc.WORLD_POINT_DTYPE, motion=0.2, rot=0.05, behind=0.05)
    Rrl = scene3d.rotation_matrix(scene3d.quat_from_axis_angle([0.1, 1.0, 0.05], 0.4)); trl = np.float32([-0.101, 0.002, 0.001])   # (a wide-angle rig: the two fields of view differ)
    tlr = (-(Rrl.astype(np.float64).T @ trl.astype(np.float64))).astype(np.float32)
    Rwc = np.ascontiguousarray(R.T)
    bounds = (0.0, 512.0, 0.0, 512.0)
    s3 = lambda a, b, c: np.float32(a + np.float32(b + c))
    mR = np.float32([[s3(Rrl[i, 0] * R[0, j], Rrl[i, 1] * R[1, j], Rrl[i, 2] * R[2, j]) for j in range(3)] for i in range(3)])
    mt = np.float32([s3(Rrl[i, 0] * t[0], Rrl[i, 1] * t[1], Rrl[i, 2] * t[2]) + trl[i] for i in range(3)])
    twc = np.float32([s3(Rwc[i, 0] * tlr[0], Rwc[i, 1] * tlr[1], Rwc[i, 2] * tlr[2]) + Ow[i] for i in range(3)])
    for far in (False, True):
        L, Rr = orc.project_in_frustum_fisheye(R, t, Ow, Rrl, trl, tlr, Rwc, cam1, cam2, bounds, 0.5, lsf, 8, pts, far_points=far, th_far_points=8.0)
        l1 = orc.project_in_frustum(R, t, Ow, 1, cam1, bounds, 0.0, 0.5, lsf, 8, pts)
        r1 = orc.project_in_frustum(mR, mt, twc, 1, cam2, bounds, 0.0, 0.5, lsf, 8, pts)
        for got, ref in ((L, l1), (Rr, r1)):
            for k in ("u", "v", "angle", "octave"):
                assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
        depthL = L["invz"]                                                   # mTrackDepth
        live = (pts["flags"] & 1).astype(bool) & ~(far & (depthL > 8.0))
        assert np.array_equal((L["flags"] & 1).astype(bool), live & (l1["flags"] & 1).astype(bool))
        assert np.array_equal((Rr["flags"] & 1).astype(bool), live & (r1["flags"] & 1).astype(bool))
        assert np.array_equal(L["flags"] & ~1, pts["flags"] & ~1)
    assert 100 < (L["flags"] & 1).sum() < len(pts) and 100 < (Rr["flags"] & 1).sum() and ((L["flags"] ^ Rr["flags"]) & 1).sum() > 10
    assert (depthL > 8.0).sum() > 50
