"""Committed golden vectors (tests/golden/orb_golden.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (guards oracle drift + the synthetic generator).
GPU: the HIP path reproduces them through the C ABI with no oracle in the loop."""
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "orb_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files} - {"search_init", "tracking"})


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden(orc, case):
    w, h, nf, idx, lap0, lap1, mono, n = GOLD[case + "/meta"].tolist()
    img = orc.synth_frame(idx, w, h)
    assert zlib.crc32(img.tobytes()) == int(GOLD[case + "/input_crc"][0])
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    m, kps, desc = o.extract(img, (lap0, lap1))
    assert (m, len(kps)) == (mono, n)
    assert np.array_equal(kps.view(np.uint8).reshape(-1, 28), GOLD[case + "/kps"])
    assert np.array_equal(desc, GOLD[case + "/desc"])
    lv = GOLD[case + "/levels"]
    for l in range(8):
        assert zlib.crc32(o.level(l).tobytes()) == lv[l, 0]
        assert zlib.crc32(o.level(l, True).tobytes()) == lv[l, 1]
        assert zlib.crc32(o.candidates(l).tobytes()) == lv[l, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_path_reproduces_golden(pkg, case):
    w, h, nf, idx, lap0, lap1, mono, n = GOLD[case + "/meta"].tolist()
    img = pkg.synth_frame(idx, w, h)
    assert zlib.crc32(img.tobytes()) == int(GOLD[case + "/input_crc"][0])
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    m, kps, desc = ex(img, None, (lap0, lap1))
    assert (m, len(kps)) == (mono, n)
    assert np.array_equal(kps.view(np.uint8).reshape(-1, 28), GOLD[case + "/kps"])
    assert np.array_equal(desc, GOLD[case + "/desc"])
    lv = GOLD[case + "/levels"]
    for l in range(8):
        assert zlib.crc32(ex.pyramid_level(0, l).tobytes()) == lv[l, 0], f"pyramid level {l}"
        assert zlib.crc32(ex.pyramid_level(0, l, True).tobytes()) == lv[l, 1], f"blurred level {l}"
        cand = ex.debug_candidates(0, l)
        assert len(cand) == lv[l, 3] and zlib.crc32(cand.tobytes()) == lv[l, 2], f"candidates level {l}"
        assert len(ex.debug_selected(0, l)) == lv[l, 4]
    ex.close()


def _search_init_inputs(mod, extract):
    img1 = mod.synth_frame(7, 640, 480)
    rng = np.random.default_rng(1)
    img2 = np.clip(np.roll(np.roll(img1, -6, axis=0), 9, axis=1).astype(np.int16) + rng.integers(-3, 4, img1.shape, dtype=np.int16), 0, 255)
    img2 = img2.astype(np.uint8)
    assert [zlib.crc32(img1.tobytes()), zlib.crc32(img2.tobytes())] == GOLD["search_init/input_crc"].tolist()
    k1, d1 = extract(img1)
    k2, d2 = extract(img2)
    return k1, d1, k2, d2, np.stack([k1["x"], k1["y"]], axis=1)


def test_oracle_reproduces_search_for_initialization_golden(orc):
    o = orc.Oracle(5000, 1.2, 8, 20, 7)
    k1, d1, k2, d2, prev = _search_init_inputs(orc, lambda im: o.extract(im, (0, 0))[1:])
    n, m12, pv = orc.search_for_initialization(k1, d1, k2, d2, prev, (0, 640, 0, 480), 100, 0.9, True)
    n2, m12b, _ = orc.search_for_initialization(k1, d1, k2, d2, pv, (0, 640, 0, 480), 100, 0.9, True)
    assert [n, n2] == GOLD["search_init/nmatches"].tolist()
    assert np.array_equal(np.stack([m12, m12b]), GOLD["search_init/matches12"])
    assert zlib.crc32(np.concatenate(orc.assign_grid(k2, (0, 640, 0, 480))).tobytes()) == int(GOLD["search_init/grid_crc"][0])


@pytest.mark.gpu
def test_hip_path_reproduces_search_for_initialization_golden(pkg):
    ex = pkg.ORBextractor(5000, 1.2, 8, 20, 7)

    def extract(im):
        _, k, d = ex(im, None, (0, 0))
        return k.copy(), d.copy()
    k1, d1, k2, d2, prev = _search_init_inputs(pkg, extract)
    m = pkg.ORBmatcher(8192, 8192)
    n, m12, pv = m.SearchForInitialization(k1, d1, k2, d2, prev, (0, 640, 0, 480), 100, 0.9, True)
    n2, m12b, _ = m.SearchForInitialization(k1, d1, k2, d2, pv, (0, 640, 0, 480), 100, 0.9, True)
    assert [n, n2] == GOLD["search_init/nmatches"].tolist()
    assert np.array_equal(np.stack([m12, m12b]), GOLD["search_init/matches12"])
    assert zlib.crc32(np.concatenate(m.AssignFeaturesToGrid(k2, (0, 640, 0, 480))).tobytes()) == int(GOLD["search_init/grid_crc"][0])
    m.close(); ex.close()


# ---- one tracking scenario (tests/golden_scenarios.py) ------------------------------------------------------
import golden_scenarios as gs  # noqa: E402


def _check_tracking(k1, d1, k2, d2, sf, sbp, sbp_points, undistort, bounds, make_voc, search_by_bow, projection_dtype):
    assert [zlib.crc32(k1.tobytes() + d1.tobytes()), zlib.crc32(k2.tobytes() + d2.tobytes())] == GOLD["tracking/input_crc"][2:].tolist()
    pr, pm, state, ur, valid = gs.tracking_inputs(k1, k2, projection_dtype)
    na, aa = sbp(k2, d2, state, sf, pr, d1, 15.0, None, 0.0)
    nb, ab = sbp(k2, d2, state, sf, pr, d1, 7.0, ur, 40.0)
    nc, ac = sbp_points(k2, d2, state, sf, pm, d1, 3.0, 0.8, ur)
    assert [na, nb, nc] == GOLD["tracking/sbp_nmatches"].tolist()
    assert np.array_equal(np.stack([aa, ab, ac]), GOLD["tracking/sbp_assign"])
    k4, dist = gs.CAMERA
    un = undistort(k2, k4, dist)
    assert np.stack([un["x"], un["y"]], axis=1).tobytes() == GOLD["tracking/undistort_xy"].tobytes()
    assert np.array_equal(np.array(bounds(752, 480, k4, dist), np.float32), GOLD["tracking/bounds"])
    parent, leaf, vdesc, weight = gs.vocabulary(d1)
    assert zlib.crc32(parent.tobytes() + leaf.tobytes() + vdesc.tobytes() + weight.tobytes()) == int(GOLD["tracking/voc_crc"][0])
    voc = make_voc(parent, leaf, vdesc, weight)
    t1, t2 = voc.transform(d1, 2), voc.transform(d2, 2)
    for tag, t in (("bow1", t1), ("bow2", t2)):
        assert np.array_equal(t["bow_word"], GOLD["tracking/%s_word" % tag])
        assert t["bow_value"].tobytes() == GOLD["tracking/%s_value" % tag].tobytes()  # doubles, bit for bit
        assert [zlib.crc32(t["fv_node"].tobytes() + t["fv_start"].tobytes() + t["fv_feature"].tobytes()),
                zlib.crc32(t["feat_word"].tobytes() + t["feat_node"].tobytes())] == GOLD["tracking/%s_fv_crc" % tag].tolist()
    nm, mm = search_by_bow(t1, k1, d1, valid, t2, k2, d2)
    assert nm == int(GOLD["tracking/bow_nmatches"][0]) and np.array_equal(mm, GOLD["tracking/bow_matches"])


def test_oracle_reproduces_tracking_golden(orc):
    img1, img2 = gs.frames(orc.synth_frame)
    assert [zlib.crc32(img1.tobytes()), zlib.crc32(img2.tobytes())] == GOLD["tracking/input_crc"][:2].tolist()
    o = orc.Oracle(1200, 1.2, 8, 20, 7)
    _, k1, d1 = o.extract(img1, (0, 0))
    _, k2, d2 = o.extract(img2, (0, 0))
    _check_tracking(k1, d1, k2, d2, o.tables()["scale"],
                    lambda kc, dc, st, sf, pr, dp, th, ur, mbf: orc.search_by_projection(kc, dc, st, gs.BOUNDS, sf, pr, dp, th, ur, mbf, False, False, True),
                    lambda kc, dc, st, sf, pm, dp, th, ratio, ur: orc.search_by_projection_points(kc, dc, st, gs.BOUNDS, sf, pm, dp, th, ratio, ur),
                    orc.undistort_keypoints, orc.compute_image_bounds,
                    lambda p, l, d, w: orc.Vocabulary(gs.VOC[0], gs.VOC[1], 0, 0, p, l, d, w),
                    lambda t1, k1_, d1_, v, t2, k2_, d2_: orc.search_by_bow(t1, k1_, d1_, v, t2, k2_, d2_, 0.7, True), orc.PROJECTION_DTYPE)


@pytest.mark.gpu
def test_hip_path_reproduces_tracking_golden(pkg):
    img1, img2 = gs.frames(pkg.synth_frame)
    assert [zlib.crc32(img1.tobytes()), zlib.crc32(img2.tobytes())] == GOLD["tracking/input_crc"][:2].tolist()
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    _, k2, d2 = ex(img2, None, (0, 0)); k2, d2 = k2.copy(), d2.copy()
    m = pkg.ORBmatcher(8192, 8192)
    _check_tracking(k1, d1, k2, d2, ex.GetScaleFactors().copy(),
                    lambda kc, dc, st, sf, pr, dp, th, ur, mbf: m.SearchByProjection(kc, dc, st, gs.BOUNDS, sf, pr, dp, th, u_right=ur, mbf=mbf),
                    lambda kc, dc, st, sf, pm, dp, th, ratio, ur: m.SearchByProjectionPoints(kc, dc, st, gs.BOUNDS, sf, pm, dp, th, ratio, u_right=ur),
                    m.UndistortKeyPoints, m.ComputeImageBounds,
                    lambda p, l, d, w: pkg.ORBVocabulary(gs.VOC[0], gs.VOC[1], 0, 0, parent=p, is_leaf=l, desc=d, weight=w),
                    lambda t1, k1_, d1_, v, t2, k2_, d2_: m.SearchByBoW(t1, k1_, d1_, v, t2, k2_, d2_, 0.7, True), pkg.PROJECTION_DTYPE)
    m.close(); ex.close()
