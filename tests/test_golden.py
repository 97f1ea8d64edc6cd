"""Committed golden vectors (tests/golden/orb_golden.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (guards oracle drift + the synthetic generator).
GPU: the HIP path reproduces them through the C ABI with no oracle in the loop."""
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "orb_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files} - {"search_init"})


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
