"""Committed golden vectors (tests/golden/orb_golden.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (guards oracle drift + the synthetic generator).
GPU: the HIP path reproduces them through the C ABI with no oracle in the loop."""
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "orb_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


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
