"""GPU parity of the Hamming matcher leafs against the CPU oracle (bit-exact integer work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _desc(rng, n, dup=0.0):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if dup > 0 and n > 4:
        k = int(n * dup)
        d[rng.integers(0, n, k)] = d[rng.integers(0, n, k)]  # exact duplicates -> distance ties
    return d


@pytest.mark.parametrize("nq,nt", [(1, 1), (7, 300), (257, 513), (1000, 1000)])
def test_hamming_matrix(pkg, orc, nq, nt):
    rng = np.random.default_rng(nq * 31 + nt)
    q, t = _desc(rng, nq), _desc(rng, nt)
    m = pkg.ORBmatcher(1024, 1024)
    got = m.hamming_matrix(q, t)
    want = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2).astype(np.int32)
    assert np.array_equal(got, want)
    assert got[0, 0] == orc.hamming256(q[0], t[0])
    m.close()


@pytest.mark.parametrize("nq,nt", [(5, 0), (5, 1), (3, 2), (64, 65), (500, 777), (4000, 4000)])
def test_knn2_matches_bfmatcher_semantics(pkg, orc, nq, nt):
    rng = np.random.default_rng(nq + 7 * nt)
    q, t = _desc(rng, nq), _desc(rng, nt, dup=0.3)
    if nt >= 3:
        q[: min(nq, 3)] = t[: min(nq, 3)]  # zero-distance matches with duplicate train rows -> index tie-break
        t[-1] = t[0]
    m = pkg.ORBmatcher(4096, 4096)
    idx, dist = m.knnMatch(q, t)
    ridx, rdist = orc.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    if nt >= 2:  # ties resolved towards the lower train index (strict `<` insertion)
        tie = dist[:, 0] == dist[:, 1]
        assert np.all(idx[tie, 0] < idx[tie, 1])
    if nt == 1:
        assert np.all(idx[:, 1] == -1) and np.all(dist[:, 1] == np.iinfo(np.int32).max)
    m.close()


def test_knn2_on_real_descriptors_and_lowe_ratio(pkg, orc):
    """config C5-style cross-match: left/right descriptors of shifted frames; ratio test as in Frame.cc:1181."""
    img = pkg.synth_frame(77, 752, 480)
    right = np.roll(img, -12, axis=1)
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=752, max_height=480)
    _, kl, dl = ex(img, None, (0, 0))
    _, kr, dr = ex(right, None, (0, 0))
    m = pkg.ORBmatcher(2048, 2048)
    idx, dist = m.knnMatch(dl, dr)
    ridx, rdist = orc.knn2(dl, dr)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    good = dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * 0.7
    assert good.sum() > 300  # the shifted copy is matched
    dx = kl["x"][good] - kr["x"][idx[good, 0]]
    assert np.median(dx) == pytest.approx(12.0, abs=1.5)
    ex.close()
    m.close()


def test_stereo_row_band_matcher(pkg, orc):
    """Hamming stage of Frame::ComputeStereoMatches (reference src/Frame.cc:841-925)."""
    img = pkg.synth_frame(90, 752, 480)
    right = np.roll(img, -20, axis=1)
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=752, max_height=480)
    _, kl, dl = ex(img, None, (0, 0))
    _, kr, dr = ex(right, None, (0, 0))
    sf = ex.GetScaleFactors()
    m = pkg.ORBmatcher(2048, 2048)
    for (mind, maxd) in [(0.0, 96.0), (0.0, 10.0), (15.0, 25.0)]:
        bi, bd = m.stereo_rows(kl, dl, kr, dr, 480, sf, mind, maxd)
        rbi, rbd = orc.stereo_rows(kl, dl, kr, dr, 480, sf, mind, maxd)
        assert np.array_equal(bi, rbi) and np.array_equal(bd, rbd)
    bi, bd = m.stereo_rows(kl, dl, kr, dr, 480, sf, 0.0, 96.0)
    ok = bd < (pkg.TH_HIGH + pkg.TH_LOW) // 2
    assert ok.sum() > 300
    assert np.median(kl["x"][ok] - kr["x"][bi[ok]]) == pytest.approx(20.0, abs=1.5)
    # empty right set
    bi, bd = m.stereo_rows(kl, dl, kr[:0], dr[:0], 480, sf, 0.0, 96.0)
    assert np.all(bi == -1) and np.all(bd == pkg.TH_HIGH)
    ex.close()
    m.close()


@pytest.mark.parametrize("shift,mb,mbf", [(20, 0.11, 47.9), (7, 0.11, 47.9), (35, 0.5373, 386.1448)])
def test_compute_stereo_matches_full(pkg, orc, shift, mb, mbf):
    """Frame::ComputeStereoMatches end to end (reference src/Frame.cc:841-1011): Hamming row-band stage + 11x11 SAD
    refinement on the device-resident pyramids + parabola fit + median outlier cut, bit-exact floats vs the oracle."""
    rng = np.random.default_rng(shift)
    left = pkg.synth_frame(200 + shift, 752, 480)
    right = np.roll(left, -shift, axis=1).copy()
    right = np.clip(right.astype(np.int16) + rng.integers(-2, 3, right.shape), 0, 255).astype(np.uint8)  # not a pure copy
    exL = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=752, max_height=480)
    exR = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=752, max_height=480)
    _, kl, dl = exL(left, None, (0, 0))
    _, kr, dr = exR(right, None, (0, 0))
    oL, oR = orc.Oracle(1200, 1.2, 8, 20, 7), orc.Oracle(1200, 1.2, 8, 20, 7)
    _, rkl, rdl = oL.extract(left, (0, 0))
    _, rkr, rdr = oR.extract(right, (0, 0))
    assert kl.tobytes() == rkl.tobytes() and kr.tobytes() == rkr.tobytes()
    m = pkg.ORBmatcher(2048, 2048)
    ur, dp = m.ComputeStereoMatches(exL, exR, kl, dl, kr, dr, mb, mbf)
    rur, rdp, nacc = orc.stereo_matches(oL, oR, rkl, rdl, rkr, rdr, mb, mbf)
    assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))
    ok = ur >= 0
    if mbf / mb > shift + 2:
        assert ok.sum() > 200
        assert np.median(kl["x"][ok] - ur[ok]) == pytest.approx(shift, abs=0.6)
    # no right keypoints at all -> nothing matched
    ur0, dp0 = m.ComputeStereoMatches(exL, exR, kl, dl, kr[:0], dr[:0], mb, mbf)
    assert np.all(ur0 == -1) and np.all(dp0 == -1)
    exL.close(); exR.close(); m.close()
