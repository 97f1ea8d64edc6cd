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


@pytest.mark.parametrize("nq,nt", [(5, 0), (5, 1), (3, 2), (64, 65), (500, 777), (4000, 4000), (257, 15), (1, 16), (300, 17), (70, 63),
                                   (1000, 129), (255, 64), (513, 1025)])
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


def test_knn2_split_merge_with_ties_across_split_boundaries(pkg, orc, monkeypatch):
    """The train rows are scanned in splits and merged by (distance, train index): duplicates of the best rows placed on both
    sides of every split boundary must still resolve to the lower index first, for any split count."""
    rng = np.random.default_rng(99)
    nq, nt = 300, 1000
    q, t = _desc(rng, nq), _desc(rng, nt)
    for b in range(0, nt, 37):          # near-copies of queries sprinkled over the train set ...
        t[b] = q[b % nq]
    for b in (62, 63, 64, 65, 124, 125, 126, 250, 499, 500, 501, 998, 999):
        t[b] = q[5]                     # ... and exact ties straddling the boundaries of 2 / 4 / 8 / 16-way splits
    t[0] = q[7]; t[999] = q[7]          # best two are the first and the last row
    ridx, rdist = orc.knn2(q, t)
    m = pkg.ORBmatcher(1024, 1024)
    for split in ("1", "2", "3", "7", "16", "1000"):
        monkeypatch.setenv("ORBX_KNN_SPLIT", split)
        idx, dist = m.knnMatch(q, t)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), split
    monkeypatch.delenv("ORBX_KNN_SPLIT")
    idx, dist = m.knnMatch(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    m.close()


def _stereo_batch(pkg, P, w=752, h=480, nf=1200, lap=(0, 0)):
    left = np.stack([pkg.synth_frame(300 + i, w, h) for i in range(P)])
    right = np.stack([np.roll(left[i], -(6 + 3 * i), axis=1) for i in range(P)])
    exl = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    exr = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
    outl = exl.extract_batch(left, lap)
    outr = exr.extract_batch(right, lap)
    return exl, exr, outl, outr


@pytest.mark.parametrize("lapping", [False, True])
def test_knn2_frames_batch_equals_single_pairs(pkg, orc, lapping):
    """orbx_match_knn2_frames: P stereo pairs in one launch on the extractors' device-resident descriptors ==
    knnMatch(desc.rowRange(mono, n), descR.rowRange(monoR, nR)) per pair (reference src/Frame.cc:1174)."""
    P = 5
    exl, exr, (nl, ml, kl, dl), (nr, mr, kr, dr) = _stereo_batch(pkg, P, lap=(200, 600) if lapping else (0, 0))
    m = pkg.ORBmatcher(4096, 4096)
    idx, dist = m.knnMatchFrames(exl, exr, P, lapping_only=lapping)
    for p in range(P):
        qb, tb = (int(ml[p]), int(mr[p])) if lapping else (0, 0)
        if lapping:
            assert 0 < qb < nl[p] and 0 < tb < nr[p]
        q, t = dl[p, qb:nl[p]], dr[p, tb:nr[p]]
        ridx, rdist = orc.knn2(q, t)
        assert np.array_equal(idx[p, : len(q)], ridx) and np.array_equal(dist[p, : len(q)], rdist), p
        sidx, sdist = m.knnMatch(q, t)  # the single-pair entry point gives the same bytes
        assert np.array_equal(sidx, ridx) and np.array_equal(sdist, rdist)
    m.close(); exl.close(); exr.close()


def test_compute_stereo_matches_frames_batch(pkg, orc):
    """orbx_compute_stereo_matches_frames: the batch form is bit-equal to P single calls and to the oracle restatement of
    Frame::ComputeStereoMatches (reference src/Frame.cc:841-1011)."""
    P = 4
    exl, exr, (nl, ml, kl, dl), (nr, mr, kr, dr) = _stereo_batch(pkg, P)
    m = pkg.ORBmatcher(4096, 4096)
    mb, mbf = 0.11, 47.9
    ur, dp = m.ComputeStereoMatchesFrames(exl, exr, P, mb, mbf)
    matched = 0
    for p in range(P):
        sur, sdp = m.ComputeStereoMatches(exl, exr, kl[p, : nl[p]], dl[p, : nl[p]], kr[p, : nr[p]], dr[p, : nr[p]], mb, mbf, frame_left=p, frame_right=p)
        assert ur[p, : nl[p]].tobytes() == sur.tobytes() and dp[p, : nl[p]].tobytes() == sdp.tobytes(), p
        matched += int((sur >= 0).sum())
    assert matched > 100 * P
    m.close(); exl.close(); exr.close()


@pytest.mark.parametrize("pipeline", ["0", "1"])
def test_knn2_other_pipelines_in_subprocess(pkg, pipeline):
    """k_knn2_scan_mfma issues the products of tile t + 1 before it handles the keys of tile t and interleaves the (unguarded) key inserts
    with them (round 5, ORBX_KNN_PIPELINE=2); 1 = pipelined with the insert guard, 0 = the tile-by-tile scan of rounds 2 - 4.  Same results:
    the kNN tests once more under each switch."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_KNN_PIPELINE") or os.environ.get("ORBX_KNN_SHARED_BOUND") or os.environ.get("ORBX_KNN_MFMA"):
        pytest.skip("already a forced run")
    env = dict(os.environ, ORBX_KNN_PIPELINE=pipeline)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "knn2 and not subprocess"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.parametrize("bound", ["0", "1", "2"])
def test_knn2_insert_bound_variants_in_subprocess(pkg, bound):
    """The insert guard of the guarded scans (ORBX_KNN_PIPELINE=1 here; the default scan has none in its whole blocks): ORBX_KNN_SHARED_BOUND=2
    the bound shared by a query's four lane groups, refreshed once per block of four tiles; 0 every lane group's own bound (rounds 2 - 4); 1
    the shared bound refreshed at every insert event (round 4's A/B).  All exact: the kNN tests once more under each switch."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_KNN_SHARED_BOUND") or os.environ.get("ORBX_KNN_MFMA"):
        pytest.skip("already a forced run")
    env = dict(os.environ, ORBX_KNN_SHARED_BOUND=bound, ORBX_KNN_PIPELINE="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "knn2 and not subprocess"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.parametrize("variant", ["0", "2"])
def test_knn2_other_scans_in_subprocess(pkg, variant):
    """The kNN-2 scan runs on the matrix cores (k_knn2_scan_mfma: +-1 bytes, q . t = 256 - 2 Hamming, 16 x 16 x 64 products); ORBX_KNN_MFMA
    (read once per process) = 0 takes the vector-ALU scan it replaced, = 2 the 32 x 32 x 32 form (k_knn2_scan_mfma32, measured slower).
    All must give the oracle's results bit for bit: run this file's kNN tests once more under the switch."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORBX_KNN_MFMA"):
        pytest.skip("already the forced run")
    env = dict(os.environ, ORBX_KNN_MFMA=variant)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "knn2 and not subprocess"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]


def test_knn2_large_sets_take_the_direct_transfers(pkg, orc):
    """A call's host arrays cross the link packed into one transfer - except pieces above 256 KB, which go straight to their buffers:
    12000 x 9000 descriptors (384 KB and 288 KB of input, 96 KB per result array) against the oracle, then a small call on the same
    matcher (its staging has been used by neither of the large inputs)."""
    rng = np.random.default_rng(991)
    q, t = _desc(rng, 12000), _desc(rng, 9000, dup=0.1)
    m = pkg.ORBmatcher(16384, 16384)
    idx, dist = m.knnMatch(q, t)
    ridx, rdist = orc.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    idx, dist = m.knnMatch(q[:300], t[:700])
    ridx, rdist = orc.knn2(q[:300], t[:700])
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    m.close()
