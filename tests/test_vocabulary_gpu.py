"""GPU parity of the DBoW2 vocabulary transform (Frame::ComputeBoW, reference src/Frame.cc:768-775 ->
Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1261) against the CPU oracle, which keeps the reference's std::map arithmetic:
word ids, node ids, the FeatureVector and the BowVector values (doubles) must be bit-identical."""
import numpy as np
import pytest

from voc_synth import features_near_words, make_vocabulary, write_text

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert set(a) == set(b)
    for key in a:
        assert a[key].dtype == b[key].dtype and a[key].tobytes() == b[key].tobytes(), key


@pytest.mark.parametrize("k,L,scoring,weighting,levelsup", [(10, 4, 0, 0, 2), (10, 3, 0, 0, 4), (8, 4, 1, 0, 2), (10, 3, 5, 0, 1),
                                                            (10, 3, 5, 1, 1), (6, 4, 2, 2, 3), (10, 3, 0, 3, 0), (20, 2, 4, 0, 1)])
def test_transform_matches_oracle(pkg, orc, k, L, scoring, weighting, levelsup):
    parent, leaf, desc, weight = make_vocabulary(k, L, seed=k * 10 + L)
    voc = pkg.ORBVocabulary(k, L, weighting, scoring, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    ref = orc.Vocabulary(k, L, scoring, weighting, parent, leaf, desc, weight)
    assert voc.size() == ref.size() == int(leaf.sum())
    for n, seed in ((1500, 1), (1, 2), (37, 3)):
        f = features_near_words(desc, leaf, n, seed)
        out, exp = voc.transform(f, levelsup), ref.transform(f, levelsup)
        _same(out, exp)
        if n > 100:
            assert len(out["bow_word"]) < n and len(out["fv_node"]) <= len(out["bow_word"]) + 1  # repeated words, shared ancestors
    if (k, L) == (10, 4):  # the largest sort: 16384 keys = 128 KB of LDS
        f = features_near_words(desc, leaf, 16384, 11)
        _same(voc.transform(f, levelsup), ref.transform(f, levelsup))
    out = voc.transform(np.zeros((0, 32), np.uint8), levelsup)
    assert len(out["bow_word"]) == 0 and len(out["fv_node"]) == 0 and out["fv_start"].tolist() == [0]
    voc.close()


def test_ragged_tree_ties_and_random_features(pkg, orc):
    """Nodes with 2..k children, leaves above depth L, sibling descriptors that are exact twins (distance ties -> the first child
    wins), uniformly random query descriptors, and every word stopped."""
    k, L = 9, 5
    parent, leaf, desc, weight = make_vocabulary(k, L, seed=77, ragged=True, p_stop=0.2, min_leaf_depth=3)
    voc = pkg.ORBVocabulary(k, L, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    ref = orc.Vocabulary(k, L, 0, 0, parent, leaf, desc, weight)
    rng = np.random.default_rng(9)
    f = np.concatenate([features_near_words(desc, leaf, 3000, 4), rng.integers(0, 256, (1000, 32), dtype=np.uint8)])
    _same(voc.transform(f, 2), ref.transform(f, 2))  # nid_level = 3 = the shallowest leaf depth
    assert (voc.transform(f, 2)["feat_word"] == -1).any()  # stopped words
    voc.close()
    w0 = np.zeros_like(weight)
    voc = pkg.ORBVocabulary(k, L, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=w0)
    out = voc.transform(f[:200], 2)
    assert len(out["bow_word"]) == 0 and len(out["fv_node"]) == 0 and (out["feat_word"] == -1).all()
    voc.close()


def test_orbvoc_scale_and_text_file(pkg, orc, tmp_path):
    """k = 10, L = 6 like ORBvoc.txt (1.1 M nodes, 10^6 words), levelsup = 4 as Frame::ComputeBoW asks; then a smaller vocabulary
    through the text format of loadFromTextFile."""
    rng = np.random.default_rng(3)
    k, L = 10, 6
    n_nodes = (k ** (L + 1) - 1) // (k - 1)
    # the same node order as the generator, built level-free: parent of node i in a complete k-ary tree laid out depth first is
    # expensive in Python, so this tree is breadth first (parent[i] = (i - 1) // k), which the loader accepts as well
    parent = ((np.arange(n_nodes, dtype=np.int64) - 1) // k).astype(np.int32); parent[0] = 0
    leaf = (np.arange(n_nodes) >= (k ** L - 1) // (k - 1)).astype(np.uint8)
    desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
    weight = np.where(leaf > 0, rng.uniform(0.5, 12.0, n_nodes), 0.0)
    voc = pkg.ORBVocabulary(k, L, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    ref = orc.Vocabulary(k, L, 0, 0, parent, leaf, desc, weight)
    assert voc.size() == 10 ** 6
    f = desc[rng.integers((k ** L - 1) // (k - 1), n_nodes, 2000)] ^ (rng.random((2000, 32)) < 0.02).astype(np.uint8)
    _same(voc.transform(f, 4), ref.transform(f, 4))
    voc.close()

    parent, leaf, desc, weight = make_vocabulary(5, 3, seed=5)
    path = tmp_path / "voc.txt"
    write_text(path, 5, 3, 0, 0, parent, leaf, desc, weight)
    voc = pkg.ORBVocabulary()
    assert voc.loadFromTextFile(str(path)) and voc.size() == int(leaf.sum())
    ref = orc.Vocabulary(5, 3, 0, 0, parent, leaf, desc, weight)
    f = features_near_words(desc, leaf, 800, 6)
    _same(voc.transform(f, 4), ref.transform(f, 4))  # levelsup > L: every feature hangs under the root
    assert not voc.loadFromTextFile(str(tmp_path / "missing.txt"))
    bad = tmp_path / "bad.txt"; bad.write_text("30 3 0 0\n")
    assert not voc.loadFromTextFile(str(bad))
    with pytest.raises(pkg.OrbxError, match="capacity"):
        pkg.ORBVocabulary(5, 3, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight).transform(np.zeros((20000, 32), np.uint8))
    with pytest.raises(pkg.OrbxError, match="earlier node"):
        pkg.ORBVocabulary(5, 3, 0, 0, parent=[0, 2, 1], is_leaf=[0, 1, 1], desc=np.zeros((3, 32), np.uint8), weight=[0, 1, 1])


def _two_views(pkg, seed, nfeat=1500):
    ex = pkg.ORBextractor(nfeat, 1.2, 8, 20, 7)
    img1 = pkg.synth_frame(seed, 640, 480)
    rng = np.random.default_rng(seed)
    img2 = np.clip(np.roll(np.roll(img1, 3, axis=0), -5, axis=1).astype(np.int16) + rng.integers(-3, 4, img1.shape, dtype=np.int16), 0, 255)
    _, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
    _, k2, d2 = ex(img2.astype(np.uint8), None, (0, 0)); k2, d2 = k2.copy(), d2.copy()
    ex.close()
    return k1, d1, k2, d2, rng


def _vocabulary_from_descriptors(d, k, L, seed):
    """A vocabulary whose words sit near real descriptors: the tree of make_vocabulary re-rooted on image descriptors (leaf
    descriptors = extractor descriptors with a little noise), so that two views of a scene fall into the same nodes."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = make_vocabulary(k, L, seed=seed, p_stop=0.02)
    # top-down: every node's descriptor = one sample descriptor, children perturb their parent's
    pick = d[rng.integers(0, len(d), len(parent))]
    depth = np.zeros(len(parent), np.int32)
    for i in range(1, len(parent)):
        depth[i] = depth[parent[i]] + 1
    for i in range(1, len(parent)):
        if depth[i] == 1:
            desc[i] = pick[i]
        else:
            flips = rng.integers(0, 256, 6 * (L - depth[i] + 1))
            desc[i] = desc[parent[i]]
            for b in flips:
                desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return parent, leaf, desc, weight


@pytest.mark.parametrize("k,L,levelsup,ratio,check", [(10, 4, 2, 0.7, True), (10, 4, 3, 0.9, True), (6, 3, 3, 0.7, False), (10, 3, 1, 0.75, True)])
def test_search_by_bow(pkg, orc, k, L, levelsup, ratio, check):
    """Tracking::TrackReferenceKeyFrame: ComputeBoW on both sides, then ORBmatcher(0.7, true).SearchByBoW(pKF, F, matches)
    (reference src/ORBmatcher1.cc:225-427): bit-exact nmatches and assignment against the oracle, FeatureVectors from the GPU
    transform."""
    k1, d1, k2, d2, rng = _two_views(pkg, 41)
    parent, leaf, desc, weight = _vocabulary_from_descriptors(d1, k, L, 13)
    voc = pkg.ORBVocabulary(k, L, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    fv1, fv2 = voc.transform(d1, levelsup), voc.transform(d2, levelsup)
    valid = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    m = pkg.ORBmatcher(8192, 8192)
    n, matches = m.SearchByBoW(fv1, k1, d1, valid, fv2, k2, d2, ratio, check)
    rn, rmatches = orc.search_by_bow(fv1, k1, d1, valid, fv2, k2, d2, ratio, check)
    assert n == rn and np.array_equal(matches, rmatches)
    assert n == int((matches >= 0).sum())
    if levelsup >= L:  # one node (the root) holds everything: the most contended case, > 64 frame features per node
        assert len(fv1["fv_node"]) == 1 and n > 100
    # no valid map points / empty sides
    n0, m0 = m.SearchByBoW(fv1, k1, d1, np.zeros(len(k1), np.uint8), fv2, k2, d2, ratio, check)
    assert n0 == 0 and (m0 == -1).all()
    empty = dict(fv_node=np.zeros(0, np.int32), fv_start=np.zeros(1, np.int32), fv_feature=np.zeros(0, np.int32))
    n0, m0 = m.SearchByBoW(empty, k1, d1, valid, fv2, k2, d2, ratio, check)
    assert n0 == 0 and (m0 == -1).all()
    if len(fv2["fv_node"]) > 1:
        bad = dict(fv2); bad["fv_node"] = fv2["fv_node"][::-1].copy()
        with pytest.raises(pkg.OrbxError, match="ascending"):
            m.SearchByBoW(fv1, k1, d1, valid, bad, k2, d2, ratio, check)
    voc.close(); m.close()


def test_search_by_bow_contention(pkg, orc):
    """Many keyframe features per node chase few frame features with near-identical descriptors: every accepted match removes a
    candidate for the following keyframe features of the node, ties in distance go to the first frame feature."""
    k1, d1, k2, d2, rng = _two_views(pkg, 43, 1200)
    n2 = len(k2)
    src = rng.integers(0, 60, n2)
    d2b = d2[src].copy()
    d2b[:, 0] ^= rng.integers(0, 4, n2, dtype=np.uint8)          # groups of near twins in the frame
    d1b = d2[rng.integers(0, 60, len(k1))].copy()
    d1b[:, 1] ^= rng.integers(0, 8, len(k1), dtype=np.uint8)
    parent, leaf, desc, weight = _vocabulary_from_descriptors(d2[:60], 5, 2, 3)
    voc = pkg.ORBVocabulary(5, 2, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    fv1, fv2 = voc.transform(d1b, 1), voc.transform(d2b, 1)
    valid = np.ones(len(k1), np.uint8)
    m = pkg.ORBmatcher(8192, 8192)
    for ratio in (0.7, 1.0, 2.0):
        n, matches = m.SearchByBoW(fv1, k1, d1b, valid, fv2, k2, d2b, ratio, True)
        rn, rmatches = orc.search_by_bow(fv1, k1, d1b, valid, fv2, k2, d2b, ratio, True)
        assert n == rn and np.array_equal(matches, rmatches)
    assert n > 20
    voc.close(); m.close()


@pytest.mark.parametrize("levelsup,ratio,check", [(2, 0.8, True), (4, 0.75, True), (1, 0.9, False)])
def test_search_by_bow_keyframes(pkg, orc, levelsup, ratio, check):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher2.cc:37-177): the keyframe - keyframe overload
    (second side masked by its own map points, strict distance test, result indexed by the first keyframe's features)."""
    k1, d1, k2, d2, rng = _two_views(pkg, 47)
    parent, leaf, desc, weight = _vocabulary_from_descriptors(d1, 10, 4, 17)
    voc = pkg.ORBVocabulary(10, 4, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    fv1, fv2 = voc.transform(d1, levelsup), voc.transform(d2, levelsup)
    v1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    v2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    m = pkg.ORBmatcher(8192, 8192)
    n, m12 = m.SearchByBoWKeyFrames(fv1, k1, d1, v1, fv2, k2, d2, v2, ratio, check)
    rn, rm12 = orc.search_by_bow_keyframes(fv1, k1, d1, v1, fv2, k2, d2, v2, ratio, check)
    assert n == rn and np.array_equal(m12, rm12)
    got = m12 >= 0
    assert n == got.sum() > 50 and (v1[got] == 1).all() and (v2[m12[got]] == 1).all() and len(set(m12[got].tolist())) == n
    n0, m0 = m.SearchByBoWKeyFrames(fv1, k1, d1, v1, fv2, k2, d2, np.zeros(len(k2), np.uint8), ratio, check)
    assert n0 == 0 and (m0 == -1).all()
    voc.close(); m.close()


def test_transform_frame_from_extractor_outputs(pkg):
    """orbx_vocabulary_transform_frame reads the descriptors where the extractor left them in HBM: same BowVector / FeatureVector as
    the host-descriptor call, for every frame of a small batch."""
    B = 3
    imgs = np.stack([pkg.synth_frame(300 + i, 640, 480) for i in range(B)])
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
    parent, leaf, vdesc, weight = _vocabulary_from_descriptors(desc[0][: n[0]], 10, 3, 5)
    voc = pkg.ORBVocabulary(10, 3, 0, 0, parent=parent, is_leaf=leaf, desc=vdesc, weight=weight)
    for f in range(B):
        a = voc.transform_frame(ex, f, 2)
        b = voc.transform(desc[f][: n[f]], 2)
        for key in a:
            assert a[key].tobytes() == b[key].tobytes(), (f, key)
        assert len(a["bow_word"]) > 50
    with pytest.raises(pkg.OrbxError, match="frame"):
        voc.transform_frame(ex, B, 2)
    voc.close(); ex.close()


def test_transform_batch_from_extractor_outputs(pkg):
    """A whole batch from pixels to BowVectors without leaving HBM in between: orbx_vocabulary_transform_batch == the per-frame
    calls, for every frame (ragged keypoint counts; an all-black frame with zero keypoints in the middle)."""
    B = 6
    imgs = np.stack([pkg.synth_frame(400 + i, 640, 480) for i in range(B)])
    imgs[3] = 0
    ex = pkg.ORBextractor(800, 1.2, 8, 20, 7, max_batch=B)
    n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
    assert n[3] == 0 and n[0] > 700
    parent, leaf, vdesc, weight = _vocabulary_from_descriptors(desc[0][: n[0]], 10, 3, 6)
    voc = pkg.ORBVocabulary(10, 3, 0, 0, parent=parent, is_leaf=leaf, desc=vdesc, weight=weight)
    out = voc.transform_batch(ex, B, 2)
    for f in range(B):
        ref = voc.transform(desc[f][: n[f]], 2)
        for key in out[f]:
            assert out[f][key].tobytes() == ref[key].tobytes(), (f, key)
    assert len(out[3]["bow_word"]) == 0 and len(out[0]["bow_word"]) > 50
    voc.close(); ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("levelsup", [2, 3])
def test_search_by_bow_over_candidate_keyframes(pkg, orc, levelsup):
    """orbx_search_by_bow_frames: the loops of Tracking::Relocalization (SearchByBoW(pKF, mCurrentFrame, ..) per candidate keyframe, reference
    src/Tracking5.cc:83) and of LoopClosing (SearchByBoW(mpCurrentKF, vpCovKFi[j], ..), src/LoopClosing1.cc:662) as ONE batch of pairs of frames
    of an extractor batch, with keypoints, descriptors and FeatureVectors where the extractor and the vocabulary left them on the device -
    per pair the same matches as the single-call oracle on host copies."""
    import torch
    from golden_scenarios import vocabulary
    P = 6
    shifts = [(0, 0), (9, -6), (-7, 4), (14, 3), (3, 11), (-12, -8)]
    base = pkg.synth_frame(78, 640, 480)
    rng = np.random.default_rng(5)
    imgs = np.stack([np.clip(np.roll(np.roll(base, dy, 0), dx, 1).astype(np.int16) + rng.integers(-3, 4, base.shape, dtype=np.int16), 0, 255).astype(np.uint8)
                     for dx, dy in shifts])
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=P)
    d_imgs = torch.from_numpy(imgs).to("cuda:0")
    ex.extract_batch_device(d_imgs.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); ex.sync()
    n, _, kk, dd = ex.download_outputs(P)
    parent, leaf, vdesc, vweight = vocabulary(dd[0, : n[0]])
    voc = pkg.ORBVocabulary(8, 3, 0, 0, parent=parent, is_leaf=leaf, desc=vdesc, weight=vweight)
    fvs = voc.transform_batch(ex, P, levelsup=levelsup)
    cap = ex.capacity
    valid = np.zeros((P, cap), np.uint8)
    for f in range(P):
        valid[f, : n[f]] = rng.random(n[f]) < 0.7
    m = pkg.ORBmatcher(8192, 8192)
    # Relocalization: candidate keyframes 1 .. P - 1 against the current frame 0 (+ a repeated candidate, + the frame against itself)
    kfs = [1, 2, 3, 4, 5, 3, 0]; fs = [0] * len(kfs)
    nm, mt = m.SearchByBoWFrames(voc, ex, kfs, fs, valid, 0.75, True)
    total = 0
    for k, (a, b) in enumerate(zip(kfs, fs)):
        rn, rm = orc.search_by_bow(fvs[a], kk[a, : n[a]], dd[a, : n[a]], valid[a, : n[a]], fvs[b], kk[b, : n[b]], dd[b, : n[b]], 0.75, True)
        assert nm[k] == rn and np.array_equal(mt[k, : n[b]], rm), (k, a, b)
        assert (mt[k, n[b]:] == -1).all()
        total += rn
    assert total > 200 and np.array_equal(mt[2], mt[5])
    # LoopClosing: the current keyframe 0 against covisible keyframes, both sides masked by their map points; no orientation check, ratio 0.8
    kf1 = [0, 0, 0, 2]; kf2 = [1, 4, 5, 3]
    nm, mt = m.SearchByBoWFrames(voc, ex, kf1, kf2, valid, 0.8, False, keyframe_pair=True)
    for k, (a, b) in enumerate(zip(kf1, kf2)):
        rn, rm = orc.search_by_bow_keyframes(fvs[a], kk[a, : n[a]], dd[a, : n[a]], valid[a, : n[a]], fvs[b], kk[b, : n[b]], dd[b, : n[b]],
                                             valid[b, : n[b]], 0.8, False)
        assert nm[k] == rn and np.array_equal(mt[k, : n[a]], rm), (k, a, b)
    assert nm.sum() > 100
    # nothing to do; a frame index beyond the batch; a transform of another extractor
    nm, mt = m.SearchByBoWFrames(voc, ex, [], [], valid)
    assert len(nm) == 0
    with pytest.raises(pkg.OrbxError, match="frames"):
        m.SearchByBoWFrames(voc, ex, [P], [0], valid)
    ex2 = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=P)
    ex2.extract_batch_device(d_imgs.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); ex2.sync()
    with pytest.raises(pkg.OrbxError, match="transform_batch"):
        m.SearchByBoWFrames(voc, ex2, [1], [0], valid)
    # the SAME extractor runs another batch of the same size after the transform: the cached FeatureVectors no longer belong to its keypoints
    # (round 4 accepted this silently and paired old feature indices with new descriptors) - refused until the batch is transformed again
    d_imgs2 = torch.from_numpy(np.ascontiguousarray(imgs[::-1])).to("cuda:0")
    ex.extract_batch_device(d_imgs2.data_ptr(), P, 640, 480, 640, 640 * 480, (0, 0)); ex.sync()
    with pytest.raises(pkg.OrbxError, match="another batch"):
        m.SearchByBoWFrames(voc, ex, [1], [0], valid)
    fvs2 = voc.transform_batch(ex, P, levelsup=levelsup)
    n2, _, kk2, dd2 = ex.download_outputs(P)
    nm, mt = m.SearchByBoWFrames(voc, ex, [1], [0], valid, 0.75, True)
    rn, rm = orc.search_by_bow(fvs2[1], kk2[1, : n2[1]], dd2[1, : n2[1]], valid[1, : n2[1]], fvs2[0], kk2[0, : n2[0]], dd2[0, : n2[0]], 0.75, True)
    assert nm[0] == rn and np.array_equal(mt[0, : n2[0]], rm)
    m.close(); voc.close(); ex.close(); ex2.close()
