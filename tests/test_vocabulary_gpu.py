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
