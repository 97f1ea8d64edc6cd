"""Synthetic DBoW2 vocabularies for the tests (the reference's ORBvoc.txt is not in its tree): a k-ary tree of depth L in the node
order TemplatedVocabulary::HKmeansStep / loadFromTextFile produce (children of a node get consecutive ids, depth first), child
descriptors = the parent's with a few bits flipped (so descents have close calls and exact ties), idf-like leaf weights with
some stopped words, optionally ragged (fewer children, leaves above depth L)."""
import numpy as np


def make_vocabulary(k=10, L=4, seed=0, ragged=False, p_stop=0.05, min_leaf_depth=1):
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = [0], [0], [np.zeros(32, np.uint8)], [0.0]

    def grow(pid, depth):
        nch = k if not ragged else int(rng.integers(2, k + 1))
        kids = []
        for _ in range(nch):  # HKmeansStep creates all children of a node first (consecutive ids) ...
            nid = len(parent)
            d = desc[pid].copy()
            flips = rng.integers(0, 256, int(rng.integers(0, 40)))  # 0 flips: a twin of the parent -> ties between siblings
            for b in flips:
                d[b >> 3] ^= np.uint8(1 << (b & 7))
            is_leaf = depth == L or (ragged and depth >= min_leaf_depth and rng.random() < 0.1)
            parent.append(pid); leaf.append(1 if is_leaf else 0); desc.append(d)
            weight.append(0.0 if (not is_leaf or rng.random() < p_stop) else float(rng.uniform(0.1, 9.0)))
            kids.append((nid, is_leaf))
        for nid, is_leaf in kids:  # ... then recurses into each
            if not is_leaf:
                grow(nid, depth + 1)

    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    grow(0, 1)
    return (np.array(parent, np.int32), np.array(leaf, np.uint8), np.stack(desc), np.array(weight, np.float64))


def features_near_words(desc, leaf, n, seed=1, noise_bits=30):
    """n query descriptors: leaf descriptors with up to `noise_bits` flipped bits (so words repeat, as in a real image)."""
    rng = np.random.default_rng(seed)
    leaves = np.flatnonzero(leaf)
    pick = leaves[rng.integers(0, min(len(leaves), max(8, n // 3)), n)]  # draw from a subset: repeated words
    f = desc[pick].copy()
    for i in range(n):
        for b in rng.integers(0, 256, int(rng.integers(0, noise_bits + 1))):
            f[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return f


def write_text(path, k, L, scoring, weighting, parent, leaf, desc, weight):
    """TemplatedVocabulary::saveToTextFile's format (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1427-1452)."""
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (k, L, scoring, weighting))
        for i in range(1, len(parent)):
            f.write("%d %d %s %r\n" % (parent[i], leaf[i], " ".join(str(int(b)) for b in desc[i]), float(weight[i])))
