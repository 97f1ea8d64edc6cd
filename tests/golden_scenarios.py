"""Inputs of the golden vectors for the widened entry points: one tracking scenario — frame 7 of the
synthetic generator and a shifted, re-noised view of it — shared by tests/golden/make_golden.py (oracle -> fixture) and
tests/test_golden.py (oracle and HIP path against the fixture).  Everything is derived from fixed seeds."""
import numpy as np

from voc_synth import make_vocabulary

BOUNDS = (0.0, 640.0, 0.0, 480.0)
CAMERA = ((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))  # EuRoC cam0
VOC = (8, 3)  # k, L


def frames(synth_frame):
    img1 = synth_frame(7, 640, 480)
    rng = np.random.default_rng(1)
    img2 = np.clip(np.roll(np.roll(img1, -6, axis=0), 9, axis=1).astype(np.int16) + rng.integers(-3, 4, img1.shape, dtype=np.int16), 0, 255)
    return img1, img2.astype(np.uint8)


def tracking_inputs(k1, k2, projection_dtype):
    """Projections of the first view's keypoints into the second (image motion (+9, -6) plus a prediction error), map-point flags,
    the second frame's occupancy, right-image coordinates — for both SearchByProjection variants."""
    rng = np.random.default_rng(2)
    n1, n2 = len(k1), len(k2)
    pr = np.zeros(n1, projection_dtype)
    pr["u"] = k1["x"] + 9 + rng.normal(0, 2.0, n1).astype(np.float32)
    pr["v"] = k1["y"] - 6 + rng.normal(0, 2.0, n1).astype(np.float32)
    pr["invz"] = (1.0 / rng.uniform(0.5, 20.0, n1)).astype(np.float32)
    pr["angle"] = k1["angle"]
    pr["octave"] = k1["octave"]
    pr["flags"] = (rng.random(n1) < 0.8).astype(np.int32) | ((rng.random(n1) < 0.7).astype(np.int32) << 1)
    state = rng.choice(np.int8([-1, -1, -1, 0, 1]), n2)
    ur = np.where(rng.random(n2) < 0.6, k2["x"] - rng.uniform(0.5, 40, n2), -1).astype(np.float32)
    pm = pr.copy()  # the same points as Frame::isInFrustum would leave them
    pm["angle"] = rng.choice(np.float32([0.5, 0.99, 0.998, 0.9985, 1.0]), n1)
    pm["invz"] = pm["u"] - rng.uniform(0.5, 40, n1).astype(np.float32)
    pm["octave"] = np.clip(k1["octave"] + rng.integers(-1, 2, n1), 0, 7)
    valid = (rng.random(n1) < 0.85).astype(np.uint8)
    return pr, pm, state, ur, valid


def vocabulary(d1):
    """A k = 8, L = 3 vocabulary around the first view's descriptors (level-1 nodes = sample descriptors, children = perturbed
    parents), so that the two views share nodes."""
    k, L = VOC
    rng = np.random.default_rng(3)
    parent, leaf, desc, weight = make_vocabulary(k, L, seed=3, p_stop=0.03)
    depth = np.zeros(len(parent), np.int32)
    for i in range(1, len(parent)):
        depth[i] = depth[parent[i]] + 1
    pick = d1[rng.integers(0, len(d1), len(parent))]
    for i in range(1, len(parent)):
        if depth[i] == 1:
            desc[i] = pick[i]
        else:
            desc[i] = desc[parent[i]]
            for b in rng.integers(0, 256, 6 * (L - depth[i] + 1)):
                desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return parent, leaf, desc, weight
