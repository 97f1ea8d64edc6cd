"""Time of one Frame::ComputeBoW-sized transform (k = 10, L = 6 vocabulary with 10^6 random words, levelsup = 4) on the GPU
(host descriptors in, BowVector / FeatureVector out) and in the CPU oracle.  Usage on the GPU box: python tests/tools/time_vocabulary.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
from oracle import oracle as orc
rng = np.random.default_rng(3)
k, L = 10, 6
n_nodes = (k ** (L + 1) - 1) // (k - 1)
parent = ((np.arange(n_nodes, dtype=np.int64) - 1) // k).astype(np.int32); parent[0] = 0
first_leaf = (k ** L - 1) // (k - 1)
leaf = (np.arange(n_nodes) >= first_leaf).astype(np.uint8)
desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
weight = np.where(leaf > 0, rng.uniform(0.5, 12.0, n_nodes), 0.0)
voc = pkg.ORBVocabulary(k, L, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
ref = orc.Vocabulary(k, L, 0, 0, parent, leaf, desc, weight)
for n in (1000, 2000, 8000):
    f = desc[rng.integers(first_leaf, n_nodes, n)] ^ (rng.random((n, 32)) < 0.02).astype(np.uint8)
    voc.transform(f, 4)
    t0 = time.perf_counter()
    for _ in range(50):
        out = voc.transform(f, 4)
    t1 = time.perf_counter()
    for _ in range(5):
        exp = ref.transform(f, 4)
    t2 = time.perf_counter()
    same = all(out[key].tobytes() == exp[key].tobytes() for key in exp)
    print("n = %5d: GPU %.3f ms per call (transfers included), CPU oracle %.3f ms, identical: %s" % (n, 1e3 * (t1 - t0) / 50, 1e3 * (t2 - t1) / 5, same))

# the benchmark batch: 1024 frames from pixels to BowVectors, descriptors never leave HBM between the two steps
B = 1024
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
frames = torch.from_numpy(np.stack([pkg.synth_frame(i, 640, 480) for i in range(B)])).cuda()
ex.extract_batch_device(frames.data_ptr(), B, 640, 480, 640, 640 * 480, (0, 1000)); ex.sync()
out = voc.transform_batch(ex, B, 4)
t0 = time.perf_counter()
for _ in range(5):
    out = voc.transform_batch(ex, B, 4)
t1 = time.perf_counter()
import ctypes
lib = pkg.load_library()
print("batch of %d frames (%d features): transform_batch %.2f ms per batch including the D2H of all results (%d words in frame 0)" %
      (B, sum(len(o["fv_feature"]) for o in out), 1e3 * (t1 - t0) / 5, len(out[0]["bow_word"])))
cap = ex.capacity
bw = np.zeros((B, cap), np.int32); bv = np.zeros((B, cap), np.float64); nb = np.zeros(B, np.int32)
vn = np.zeros((B, cap), np.int32); vs = np.zeros((B, cap + 1), np.int32); vf = np.zeros((B, cap), np.int32); nv = np.zeros(B, np.int32)
args = (voc._h, ex._h, B, 4, bw.ctypes.data, bv.ctypes.data, nb.ctypes.data, vn.ctypes.data, vs.ctypes.data, vf.ctypes.data, nv.ctypes.data)
lib.orbx_vocabulary_transform_batch(*args)
t0 = time.perf_counter()
for _ in range(10):
    lib.orbx_vocabulary_transform_batch(*args)
t1 = time.perf_counter()
print("  the C call alone (two launches + 25 MB of results to pageable host memory): %.2f ms per batch = %.0f frames/s" %
      (1e3 * (t1 - t0) / 10, B * 10 / (t1 - t0)))
