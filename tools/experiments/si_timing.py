"""Where k_search_init_lds (one wave per frame pair) spends its time: s_memtime stamps of a -DSI_TIMING build
(make -C wut-cuda-orb-slam3_amd/csrc OUT=../liborbx_t.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DSI_TIMING").
Usage on the GPU box: ORBX_LIB=wut-cuda-orb-slam3_amd/liborbx_t.so python tools/experiments/si_timing.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
lib = pkg.load_library()
ex = pkg.ORBextractor(5000, 1.2, 8, 20, 7)
img1 = pkg.synth_frame(7, 640, 480); img2 = np.roll(np.roll(img1, -6, 0), 9, 1)
_, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
_, k2, d2 = ex(img2, None, (0, 0)); k2, d2 = k2.copy(), d2.copy()
m = pkg.ORBmatcher(8192, 8192)
prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
for _ in range(3):
    n, m12, _p = m.SearchForInitialization(k1, d1, k2, d2, prev.copy(), (0, 640, 0, 480), 100, 0.9, True)
buf = np.zeros(16, dtype=np.uint64)
assert lib.orbx_debug_si_timing(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
names = ["init + octave-0 subsets", "cells of F2's subset", "stable placement", "descriptors -> LDS", "walk", "bins, stolen matches, histogram, write-out"]
print("%d matches of %d level-0 keypoints; walk blocks: %d eight at a time, %d step by step" % (n, int((k1["octave"] == 0).sum()), t[7], t[8]))
for i, nm in enumerate(names):
    print("%-45s %8d cycles" % (nm, t[i + 1] - t[i]))
print("%-45s %8d cycles" % ("total", t[6] - t[0]))
