"""Where a single frame's k_sbp_resolve spends its time: s_memtime stamps of a -DSBP_TIMING build
(make -C wut-cuda-orb-slam3_amd/csrc OUT=../liborbx_t.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DSBP_TIMING").
Usage on the GPU box: ORBX_LIB=wut-cuda-orb-slam3_amd/liborbx_t.so python tools/experiments/sbp_timing.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
if os.environ.get("ORBX_LIB"):
    pkg.LIB_PATH = os.path.abspath(os.environ["ORBX_LIB"])
lib = pkg.load_library()
ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7)
img1 = pkg.synth_frame(17, 640, 480); img2 = np.roll(np.roll(img1, -4, 0), 6, 1)
_, k1, d1 = ex(img1, None, (0, 0)); k1, d1 = k1.copy(), d1.copy()
_, k2, d2 = ex(img2, None, (0, 0)); k2, d2 = k2.copy(), d2.copy()
sf = ex.GetScaleFactors().copy()
rng = np.random.default_rng(0)
pr = np.zeros(len(k1), pkg.PROJECTION_DTYPE)
pr["u"] = k1["x"] + 6 + rng.normal(0, 2, len(k1)); pr["v"] = k1["y"] - 4 + rng.normal(0, 2, len(k1)); pr["invz"] = 0.2
pr["angle"] = k1["angle"]; pr["octave"] = k1["octave"]; pr["flags"] = 3
state = np.full(len(k2), -1, np.int8)
m = pkg.ORBmatcher(8192, 8192)
for th in (15, 30):
    for _ in range(3):
        n, _a = m.SearchByProjection(k2, d2, state, (0, 640, 0, 480), sf, pr, d1, th)
    buf = np.zeros(8, dtype=np.uint64)
    assert lib.orbx_debug_sbp_timing(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.astype(np.int64)
    print("th=%d: %d matches; %d rounds; cycles: setup %d, rounds %d, histogram + assignment %d, total %d"
          % (th, n, t[4], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] - t[0]))
