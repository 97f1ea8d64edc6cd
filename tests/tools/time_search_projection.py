"""Times orbx_search_by_projection (H2D + grid + candidate lists + fixed-point resolve + D2H) against the single-threaded oracle."""
import sys, time, importlib, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
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
    n, _ = m.SearchByProjection(k2, d2, state, (0, 640, 0, 480), sf, pr, d1, th)
    t = time.time()
    for _ in range(20): n, a = m.SearchByProjection(k2, d2, state, (0, 640, 0, 480), sf, pr, d1, th)
    tg = (time.time() - t) * 50
    t = time.time()
    for _ in range(20): rn, ra = orc.search_by_projection(k2, d2, state, (0, 640, 0, 480), sf, pr, d1, th)
    tc = (time.time() - t) * 50
    print("th=%d: GPU %.3f ms per call (incl. transfers), oracle (1 CPU thread) %.3f ms; %d projections, %d keypoints, %d matches, equal=%s"
          % (th, tg, tc, len(pr), len(k2), n, n == rn and np.array_equal(a, ra)))
