"""Single-pair latency of orbx_search_for_initialization_frames (5000-feature frames, 752x480) for two window sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
w, h, nf = 752, 480, 5000
k4 = np.float32([458.654, 457.296, 367.215, 248.375]); dist = np.float32([])
P = 1
base = np.stack([pkg.synth_frame(500 + i, w, h) for i in range(P)])
nxt = np.stack([np.roll(np.roll(base[i], -5, axis=0), 8, axis=1) for i in range(P)])
ex1 = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
ex2 = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P)
ex1.extract_batch(base, (0, 0)); ex2.extract_batch(nxt, (0, 0))
m = pkg.ORBmatcher(8192, 8192)
bounds = (0.0, float(w), 0.0, float(h))
for win in (100, 30, 100):
    m.SearchForInitializationFrames(ex1, ex2, P, k4, dist, bounds, win, 0.9, True, download=False); m.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        m.SearchForInitializationFrames(ex1, ex2, P, k4, dist, bounds, win, 0.9, True, download=False)
    m.sync()
    print("window", win, "ms per call", round((time.perf_counter() - t0) / 20 * 1e3, 4))
