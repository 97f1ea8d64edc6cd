"""Phase timing inside a k_pyr_fast workgroup from a -DPF_TIMING build
(make -C wut-cuda-orb-slam3_amd/csrc OUT=../liborbx_t.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DPF_TIMING").
Wave 0 stamps s_memtime at the phase boundaries, so the differences include the barrier waits for the other seven waves.
Usage on the GPU box: ORBX_LIB=wut-cuda-orb-slam3_amd/liborbx_t.so python tools/pf_timing.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
if os.environ.get("ORBX_LIB"):
    pkg.LIB_PATH = os.path.abspath(os.environ["ORBX_LIB"])
lib = pkg.load_library()
B = 256
ex = pkg.ORBextractor(1000, 1.2, 2, 20, 7, max_width=640, max_height=480, max_batch=B)  # 2 levels: level 0 is the last one with a resize
frames = np.stack([pkg.synth_frame(i, 640, 480) for i in range(B)])
ex.extract_batch(frames)
ex.extract_batch(frames)
buf = np.zeros((8192, 16), dtype=np.uint64)
rc = lib.orbx_debug_pf_timing(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
names = [(0, 1, "tile load (global -> LDS, barrier)"), (1, 2, "LDS setup: zero, cell table, resize tables"), (2, 3, "resize -> next level"),
         (3, 4, "round 0 (A) pre-test"), (4, 5, "round 0 (B) ring test"), (5, 6, "round 0 (C) score"), (6, 7, "round 0 (D) NMS"),
         (7, 12, "round 1 (empty cells) + emission")]
ok = (t[:, 12] > t[:, 0]) & (t[:, 7] > t[:, 0])
t = t[ok]
print("workgroups sampled:", len(t), " cycles per workgroup: %.0f" % np.mean(t[:, 12] - t[:, 0]))
for a, b, nme in names:
    d = t[:, b] - t[:, a]
    print("%-45s mean %7.0f  median %7.0f  (%4.1f %%)" % (nme, d.mean(), np.median(d), 100.0 * d.mean() / np.mean(t[:, 12] - t[:, 0])))
r1 = t[:, 11] > t[:, 7]
print("workgroups that ran round 1: %.1f %%" % (100.0 * r1.mean()))
if r1.any():
    print("  round 1 phases A..D (those workgroups): %.0f cycles, emission %.0f" % (np.mean(t[r1, 11] - t[r1, 7]), np.mean(t[r1, 12] - t[r1, 11])))
    print("  emission without round 1: %.0f" % np.mean(t[~r1, 12] - t[~r1, 7]) if (~r1).any() else "")
