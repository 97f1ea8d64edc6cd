"""Phase timing of k_orient_desc from a -DOD_TIMING build (tools: make OUT=../liborbx_t.so CXXFLAGS="... -DOD_TIMING").
Usage on the GPU box: ORBX_LIB=wut-cuda-orb-slam3_amd/liborbx_t.so python tools/od_timing.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
if os.environ.get("ORBX_LIB"):
    pkg.LIB_PATH = os.path.abspath(os.environ["ORBX_LIB"])
lib = pkg.load_library()
B = 256
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
frames = np.stack([pkg.synth_frame(i, 640, 480) for i in range(B)])
ex.extract_batch(frames)
ex.extract_batch(frames)
buf = np.zeros((8192, 8), dtype=np.uint64)
rc = lib.orbx_debug_od_timing(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
ok = t[:, 4] > t[:, 0]
t = t[ok]
names = ["prologue (key / table loads, weights)", "pass 1 moments", "pass 2 angle, cos/sin", "pass 3 descriptors"]
print("workgroups sampled:", len(t), " total cycles/wave: %.0f" % np.mean(t[:, 4] - t[:, 0]))
for i, nme in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("%-40s mean %8.0f  median %8.0f cycles" % (nme, d.mean(), np.median(d)))
