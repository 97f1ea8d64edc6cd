"""Latency of one STEREO frame the way Frame::Frame(imLeft, imRight, ..) drives the front end (reference src/Frame.cc:124-143): the two
ORBextractor::operator() calls in two threads, then Frame::ComputeStereoMatches.  Shape of the reference's own recorded run (ExecMean.txt:
KITTI 03, 1242 x 375, KITTI03.yaml: 2000 features, 8 levels, 1.2): "ORB extraction 38.5 +- 5.1 ms, stereo matching 7.7 +- 1.6 ms" per
stereo frame on unstated hardware - quoted beside this measurement, not as a baseline of the bench metric.
Usage on the GPU box: python tools/time_stereo_pair.py [w h nfeatures]"""
import importlib, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1242
h = int(sys.argv[2]) if len(sys.argv) > 2 else 375
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
left = pkg.synth_frame(5, w, h)
right = np.roll(left, -14, axis=1).copy()
exl = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
exr = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
m = pkg.ORBmatcher(8192, 8192)
mb, mbf = 0.54, 386.1448  # KITTI03.yaml: Camera.bf / fx


from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(2)  # two persistent workers, like the reference's two std::threads per frame without Python's thread start-up cost


def frame_threads():
    fl = pool.submit(exl, left, None, (0, 0))
    fr = pool.submit(exr, right, None, (0, 0))
    return [fl.result(), fr.result()]


def frame_serial():
    return [exl(left, None, (0, 0)), exr(right, None, (0, 0))]


for mode, fn in (("two threads (as the reference)", frame_threads), ("one thread", frame_serial)):
    for _ in range(5):
        (_, kl, dl), (_, kr, dr) = fn()
        m.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, mb, mbf)
    N = 100
    te = ts = 0.0
    for _ in range(N):
        t0 = time.perf_counter()
        (_, kl, dl), (_, kr, dr) = fn()
        t1 = time.perf_counter()
        ur, dp = m.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, mb, mbf)
        t2 = time.perf_counter()
        te += t1 - t0; ts += t2 - t1
    print("%dx%d, %d features, %s: ORB extraction L+R %.3f ms, stereo matching %.3f ms, total %.3f ms per stereo frame (%d / %d keypoints, %d matches)"
          % (w, h, nf, mode, 1e3 * te / N, 1e3 * ts / N, 1e3 * (te + ts) / N, len(kl), len(kr), int((dp > 0).sum())))
