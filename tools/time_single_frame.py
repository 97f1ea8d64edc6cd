"""Latency of ONE ORBextractor::operator() call (640x480, 1000 features, 8 levels) the way a SLAM front end drives it: host image in,
keypoints + descriptors out (src/Frame.cc:313 ExtractORB); and of the device-resident call (image already in HBM, results left in
HBM, one sync).  Usage on the GPU box: python tools/time_single_frame.py [batch]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
print("imported", flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
frames = np.stack([pkg.synth_frame(i, 640, 480) for i in range(B)])
for _ in range(5):
    ex.extract_batch(frames) if B > 1 else ex(frames[0], None, (0, 0))
print("warm", flush=True)
N = 200 if B <= 64 else 50
t0 = time.perf_counter()
for _ in range(N):
    ex.extract_batch(frames) if B > 1 else ex(frames[0], None, (0, 0))
t1 = time.perf_counter()
print("host-pointer call, batch %d: %.3f ms per call (%.3f ms per frame)" % (B, 1e3 * (t1 - t0) / N, 1e3 * (t1 - t0) / N / B))
if B == 1:  # the caller's frame buffer page-locked in place (orbx_host_register): what a camera ring would do once
    with pkg.host_registered(frames) as reg:
        for _ in range(5):
            ex(reg[0], None, (0, 0))
        t0 = time.perf_counter()
        for _ in range(N):
            ex(reg[0], None, (0, 0))
        t1 = time.perf_counter()
    print("host-pointer call from a registered (page-locked) frame buffer: %.3f ms per call" % (1e3 * (t1 - t0) / N))
d = torch.from_numpy(frames).cuda()
for _ in range(5):
    ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480); ex.sync()
t0 = time.perf_counter()
for _ in range(N):
    ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480); ex.sync()
t1 = time.perf_counter()
print("device-resident call + sync, batch %d: %.3f ms per call" % (B, 1e3 * (t1 - t0) / N))
ex.set_profiling(True)
ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480); ex.sync()
print("stage times (ms):", ex.stage_times())
