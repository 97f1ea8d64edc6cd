"""Single-frame latency over image shapes / parameters: looks for cliffs (a 2-D host-to-device copy once cost 3 ms per 1242-wide frame)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
cases = [(640, 480, 1000, 1.2, 8), (641, 479, 1000, 1.2, 8), (1241, 376, 2000, 1.2, 8), (1242, 375, 2000, 1.2, 8), (752, 480, 1200, 1.2, 8),
         (512, 512, 1500, 1.2, 8), (1920, 1080, 4000, 1.2, 8), (640, 480, 5000, 1.2, 8), (640, 480, 500, 1.2, 8), (640, 480, 1000, 1.1, 12),
         (640, 480, 1000, 1.4, 4), (320, 240, 500, 1.2, 6), (1280, 720, 2000, 1.2, 8), (848, 480, 1000, 1.2, 8), (4000, 3000, 8000, 1.2, 8)]
for (w, h, nf, sf, nl) in cases:
    try:
        ex = pkg.ORBextractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
        img = pkg.synth_frame(5, w, h)
        for _ in range(3):
            r = ex(img, None, (0, 0))
        t0 = time.perf_counter()
        for _ in range(20):
            r = ex(img, None, (0, 0))
        dt = (time.perf_counter() - t0) / 20 * 1e3
        d = torch.from_numpy(img[None]).cuda()
        for _ in range(3):
            ex.extract_batch_device(d.data_ptr(), 1, w, h, w, w * h); ex.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ex.extract_batch_device(d.data_ptr(), 1, w, h, w, w * h); ex.sync()
        dd = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{w}x{h} nf={nf} scale={sf} levels={nl}: host call {dt:.3f} ms, device-resident {dd:.3f} ms, {len(r[1])} keypoints, {w * h / 1e6 / dt:.2f} Mpx/ms", flush=True)
        ex.close()
    except Exception as e:  # noqa: BLE001
        print(f"{w}x{h} nf={nf} scale={sf} levels={nl}: {e}", flush=True)
