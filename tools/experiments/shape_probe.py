import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
for (w,h,nf) in ((1242,375,2000),(1242,375,1000),(1024,375,2000),(752,480,2000)):
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    img = pkg.synth_frame(5, w, h)
    for _ in range(3): ex(img, None, (0,0))
    t0=time.perf_counter()
    for _ in range(20): r = ex(img, None, (0,0))
    dt=(time.perf_counter()-t0)/20*1e3
    d = torch.from_numpy(img[None]).cuda()
    ex.set_profiling(True)
    ex.extract_batch_device(d.data_ptr(), 1, w, h, w, w*h); ex.sync()
    print(w,h,nf, "ms/call %.3f"%dt, "n", len(r[1]), "stages", ex.stage_times())
    ex.close()
