"""LDS footprint / workgroups per CU of k_pyr_fast for the bench workloads (ORBX_DEBUG prints them at the first launch of a level)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ORBX_DEBUG"] = "1"
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
for (w, h, nf) in ((640, 480, 1000), (752, 480, 1200), (512, 512, 1500), (1920, 1080, 4000), (1242, 375, 2000)):
    print(f"== {w}x{h} nf={nf}, max_batch 32", flush=True)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=32)
    imgs = np.stack([pkg.synth_frame(i, w, h) for i in range(17)])
    ex.extract_batch(imgs, (0, 0))
    ex.close()
