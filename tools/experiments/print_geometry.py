"""Tile geometry of k_pyr_fast per level (ORBX_DEBUG lines of the library) for a frame size.  Usage: print_geometry.py [w h nfeatures batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ORBX_DEBUG"] = "1"
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
w, h, nf, b = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (640, 480, 1000, 1024)))
ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=b)
imgs = np.stack([pkg.synth_frame(i, w, h) for i in range(b)])
ex.extract_batch(imgs, (0, 0))
