"""Generates tests/golden/orb_golden.npz from the CPU oracle (oracle/), on seeded synthetic frames.

The reference's own tests hold no numeric vectors for this path and the reference cannot be built here (needs
OpenCV >= 4.4), so these vectors pin the ORACLE (against drift) and give the GPU tests an oracle-free target.
Inputs are regenerated from the frame index by the deterministic generator (include/orbx_synth.h); only their
CRC32 is stored.  Run:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

CASES = [  # name, w, h, nfeatures, frame index, lapping area
    ("c1_640x480_f0", 640, 480, 1000, 0, (0, 1000)),
    ("c1_640x480_f1", 640, 480, 1000, 1, (0, 1000)),
    ("c1_640x480_f2_lap", 640, 480, 1000, 2, (200, 420)),
    ("c3_752x480_f0", 752, 480, 1200, 0, (0, 0)),
    ("c4_512x512_f0", 512, 512, 1500, 0, (0, 1000)),
]


def main():
    orc = graft.load_oracle()
    out = {}
    for name, w, h, nf, idx, lap in CASES:
        img = orc.synth_frame(idx, w, h)
        o = orc.Oracle(nf, 1.2, 8, 20, 7)
        mono, kps, desc = o.extract(img, lap)
        out[name + "/meta"] = np.array([w, h, nf, idx, lap[0], lap[1], mono, len(kps)], np.int64)
        out[name + "/input_crc"] = np.array([zlib.crc32(img.tobytes())], np.uint32)
        out[name + "/kps"] = kps.view(np.uint8).reshape(-1, 28)
        out[name + "/desc"] = desc
        crcs = []
        for l in range(8):
            crcs.append([zlib.crc32(o.level(l).tobytes()), zlib.crc32(o.level(l, True).tobytes()),
                         zlib.crc32(o.candidates(l).tobytes()), len(o.candidates(l)), len(o.level_keypoints(l))])
        out[name + "/levels"] = np.array(crcs, np.int64)
    path = os.path.join(ROOT, "tests", "golden", "orb_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
