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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import golden_scenarios as gs  # noqa: E402

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
    # ORBmatcher::SearchForInitialization on a frame and a shifted, re-noised view of it (5 x nFeatures
    # extractor, window 100, ORBmatcher(0.9, true): src/Tracking3.cc:740-741); a second call on the updated vbPrevMatched
    img1 = orc.synth_frame(7, 640, 480)
    rng = np.random.default_rng(1)
    img2 = np.clip(np.roll(np.roll(img1, -6, axis=0), 9, axis=1).astype(np.int16) + rng.integers(-3, 4, img1.shape, dtype=np.int16), 0, 255)
    img2 = img2.astype(np.uint8)
    o = orc.Oracle(5000, 1.2, 8, 20, 7)
    _, k1, d1 = o.extract(img1, (0, 0))
    _, k2, d2 = o.extract(img2, (0, 0))
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    n, m12, pv = orc.search_for_initialization(k1, d1, k2, d2, prev, (0, 640, 0, 480), 100, 0.9, True)
    n2, m12b, _ = orc.search_for_initialization(k1, d1, k2, d2, pv, (0, 640, 0, 480), 100, 0.9, True)
    out["search_init/input_crc"] = np.array([zlib.crc32(img1.tobytes()), zlib.crc32(img2.tobytes())], np.uint32)
    out["search_init/nmatches"] = np.array([n, n2], np.int64)
    out["search_init/matches12"] = np.stack([m12, m12b])
    out["search_init/grid_crc"] = np.array([zlib.crc32(np.concatenate(orc.assign_grid(k2, (0, 640, 0, 480))).tobytes())], np.uint32)
    # SURVEY 8(f) ranks 2-4 on one tracking scenario (tests/golden_scenarios.py): both SearchByProjection variants,
    # UndistortKeyPoints / ComputeImageBounds, the vocabulary transform of both views and SearchByBoW between them
    img1, img2 = gs.frames(orc.synth_frame)
    o = orc.Oracle(1200, 1.2, 8, 20, 7)
    _, k1, d1 = o.extract(img1, (0, 0))
    _, k2, d2 = o.extract(img2, (0, 0))
    sf = o.tables()["scale"]
    pr, pm, state, ur, valid = gs.tracking_inputs(k1, k2, orc.PROJECTION_DTYPE)
    na, aa = orc.search_by_projection(k2, d2, state, gs.BOUNDS, sf, pr, d1, 15.0, None, 0.0, False, False, True)
    nb, ab = orc.search_by_projection(k2, d2, state, gs.BOUNDS, sf, pr, d1, 7.0, ur, 40.0, False, False, True)
    nc, ac = orc.search_by_projection_points(k2, d2, state, gs.BOUNDS, sf, pm, d1, 3.0, 0.8, ur)
    out["tracking/input_crc"] = np.array([zlib.crc32(img1.tobytes()), zlib.crc32(img2.tobytes()), zlib.crc32(k1.tobytes() + d1.tobytes()),
                                          zlib.crc32(k2.tobytes() + d2.tobytes())], np.uint32)
    out["tracking/sbp_nmatches"] = np.array([na, nb, nc], np.int64)
    out["tracking/sbp_assign"] = np.stack([aa, ab, ac])
    k4, dist = gs.CAMERA
    un = orc.undistort_keypoints(k2, k4, dist)
    out["tracking/undistort_xy"] = np.stack([un["x"], un["y"]], axis=1)
    out["tracking/bounds"] = np.array(orc.compute_image_bounds(752, 480, k4, dist), np.float32)
    parent, leaf, vdesc, weight = gs.vocabulary(d1)
    V = orc.Vocabulary(gs.VOC[0], gs.VOC[1], 0, 0, parent, leaf, vdesc, weight)
    t1, t2 = V.transform(d1, 2), V.transform(d2, 2)
    out["tracking/voc_crc"] = np.array([zlib.crc32(parent.tobytes() + leaf.tobytes() + vdesc.tobytes() + weight.tobytes())], np.uint32)
    for tag, t in (("bow1", t1), ("bow2", t2)):
        out["tracking/%s_word" % tag] = t["bow_word"]
        out["tracking/%s_value" % tag] = t["bow_value"]
        out["tracking/%s_fv_crc" % tag] = np.array([zlib.crc32(t["fv_node"].tobytes() + t["fv_start"].tobytes() + t["fv_feature"].tobytes()),
                                                    zlib.crc32(t["feat_word"].tobytes() + t["feat_node"].tobytes())], np.uint32)
    nm, mm = orc.search_by_bow(t1, k1, d1, valid, t2, k2, d2, 0.7, True)
    out["tracking/bow_nmatches"] = np.array([nm], np.int64)
    out["tracking/bow_matches"] = mm
    path = os.path.join(ROOT, "tests", "golden", "orb_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
