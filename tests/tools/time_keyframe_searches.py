#!/usr/bin/env python3
"""GPU entry point vs single-thread CPU oracle for the keyframe searches of round 3 (host arrays in and out, one call each):
SearchForTriangulation, Fuse / SearchBySim3 window searches, SearchByProjection(KeyFrame, Sim3) rounds, ComputeDistinctiveDescriptors.
Usage on the GPU box: python tests/tools/time_keyframe_searches.py > gpurun_out/keyframe_searches.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg, orc = graft.load_package(), graft.load_oracle()
import test_fuse_sim3_gpu as tf  # noqa: E402
import test_triangulation_search_gpu as tt  # noqa: E402
from voc_synth import make_vocabulary  # noqa: E402


def bench(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


m = pkg.ORBmatcher(8192, 8192)
k1, d1, k2, d2, F, ep, rng = tt._pinhole_scene(pkg, 2000, seed=1)
parent, leaf, desc, weight = make_vocabulary(10, 4, seed=3)
voc = pkg.ORBVocabulary(10, 4, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
fv1, fv2 = voc.transform(d1, 2), voc.transform(d2, 2)
a = pkg.keyframe_view(k1, d1, np.zeros(len(k1), np.uint8), fv1, tt.SF, tt.SIGMA2)
b = pkg.keyframe_view(k2, d2, np.zeros(len(k2), np.uint8), fv2, tt.SF, tt.SIGMA2)
g = pkg.triangulation_geometry(ep, F)
print(f"SearchForTriangulation 2000 x 2250 features, {len(fv1['fv_node'])} nodes: GPU {bench(lambda: m.SearchForTriangulation(a, b, g)):.3f} ms, "
      f"oracle 1 thread {bench(lambda: orc.search_for_triangulation(a, b, g), 5):.3f} ms")
k, d, ur, pts, pd = tf._scene(pkg, orc, 2000, 1500, seed=5)
print(f"Fuse window search 1500 points x 2000 keypoints: GPU {bench(lambda: m.Fuse(k, d, tf.BOUNDS, pts, pd, 8, u_right=ur, inv_level_sigma2=tf.INV_SIGMA2)):.3f} ms, "
      f"oracle {bench(lambda: orc.search_keyframe_windows(k, d, tf.BOUNDS, pts, pd, u_right=ur, inv_level_sigma2=tf.INV_SIGMA2, chi2_test=True), 5):.3f} ms")
matched = np.zeros(len(k), np.uint8)
print(f"SearchByProjection(KeyFrame, Sim3) 1500 points x 2000 keypoints: GPU {bench(lambda: m.SearchByProjectionSim3(k, d, tf.BOUNDS, pts, pd, matched)):.3f} ms, "
      f"oracle {bench(lambda: orc.search_by_projection_sim3(k, d, tf.BOUNDS, pts, pd, matched), 5):.3f} ms")
counts = rng.integers(2, 15, 5000)
start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
dd = rng.integers(0, 256, (int(start[-1]), 32), dtype=np.uint8)
print(f"ComputeDistinctiveDescriptors 5000 map points, {int(start[-1])} observations: GPU {bench(lambda: m.ComputeDistinctiveDescriptors(dd, start)):.3f} ms, "
      f"oracle {bench(lambda: orc.distinctive_descriptors(dd, start), 3):.3f} ms")
valid = np.ones(len(k1), np.uint8)
for lv in (2, 3):  # levelsup 2: ~100 nodes (ORB-SLAM3 uses 4 of 6 levels: ~100 nodes for 1000-2000 features); 3: ~10 nodes
    f1, f2 = voc.transform(d1, lv), voc.transform(d2, lv)
    print(f"SearchByBoW(KeyFrame, Frame) {len(k1)} x {len(k2)} features, {len(f1['fv_node'])} nodes: GPU "
          f"{bench(lambda: m.SearchByBoW(f1, k1, d1, valid, f2, k2, d2, 0.7, True)):.3f} ms, "
          f"oracle {bench(lambda: orc.search_by_bow(f1, k1, d1, valid, f2, k2, d2, 0.7, True), 5):.3f} ms")

# the candidate loop of Tracking::Relocalization as ONE batch of device-resident pairs (orbx_search_by_bow_frames) against a loop of single calls
import torch  # noqa: E402
from golden_scenarios import vocabulary  # noqa: E402
P = 21
base = pkg.synth_frame(78, 752, 480)
imgs = np.stack([np.clip(np.roll(np.roll(base, int(rng.integers(-12, 13)), 0), int(rng.integers(-16, 17)), 1).astype(np.int16) +
                         rng.integers(-3, 4, base.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(P)])
ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=P)
d_imgs = torch.from_numpy(imgs).to("cuda:0")
ex.extract_batch_device(d_imgs.data_ptr(), P, 752, 480, 752, 752 * 480, (0, 0)); ex.sync()
n, _, kk, dd2 = ex.download_outputs(P)
pr, lf, vd, vw = vocabulary(dd2[0, : n[0]])
voc2 = pkg.ORBVocabulary(8, 3, 0, 0, parent=pr, is_leaf=lf, desc=vd, weight=vw)
fvs = voc2.transform_batch(ex, P, levelsup=2)
val = np.ones((P, ex.capacity), np.uint8)
kfs = list(range(1, P)); fs = [0] * (P - 1)
t_batch = bench(lambda: m.SearchByBoWFrames(voc2, ex, kfs, fs, val, 0.75, True))
t_loop = bench(lambda: [m.SearchByBoW(fvs[a_], kk[a_, : n[a_]], dd2[a_, : n[a_]], val[a_, : n[a_]], fvs[0], kk[0, : n[0]], dd2[0, : n[0]], 0.75, True)
                        for a_ in kfs], 5)
t_orc = bench(lambda: [orc.search_by_bow(fvs[a_], kk[a_, : n[a_]], dd2[a_, : n[a_]], val[a_, : n[a_]], fvs[0], kk[0, : n[0]], dd2[0, : n[0]], 0.75, True)
                       for a_ in kfs], 3)
print(f"SearchByBoW, {P - 1} candidate keyframes against one frame ({int(n.mean())} features, {len(fvs[0]['fv_node'])} nodes): one device-resident batch "
      f"{t_batch:.3f} ms (valid flags up, matches down), {P - 1} single calls with host arrays {t_loop:.3f} ms, oracle 1 thread {t_orc:.3f} ms")

# LocalMapping::CreateNewMapPoints: keyframe 0 against its 20 neighbours (orbx_search_for_triangulation_frames: two launches for all of them)
has = np.zeros((P, ex.capacity), np.uint8)
k4 = np.float32([458.654, 457.296, 367.215, 248.375])
Kc = np.array([[k4[0], 0, k4[2]], [0, k4[1], k4[3]], [0, 0, 1]], np.float64)
tdir = np.array([1.0, 0.3, 0.0]) / np.hypot(1.0, 0.3)
tx = np.array([[0, -tdir[2], tdir[1]], [tdir[2], 0, -tdir[0]], [-tdir[1], tdir[0], 0]])
F12 = (np.linalg.inv(Kc).T @ tx @ np.linalg.inv(Kc)).astype(np.float32)
geoms = [pkg.triangulation_geometry((1e6 * tdir[0], 1e6 * tdir[1]), F12)] * (P - 1)
sf2 = ex.GetScaleFactors().copy(); s22 = ex.GetScaleSigmaSquares().copy()
t_batch = bench(lambda: m.SearchForTriangulationFrames(voc2, ex, 0, kfs, k4, np.zeros(0, np.float32), geoms, has))
va = pkg.keyframe_view(kk[0, : n[0]], dd2[0, : n[0]], has[0, : n[0]], fvs[0], sf2, s22)
views = [pkg.keyframe_view(kk[a_, : n[a_]], dd2[a_, : n[a_]], has[a_, : n[a_]], fvs[a_], sf2, s22) for a_ in kfs]
t_loop = bench(lambda: [m.SearchForTriangulation(va, vb, geoms[0]) for vb in views], 5)
print(f"SearchForTriangulation, one keyframe against {P - 1} neighbours ({int(n.mean())} features): one device-resident batch {t_batch:.3f} ms "
      f"(flags up, matches down), {P - 1} single calls with host arrays {t_loop:.3f} ms")
