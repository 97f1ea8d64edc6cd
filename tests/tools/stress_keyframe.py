"""Randomised parity sweep of the keyframe-side searches against the CPU oracle: SearchForTriangulation, Fuse / window search, SearchBySim3's
window searches, SearchByProjection(KeyFrame, Sim3), SearchByProjection(Frame, KeyFrame), both SearchByBoW overloads and
ComputeDistinctiveDescriptors, with random sizes, vocabularies and parameters per case (the scene builders of the GPU tests).
Usage on the GPU box: python tests/tools/stress_keyframe.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg, orc = graft.load_package(), graft.load_oracle()
import test_fuse_sim3_gpu as tf  # noqa: E402
import test_triangulation_search_gpu as tt  # noqa: E402
from voc_synth import make_vocabulary  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
m = pkg.ORBmatcher(8192, 8192)


def check(tag, ok):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH", tag)


for c in range(cases):
    seed = int(rng.integers(0, 1 << 30))
    # two views + a random vocabulary: triangulation search and both BoW searches
    n = int(rng.integers(50, 3000))
    k1, d1, k2, d2, F, ep, _ = tt._pinhole_scene(pkg, n, seed=seed, outliers=float(rng.uniform(0, 0.4)))
    kk, L = int(rng.integers(3, 11)), int(rng.integers(1, 5))
    parent, leaf, desc, weight = make_vocabulary(kk, L, seed=seed % 1000)
    voc = pkg.ORBVocabulary(kk, L, 0, 0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)
    lv = int(rng.integers(0, L + 1))
    fv1, fv2 = voc.transform(d1, lv), voc.transform(d2, lv)
    mp1 = (rng.random(len(k1)) < rng.uniform(0, 0.6)).astype(np.uint8); mp2 = (rng.random(len(k2)) < rng.uniform(0, 0.6)).astype(np.uint8)
    a = pkg.keyframe_view(k1, d1, mp1, fv1, tt.SF, tt.SIGMA2); b = pkg.keyframe_view(k2, d2, mp2, fv2, tt.SF, tt.SIGMA2)
    g = pkg.triangulation_geometry(ep, F)
    coarse, chk = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    got = m.SearchForTriangulation(a, b, g, False, coarse, chk)
    ref = orc.search_for_triangulation(a, b, g, False, coarse, chk)
    check("SearchForTriangulation case %d" % c, got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]))
    ratio = float(rng.choice([0.6, 0.7, 0.9, 1.0])); v1 = 1 - mp1; v2 = 1 - mp2
    got = m.SearchByBoW(fv1, k1, d1, v1, fv2, k2, d2, ratio, chk)
    ref = orc.search_by_bow(fv1, k1, d1, v1, fv2, k2, d2, ratio, chk)
    check("SearchByBoW case %d" % c, got[0] == ref[0] and np.array_equal(got[1], ref[1]))
    got = m.SearchByBoWKeyFrames(fv1, k1, d1, v1, fv2, k2, d2, v2, ratio, chk)
    ref = orc.search_by_bow_keyframes(fv1, k1, d1, v1, fv2, k2, d2, v2, ratio, chk)
    check("SearchByBoW(KF, KF) case %d" % c, got[0] == ref[0] and np.array_equal(got[1], ref[1]))
    voc.close()
    # projected map points: window search / Fuse, the projection searches with their rounds
    nk, npt, dense = int(rng.integers(20, 2500)), int(rng.integers(1, 3000)), bool(rng.integers(0, 2))
    k, d, ur, pts, pd = tf._scene(pkg, orc, nk, npt, seed=seed + 1, dense=dense)
    chi2 = bool(rng.integers(0, 2))
    got = m.Fuse(k, d, tf.BOUNDS, pts, pd, tf.LEVELS, u_right=ur if chi2 else None, inv_level_sigma2=tf.INV_SIGMA2 if chi2 else None, chi2_test=chi2)
    ri, rd = orc.search_keyframe_windows(k, d, tf.BOUNDS, pts, pd, u_right=ur if chi2 else None, inv_level_sigma2=tf.INV_SIGMA2 if chi2 else None, chi2_test=chi2)
    fused = np.where((ri >= 0) & (rd <= 50), ri, -1)
    check("Fuse case %d" % c, np.array_equal(got[1], fused) and got[0] == int((fused >= 0).sum()))
    matched = (rng.random(nk) < rng.uniform(0, 0.5)).astype(np.uint8)
    rh = float(rng.choice([0.5, 0.8, 1.0, 1.5]))
    got = m.SearchByProjectionSim3(k, d, tf.BOUNDS, pts, pd, matched, rh, tf.LEVELS)
    ref = orc.search_by_projection_sim3(k, d, tf.BOUNDS, pts, pd, matched, rh)
    check("SearchByProjection(KF, Sim3) case %d" % c, got[0] == ref[0] and np.array_equal(got[1], ref[1]))
    k["angle"] = rng.uniform(0, 360, nk).astype(np.float32)
    pa = ((rng.normal(10, 4, npt) + np.where(rng.random(npt) < 0.2, rng.uniform(0, 360, npt), 0)) % 360).astype(np.float32)
    dist, chk2 = int(rng.choice([50, 64, 100])), bool(rng.integers(0, 2))
    got = m.SearchByProjectionKeyFrame(k, d, tf.BOUNDS, pts, pd, pa, matched, dist, chk2, tf.LEVELS)
    ref = orc.search_by_projection_keyframe(k, d, tf.BOUNDS, pts, pd, pa, matched, dist, chk2)
    check("SearchByProjection(Frame, KF) case %d" % c, got[0] == ref[0] and np.array_equal(got[1], ref[1]))
    # distinctive descriptors
    counts = rng.integers(0, 40, int(rng.integers(1, 3000)))
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    dd = rng.integers(0, 256, (int(start[-1]), 32), dtype=np.uint8)
    if len(dd) > 10:
        dd[rng.integers(0, len(dd), len(dd) // 5)] = dd[rng.integers(0, len(dd), len(dd) // 5)]
    check("ComputeDistinctiveDescriptors case %d" % c, np.array_equal(m.ComputeDistinctiveDescriptors(dd, start), orc.distinctive_descriptors(dd, start)))
    if c % 10 == 0:
        print("case %d ok so far (%d x %d features / %d nodes; %d keypoints x %d points)" % (c, len(k1), len(k2), len(fv1["fv_node"]), nk, npt), flush=True)
print("cases: %d, mismatches: %d" % (cases, bad))
sys.exit(1 if bad else 0)
