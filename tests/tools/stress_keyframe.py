"""Randomised parity sweep of the keyframe-side searches against the CPU oracle: SearchForTriangulation, Fuse / window search, SearchBySim3's
window searches, SearchByProjection(KeyFrame, Sim3), SearchByProjection(Frame, KeyFrame), both SearchByBoW overloads and
ComputeDistinctiveDescriptors, with random sizes, vocabularies and parameters per case (the scene builders of the GPU tests); every
fourth case also the device-resident batch forms (SearchByBoWFrames in both overloads, SearchForTriangulationFrames) over an extractor batch.
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
    # every fourth case: the device-resident batch forms over the frames of an extractor batch (orbx_search_by_bow_frames, both overloads;
    # orbx_search_for_triangulation_frames) against the single-call oracle on host copies
    if c % 4 == 0:
        import torch
        from golden_scenarios import vocabulary
        P = int(rng.integers(2, 7))
        w_, h_ = int(rng.choice([320, 480, 640])), int(rng.choice([240, 360, 480]))
        base = pkg.synth_frame(seed % 1000, w_, h_)
        imgs = np.stack([np.clip(np.roll(np.roll(base, int(rng.integers(-9, 10)), 0), int(rng.integers(-12, 13)), 1).astype(np.int16) +
                                 rng.integers(-3, 4, base.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(P)])
        exb = pkg.ORBextractor(int(rng.integers(300, 1500)), 1.2, int(rng.integers(3, 9)), 20, 7, max_width=w_, max_height=h_, max_batch=P)
        d_imgs = torch.from_numpy(imgs).to("cuda:0")
        exb.extract_batch_device(d_imgs.data_ptr(), P, w_, h_, w_, w_ * h_, (0, 0)); exb.sync()
        nb, _, kb, db = exb.download_outputs(P)
        pr, lf, vd, vw = vocabulary(db[0, : nb[0]])
        vocb = pkg.ORBVocabulary(8, 3, 0, 0, parent=pr, is_leaf=lf, desc=vd, weight=vw)
        lvb = int(rng.integers(1, 4))
        fvb = vocb.transform_batch(exb, P, levelsup=lvb)
        capb = exb.capacity
        val = np.zeros((P, capb), np.uint8)
        for f_ in range(P):
            val[f_, : nb[f_]] = rng.random(nb[f_]) < rng.uniform(0.2, 1.0)
        K_ = int(rng.integers(1, 9))
        fa = rng.integers(0, P, K_); fb = rng.integers(0, P, K_)
        rt, ck = float(rng.choice([0.6, 0.75, 0.9])), bool(rng.integers(0, 2))
        nm, mt = m.SearchByBoWFrames(vocb, exb, fa, fb, val, rt, ck)
        okb = True
        for k_, (a_, b_) in enumerate(zip(fa, fb)):
            rn, rm = orc.search_by_bow(fvb[a_], kb[a_, : nb[a_]], db[a_, : nb[a_]], val[a_, : nb[a_]], fvb[b_], kb[b_, : nb[b_]], db[b_, : nb[b_]], rt, ck)
            okb = okb and nm[k_] == rn and np.array_equal(mt[k_, : nb[b_]], rm) and (mt[k_, nb[b_]:] == -1).all()
        check("SearchByBoWFrames case %d" % c, okb)
        nm, mt = m.SearchByBoWFrames(vocb, exb, fa, fb, val, rt, ck, keyframe_pair=True)
        okb = True
        for k_, (a_, b_) in enumerate(zip(fa, fb)):
            rn, rm = orc.search_by_bow_keyframes(fvb[a_], kb[a_, : nb[a_]], db[a_, : nb[a_]], val[a_, : nb[a_]], fvb[b_], kb[b_, : nb[b_]],
                                                 db[b_, : nb[b_]], val[b_, : nb[b_]], rt, ck)
            okb = okb and nm[k_] == rn and np.array_equal(mt[k_, : nb[a_]], rm)
        check("SearchByBoWFrames(KF, KF) case %d" % c, okb)
        # triangulation: keyframe fa[0] against the others, pinhole geometry of a sideways translation
        k4b = np.float32([400.0, 400.0, w_ / 2, h_ / 2])
        Kb = np.array([[k4b[0], 0, k4b[2]], [0, k4b[1], k4b[3]], [0, 0, 1]], np.float64)
        geoms = []
        for _ in range(K_):
            th_ = rng.uniform(0, 2 * np.pi); t_ = np.array([np.cos(th_), np.sin(th_), 0.0])
            tx_ = np.array([[0, -t_[2], t_[1]], [t_[2], 0, -t_[0]], [-t_[1], t_[0], 0]])
            geoms.append(pkg.triangulation_geometry((1e6 * t_[0], 1e6 * t_[1]), (np.linalg.inv(Kb).T @ tx_ @ np.linalg.inv(Kb)).astype(np.float32)))
        sfb = exb.GetScaleFactors().copy(); s2b = exb.GetScaleSigmaSquares().copy()
        has = 1 - val
        ref_f = int(fa[0]); cs = bool(rng.integers(0, 2))
        nm, m12 = m.SearchForTriangulationFrames(vocb, exb, ref_f, fb, k4b, np.zeros(0, np.float32), geoms, has, bCoarse=cs, checkOrientation=ck)
        okb = True
        va_ = pkg.keyframe_view(kb[ref_f, : nb[ref_f]], db[ref_f, : nb[ref_f]], has[ref_f, : nb[ref_f]], fvb[ref_f], sfb, s2b)
        for k_, b_ in enumerate(fb):
            vb_ = pkg.keyframe_view(kb[b_, : nb[b_]], db[b_, : nb[b_]], has[b_, : nb[b_]], fvb[b_], sfb, s2b)
            rn, _, rm = orc.search_for_triangulation(va_, vb_, geoms[k_], False, cs, ck)
            okb = okb and nm[k_] == rn and np.array_equal(m12[k_, : nb[ref_f]], rm)
        check("SearchForTriangulationFrames case %d" % c, okb)
        vocb.close(); exb.close()
    if c % 10 == 0:
        print("case %d ok so far (%d x %d features / %d nodes; %d keypoints x %d points)" % (c, len(k1), len(k2), len(fv1["fv_node"]), nk, npt), flush=True)
print("cases: %d, mismatches: %d" % (cases, bad))
sys.exit(1 if bad else 0)
