#!/usr/bin/env python3
"""Pins the CPU oracle's restatement of OpenCV's internals against a REAL OpenCV (cv2), stage by stage and byte for byte.

Why: the reference's image-dependent numerics live in OpenCV 4.x (cv::resize, cv::FAST, cv::GaussianBlur, cv::fastAtan2,
cv::BFMatcher, cv::undistortPoints); oracle/orb_oracle.cpp restates them from OpenCV 4.5.0's published source because no
OpenCV exists in the build image ("parity unpinned").  Wherever `import cv2` works, this script (and
tests/test_opencv_pin.py, which calls the same checks under pytest) turns that into a measured statement:

    python tests/tools/validate_against_opencv.py            # all checks, JSON report on stdout, exit code 1 on a mismatch
    python tests/tools/validate_against_opencv.py --exhaustive-trig   # + glibc cosf / sinf restatement on all floats in [0, 6.5]

It reports which `blur_variant` (orbx_params.blur_variant: 0 = OpenCV <= 4.5.0 taps, 1 = >= 4.5.1) the installed OpenCV
matches, so an integrator knows which value to pass.  A build of OpenCV with IPP (e.g. the PyPI wheels) may route 8-bit resize /
GaussianBlur through IPP with different last bits: the report says so instead of failing when
cv2.getBuildInformation() lists IPP and only those two stages differ.

Nothing here is imported by the product; the oracle is test infrastructure.  This file cannot run in the build image (no cv2).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

EDGE_THRESHOLD = 19  # reference src/ORBextractor.cc:101
SIZES = [(640, 480), (752, 480), (512, 512), (317, 243)]


def _frames(orc, n=3):
    out = []
    for i, (w, h) in enumerate(SIZES[:n] + [SIZES[-1]]):
        out.append(orc.synth_frame(1000 + i, w, h))
    rng = np.random.default_rng(5)
    out.append(rng.integers(0, 256, (200, 333), dtype=np.uint8))  # white noise: dense corners, saturated blur sums
    return out


def check_resize(cv2, orc):
    """cv::resize(src, dst, sz, 0, 0, INTER_LINEAR) on 8UC1 (reference src/ORBextractor.cc:1320) - SURVEY App. A.1."""
    bad = []
    for img in _frames(orc):
        h, w = img.shape
        for s in (1.2, 1.44, 2.0736, 1.3):
            dw, dh = int(round(w / s)), int(round(h / s))
            want = cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR)
            got = orc.resize_linear(img, dw, dh)
            if not np.array_equal(got, want):
                bad.append({"src": [w, h], "dst": [dw, dh], "differing_pixels": int((got != want).sum()),
                            "max_abs_diff": int(np.abs(got.astype(int) - want.astype(int)).max())})
    return {"ok": not bad, "mismatches": bad}


def check_blur(cv2, orc):
    """cv::GaussianBlur(img, img, Size(7, 7), 2, 2, BORDER_REFLECT_101) (reference src/ORBextractor.cc:1270-1273) - App. A.2.
    Reports which tap table (blur_variant) the installed OpenCV uses."""
    res = {}
    for variant in (0, 1):
        diff = 0
        for img in _frames(orc):
            want = cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
            diff += int((orc.gaussian_blur7(img, variant) != want).sum())
        res[f"variant_{variant}_differing_pixels"] = diff
    match = [v for v in (0, 1) if res[f"variant_{v}_differing_pixels"] == 0]
    res["matching_blur_variant"] = match[0] if match else None
    res["ok"] = bool(match)
    return res


def check_fast(cv2, orc):
    """cv::FAST(roi, kps, threshold, true) = FastFeatureDetector TYPE_9_16 with non-max suppression (reference
    src/ORBextractor.cc:908-925) - App. A.3: same corners, same order, same response (the score)."""
    bad = []
    for img in _frames(orc):
        for th in (20, 7, 1, 60):
            for (y0, y1, x0, x1) in ((16, 16 + 44, 16, 16 + 42), (0, img.shape[0], 0, img.shape[1]), (30, 37, 30, 100), (5, 60, 9, 16)):
                roi = np.ascontiguousarray(img[y0:y1, x0:x1])
                det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
                kps = det.detect(roi, None)
                want = [(int(round(k.pt[0])), int(round(k.pt[1])), int(round(k.response))) for k in kps]
                packed = orc.fast_roi(roi, th)
                got = [(int(p & 0xFFF), int((p >> 12) & 0xFFF), int(p >> 24)) for p in packed]
                if got != want:
                    bad.append({"roi": [y0, y1, x0, x1], "th": th, "n_cv": len(want), "n_oracle": len(got)})
    return {"ok": not bad, "mismatches": bad[:10], "n_mismatches": len(bad)}


def check_cell_candidates(cv2, orc):
    """tileCalcKeypoints (reference src/ORBextractor.cc:867-950) rebuilt from cv2 calls - pyramid by cv2.resize of the previous
    level's interior, per-cell cv2 FAST with the iniThFAST / minThFAST retry - against the oracle's per-level candidate lists."""
    bad = []
    nl, sf, ini, mn = 8, 1.2, 20, 7
    for img in _frames(orc, 2)[:2]:
        ref = orc.Oracle(1000, sf, nl, ini, mn)
        ref.extract(img, (0, 0))
        inv = ref.tables()["inv_scale"]
        level = img
        for l in range(nl):
            if l > 0:
                w, h = int(np.rint(np.float32(img.shape[1]) * inv[l])), int(np.rint(np.float32(img.shape[0]) * inv[l]))  # cvRound
                level = cv2.resize(level, (w, h), interpolation=cv2.INTER_LINEAR)
            if not np.array_equal(level, ref.level(l)):
                bad.append({"level": l, "stage": "pyramid"})
                break
            rows, cols = level.shape
            minBX, minBY = EDGE_THRESHOLD - 3, EDGE_THRESHOLD - 3
            maxBX, maxBY = cols - EDGE_THRESHOLD + 3, rows - EDGE_THRESHOLD + 3
            width, height = float(maxBX - minBX), float(maxBY - minBY)
            nCols, nRows = int(width / 35), int(height / 35)
            wCell, hCell = int(np.ceil(width / nCols)), int(np.ceil(height / nRows))
            want = []
            for i in range(nRows):
                iniY = minBY + i * hCell
                maxY = iniY + hCell + 6
                if iniY >= maxBY - 3:
                    continue
                maxY = min(maxY, maxBY)
                for j in range(nCols):
                    iniX = minBX + j * wCell
                    maxX = iniX + wCell + 6
                    if iniX >= maxBX - 6:
                        continue
                    maxX = min(maxX, maxBX)
                    roi = np.ascontiguousarray(level[iniY:maxY, iniX:maxX])
                    kps = cv2.FastFeatureDetector_create(ini, True, cv2.FAST_FEATURE_DETECTOR_TYPE_9_16).detect(roi, None)
                    if not kps:
                        kps = cv2.FastFeatureDetector_create(mn, True, cv2.FAST_FEATURE_DETECTOR_TYPE_9_16).detect(roi, None)
                    want += [(int(round(k.pt[0])) + j * wCell, int(round(k.pt[1])) + i * hCell, int(round(k.response))) for k in kps]
            got = [(int(p & 0xFFF), int((p >> 12) & 0xFFF), int(p >> 24)) for p in ref.candidates(l)]
            if got != want:
                bad.append({"level": l, "stage": "candidates", "n_cv": len(want), "n_oracle": len(got)})
            blur = cv2.GaussianBlur(level, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
            if not any(np.array_equal(blur, orc.gaussian_blur7(level, v)) for v in (0, 1)):  # (which variant: check_blur)
                bad.append({"level": l, "stage": "blurred level"})
    return {"ok": not bad, "mismatches": bad[:10], "n_mismatches": len(bad)}


def check_fast_atan2(cv2, orc):
    """cv::fastAtan2(y, x) in degrees (IC_Angle, reference src/ORBextractor.cc:85) - App. A.4."""
    rng = np.random.default_rng(3)
    ys = np.concatenate([rng.integers(-200000, 200000, 20000), [0, 0, 1, -1, 0, 5, -5]]).astype(np.float32)
    xs = np.concatenate([rng.integers(-200000, 200000, 20000), [0, 1, 0, 0, -1, 5, -5]]).astype(np.float32)
    bad = 0
    for y, x in zip(ys, xs):
        a = np.float32(cv2.fastAtan2(float(y), float(x)))
        b = np.float32(orc.fast_atan2(float(y), float(x)))
        bad += int(a.tobytes() != b.tobytes())
    return {"ok": bad == 0, "differing_values": bad, "tested": int(len(ys))}


def check_knn2(cv2, orc):
    """cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, matches, 2) (reference src/Frame.cc:45, 1174) - App. A.7, ties included."""
    rng = np.random.default_rng(11)
    bad = []
    for nq, nt in ((5, 1), (64, 65), (500, 777), (1200, 1200)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nt > 3:
            t[nt // 2] = t[0]; t[-1] = t[0]; q[0] = t[0]      # exact ties: lower train index first
            t[1:nt:7] = t[1]
        matches = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q, t, k=2)
        idx, dist = orc.knn2(q, t)
        for i, ms in enumerate(matches):
            want = [(m.trainIdx, int(m.distance)) for m in ms]
            got = [(int(idx[i, k]), int(dist[i, k])) for k in range(2) if idx[i, k] >= 0]
            if want != got:
                bad.append({"nq": nq, "nt": nt, "query": i, "cv": want, "oracle": got})
                break
    return {"ok": not bad, "mismatches": bad}


def check_undistort(cv2, orc):
    """cv::undistortPoints(mat, mat, K, distCoef, Mat(), K) (Frame::UndistortKeyPoints, reference src/Frame.cc:777-811)."""
    rng = np.random.default_rng(17)
    bad = []
    cams = [((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)),           # EuRoC
            ((517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)),                                # TUM1
            ((600.0, 600.0, 320.0, 240.0), (0.1, -0.2, 0.001, 0.002, 0.05, 0.01, -0.02, 0.003))]
    for k4, dist in cams:
        pts = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
        K = np.array([[k4[0], 0, k4[2]], [0, k4[1], k4[3]], [0, 0, 1]], np.float32)
        want = cv2.undistortPoints(pts.reshape(-1, 1, 2), K, np.array(dist, np.float32), None, None, K).reshape(-1, 2)
        kps = np.zeros(len(pts), orc.KEYPOINT_DTYPE)
        kps["x"], kps["y"] = pts[:, 0], pts[:, 1]
        got = orc.undistort_keypoints(kps, k4, dist)
        g = np.stack([got["x"], got["y"]], 1)
        if g.tobytes() != want.astype(np.float32).tobytes():
            bad.append({"camera": k4, "differing_points": int((g != want).any(axis=1).sum()), "max_abs_diff": float(np.abs(g - want).max())})
    return {"ok": not bad, "mismatches": bad}


def check_trig_exhaustive():
    """include/orbx_trig.h against this host's libm cosf / sinf on every float in [0, 6.5]."""
    import subprocess
    import tempfile
    d = tempfile.mkdtemp()
    src = os.path.join(d, "t.c")
    open(src, "w").write(r'''
#include <math.h>
#include <stdio.h>
#include "orbx_trig.h"
int main(void) { long bad = 0; for (uint32_t b = 0; b <= 0x40d00000u; ++b) { float y; memcpy(&y, &b, 4);
  if (orbx_glibc_sincosf(y, 1) != cosf(y) || orbx_glibc_sincosf(y, 0) != sinf(y)) ++bad; } printf("%ld\n", bad); return bad != 0; }
''')
    out = {}
    for name, flags in (("unfused", ["-ffp-contract=off"]), ("fused", ["-ffp-contract=fast", "-mfma"])):
        exe = os.path.join(d, "t_" + name)
        subprocess.run(["gcc", "-O2", "-std=c99", *flags, "-I", os.path.join(ROOT, "include"), "-o", exe, src, "-lm"], check=True)
        r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
        out[name + "_mismatches"] = int(r.stdout.strip())
    out["ok"] = all(v == 0 for k, v in out.items() if k.endswith("_mismatches"))
    return out


CHECKS = {"resize": check_resize, "gaussian_blur": check_blur, "fast": check_fast, "cell_candidates": check_cell_candidates,
          "fast_atan2": check_fast_atan2, "knn2": check_knn2, "undistort_points": check_undistort}


def run_all(cv2, orc):
    return {name: fn(cv2, orc) for name, fn in CHECKS.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exhaustive-trig", action="store_true")
    ap.add_argument("--json", action="store_true", help="(the report is always JSON on stdout; accepted so that the acceptance step of "
                                                        "INTEGRATION.md section 5 reads the same everywhere)")
    ap.add_argument("--out", default=None, help="also write the report to this file (attach it to the integration change)")
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print(json.dumps({"status": "skipped", "reason": "cv2 is not importable here; run this where an OpenCV 4.x Python binding exists"}))
        return 0
    orc = graft.load_oracle()
    rep = {"opencv_version": cv2.__version__, "ipp": "IPP" in cv2.getBuildInformation() and "Intel IPP:                   NO" not in cv2.getBuildInformation()}
    rep.update(run_all(cv2, orc))
    if args.exhaustive_trig:
        rep["glibc_sincosf"] = check_trig_exhaustive()
    failed = [k for k, v in rep.items() if isinstance(v, dict) and not v.get("ok", True)]
    rep["status"] = "pinned" if not failed else "mismatch"
    rep["failed"] = failed
    rep["use_blur_variant"] = rep["gaussian_blur"].get("matching_blur_variant")
    # what to pass to orbx_create: blur_variant from the Gaussian check; trig_mode cannot be read from cv2 (it is a property of how the
    # reference's own translation unit resolves cos/sin) - the report says how to find it
    rep["orbx_params"] = {"blur_variant": rep["use_blur_variant"],
                          "trig_mode": "nm -C ORBextractor.o | grep -w 'cosf\\|cos'  ->  cos: 0 (ORBX_TRIG_DOUBLE), cosf: 1 (ORBX_TRIG_FLOAT)"}
    text = json.dumps(rep, indent=1, default=str)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
