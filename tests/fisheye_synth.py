"""Synthetic fisheye stereo rig for the KannalaBrandt8 triangulation tests: an independent double-precision numpy model of the
camera (project / pose), TUM-VI-like parameters."""
import numpy as np

CAM1 = np.array([190.978, 190.973, 254.932, 256.897, 0.00348, 0.00071, -0.00205, 0.00020], np.float32)
CAM2 = np.array([190.442, 190.434, 252.598, 254.917, 0.00340, 0.00176, -0.00266, 0.00032], np.float32)


def rig(baseline=0.101, yaw=0.02, pitch=-0.01):
    """(R12, t12) = Frame::mRlr / mtlr: x_left = R12 x_right + t12."""
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    t = np.array([baseline, 0.002, -0.001])
    return R.astype(np.float32), t.astype(np.float32)


def project(cam, X):
    """KB8 projection of points X[n, 3] (double precision)."""
    X = np.asarray(X, np.float64)
    r = np.hypot(X[:, 0], X[:, 1])
    theta = np.arctan2(r, X[:, 2])
    psi = np.arctan2(X[:, 1], X[:, 0])
    c = cam.astype(np.float64)
    rd = theta + c[4] * theta**3 + c[5] * theta**5 + c[6] * theta**7 + c[7] * theta**9
    return np.stack([c[0] * rd * np.cos(psi) + c[2], c[1] * rd * np.sin(psi) + c[3]], 1)


def scene(n, seed=0):
    """n points in front of the left camera, their pixel positions in both cameras."""
    rng = np.random.default_rng(seed)
    X1 = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(0.4, 6.0, n)], 1)
    R12, t12 = rig()
    X2 = (X1 - t12.astype(np.float64)) @ R12.astype(np.float64)   # x_right = R12^T (x_left - t12)
    return X1, project(CAM1, X1), project(CAM2, X2), R12, t12
