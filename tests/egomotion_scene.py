"""Synthetic stereo-odometry scenes for the egomotion tests: 3-d points seen by a rectified
stereo rig before and after a rigid motion, projected to integer pixel positions (features sit
on pixels), with a share of gross outliers -- as p_match records (Matcher::p_match)."""
import numpy as np


def rot(rx, ry, rz):
    sx, cx, sy, cy, sz, cz = np.sin(rx), np.cos(rx), np.sin(ry), np.cos(ry), np.sin(rz), np.cos(rz)
    return np.array([[+cy * cz, -cy * sz, +sy],
                     [+sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy],
                     [-cx * sy * cz + sx * sz, +cx * sy * sz + sx * cz, +cx * cy]])


def scene(dtype, n, seed, tr=(0.004, -0.012, 0.002, 0.03, -0.01, -0.85), outliers=0.25, f=645.24, cu=635.96, cv=194.13,
          base=0.5707, W=1241, H=376, noise=0.0):
    """-> (p_match[n], true tr).  tr maps previous-frame coordinates to current-frame coordinates
    (p_t = R p_{t-1} + t, src/viso.h:80-86)."""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype)
    R, t = rot(*tr[:3]), np.array(tr[3:])
    k = 0
    while k < n:
        Z = rng.uniform(4, 60); X = rng.uniform(-1, 1) * Z * 0.9; Y = rng.uniform(-0.3, 0.25) * Z
        P = np.array([X, Y, Z]); Q = R @ P + t
        if Q[2] < 2:
            continue
        u1p, v1p, u2p = f * P[0] / P[2] + cu, f * P[1] / P[2] + cv, f * (P[0] - base) / P[2] + cu
        u1c, v1c, u2c = f * Q[0] / Q[2] + cu, f * Q[1] / Q[2] + cv, f * (Q[0] - base) / Q[2] + cu
        vals = np.array([u1p, v1p, u2p, v1p, u1c, v1c, u2c, v1c]) + (rng.normal(0, noise, 8) if noise else 0)
        vals = np.round(vals)
        if rng.random() < outliers:
            vals[4:] += rng.integers(-40, 41, 4)
        if not (np.all(vals[[0, 2, 4, 6]] >= 0) and np.all(vals[[0, 2, 4, 6]] < W) and np.all(vals[[1, 3, 5, 7]] >= 0) and np.all(vals[[1, 3, 5, 7]] < H)):
            continue
        if vals[0] < vals[2] or vals[4] < vals[6]:
            continue
        r = out[k]
        r["u1p"], r["v1p"], r["u2p"], r["v2p"], r["u1c"], r["v1c"], r["u2c"], r["v2c"] = vals
        r["i1p"] = r["i2p"] = r["i1c"] = r["i2c"] = k
        k += 1
    return out, np.array(tr)


def mono_scene(dtype, n, seed, tr=(0.002, -0.01, 0.001, 0.02, -0.005, -0.9), outliers=0.2, ground=0.45, height=1.65, f=645.24, cu=635.96,
               cv=194.13, W=1241, H=376, noise=0.0):
    """Flow matches of ONE camera (right-camera fields = -1, as Matcher::matching method 0 emits them):
    a share `ground` of the points lies on the road plane Y = height below the camera (the mono
    estimator scales its translation by that plane), the rest is structure above it."""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype)
    for name in out.dtype.names:
        out[name] = -1
    R, t = rot(*tr[:3]), np.array(tr[3:])
    k = 0
    while k < n:
        Z = rng.uniform(4, 50)
        if rng.random() < ground:
            X, Y = rng.uniform(-0.8, 0.8) * Z * 0.6, height
        else:
            X, Y = rng.uniform(-1, 1) * Z * 0.9, rng.uniform(-0.28, 0.02) * Z
        P = np.array([X, Y, Z]); Q = R @ P + t
        if Q[2] < 2:
            continue
        vals = np.array([f * P[0] / P[2] + cu, f * P[1] / P[2] + cv, f * Q[0] / Q[2] + cu, f * Q[1] / Q[2] + cv])
        vals = np.round(vals + (rng.normal(0, noise, 4) if noise else 0))
        if rng.random() < outliers:
            vals[2:] += rng.integers(-40, 41, 2)
        if not (0 <= vals[0] < W and 0 <= vals[2] < W and 0 <= vals[1] < H and 0 <= vals[3] < H):
            continue
        r = out[k]
        r["u1p"], r["v1p"], r["u1c"], r["v1c"] = vals
        r["i1p"] = r["i1c"] = k
        k += 1
    return out, np.array(tr)
