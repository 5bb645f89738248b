#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: tests/golden/mono.npz from the REFERENCE's own
VisualOdometryMono::estimateMotion and Matrix::svd (oracle/_ref, src/viso_mono.cpp:41-160,
src/matrix.cpp:579-802) on the synthetic scenes of tests/egomotion_scene.py.  Run in the build
container (needs /root/reference):

    python oracle/gen_golden_mono.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob
from egomotion_scene import mono_scene

ref = ob.Reference(); oracle = ob.Oracle()
out = {}
CASES = {
    "m400": (400, 1, 0.2, 0.0, {}), "m200_noisy": (200, 2, 0.3, 0.3, {}), "m1500": (1500, 3, 0.1, 0.5, {}),
    "m300_hard": (300, 6, 0.5, 0.2, {}), "m400_pitch": (400, 7, 0.2, 0.2, {"pitch": -0.03}),
    "m250_few_iters": (250, 8, 0.3, 0.4, {"ransac_iters": 300, "inlier_threshold": 0.00002}), "m12": (12, 4, 0.0, 0.0, {}),
    "m9": (9, 5, 0.0, 0.0, {})}
for name, (n, seed, outliers, noise, kw) in CASES.items():
    pm, _ = mono_scene(ob.P_MATCH_DTYPE, n, seed, outliers=outliers, noise=noise)
    e = ob.MonoParams.default(f=645.24, cu=635.96, cv=194.13, height=1.65, **kw)
    ok, tr, inl = ref.estimate_motion_mono(e, pm)
    samples = oracle.draw_samples_n(n, 8, e.ransac_iters) if n >= 10 else np.zeros((e.ransac_iters, 8), np.int32)
    ok2, tr2, inl2 = oracle.estimate_motion_mono(e, pm, samples)
    assert ok == ok2 and tr.tobytes() == tr2.tobytes() and np.array_equal(inl, inl2), name
    out[name + "__pm"] = pm.view(np.uint8).reshape(n, 48)
    out[name + "__samples"] = samples.astype(np.int16) if n < 32768 else samples
    out[name + "__mono"] = np.array([e.ransac_iters, e.inlier_threshold, e.motion_threshold, e.height, e.pitch, e.f, e.cu, e.cv], np.float64)
    out[name + "__ok"] = np.array(int(ok)); out[name + "__tr"] = tr; out[name + "__inliers"] = inl
    print(name, ok, len(inl), tr)
# Matrix::svd of the shapes the estimator uses (8x9 sample system, 3x3, 4x4 triangulation, tall refit systems)
rng = np.random.default_rng(11)
for k, (m, n) in enumerate([(3, 3), (3, 3), (4, 4), (8, 9), (8, 9), (40, 9), (4, 4)]):
    a = rng.normal(size=(m, n))
    if k == 1:
        a[2] = a[0] - 2 * a[1]   # rank 2
    if k == 6:
        a = np.round(a * 3)      # small integers: exact zeros and ties on the way
    U, W, V = ref.svd(a)
    U2, W2, V2 = oracle.svd(a)
    assert U.tobytes() == U2.tobytes() and W.tobytes() == W2.tobytes() and V.tobytes() == V2.tobytes(), (m, n)
    out[f"svd{k}__a"] = a; out[f"svd{k}__U"] = U; out[f"svd{k}__W"] = W; out[f"svd{k}__V"] = V
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mono.npz"), **out)
print("wrote tests/golden/mono.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "mono.npz")), "bytes")
