#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: tests/golden/egomotion.npz from the REFERENCE's own
VisualOdometryStereo::estimateMotion (oracle/_ref, src/viso_stereo.cpp:54-157) on the synthetic
scenes of tests/egomotion_scene.py.  Run in the build container (needs /root/reference):

    python oracle/gen_golden_ego.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob
from egomotion_scene import scene

ref = ob.Reference(); oracle = ob.Oracle()
out = {}
for name, (n, seed, outliers, noise, kw) in {
        "s400": (400, 1, 0.25, 0.0, {}), "s60_noisy": (60, 2, 0.4, 0.3, {}), "s1500": (1500, 3, 0.1, 0.5, {}),
        "s250_hard": (250, 6, 0.7, 0.2, {}), "s400_plain": (400, 7, 0.2, 0.2, {"reweighting": 0}),
        "s300_tight": (300, 8, 0.3, 0.4, {"ransac_iters": 50, "inlier_threshold": 1.0}), "s9": (9, 5, 0.5, 0.0, {})}.items():
    pm, _ = scene(ob.P_MATCH_DTYPE, n, seed, outliers=outliers, noise=noise)
    e = ob.EgoParams.default(f=645.24, cu=635.96, cv=194.13, base=0.5707, **kw)
    ok, tr, inl = ref.estimate_motion_stereo(e, pm)
    samples = oracle.draw_samples(n, e.ransac_iters)   # rand() after srand(0), as the reference drew them
    ok2, tr2, inl2 = oracle.estimate_motion_stereo(e, pm, samples)
    assert ok == ok2 and tr.tobytes() == tr2.tobytes() and np.array_equal(inl, inl2), name
    out[name + "__pm"] = pm.view(np.uint8).reshape(n, 48)
    out[name + "__samples"] = samples
    out[name + "__ego"] = np.array([e.ransac_iters, e.reweighting, e.inlier_threshold, e.f, e.cu, e.cv, e.base], np.float64)
    out[name + "__ok"] = np.array(int(ok)); out[name + "__tr"] = tr; out[name + "__inliers"] = inl
    print(name, ok, len(inl), tr)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "egomotion.npz"), **out)
