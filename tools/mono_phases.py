#!/usr/bin/env python3
"""Where mono_final's time goes: phase clocks of workgroup 0 on a -DVH_MONO_TIMING build
(make VARIANT=mtime EXTRA=-DVH_MONO_TIMING), same lists as tools/mono_timing.py.  usage: mono_phases.py [S]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VISO_HIP_LIB", os.path.join(ROOT, "hls-final-visual-odometry_amd", "libviso_hip_mtime.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
from egomotion_scene import mono_scene
pkg = entry.load_package()
lib = C.CDLL(os.environ["VISO_HIP_LIB"])
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mono = pkg.MonoParams.default(f=645.24, cu=635.96, cv=194.13, height=1.65)
raw = np.random.default_rng(1).integers(0, 2 ** 31 - 1, (S, mono.ransac_iters, 8)).astype(np.int32)
names = ["winner's F", "inlier list", "refit system", "SVD of it", "F, E, R|t (one lane)", "-", "points in front", "median", "ground-plane vote", "rest"]  # (mono_final_a: 0-4, mono_final_c: 6-8; the triangulation is a kernel of its own)
out = (C.c_ulonglong * 16)()
for n in (400, 2000, 9000):
    lists = [mono_scene(pkg.P_MATCH_DTYPE, n, 500 + (s % 8), outliers=0.2, noise=0.3)[0] for s in range(8)]
    lists = [lists[s % 8] for s in range(S)]
    pkg.estimate_motion_mono(mono, lists, raw)
    lib.vh_debug_mono_timing(out, 1)
    reps = 3
    for _ in range(reps):
        tr, ok, inl = pkg.estimate_motion_mono(mono, lists, raw)
    lib.vh_debug_mono_timing(out, 1)
    t = [out[k] / 100.0 / reps for k in range(10)]  # us (100 MHz)
    print(f"{n} matches, {len(inl[0])} inliers in list 0: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, t) if v > 0) + f"  (sum {sum(t):.0f} us); hypotheses queued for the signed recount: {out[15] / reps:.0f} of {S * mono.ransac_iters}", flush=True)
