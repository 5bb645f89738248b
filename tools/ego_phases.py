#!/usr/bin/env python3
"""Where ego_kernel's time goes: phase clocks of workgroup 0 on a -DVH_EGO_TIMING build
(make VARIANT=etime EXTRA=-DVH_EGO_TIMING).  usage: ego_phases.py [S]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VISO_HIP_LIB", os.path.join(ROOT, "hls-final-visual-odometry_amd", "libviso_hip_etime.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
from egomotion_scene import scene
pkg = entry.load_package()
lib = C.CDLL(os.environ["VISO_HIP_LIB"])
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ego = pkg.EgoParams.default(f=645.24, cu=635.96, cv=194.13, base=0.5707)
raw = np.random.default_rng(1).integers(0, 2 ** 31 - 1, (S, ego.ransac_iters, 3)).astype(np.int32)
names = ["3-d points", "Gauss-Newton on the sample (wave 0)", "inlier counts + arg max", "inlier list", "refit"]
out = (C.c_ulonglong * 8)()
for n in (400, 2000, 9000):
    lists = [scene(pkg.P_MATCH_DTYPE, n, 500 + (s % 8), outliers=0.2, noise=0.3)[0] for s in range(8)]
    lists = [lists[s % 8] for s in range(S)]
    pkg.estimate_motion_stereo(ego, lists, raw)
    lib.vh_debug_ego_timing(out, 1)
    reps = 3
    for _ in range(reps):
        r = pkg.estimate_motion_stereo(ego, lists, raw)
    lib.vh_debug_ego_timing(out, 1)
    t = [out[k] / 100.0 / reps for k in range(5)]
    print(f"{n} matches: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, t)) + f"  (sum {sum(t):.0f} us)", flush=True)
