#!/usr/bin/env python3
"""Context numbers for DESIGN.md section 6: the batched monocular estimator (vh_estimate_motion_mono /
vh_group_estimate_motion_mono) at the sizes of the reference's mono loop -- bucketed flow matches
(bucketFeatures(2, 50, 50): ~400 per frame) and the unbucketed lists of a 256-stream group --
2000 hypotheses each (src/viso_mono.h:42)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as entry
from egomotion_scene import mono_scene
pkg = entry.load_package()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mono = pkg.MonoParams.default(f=645.24, cu=635.96, cv=194.13, height=1.65)
raw = np.random.default_rng(1).integers(0, 2 ** 31 - 1, (S, mono.ransac_iters, 8)).astype(np.int32)
for n in (400, 2000):
    lists = [mono_scene(pkg.P_MATCH_DTYPE, n, 500 + (s % 8), outliers=0.2, noise=0.3)[0] for s in range(8)]
    lists = [lists[s % 8] for s in range(S)]
    pkg.estimate_motion_mono(mono, lists, raw)
    t0 = time.perf_counter()
    for _ in range(3):
        tr, ok, inl = pkg.estimate_motion_mono(mono, lists, raw)
    dt = (time.perf_counter() - t0) / 3
    print(f"vh_estimate_motion_mono: {S} lists x {n} matches x {mono.ransac_iters} hypotheses, upload + 6 kernels + download: "
          f"{1e3 * dt:.2f} ms per call ({1e6 * dt / S:.1f} us per list), ok {int(ok.sum())}/{S}, inliers {int(np.mean([len(i) for i in inl]))}", flush=True)
W, H = 1241, 376
dims = [W, H, pkg.synth.bytes_per_line(W)]
seq = pkg.synth.stereo_sequence(W, H, 2, 12)
g = pkg.StreamGroup(S, pkg.Params.default(), max_features=16384, max_matches=16384)
for l, r in seq:
    g.pushBack(np.stack([l] * S), None, dims, False)
g.matchFeatures(pkg.METHOD_FLOW)
g.estimateMotionMono(mono, raw)
t0 = time.perf_counter()
for _ in range(3):
    tr, ok, ninl = g.estimateMotionMono(mono, raw)
dt = (time.perf_counter() - t0) / 3
nm = g.getCounts()[1]
print(f"vh_group_estimate_motion_mono: {S} streams x {int(nm.mean())} device-resident flow matches (unbucketed): {1e3 * dt:.2f} ms per call "
      f"({1e6 * dt / S:.1f} us per stream), ok {int(ok.sum())}/{S}, inliers {int(ninl.mean())}")
g.close()
