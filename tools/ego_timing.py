#!/usr/bin/env python3
"""Context number for DESIGN.md: batched stereo egomotion on the device-resident quad match lists of
a 256-stream group (vh_group_estimate_motion), and on bucketed lists (vh_estimate_motion_stereo)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
W, H = 1241, 376
dims = [W, H, pkg.synth.bytes_per_line(W)]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seq = pkg.synth.stereo_sequence(W, H, 2, 12)
g = pkg.StreamGroup(S, pkg.Params.default(), max_features=32768, max_matches=32768)
for l, r in seq:
    g.pushBack(np.stack([l] * S), np.stack([r] * S), dims, False)
g.matchFeatures(pkg.METHOD_QUAD)
ego = pkg.EgoParams.default(f=645.24, cu=635.96, cv=194.13, base=0.5707)
raw = np.random.default_rng(1).integers(0, 2 ** 31 - 1, (S, 200, 3)).astype(np.int32)
g.estimateMotion(ego, raw)
t0 = time.perf_counter()
for _ in range(5):
    tr, ok, ninl = g.estimateMotion(ego, raw)
dt = (time.perf_counter() - t0) / 5
nm = g.getCounts()[1]
print(f"vh_group_estimate_motion: {S} streams x {int(nm.mean())} matches, 200 hypotheses each: {1e3 * dt:.2f} ms per call "
      f"({1e6 * dt / S:.1f} us per stream), ok {int(ok.sum())}/{S}, inliers {int(ninl.mean())}, tr[0] {tr[0]}")
pm = g.getMatches(0)
b = pm[np.random.default_rng(2).choice(len(pm), 400, replace=False)]  # a bucketed-size list (2 per 50x50 bucket ~ 400)
lists = [b] * S
pkg.estimate_motion_stereo(ego, lists, raw)
t0 = time.perf_counter()
for _ in range(5):
    pkg.estimate_motion_stereo(ego, lists, raw)
dt = (time.perf_counter() - t0) / 5
print(f"vh_estimate_motion_stereo: {S} lists x 400 matches (after bucketing), upload + kernel + download: {1e3 * dt:.2f} ms per call ({1e6 * dt / S:.1f} us per list)")
g.close()
