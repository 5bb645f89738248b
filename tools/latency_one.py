#!/usr/bin/env python3
"""One camera stream, the literal drop-in scenario: pushBack + matchFeatures(2) + getMatches per
stereo pair at KITTI size (VisualOdometryStereo::process, src/viso_stereo.cpp:33-52), host images.
  python tools/latency_one.py [pinned]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
W, H = 1241, 376
bpl = pkg.synth.bytes_per_line(W)
dims = [W, H, bpl]
seq = pkg.synth.stereo_sequence(W, H, 8, 12)
pinned = len(sys.argv) > 1 and sys.argv[1] == "pinned"
if pinned:
    q = []
    for l, r in seq:
        a = pkg.pinned_empty(l.shape); a[...] = l; b = pkg.pinned_empty(r.shape); b[...] = r
        q.append((a, b))
    seq = q
m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
for l, r in seq[:4]:
    m.pushBack(l, r, dims, False); m.matchFeatures(2); m.getMatches()
reps = int(os.environ.get("REPS", "20"))
tp = tm = tg = 0.0
t00 = time.perf_counter(); n = 0
for rep in range(reps):
    for l, r in seq:
        t0 = time.perf_counter(); m.pushBack(l, r, dims, False)
        t1 = time.perf_counter(); m.matchFeatures(2)
        t2 = time.perf_counter(); pm = m.getMatches()
        t3 = time.perf_counter(); tp += t1 - t0; tm += t2 - t1; tg += t3 - t2; n += 1
dt = time.perf_counter() - t00
print(f"one stream, {'page-locked' if pinned else 'pageable'} host images: {1e3 * dt / n:.3f} ms per pushBack+matchFeatures+getMatches "
      f"(pushBack {1e6 * tp / n:.0f} us, matchFeatures {1e6 * tm / n:.0f} us, getMatches {1e6 * tg / n:.0f} us), {len(pm)} matches")
m.close()
