#!/usr/bin/env python3
"""Run ON THE GPU BOX: the mono flow loop of bench.py's flow_pinned alone (S streams, pushBack of one camera +
matchFeatures(0)), for timelines:   rocprofv3 --kernel-trace ... -- python3 tools/flow_loop.py [steps] [streams]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pkg = entry.load_package()
W, H = 1241, 376
bpl = pkg.synth.bytes_per_line(W)
T = 8
fr = np.zeros((T, S, H, bpl), np.uint8)
base = [pkg.synth.frame(W, H, (5 * k) % 20, k % 20, 8, 1, 1 + s) for s in range(8) for k in range(T)]
for t in range(T):
    for s in range(S):
        fr[t, s] = base[(s % 8) * T + (t + s // 8) % T]
frames = torch.from_numpy(fr).cuda()
g = pkg.StreamGroup(S, pkg.Params.default(), max_features=32768, max_matches=32768)
g.setStream(torch.cuda.current_stream().cuda_stream)
dims = [W, H, bpl]
for k in range(5):
    g.pushBackDevice(frames[k % T].data_ptr(), None, H * bpl, dims, False)
    g.matchFeatures(pkg.METHOD_FLOW)
g.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    g.pushBackDevice(frames[k % T].data_ptr(), None, H * bpl, dims, False)
    g.matchFeatures(pkg.METHOD_FLOW)
g.synchronize(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"flow loop: {S * steps / dt:.0f} frames/s, {1e3 * dt / steps:.3f} ms per step")
g.close()
