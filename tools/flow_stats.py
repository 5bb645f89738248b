#!/usr/bin/env python3
"""Debug: what the flow search executes on the benchmark workload (needs a library built
with EXTRA=-DVH_FLOW_STATS; VISO_HIP_LIB selects it).  One stream, a few quad steps."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
lib = pkg._lib()
W, H = 1241, 376
dims = [W, H, pkg.synth.bytes_per_line(W)]
seq = pkg.synth.stereo_sequence(W, H, 3)
m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
out = (C.c_ulonglong * 12)()
for t, (l, r) in enumerate(seq):
    m.pushBack(l, r, dims, False)
    if t:
        m.synchronize(); lib.vh_debug_flow_stats(out, 1)
        m.matchFeatures(pkg.METHOD_QUAD); m.synchronize()
        lib.vh_debug_flow_stats(out, 1)
        tiles, chunks, t0, t1, t2, nq, ncol, nredo = [int(x) for x in out[:8]]
        P = int(os.environ.get("P", "4")); Q = int(os.environ.get("Q", "2"))
        T = 64 * Q // P
        trips = t0 + t1 + t2
        print(f"step {t}: tiles {tiles} queries {nq} (fill {nq / max(tiles,1) / T:.2f}) columns/tile {ncol / max(tiles,1):.1f} chunks/tile {chunks / max(tiles,1):.1f} "
              f"candidates/tile {trips * 2 * P / max(tiles,1):.0f}  trips none/v/full {t0} {t1} {t2} ({t0 / trips:.2f} {t1 / trips:.2f} {t2 / trips:.2f})  "
              f"evaluated lane-pairs {trips * 2 * P * T:.3e}  re-searched queries {nredo} ({nredo / max(nq, 1):.4f})")
        st, strips, sredo, sq = [int(x) for x in out[8:12]]
        print(f"        stereo: tiles {st} queries {sq} candidates/tile {strips * 2 * P / max(st, 1):.0f} re-searched {sredo} ({sredo / max(sq, 1):.4f})")
m.close()
