#!/usr/bin/env python3
"""Debug: where the workgroups of emit_features (library built with EXTRA=-DVH_EMIT_TIMING) or of
detect_nms_fast (EXTRA=-DVH_EMIT_TIMING=2, TIMING_OF=detect) spend their time on the benchmark
workload; VISO_HIP_LIB selects the library.  256 streams, a few steps."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
lib = pkg._lib()
W, H, S = 1241, 376, int(os.environ.get("S", "256"))
dims = [W, H, pkg.synth.bytes_per_line(W)]
seq = pkg.synth.stereo_sequence(W, H, 3)
bpl = dims[2]
grp = pkg.StreamGroup(S, pkg.Params.default())
out = (C.c_ulonglong * 8)()
for t, (l, r) in enumerate(seq):
    L = np.ascontiguousarray(np.broadcast_to(l, (S,) + l.shape)); R = np.ascontiguousarray(np.broadcast_to(r, (S,) + r.shape))
    grp.synchronize(); lib.vh_debug_emit_timing(out, 1)
    grp.pushBack(L, R, dims)
    grp.synchronize(); lib.vh_debug_emit_timing(out, 1)
    v = [int(x) for x in out]
    n = max(v[5], 1); tot = sum(v[:5])
    names = (["image tile", "filters", "block extrema", "window checks", "records"] if os.environ.get("TIMING_OF") == "detect"
             else ["prefix", "A compaction", "A2 row ranks", "bin slots+staging", "B descriptors"])
    span = v[7] - v[6] if v[7] > v[6] else 0
    print(f"        first start to last end {span} ticks; workgroups in flight on average {tot / max(span, 1):.0f}")
    print(f"step {t}: {v[5]} workgroups, {tot / n:.0f} ticks per workgroup: " + ", ".join(f"{nm} {x / n:.0f} ({x / max(tot,1):.2f})" for nm, x in zip(names, v[:5])))
grp.close()
