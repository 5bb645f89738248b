#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline`: the reference's own SSE code
(oracle/_ref, compiled where /root/reference lies) and the plain-C port that travels to the
GPU box (oracle/viso_oracle.c), timed on the SAME core of THIS container on cfg-1 (KITTI
1241x376, mono, frames 0 -> 1, flow matching): Matcher::computeFeatures
(src/matcher.cpp:585-672) on both frames + Matcher::matching flow (src/matcher.cpp:274-344).
Median of >= 20 repetitions after 3 warm-ups, perf_counter (steady clock), one thread.

  python tools/calibrate_cpu_baseline.py [reps]      -> profiles/r02_cpu_calibration.json

The reference binary stays in this container; bench.py prints the committed ratio as
`cpu_baseline.port_over_reference_sse` next to the port's rate measured on the GPU box."""
import json, os, platform, statistics, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
pkg = g.load_package(); ob = g.load_oracle()
if not ob.Reference.available():
    ob.build(ref=True)
ref = ob.Reference(); port = ob.Oracle(); p = ob.Params.default()
W, H = 1241, 376
dims = [W, H, pkg.synth.bytes_per_line(W)]
Ip, Ic = pkg.synth.frame(W, H, 0, 0), pkg.synth.frame(W, H, 5, 1)


def one(impl):
    t0 = time.perf_counter()
    fp = impl.compute_features(p, Ip, dims)[1]
    fc = impl.compute_features(p, Ic, dims)[1]
    t1 = time.perf_counter()
    pm = impl.matching_flow(p, dims, fp, fc) if impl is ref else impl.matching(p, dims, 0, fp, None, fc, None)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, fp, fc, pm


out = {}
res = {}
for name, impl in (("reference_sse", ref), ("port", port)):
    for _ in range(3):
        one(impl)
    det, mat = [], []
    for _ in range(reps):
        d, m, fp, fc, pm = one(impl)
        det.append(d); mat.append(m)
    res[name] = (fp, fc, pm)
    out[name] = {"detect_2_images_ms": 1e3 * statistics.median(det), "flow_matching_ms": 1e3 * statistics.median(mat),
                 "total_ms": 1e3 * statistics.median([a + b for a, b in zip(det, mat)]),
                 "min_total_ms": 1e3 * min(a + b for a, b in zip(det, mat))}
same = all(np.array_equal(a, b) for a, b in zip(res["reference_sse"][:2], res["port"][:2])) and \
    res["reference_sse"][2].tobytes() == res["port"][2].tobytes()
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
out.update({"workload": "cfg-1: KITTI 1241x376 mono, frames (0,0) -> (5,1), computeFeatures x2 + flow matching, defaults",
            "features": [int(len(res["port"][0])), int(len(res["port"][1]))], "matches": int(len(res["port"][2])),
            "results_identical": bool(same), "repetitions": reps, "warmups": 3, "threads": 1,
            "cpu": cpu[0] if cpu else platform.processor(), "compiler_flags": {"reference_sse": "g++ -std=gnu++11 -O2 -msse3 (oracle/Makefile _ref)",
                                                                               "port": "gcc -O3 -msse2 (oracle/Makefile)"},
            "port_over_reference_sse": out["reference_sse"]["total_ms"] / out["port"]["total_ms"],
            "note": "ratio of RATES (port pairs/s over reference pairs/s) = reference time / port time on the same core; "
                    "> 1 means the port is faster than the reference's SSE code"})
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_cpu_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
