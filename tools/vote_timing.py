#!/usr/bin/env python3
"""Run ON THE GPU BOX: the device form of removeOutliers + bucketFeatures (vh_remove_outliers_device) on KITTI-sized
flow-match lists -- parity against the oracle for every distinct list, then the sweep kernel's time for P lists at
several lanes-per-wave settings.

  python tools/vote_timing.py [--lists 256] [--lanes 1,2,4,8,16,64] [--distinct 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lists", type=int, default=256)
    ap.add_argument("--lanes", default="1,2,4,8,16,64")
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    pkg = entry.load_package()
    ob = entry.load_oracle()
    o = ob.Oracle()
    p = ob.Params.default()
    W, H = 1241, 376
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    rng = np.random.default_rng(1)
    base = []
    for k in range(args.distinct):
        _, a = o.compute_features(p, pkg.synth.frame(W, H, (5 * k) % 20, k % 20, 8, 1, 1 + k % 8), dims)
        _, b = o.compute_features(p, pkg.synth.frame(W, H, (5 * k + 5) % 20, (k + 1) % 20, 8, 1, 1 + k % 8), dims)
        pm = o.matching(p, dims, 0, m1p=a, m1c=b)
        idx = rng.choice(len(pm), len(pm) // 10, replace=False)  # a tenth of the flows disturbed: the vote has something to remove
        pm["u1p"][idx] += rng.integers(-25, 26, len(idx)).astype(np.float32)
        pm["v1p"][idx] += rng.integers(-9, 10, len(idx)).astype(np.float32)
        base.append(pm)
    t0 = time.perf_counter()
    want = [o.remove_outliers(pm)[0] for pm in base]
    cpu_ms = 1e3 * (time.perf_counter() - t0) / len(base)
    want_b = [o.bucket_features(w, 2, 50, 50) for w in want]
    print(f"{len(base)} distinct lists, {np.mean([len(b) for b in base]):.0f} matches each, {np.mean([len(w) for w in want]):.0f} kept, "
          f"{np.mean([len(w) for w in want_b]):.0f} bucketed; oracle vote {cpu_ms:.2f} ms per list", flush=True)
    t0 = time.perf_counter()
    for pm in base:
        pkg.remove_outliers(pm)
    host_ms = 1e3 * (time.perf_counter() - t0) / len(base)
    res = {"lists": args.lists, "matches_per_list": float(np.mean([len(b) for b in base])), "host_vote_ms_per_list": host_ms, "runs": []}
    lists = [base[i % len(base)] for i in range(args.lists)]
    for lanes in [int(x) for x in args.lanes.split(",")]:
        got, ntri, ms = pkg.remove_outliers_device(lists, lanes_per_wave=lanes)
        ok = all(got[i].tobytes() == want[i % len(base)].tobytes() for i in range(args.lists))
        got_b, _, ms_b = pkg.remove_outliers_device(lists, lanes_per_wave=lanes, max_features=2, bucket_width=50.0, bucket_height=50.0)
        ok_b = all(got_b[i].tobytes() == want_b[i % len(base)].tobytes() for i in range(args.lists))
        print(f"lanes {lanes:2d}: sweep {ms:8.2f} ms for {args.lists} lists ({ms / args.lists:.3f} ms per list amortised), "
              f"vote parity {ok}, vote+bucket parity {ok_b}, triangles {ntri.mean():.0f}", flush=True)
        res["runs"].append({"lanes_per_wave": lanes, "sweep_ms": ms, "sweep_ms_second_call": ms_b, "parity_vote": bool(ok), "parity_bucket": bool(ok_b)})
        if not (ok and ok_b):
            bad = [i for i in range(args.lists) if got[i].tobytes() != want[i % len(base)].tobytes()][:3]
            print("  first deviating lists:", bad, [(len(got[i]), len(want[i % len(base)])) for i in bad])
    print(f"host (csrc/outliers.cpp) {host_ms:.2f} ms per list and thread")
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    return 0 if all(r["parity_vote"] and r["parity_bucket"] for r in res["runs"]) else 1


if __name__ == "__main__":
    sys.exit(main())
