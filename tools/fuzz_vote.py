#!/usr/bin/env python3
"""Run ON THE GPU BOX: soak of the device form of removeOutliers + bucketFeatures (vh_remove_outliers_device,
csrc/kernels_vote.hip) against the pinned oracle (oracle/viso_outliers.c): random match lists -- sizes 0..12 000, image
shapes, flow fields with outliers, duplicate and collinear points, textures on a grid (co-circular quadruples) -- in
batches through every lanes-per-wave setting.  Prints one line per batch and a summary; exit code 1 on any deviation.

  python tools/fuzz_vote.py [--seconds 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def random_list(pkg, rng):
    kind = rng.integers(0, 8)
    W, H = int(rng.integers(64, 1400)), int(rng.integers(48, 500))
    n = int(min(rng.integers(0, 12001) if kind else rng.integers(0, 40), W * H // 2))
    pm = np.zeros(n, pkg.P_MATCH_DTYPE)
    if kind == 1 and n:  # a regular grid: every quadruple of neighbours is co-circular
        step = max(2, int(np.sqrt(W * H / max(n, 1))))
        xs, ys = np.meshgrid(np.arange(0, W, step), np.arange(0, H, step))
        k = min(n, xs.size)
        pm = pm[:k]
        sel = rng.permutation(xs.size)[:k]
        pm["u1c"] = xs.ravel()[sel]; pm["v1c"] = ys.ravel()[sel]
    elif n:
        cells = rng.choice(W * H, n, replace=False)
        pm["u1c"] = (cells % W).astype(np.float32); pm["v1c"] = (cells // W).astype(np.float32)
    n = len(pm)
    fu, fv = rng.integers(-6, 7), rng.integers(-3, 4)
    pm["u1p"] = pm["u1c"] + fu + (pm["u1c"] / max(W, 1) * rng.integers(0, 4)).astype(np.int32)
    pm["v1p"] = pm["v1c"] + fv
    if n:
        k = rng.choice(n, max(1, int(n * rng.uniform(0, 0.4))), replace=False)
        pm["u1p"][k] += rng.integers(-30, 31, len(k)).astype(np.float32)
        pm["v1p"][k] += rng.integers(-12, 13, len(k)).astype(np.float32)
    if kind == 2 and n > 20:  # duplicates
        k = rng.choice(n, n // 10, replace=False)
        pm["u1c"][k] = pm["u1c"][(k + 1) % n]; pm["v1c"][k] = pm["v1c"][(k + 1) % n]
    if kind == 3 and n > 5:  # collinear
        pm["v1c"] = float(rng.integers(0, H))
    pm["i1c"] = np.arange(n); pm["i1p"] = np.arange(n)
    for f in ("u2p", "v2p", "u2c", "v2c"):
        pm[f] = -1
    pm["i2p"] = -1; pm["i2c"] = -1
    return pm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    pkg = entry.load_package(); ob = entry.load_oracle(); o = ob.Oracle()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    lists_n = matches_n = bad = batches = 0
    while time.time() < t_end:
        P = int(rng.integers(1, 70))
        lists = [random_list(pkg, rng) for _ in range(P)]
        lanes = int(rng.choice([1, 2, 5, 16, 64]))
        mf, bw, bh = int(rng.integers(1, 5)), float(rng.choice([50, 33, 64, 100])), float(rng.choice([50, 40, 25]))
        want = [o.remove_outliers(pm)[0] for pm in lists]
        try:
            got, _, _ = pkg.remove_outliers_device(lists, lanes_per_wave=lanes)
            got_b, _, _ = pkg.remove_outliers_device(lists, lanes_per_wave=lanes, max_features=mf, bucket_width=bw, bucket_height=bh)
        except pkg.VisoHipError as ex:  # which list, which call?
            print(f"  REFUSED: batch {batches}: {ex}", flush=True)
            for k, pm in enumerate(lists):
                for mode in (0, mf):
                    try:
                        pkg.remove_outliers_device([pm], lanes_per_wave=lanes, max_features=mode, bucket_width=bw, bucket_height=bh)
                    except pkg.VisoHipError as ex1:
                        depth = o.remove_outliers(pm)[1]
                        print(f"    list {k} alone (n {len(pm)}, max_features {mode}, oracle flip-stack depth {depth}): {ex1}", flush=True)
                        np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_vote_fail_{batches}_{k}.npy"), pm)
            bad += 1; batches += 1
            continue
        nb = 0
        for k in range(P):
            if got[k].tobytes() != want[k].tobytes() or got_b[k].tobytes() != o.bucket_features(want[k], mf, bw, bh).tobytes():
                nb += 1
                print(f"  DEVIATION: batch {batches} list {k}: n {len(lists[k])}, kept {len(got[k])} vs {len(want[k])}", flush=True)
        bad += nb; batches += 1; lists_n += P; matches_n += sum(len(x) for x in lists)
        print(f"batch {batches:4d}: {P:3d} lists, lanes {lanes:2d}, bucket {mf}/{bw:.0f}x{bh:.0f}: {'ok' if not nb else str(nb) + ' deviating'}", flush=True)
    print(f"fuzz_vote: {batches} batches, {lists_n} lists, {matches_n} matches, {bad} deviating (seed {args.seed})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
