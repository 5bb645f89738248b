#!/usr/bin/env python3
"""Randomised parity soak of the batched egomotion estimators on the GPU box: vh_estimate_motion_stereo and
vh_estimate_motion_mono against the CPU oracle (itself pinned bit for bit to the reference's estimateMotion) over
random scenes -- list lengths from below the minimum to a few thousand, outlier shares, pixel noise, motions,
RANSAC iteration counts, thresholds, pitch -- many lists per launch.  Inlier sets must be equal, ok flags equal,
tr within 1e-9 relative (the tolerance the tests assert: parallel summation / device exp, asin, cos).

    python tools/fuzz_ego.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from egomotion_scene import mono_scene, scene  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pkg = entry.load_package(); ob = entry.load_oracle(); oracle = ob.Oracle()
rng = np.random.default_rng(seed0)
t_end = time.time() + budget
stats = {"stereo_lists": 0, "mono_lists": 0, "stereo_ok": 0, "mono_ok": 0, "fail": 0, "ill_conditioned": 0, "max_rel": 0.0}


def ill_conditioned(run, e, field):
    """A list whose ORACLE result moves by more than the tolerance -- or flips its success flag -- when one input
    changes by one unit in the last place is ill-conditioned: the estimate is a chaotic function of rounding
    there (minimal or degenerate samples, every match an inlier of a loose threshold), and the GPU's rounding
    (device sin/cos/exp, parallel summation) is such a change.  Those lists are counted apart, not as failures."""
    ok0, tr0, inl0 = run(e)
    v = getattr(e, field)
    worst = 0.0
    for k in (1, -1, 2):
        setattr(e, field, float(np.nextafter(v, v + k)) if abs(k) == 1 else float(np.nextafter(np.nextafter(v, v + 1), v + 1)))
        ok1, tr1, inl1 = run(e)
        if ok1 != ok0 or not np.array_equal(inl0, inl1):
            worst = np.inf
        elif ok0:
            worst = max(worst, float(np.max(np.abs(tr1 - tr0) / np.maximum(np.abs(tr0), 1e-3))))
    setattr(e, field, v)
    return worst > 1e-9, worst


def close(a, b):
    d = np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) if len(a) else 0.0
    stats["max_rel"] = max(stats["max_rel"], float(d))
    return np.allclose(a, b, rtol=1e-9, atol=1e-12)


trial = 0
while time.time() < t_end:
    trial += 1
    n_lists = int(rng.integers(1, 24))
    # ---- stereo
    iters = int(rng.choice([20, 50, 200, 333]))
    e = ob.EgoParams.default(f=float(rng.uniform(300, 900)), cu=float(rng.uniform(300, 700)), cv=float(rng.uniform(100, 250)),
                             base=float(rng.uniform(0.2, 0.8)), ransac_iters=iters, reweighting=int(rng.integers(0, 2)),
                             inlier_threshold=float(rng.choice([0.5, 1.0, 2.0, 4.0])))
    ge = pkg.EgoParams(ransac_iters=e.ransac_iters, reweighting=e.reweighting, inlier_threshold=e.inlier_threshold, f=e.f, cu=e.cu, cv=e.cv, base=e.base)
    lists = []
    for s in range(n_lists):
        n = int(rng.choice([0, 3, 5, 6, 7, int(rng.integers(8, 400)), int(rng.integers(400, 2500))]))
        tr = (rng.normal(0, 0.01), rng.normal(0, 0.02), rng.normal(0, 0.005), rng.normal(0, 0.05), rng.normal(0, 0.02), -abs(rng.normal(0.8, 0.4)))
        pm = scene(ob.P_MATCH_DTYPE, n, int(rng.integers(1, 1 << 30)), tr=tr, outliers=float(rng.uniform(0, 0.7)), noise=float(rng.uniform(0, 0.8)),
                   f=e.f, cu=e.cu, cv=e.cv, base=e.base)[0] if n else np.zeros(0, ob.P_MATCH_DTYPE)
        if n and rng.random() < 0.05:
            pm[:] = pm[0]  # degenerate: singular normal equations everywhere
        lists.append(pm)
    raw = rng.integers(-2 ** 31, 2 ** 31 - 1, (n_lists, iters, 3)).astype(np.int32)  # any 32-bit values: the sign bit is dropped
    tr_g, ok_g, inl_g = pkg.estimate_motion_stereo(ge, lists, raw)
    for s, pm in enumerate(lists):
        ok_o, tr_o, inl_o = oracle.estimate_motion_stereo(e, pm, oracle.draw_samples(len(pm), iters, (raw[s].reshape(-1) & 0x7FFFFFFF).astype(np.int32))) \
            if len(pm) >= 6 else (False, np.zeros(6), np.zeros(0, np.int32))
        good = ok_g[s] == ok_o and np.array_equal(inl_g[s], inl_o) and close(tr_g[s], tr_o)
        stats["stereo_lists"] += 1; stats["stereo_ok"] += int(ok_o)
        if not good:
            smp = oracle.draw_samples(len(pm), iters, (raw[s].reshape(-1) & 0x7FFFFFFF).astype(np.int32))
            ill, worst = ill_conditioned(lambda q: oracle.estimate_motion_stereo(q, pm, smp), e, "cu")
            same = len(pm) > 0 and len(np.unique(pm)) == 1  # every match identical: singular normal equations in every hypothesis
            if ill or same:
                stats["ill_conditioned"] += 1
                print(f"ill-conditioned stereo trial {trial} list {s}: n {len(pm)}, " + ("all matches identical (singular systems)" if same else
                      f"the oracle itself moves by {worst:.1e} under a 1-ulp change of cu"), flush=True)
                continue
            stats["fail"] += 1
            print(f"FAIL stereo trial {trial} list {s}: n {len(pm)} iters {iters} ok {ok_g[s]}/{ok_o} inl {len(inl_g[s])}/{len(inl_o)} tr {tr_g[s]} vs {tr_o}", flush=True)
    # ---- mono
    iters = int(rng.choice([30, 200, 700, 2000]))
    m = ob.MonoParams.default(f=float(rng.uniform(300, 900)), cu=float(rng.uniform(300, 700)), cv=float(rng.uniform(100, 250)), height=float(rng.uniform(1.0, 2.0)),
                              pitch=float(rng.choice([0.0, -0.02, -0.05])), ransac_iters=iters, inlier_threshold=float(rng.choice([0.000005, 0.00001, 0.00003])),
                              motion_threshold=float(rng.choice([30.0, 100.0, 300.0])))
    gm = pkg.MonoParams.default(ransac_iters=m.ransac_iters, inlier_threshold=m.inlier_threshold, motion_threshold=m.motion_threshold, height=m.height,
                                pitch=m.pitch, f=m.f, cu=m.cu, cv=m.cv)
    lists = []
    for s in range(n_lists):
        n = int(rng.choice([0, 5, 9, 10, 11, int(rng.integers(12, 500)), int(rng.integers(500, 1500))]))
        tr = (rng.normal(0, 0.004), rng.normal(0, 0.02), rng.normal(0, 0.003), rng.normal(0, 0.05), rng.normal(0, 0.02), -abs(rng.normal(0.8, 0.3)))
        pm = mono_scene(ob.P_MATCH_DTYPE, n, int(rng.integers(1, 1 << 30)), tr=tr, outliers=float(rng.uniform(0, 0.6)), noise=float(rng.uniform(0, 0.6)),
                        ground=float(rng.uniform(0.1, 0.7)), height=m.height, f=m.f, cu=m.cu, cv=m.cv)[0] if n else np.zeros(0, ob.P_MATCH_DTYPE)
        lists.append(pm)
    raw = rng.integers(0, 2 ** 31 - 1, (n_lists, iters, 8)).astype(np.int32)
    tr_g, ok_g, inl_g = pkg.estimate_motion_mono(gm, lists, raw)
    for s, pm in enumerate(lists):
        ok_o, tr_o, inl_o = oracle.estimate_motion_mono(m, pm, oracle.draw_samples_n(len(pm), 8, iters, raw[s].reshape(-1))) \
            if len(pm) >= 10 else (False, np.zeros(6), np.zeros(0, np.int32))
        good = ok_g[s] == ok_o and np.array_equal(inl_g[s], inl_o) and close(tr_g[s], tr_o)
        stats["mono_lists"] += 1; stats["mono_ok"] += int(ok_o)
        if not good:
            smp = oracle.draw_samples_n(len(pm), 8, iters, raw[s].reshape(-1))
            ill, worst = ill_conditioned(lambda q: oracle.estimate_motion_mono(q, pm, smp), m, "height")
            if ill:
                stats["ill_conditioned"] += 1
                print(f"ill-conditioned mono trial {trial} list {s}: n {len(pm)}, the oracle itself moves by {worst:.1e} under a 1-ulp change of height", flush=True)
                continue
            stats["fail"] += 1
            print(f"FAIL mono trial {trial} list {s}: n {len(pm)} iters {iters} ok {ok_g[s]}/{ok_o} inl {len(inl_g[s])}/{len(inl_o)} tr {tr_g[s]} vs {tr_o}", flush=True)
    if trial % 10 == 0:
        print(f"[{trial} launches, {stats}]", flush=True)
print(f"done: {trial} launch pairs, {stats['stereo_lists']} stereo lists ({stats['stereo_ok']} with a pose), {stats['mono_lists']} mono lists "
      f"({stats['mono_ok']} with a pose), {stats['fail']} failing, {stats['ill_conditioned']} ill-conditioned (the oracle itself moves beyond the "
      f"tolerance or flips its outcome under a 1-ulp change of one input, or every match of the list is identical)")
