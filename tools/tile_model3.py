#!/usr/bin/env python3
"""Offline cost model, round 5: which share of the flow search's evaluated (query, candidate) pairs lies inside the
queries' own windows, for different ORDERS of the query tiles and different granularities of the candidate storage.
CPU only (oracle features of the SURVEY Appendix-B frame pair).

  python tools/tile_model3.py

For every tile of T queries: in = pairs inside each query's own +-radius window (what findMatch must evaluate,
src/matcher.cpp:237-249), union = candidates in the union of the tile's windows x queries, walked = the same rounded
outward to whole storage cells, slots = walked x T (idle lanes of a partial tile included).  Round 4's kernel is the
row `T=32 col G=50x50` (bin order, tiles over a whole class: 1.38; measured 1.41), round 5's `T=32 snake G=50x50`
(1.32; measured 1.33).  A finer storage order (G=10x10) would buy 3 % more, 16-query tiles 5 % at twice the tiles.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); ob = g.load_oracle(); o = ob.Oracle(); p = ob.Params.default()
W, H = 1241, 376; dims = [W, H, 1248]; r = 200
fq = o.compute_features(p, pkg.synth.frame(W, H, 5, 1), dims)[1]
fc = o.compute_features(p, pkg.synth.frame(W, H, 0, 0), dims)[1]


def integral(cd):
    cnt = np.zeros((H + 1, W + 1), np.int64)
    np.add.at(cnt, (cd[:, 1] + 1, cd[:, 0] + 1), 1)
    return cnt.cumsum(0).cumsum(1)


def rect(I, u0, u1, v0, v1):  # candidates in the inclusive pixel rectangle, clipped to the image
    u0 = max(u0, 0); v0 = max(v0, 0); u1 = min(u1, W - 1); v1 = min(v1, H - 1)
    if u1 < u0 or v1 < v0:
        return 0
    return int(I[v1 + 1, u1 + 1] - I[v0, u1 + 1] - I[v1 + 1, u0] + I[v0, u0])


IN = {}


def model(T, order, GU=50, GV=50, A=50, colbreak=False):
    """order: 'col' = (class, u // A, v-bin, index) -- the bin order; 'snake' = columns of width A, even ones top-down, odd ones
    bottom-up (kernels_bin.hip: make_tiles); colbreak: tiles end at column ends."""
    tot_in = tot_union = tot_walk = tot_slots = ntile = 0
    for c in range(4):
        q = fq[fq[:, 3] == c]; cd = fc[fc[:, 3] == c]
        I = integral(cd)
        col = q[:, 0] // A
        if order == "col":
            od = np.lexsort((np.arange(len(q)), q[:, 1] // 50, col))
        else:
            od = np.lexsort((q[:, 0], np.where(col % 2 == 0, q[:, 1], -q[:, 1]), col))
        q = q[od]; col = col[od]
        if c not in IN:
            IN[c] = sum(rect(I, a - r, a + r, b - r, b + r) for a, b in zip(q[:, 0], q[:, 1]))
        tot_in += IN[c]
        i = 0
        while i < len(q):
            j = min(i + T, len(q))
            if colbreak:
                k = i
                while k < j and col[k] == col[i]:
                    k += 1
                j = k
            t = q[i:j]; u = t[:, 0]; v = t[:, 1]
            U0, U1, V0, V1 = u.min() - r, u.max() + r, v.min() - r, v.max() + r
            un = rect(I, U0, U1, V0, V1)
            wk = rect(I, max(U0, 0) // GU * GU, min(U1, W - 1) // GU * GU + GU - 1, max(V0, 0) // GV * GV, min(V1, H - 1) // GV * GV + GV - 1)
            tot_union += un * len(t); tot_walk += wk * len(t); tot_slots += wk * T; ntile += 1
            i = j
    return ntile, tot_union / tot_in, tot_walk / tot_in, tot_slots / tot_in


print(f"{len(fq)} queries, {len(fc)} candidates; pairs relative to the in-window pairs of one flow pass")
print("T   order  colbreak  G       tiles  union  walked  slots")
for T, order, cb, GU, GV in [(32, "col", True, 50, 50), (32, "col", False, 50, 50), (32, "snake", False, 50, 50), (32, "snake", False, 10, 10),
                             (16, "col", False, 50, 50), (16, "snake", False, 50, 50), (16, "snake", False, 10, 10), (64, "snake", False, 50, 50)]:
    nt, a, b, c_ = model(T, order, GU, GV, 50, cb)
    print(f"{T:<3d} {order:6s} {str(cb):8s}  {GU}x{GV:<4d} {nt:5d}  {a:.3f}  {b:.3f}   {c_:.3f}")
