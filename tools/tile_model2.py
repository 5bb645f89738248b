#!/usr/bin/env python3
"""Offline cost model, round 2: flow-search tilings with T queries per wave (64/T
phases sharing one candidate stream), internal walk cells finer than the
reference's bins, and the partial-SAD filter.  CPU only (oracle features).

  python tools/tile_model2.py [noise]
"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); ob = g.load_oracle(); o = ob.Oracle(); p = ob.Params.default()
W, H = 1241, 376; dims = [W, H, 1248]; r = 200
noise = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(1)
def img(dx, dy):
    im = pkg.synth.frame(W, H, dx, dy)
    if noise:
        im = np.clip(im.astype(np.int32) + rng.integers(-noise, noise + 1, im.shape), 0, 255).astype(np.uint8)
        im[:, W:] = 0
    return im
fq = o.compute_features(p, img(5, 1), dims)[1]
fc = o.compute_features(p, img(0, 0), dims)[1]
print(f"noise {noise}: {len(fq)} queries, {len(fc)} candidates")

COST = {0: 9.25, 1: 12.25, 2: 13.25}

def model(T, GU, GV, order="col"):
    """queries ordered (class, u//GU, v//GV, idx); tiles of T consecutive per class;
    walk: columns of GU px, cells of GV px; a cell is skipped if outside the union window,
    untested if inside every lane's window, v-tested if its column is inside every lane's u window."""
    ubn = -(-W // GU); vbn = -(-H // GV)
    tot_ops = 0.0; tot_eval = 0; tot_in = 0; ntile = 0; nchunk = 0
    for c in range(4):
        q = fq[fq[:, 3] == c]; cd = fc[fc[:, 3] == c]
        cu = np.minimum(cd[:, 0] // GU, ubn - 1); cv = np.minimum(cd[:, 1] // GV, vbn - 1)
        cnt = np.zeros((ubn, vbn), np.int64); np.add.at(cnt, (cu, cv), 1)
        qu = np.minimum(q[:, 0] // GU, ubn - 1); qv = np.minimum(q[:, 1] // GV, vbn - 1)
        if order == "col":
            od = np.lexsort((qv, qu))
        else:  # morton over (qu, qv)
            def part(x):
                x = x.astype(np.int64); y = np.zeros_like(x)
                for b in range(10): y |= ((x >> b) & 1) << (2 * b)
                return y
            od = np.argsort(part(qu) | (part(qv) << 1), kind="stable")
        q = q[od]
        for i in range(0, len(q), T):
            t = q[i:i + T]; u = t[:, 0]; v = t[:, 1]
            ulo, uhi, vlo, vhi = u - r, u + r, v - r, v + r
            UB0 = max(ulo.min(), 0) // GU; UB1 = min(uhi.max() // GU, ubn - 1)
            VB0 = max(vlo.min(), 0) // GV; VB1 = min(vhi.max() // GV, vbn - 1)
            ULO, UHI, VLO, VHI = ulo.max(), uhi.min(), vlo.max(), vhi.min()
            ev = 0; ops = 0.0
            for ub in range(UB0, UB1 + 1):
                interior = ub * GU >= ULO and min(ub * GU + GU - 1, W - 1) <= UHI
                col = cnt[ub, VB0:VB1 + 1]
                n = int(col.sum())
                if not n: continue
                nchunk += -(-n // 64)
                if not interior:
                    ops += n * COST[2]
                else:
                    va0 = max(VB0, -(-max(VLO, 0) // GV)); va1 = min(VB1, (VHI + 1) // GV - 1)
                    n0 = int(cnt[ub, va0:va1 + 1].sum()) if va0 <= va1 else 0
                    ops += n0 * COST[0] + (n - n0) * COST[1]
                ev += n
            inw = ((np.abs(cd[None, :, 0] - u[:, None]) <= r) & (np.abs(cd[None, :, 1] - v[:, None]) <= r)).sum()
            tot_ops += ops * T  # lane-ops the wave issues for this tile (idle lanes of a partial tile included)
            tot_eval += ev * len(t); tot_in += int(inw); ntile += 1
    # wave-ops: each evaluated (lane, candidate) pair costs COST/64 of a wave instruction when all lanes are busy;
    # a T-query tile with 64/T phases keeps all 64 lanes busy on T queries
    return tot_ops, tot_eval, tot_in, ntile, nchunk

print("T   GU  GV  order   tiles  eff    lane-ops/in-window pair   (+ overhead 40/tile + 12/chunk wave-ops)")
for order in ("col", "morton"):
    for T in (64, 32, 16, 8):
        for GU, GV in ((50, 50), (50, 25), (50, 10), (25, 25), (25, 10), (32, 16), (16, 16)):
            ops, ev, inw, nt, nch = model(T, GU, GV, order)
            over = (40.0 * nt + 12.0 * nch) * 64
            print(f"{T:3d} {GU:3d} {GV:3d} {order:7s} {nt:5d}  {inw / ev:.3f}  {ops / inw:6.2f}   {(ops + over) / inw:6.2f}")
