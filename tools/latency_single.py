#!/usr/bin/env python3
"""Single-stream and host-image (PCIe-inclusive) rates of the Matcher surface --
context numbers for DESIGN.md section 4.4; never the bench `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
W, H = 1241, 376
bpl = pkg.synth.bytes_per_line(W)
dims = [W, H, bpl]
seq = pkg.synth.stereo_sequence(W, H, 8, 12)

for outlier_removal in (False, True):
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=outlier_removal)
    for l, r in seq[:3]:
        m.pushBack(l, r, dims, False); m.matchFeatures(2); m.getMatches()
    t0 = time.perf_counter(); n = 0
    for rep in range(5):
        for l, r in seq:
            m.pushBack(l, r, dims, False); m.matchFeatures(2); pm = m.getMatches(); n += 1
    dt = time.perf_counter() - t0
    print(f"single stream, host images, getMatches every frame (the VisualOdometryStereo::process pattern), "
          f"removeOutliers {'on (host)' if outlier_removal else 'off'}: "
          f"{1e3 * dt / n:.3f} ms per pair = {n / dt:.0f} pairs/s, {len(pm)} matches")
    m.close()

for S, pinned in ((64, False), (256, False), (256, True)):
    g = pkg.StreamGroup(S, pkg.Params.default(), max_features=32768, max_matches=32768)
    L = [np.ascontiguousarray(np.stack([seq[t][0]] * S)) for t in range(8)]
    R = [np.ascontiguousarray(np.stack([seq[t][1]] * S)) for t in range(8)]
    if pinned:  # page-locked image buffers (vh_host_alloc)
        for buf in (L, R):
            for t in range(8):
                q = pkg.pinned_empty(buf[t].shape); q[...] = buf[t]; buf[t] = q
    for t in range(3):
        g.pushBack(L[t], R[t], dims, False); g.matchFeatures(2)
    g.synchronize()
    t0 = time.perf_counter(); k = 0
    for rep in range(2):
        for t in range(8):
            g.pushBack(L[t], R[t], dims, False); g.matchFeatures(2); k += 1
    g.synchronize()
    dt = time.perf_counter() - t0
    print(f"group of {S} streams, HOST images ({'page-locked' if pinned else 'pageable'}, H2D inside pushBack): {S * k / dt:.0f} pairs/s")
    # images in AND matches out (vh_group_get_matches_all) every step
    mbuf = pkg.pinned_empty((S, 16384), pkg.P_MATCH_DTYPE) if pinned else np.zeros((S, 16384), pkg.P_MATCH_DTYPE)
    t0 = time.perf_counter(); k = 0
    for rep in range(2):
        for t in range(8):
            g.pushBack(L[t], R[t], dims, False); g.matchFeatures(2); _, cnt = g.getMatchesAll(out=mbuf); k += 1
    dt = time.perf_counter() - t0
    print(f"group of {S} streams, HOST images in + all matches out ({'page-locked' if pinned else 'pageable'} buffers, "
          f"{int(cnt.mean())} matches per pair): {S * k / dt:.0f} pairs/s")
    if pinned:
        # the same, pipelined: the download of step k runs while step k+1 is uploaded and computed
        mb = [pkg.pinned_empty((S, 10240), pkg.P_MATCH_DTYPE) for _ in range(2)]
        cb = [pkg.pinned_empty((S,), np.int32) for _ in range(2)]
        t0 = time.perf_counter(); k = 0
        for rep in range(2):
            for t in range(8):
                g.pushBack(L[t], R[t], dims, False); g.matchFeatures(2)
                g.waitDownload()                       # step k-1's lists are now in mb[(k-1) % 2]
                g.downloadMatchesAsync(mb[k % 2], cb[k % 2]); k += 1
        g.waitDownload()
        dt = time.perf_counter() - t0
        print(f"group of {S} streams, HOST images in + all matches out, asynchronous download "
              f"({int(cb[(k - 1) % 2].mean())} matches per pair): {S * k / dt:.0f} pairs/s")
    g.close()
