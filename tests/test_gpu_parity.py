"""GPU suite: the HIP path, called through the C ABI (libviso_hip.so), must be
bit-identical to the oracle -- and to the golden vectors generated from the
reference's own code -- on the same inputs.  Integer/byte work: the bar is
exact equality everywhere (no tolerances)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names, load_golden

pytestmark = pytest.mark.gpu


def feats_for(oracle, p, dims, imgs):
    return [oracle.compute_features(p, im, dims)[1] for im in imgs]


# ------------------------------------------------------------------ a2-a4: filters
@pytest.mark.parametrize("shape", [(64, 48), (333, 100), (1248, 376)])
def test_filters_planes(shape, pkg, oracle, gpu):
    bpl, H = shape
    rng = np.random.default_rng(bpl)
    img = rng.integers(0, 256, (H, bpl), dtype=np.uint8)
    got = pkg.filters(img)
    want = oracle.filters(img)
    for name, a, b in zip(("du", "dv", "f1", "f2"), got, want):
        assert np.array_equal(a, b), name


# ------------------------------------------------ a1,a5-a7,a12: computeFeatures (golden)
@pytest.mark.parametrize("name", golden_names())
def test_compute_features_golden(name, pkg, ob, oracle, gpu):
    p, dims, Ip, Ic, z = load_golden(name, pkg, pkg.Params)
    m1p, m2p = pkg.compute_features(p, Ip, dims)
    m1c, m2c, du, dv = pkg.compute_features(p, Ic, dims, planes=True)
    assert np.array_equal(m2p, z["max2p"]) and np.array_equal(m2c, z["max2c"])
    assert np.array_equal(m1p, z["max1p"]) and np.array_equal(m1c, z["max1c"])
    assert oracle.fnv(np.ascontiguousarray(du[2:-2, 2:du.shape[1] - 16])) == int(z["du_interior_fnv"])
    assert oracle.fnv(np.ascontiguousarray(dv[2:-2, 2:dv.shape[1] - 16])) == int(z["dv_interior_fnv"])


@pytest.mark.parametrize("name", golden_names())
def test_index_findmatch_matching_golden(name, pkg, ob, oracle, gpu):
    """a8 createIndexVector, a9 findMatch (all queries), a10 flow matching."""
    p, dims, Ip, Ic, z = load_golden(name, pkg, pkg.Params)
    bs, lst = pkg.create_index(p, dims, z["max2c"])
    assert np.array_equal(bs, z["bin_start"]) and np.array_equal(lst, z["bin_list"])
    assert np.array_equal(pkg.match_all(p, dims, z["max2c"], z["max2p"], flow=True), z["fwd"])
    pm = pkg.match(p, dims, pkg.METHOD_FLOW, m1p=z["max2p"], m1c=z["max2c"])
    assert pm.tobytes() == z["p_match"].tobytes()


def test_find_match_prior_term(pkg, ob, oracle, gpu):
    """a9, optional branch: cost + 4*||(u2,v2)-(u_,v_)|| in double (src/matcher.cpp:257-262),
    against the golden table produced by the reference itself and against the oracle."""
    p, dims, Ip, Ic, z = load_golden("small_default", pkg, pkg.Params)
    g = np.load(os.path.join(GOLDEN, "find_match_prior.npz"))
    got = pkg.match_all_prior(p, dims, z["max2c"], z["max2p"], float(g["u_"]), float(g["v_"]))
    assert np.array_equal(got, g["best"])
    po = ob.Params.default(match_radius=90, match_binsize=37)
    pg = pkg.Params.default(match_radius=90, match_binsize=37)
    for u_, v_, flow in ((10.5, 300.25, True), (0.0, 0.0, False), (-1.0, 50.0, True)):
        want = oracle.match_all_prior(po, dims, z["max2c"], z["max2p"], u_, v_, flow=flow)
        assert np.array_equal(pkg.match_all_prior(pg, dims, z["max2c"], z["max2p"], u_, v_, flow=flow), want)


@pytest.mark.parametrize("case", ["kitti_1241x376", "seq_1024x284"])
def test_known_answers_full_size(case, pkg, oracle, gpu):
    """SURVEY Appendix-B sizes through the stateful Matcher surface (mono flow = configs[0])."""
    z = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    W, H, blur, gain, seed, dx, dy = [int(v) for v in z[case + "__gen"]]
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    assert m.pushBack(pkg.synth.frame(W, H, 0, 0, blur, gain, seed), None, dims, False)
    assert m.pushBack(pkg.synth.frame(W, H, dx, dy, blur, gain, seed), None, dims, False)
    m.matchFeatures(pkg.METHOD_FLOW)
    fp, fc, pm = m.getFeatures(pkg.SET_1P), m.getFeatures(pkg.SET_1C), m.getMatches()
    m.close()
    assert (len(fp), len(fc), len(pm)) == (int(z[case + "__n2p"]), int(z[case + "__n2c"]), int(z[case + "__n_match"]))
    assert oracle.fnv(fp) == int(z[case + "__fnv_max2p"]) and oracle.fnv(fc) == int(z[case + "__fnv_max2c"])
    assert oracle.fnv(pm) == int(z[case + "__fnv_p_match"])
    assert pm[:32].tobytes() == z[case + "__head_p_match"].tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_every_filter_layout_of_detect_nms(n, pkg, ob, oracle, gpu):
    """detect_nms_fast<N> lays its filter pass out per nms_n (2 or 4 columns per lane, 3..7 row
    segments; packed block extrema for odd n) and nms_n = 5 takes the generic kernel: each layout
    on an image of several tiles with partial edge tiles, on a smooth and on a tie-rich texture."""
    W, H = 777, 301
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    for blur, gain, tau in ((6, 1, 50), (2, 3, 20)):
        img = pkg.synth.frame(W, H, 3, 1, blur, gain, 40 + n)
        if gain == 3:  # flat and saturated patches: equal responses, first-in-scan-order ties
            img = img.copy(); img[40:90, 100:260] = 255; img[150:200, 300:520] = 17
        p, po = pkg.Params.default(nms_n=n, nms_tau=tau), ob.Params.default(nms_n=n, nms_tau=tau)
        got = pkg.compute_features(p, img, dims)
        want = oracle.compute_features(po, img, dims)
        assert len(want[1]) > 200 and np.array_equal(got[1], want[1]), (n, blur, gain, tau)


def test_unaligned_stride_generic_kernels(pkg, ob, oracle, gpu):
    """Rows that do not start on 4-byte boundaries (bpl = odd width, as the
    reference's dims[2] >= dims[0] contract allows) take the generic detection kernel
    (emit_features fetches its patches byte-granular either way)."""
    rng = np.random.default_rng(3)
    for W, H, n in ((201, 100, 2), (333, 121, 3), (255, 97, 6)):
        img = pkg.synth.frame(W, H, blur=3, seed=W)[:, :W].copy()  # stride == W
        dims = [W, H, W]
        p, po = pkg.Params.default(nms_n=n), ob.Params.default(nms_n=n)
        got = pkg.compute_features(p, img, dims)
        want = oracle.compute_features(po, img, dims)
        assert len(want[1]) > 100 and np.array_equal(got[1], want[1])


# ------------------------------------------------ random parameter sweep vs the oracle
@pytest.mark.parametrize("trial", range(8))
def test_random_configs_vs_oracle(trial, pkg, ob, oracle, gpu):
    rng = np.random.default_rng(100 + trial)
    W = int(rng.integers(120, 500)); H = int(rng.integers(90, 300))
    over = {"nms_n": int(rng.integers(1, 6)), "nms_tau": int(rng.integers(5, 90)),
            "match_binsize": int(rng.integers(10, 120)), "match_radius": int(rng.integers(5, 300)),
            "match_disp_tolerance": int(rng.integers(0, 5)),
            "half_resolution": int(trial % 2), "multi_stage": int((trial // 2) % 2)}
    p, po = pkg.Params.default(**over), ob.Params.default(**over)
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    blur, gain, seed = int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(1, 10000))
    disp = int(rng.integers(0, 10))
    imgs = []
    for t in range(2):
        dx, dy = int(rng.integers(0, 8)) * t, int(rng.integers(0, 5)) * t
        imgs += [pkg.synth.frame(W, H, dx, dy, blur, gain, seed), pkg.synth.frame(W, H, dx + disp, dy, blur, gain, seed)]
    for im in imgs[:2]:
        got = pkg.compute_features(p, im, dims)
        want = oracle.compute_features(po, im, dims)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), over
    f = feats_for(oracle, po, dims, imgs)
    for flow in (True, False):
        assert np.array_equal(pkg.match_all(p, dims, f[2], f[0], flow=flow), oracle.match_all(po, dims, f[2], f[0], flow=flow)), over
    for method in (0, 1, 2):
        got = pkg.match(p, dims, method, *f)
        want = oracle.matching(po, dims, method, *f)
        assert got.tobytes() == want.tobytes(), (over, method)


@pytest.mark.parametrize("binsize,tau", [(2, 70), (3, 80), (4, 80)])
def test_query_tiles_across_many_sparse_columns(binsize, tau, pkg, ob, oracle, gpu):
    """Snake-ordered query tiles (csrc/kernels_bin.hip: make_tiles) on sets with far fewer features than (class, u-bin)
    columns: a tile of 32 queries then runs through dozens of columns -- more than the 63 column starts a wave holds at a
    time for the snake index -> position mapping (kernels_match.hip: flow_tile), so the table is re-fetched mid-tile -- and
    through empty ones.  Caller-supplied feature sets (histogram / fill / sort path) and the detector's own."""
    W, H = 1400, 120
    over = {"nms_tau": tau, "match_binsize": binsize, "match_radius": 150}
    p, po = pkg.Params.default(**over), ob.Params.default(**over)
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    imgs = [pkg.synth.frame(W, H, dx, dy, 9, 1, 77) for dx, dy in ((0, 0), (9, 0), (4, 1), (13, 1))]
    f = feats_for(oracle, po, dims, imgs)
    ncol = -(-W // binsize)
    per_class = [int((f[0][:, 3] == c).sum()) for c in range(4)]
    assert 0 < min(per_class) and max(per_class) * 2 < ncol, (per_class, ncol)  # (sparse: a 32-query tile spans > 64 columns on average)
    for flow in (True, False):
        assert np.array_equal(pkg.match_all(p, dims, f[2], f[0], flow=flow), oracle.match_all(po, dims, f[2], f[0], flow=flow))
    for method in (0, 1, 2):
        assert pkg.match(p, dims, method, *f).tobytes() == oracle.matching(po, dims, method, *f).tobytes(), method
    m = pkg.Matcher(p, outlier_removal=False)
    m.pushBack(imgs[0], imgs[1], dims, False)
    m.pushBack(imgs[2], imgs[3], dims, False)
    m.matchFeatures(pkg.METHOD_QUAD)
    assert m.getMatches().tobytes() == oracle.matching(po, dims, 2, *f).tobytes()
    m.close()


# ------------------------------------------------ stateful Matcher: ring buffer, methods
def test_matcher_ring_buffer_and_methods(pkg, ob, oracle, gpu):
    """pushBack ring (src/matcher.cpp:64-79), replace flag, stereo + quad + flow on one handle."""
    W, H = 400, 200
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    p, po = pkg.Params.default(), ob.Params.default()
    seq = pkg.synth.stereo_sequence(W, H, 4, disparity=8, blur=4, seed=21)
    F = [[oracle.compute_features(po, im, dims)[1] for im in pair] for pair in seq]
    m = pkg.Matcher(p, outlier_removal=False)
    m.pushBack(seq[0][0], seq[0][1], dims, False)
    # one frame only: previous sets are empty -> no flow/quad matches, stereo works
    m.matchFeatures(pkg.METHOD_QUAD)
    assert len(m.getMatches()) == 0
    m.matchFeatures(pkg.METHOD_STEREO)
    assert m.getMatches().tobytes() == oracle.matching(po, dims, 1, m1c=F[0][0], m2c=F[0][1]).tobytes()
    for t in (1, 2):
        m.pushBack(seq[t][0], seq[t][1], dims, False)
        for k, want in enumerate((F[t - 1][0], F[t - 1][1], F[t][0], F[t][1])):
            assert np.array_equal(m.getFeatures(k), want)
        for method in (2, 0, 1):
            m.matchFeatures(method)
            assert m.getMatches().tobytes() == oracle.matching(po, dims, method, F[t - 1][0], F[t - 1][1], F[t][0], F[t][1]).tobytes()
    # replace=True overwrites the current pair and keeps the previous one
    m.pushBack(seq[3][0], seq[3][1], dims, True)
    assert np.array_equal(m.getFeatures(pkg.SET_1P), F[1][0]) and np.array_equal(m.getFeatures(pkg.SET_1C), F[3][0])
    m.matchFeatures(pkg.METHOD_QUAD)
    assert m.getMatches().tobytes() == oracle.matching(po, dims, 2, F[1][0], F[1][1], F[3][0], F[3][1]).tobytes()
    # mono push after stereo: the right sets of the new pair are empty
    m.pushBack(seq[0][0], None, dims, False)
    assert len(m.getFeatures(pkg.SET_2C)) == 0 and np.array_equal(m.getFeatures(pkg.SET_1P), F[3][0])
    m.matchFeatures(pkg.METHOD_FLOW)
    assert m.getMatches().tobytes() == oracle.matching(po, dims, 0, m1p=F[3][0], m1c=F[0][0]).tobytes()
    m.close()


def test_dimension_mismatch_and_errors(pkg, gpu, capsys):
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    img = np.zeros((50, 64), np.uint8)
    assert m.pushBack(img, None, [64, 50, 32], False) is False  # bpl < width (src/matcher.cpp:59-62)
    assert "Image dimension mismatch" in capsys.readouterr().out
    assert m.pushBack(img, None, [0, 50, 64], False) is False
    with pytest.raises(pkg.VisoHipError) as e:
        m.matchFeatures(0)  # nothing pushed yet
    assert e.value.code == pkg.VH_ERR_STATE
    with pytest.raises(pkg.VisoHipError):
        m.matchFeatures(7)
    m.close()


def test_tiny_and_empty_inputs(pkg, ob, oracle, gpu):
    p, po = pkg.Params.default(), ob.Params.default()
    # image too small for a single NMS block, and a flat image: no features, no crash
    for W, H, fill in ((16, 16, 0), (18, 40, 0), (200, 100, 77)):
        img = np.full((H, pkg.synth.bytes_per_line(W)), fill, np.uint8)
        a = pkg.compute_features(p, img, [W, H, img.shape[1]])
        assert len(a[0]) == 0 and len(a[1]) == 0
    dims = [200, 100, 208]
    f = oracle.compute_features(po, pkg.synth.frame(200, 100, blur=3, seed=2), dims)[1]
    empty = np.zeros((0, 12), np.int32)
    # empty candidate set: findMatch returns min_ind = 0 for every query (src/matcher.cpp:221)
    assert np.array_equal(pkg.match_all(p, dims, f, empty), np.zeros(len(f), np.int32))
    assert len(pkg.match_all(p, dims, empty, f)) == 0
    for method in (0, 1, 2):
        assert len(pkg.match(p, dims, method, m1p=empty, m2p=f, m1c=f, m2c=empty)) == 0
    # a single feature on both sides closes its own circle
    one = f[:1]
    assert pkg.match(p, dims, 0, m1p=one, m1c=one).tobytes() == oracle.matching(po, dims, 0, m1p=one, m1c=one).tobytes()
    # queries whose window holds no candidate fall back to index 0 and must not close a circle by accident
    far = f.copy(); far[:, 0] = np.minimum(far[:, 0], 30)
    got = pkg.match_all(pkg.Params.default(match_radius=3), dims, f, far)
    assert np.array_equal(got, oracle.match_all(ob.Params.default(match_radius=3), dims, f, far))


def test_pixel_dedup_mask(pkg, ob, oracle, gpu):
    """The fork's first-writer-per-pixel mask M (src/matcher.cpp:331-334): two
    current features on the same pixel (different classes) that both close."""
    dims = [300, 200, 304]
    p, po = pkg.Params.default(), ob.Params.default()
    rng = np.random.default_rng(5)
    n = 400
    prev = np.zeros((n, 12), np.int32)
    prev[:, 0] = rng.integers(10, 290, n); prev[:, 1] = rng.integers(10, 190, n)
    prev[:, 3] = rng.integers(0, 4, n)
    prev[:, 4:] = rng.integers(0, 2**31 - 1, (n, 8))
    cur = prev.copy()
    # force pixel collisions: pairs (2k, 2k+1) share a pixel; classes differ so both can match
    cur[1::2, 0:2] = cur[0::2, 0:2]
    prev[1::2, 0:2] = prev[0::2, 0:2]
    prev[0::2, 3] = 0; cur[0::2, 3] = 0; prev[1::2, 3] = 1; cur[1::2, 3] = 1
    want = oracle.matching(po, dims, 0, m1p=prev, m1c=cur)
    got = pkg.match(p, dims, 0, m1p=prev, m1c=cur)
    assert 0 < len(want) < n and got.tobytes() == want.tobytes()


def test_tie_breaking_order(pkg, ob, oracle, gpu):
    """Identical descriptors everywhere: the winner is decided purely by the
    reference's visiting order (u_bin, v_bin, list position; src/matcher.cpp:243-267)."""
    dims = [640, 480, 640]
    rng = np.random.default_rng(11)
    n = 3000
    a = np.zeros((n, 12), np.int32)
    a[:, 0] = rng.integers(0, 640, n); a[:, 1] = rng.integers(0, 480, n); a[:, 3] = rng.integers(0, 4, n)
    a[:, 4:] = 0x40404040
    b = a.copy(); rng.shuffle(b, axis=0)
    for bs, r in ((50, 200), (33, 77), (100, 40)):
        p, po = pkg.Params.default(match_binsize=bs, match_radius=r), ob.Params.default(match_binsize=bs, match_radius=r)
        for flow in (True, False):
            assert np.array_equal(pkg.match_all(p, dims, a, b, flow=flow), oracle.match_all(po, dims, a, b, flow=flow))
        assert pkg.match(p, dims, 2, a, b, b, a).tobytes() == oracle.matching(po, dims, 2, a, b, b, a).tobytes()


def test_capacity_errors(pkg, gpu):
    W, H = 320, 160
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    img = pkg.synth.frame(W, H, blur=4, seed=3)
    m = pkg.Matcher(pkg.Params.default(), max_features=100, max_matches=50, outlier_removal=False)
    m.pushBack(img, None, dims, False)
    n = C.c_int32(0)
    buf = np.zeros((100, 12), np.int32)
    rc = pkg._lib().vh_get_features(m._h, pkg.SET_1C, buf.ctypes.data_as(C.c_void_p), 100, C.byref(n))
    assert rc == pkg.VH_ERR_CAPACITY and n.value > 100  # true count is still reported
    m.close()
    full = pkg.compute_features(pkg.Params.default(), img, dims)[1]
    assert np.array_equal(buf, full[:100])  # the first `capacity` records are exact


def test_bucket_features(pkg, ob, oracle, gpu):
    """f-2: Matcher::bucketFeatures incl. the LFSR shuffle (src/matcher.cpp:113-187)."""
    g = np.load(os.path.join(GOLDEN, "bucket_1024x284.npz"))
    dims = [1024, 284, 1024]
    for key in g.files:
        _, mf, bw, bh = key.split("_")
        m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
        m.pushBack(pkg.synth.frame(1024, 284, 0, 0), None, dims, False)
        m.pushBack(pkg.synth.frame(1024, 284, 5, 1), None, dims, False)
        m.matchFeatures(pkg.METHOD_FLOW)
        m.bucketFeatures(int(mf), float(bw), float(bh))
        assert m.getMatches().tobytes() == g[key].tobytes(), key
        m.close()
    # beyond the reference's fixed bucket array (1241x376 needs 200 buckets): vs the oracle
    po = ob.Params.default()
    d2 = [1241, 376, 1248]
    fp = oracle.compute_features(po, pkg.synth.frame(1241, 376, 0, 0), d2)[1]
    fc = oracle.compute_features(po, pkg.synth.frame(1241, 376, 5, 1), d2)[1]
    want = oracle.bucket_features(oracle.matching(po, d2, 0, m1p=fp, m1c=fc), 2, 50, 50)
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    m.pushBack(pkg.synth.frame(1241, 376, 0, 0), None, d2, False)
    m.pushBack(pkg.synth.frame(1241, 376, 5, 1), None, d2, False)
    m.matchFeatures(0); m.bucketFeatures(2, 50, 50)
    assert m.getMatches().tobytes() == want.tobytes()
    m.close()


# ------------------------------------------------ multi-stream group
def test_stream_group_equals_independent_matchers(pkg, ob, oracle, gpu):
    """configs[3] in miniature: S independent sequences stepped together give,
    stream by stream, exactly what one Matcher per sequence gives."""
    S, W, H = 5, 360, 180
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    po = ob.Params.default()
    seqs = [pkg.synth.stereo_sequence(W, H, 3, disparity=5 + s, blur=3 + s % 3, seed=40 + s) for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    for t in range(3):
        L = np.stack([seqs[s][t][0] for s in range(S)]); R = np.stack([seqs[s][t][1] for s in range(S)])
        g.pushBack(L, R, dims, False)
        if t == 0:
            continue
        for method in (2, 0, 1):
            g.matchFeatures(method)
            nf, nm = g.getCounts()
            for s in range(S):
                f = feats_for(oracle, po, dims, (seqs[s][t - 1][0], seqs[s][t - 1][1], seqs[s][t][0], seqs[s][t][1]))
                assert [len(x) for x in f] == list(nf[s])
                for k in range(4):
                    assert np.array_equal(g.getFeatures(s, k), f[k])
                want = oracle.matching(po, dims, method, *f)
                assert nm[s] == len(want) and g.getMatches(s).tobytes() == want.tobytes()
    g.close()


def test_pipelined_steps_without_host_sync(pkg, ob, oracle, gpu):
    """pushBack(t+1) is issued while matchFeatures(t) is still running (two
    internal streams, three ring slots): queue many steps without ever
    synchronising, mixing methods and a replace, and only then read back."""
    S, W, H, T = 3, 320, 160, 7
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    po = ob.Params.default()
    seqs = [pkg.synth.stereo_sequence(W, H, T, disparity=4 + s, blur=4, seed=60 + s) for s in range(S)]
    F = [[[oracle.compute_features(po, im, dims)[1] for im in pair] for pair in seqs[s]] for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    stack = lambda t, c: np.stack([seqs[s][t][c] for s in range(S)])
    methods = [2, 2, 0, 2, 1, 2]
    for t in range(T):
        g.pushBack(stack(t, 0), stack(t, 1), dims, False)
        if t:
            g.matchFeatures(methods[t - 1])
    # replace the newest pair by an older frame and match again, still without a sync
    g.pushBack(stack(2, 0), stack(2, 1), dims, True)
    g.matchFeatures(2)
    for s in range(S):
        want = oracle.matching(po, dims, 2, F[s][T - 2][0], F[s][T - 2][1], F[s][2][0], F[s][2][1])
        assert g.getMatches(s).tobytes() == want.tobytes()
        assert np.array_equal(g.getFeatures(s, pkg.SET_1P), F[s][T - 2][0])
        assert np.array_equal(g.getFeatures(s, pkg.SET_2C), F[s][2][1])
    # and every intermediate step, re-run with a read-back after each match, agrees with the oracle
    g2 = pkg.StreamGroup(S, pkg.Params.default())
    for t in range(T):
        g2.pushBack(stack(t, 0), stack(t, 1), dims, False)
        if t:
            g2.matchFeatures(methods[t - 1])
            for s in range(S):
                want = oracle.matching(po, dims, methods[t - 1], F[s][t - 1][0], F[s][t - 1][1], F[s][t][0], F[s][t][1])
                assert g2.getMatches(s).tobytes() == want.tobytes(), (t, s)
    g.close(); g2.close()


# ------------------------------------------------ full-size configs: exact where the oracle is fast, properties beyond
def test_kitti_stereo_quad_full_size(pkg, ob, oracle, gpu):
    """configs[1]: KITTI 1241x376 stereo quad-match, default parameters."""
    W, H = 1241, 376
    dims = [W, H, 1248]
    po = ob.Params.default()
    seq = pkg.synth.stereo_sequence(W, H, 2, disparity=12)
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    for l, r in seq:
        m.pushBack(l, r, dims, False)
    m.matchFeatures(pkg.METHOD_QUAD)
    got = m.getMatches()
    f = [m.getFeatures(k) for k in range(4)]
    m.close()
    want_f = feats_for(oracle, po, dims, (seq[0][0], seq[0][1], seq[1][0], seq[1][1]))
    for a, b in zip(f, want_f):
        assert np.array_equal(a, b)
    want = oracle.matching(po, dims, 2, *want_f)
    assert len(want) > 5000 and got.tobytes() == want.tobytes()
    ok = (got["u1p"] - got["u2p"] == 12) & (got["u1c"] - got["u2c"] == 12) & (got["u1p"] - got["u1c"] == 5) & (got["v1p"] - got["v1c"] == 1)
    assert ok.mean() > 0.95  # the generator's ground truth


def _check_match_properties(pkg, oracle, po, dims, f, got, method, rng, nsample):
    """Size-independent checks: emission order, sign constraints, circle
    closure re-derived with the oracle's findMatch on a sample of matches."""
    drive = "i1p" if method == 2 else "i1c"
    assert np.all(np.diff(got[drive]) > 0)
    if method == 2:
        assert np.all(got["u1p"] >= got["u2p"]) and np.all(got["u1c"] >= got["u2c"])
    idx = rng.choice(len(got), size=min(nsample, len(got)), replace=False)
    ubn = -(-dims[0] // po.match_binsize); vbn = -(-dims[1] // po.match_binsize)
    index = [oracle.create_index(po, x, dims) for x in f]
    fm = oracle.lib.vo_find_match
    fm.restype = C.c_int32

    def find(a, i, b, flow):
        bs, lst = index[b]
        return fm(C.byref(po), f[a].ctypes.data_as(C.c_void_p), int(i), f[b].ctypes.data_as(C.c_void_p),
                  bs.ctypes.data_as(C.c_void_p), lst.ctypes.data_as(C.c_void_p), ubn, vbn, int(flow), -1.0, -1.0)
    for k in idx:
        r = got[k]
        if method == 2:
            assert find(0, r["i1p"], 1, 0) == r["i2p"] and find(1, r["i2p"], 3, 1) == r["i2c"]
            assert find(3, r["i2c"], 2, 0) == r["i1c"] and find(2, r["i1c"], 0, 1) == r["i1p"]
        else:
            assert find(2, r["i1c"], 0, 1) == r["i1p"] and find(0, r["i1p"], 2, 1) == r["i1c"]
        for tag, s in (("1p", 0), ("2p", 1), ("1c", 2), ("2c", 3)):
            i = r["i" + tag]
            if i >= 0:
                assert r["u" + tag] == f[s][i, 0] and r["v" + tag] == f[s][i, 1]


def test_1080p_all_classes_radius200(pkg, ob, oracle, gpu):
    """configs[2]: 1920x1080 stereo, half_resolution=0, match_radius=200."""
    W, H = 1920, 1080
    dims = [W, H, 1920]
    po = ob.Params.default()
    seq = pkg.synth.stereo_sequence(W, H, 2, disparity=10, seed=2)
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    for l, r in seq:
        m.pushBack(l, r, dims, False)
    m.matchFeatures(pkg.METHOD_QUAD)
    got = m.getMatches()
    f = [m.getFeatures(k) for k in range(4)]
    m.close()
    want_f = feats_for(oracle, po, dims, (seq[0][0], seq[0][1], seq[1][0], seq[1][1]))
    for a, b in zip(f, want_f):
        assert np.array_equal(a, b)
    assert min(len(x) for x in f) > 30000 and len(got) > 20000
    assert set(np.unique(f[2][:, 3])) == {0, 1, 2, 3}
    # the whole p_match list of the quad match, byte for byte (the oracle needs ~1 s for it at this size)
    want = oracle.matching(po, dims, 2, *want_f)
    assert len(want) == len(got) and got.tobytes() == want.tobytes()
    _check_match_properties(pkg, oracle, po, dims, f, got, 2, np.random.default_rng(1), 50)
    # ... and one whole findMatch table
    assert np.array_equal(pkg.match_all(pkg.Params.default(), dims, f[2], f[3], flow=False),
                          oracle.match_all(po, dims, f[2], f[3], flow=False))


def test_4k_dense_small_bins(pkg, ob, oracle, gpu):
    """configs[4]: 3840x2160, nms_n=3, match_binsize=25 (53 592 bins)."""
    W, H = 3840, 2160
    dims = [W, H, 3840]
    over = {"nms_n": 3, "match_binsize": 25}
    p, po = pkg.Params.default(**over), ob.Params.default(**over)
    Ip = pkg.synth.frame(W, H, 0, 0, seed=3); Ic = pkg.synth.frame(W, H, 5, 1, seed=3)
    m = pkg.Matcher(p, outlier_removal=False)
    m.pushBack(Ip, None, dims, False)
    m.pushBack(Ic, None, dims, False)
    m.matchFeatures(pkg.METHOD_FLOW)
    got = m.getMatches()
    fp, fc = m.getFeatures(pkg.SET_1P), m.getFeatures(pkg.SET_1C)
    m.close()
    assert np.array_equal(fp, oracle.compute_features(po, Ip, dims)[1])
    assert np.array_equal(fc, oracle.compute_features(po, Ic, dims)[1])
    assert len(fc) > 100000 and len(got) > 80000
    f = [fp, np.zeros((0, 12), np.int32), fc, np.zeros((0, 12), np.int32)]
    want = oracle.matching(po, dims, 0, fp, f[1], fc, f[3])  # the whole flow p_match list (~4 s of oracle time)
    assert len(want) == len(got) and got.tobytes() == want.tobytes()
    _check_match_properties(pkg, oracle, po, dims, f, got, 0, np.random.default_rng(2), 50)
    assert np.mean((got["u1p"] - got["u1c"] == 5) & (got["v1p"] - got["v1c"] == 1)) > 0.9
    # no pixel of the current image is matched twice (mask M)
    pix = got["v1c"].astype(np.int64) * W + got["u1c"].astype(np.int64)
    assert len(np.unique(pix)) == len(pix)
    bs, lst = pkg.create_index(p, dims, fc)
    bo, lo = oracle.create_index(po, fc, dims)
    assert len(bs) == 4 * 154 * 87 + 1 and np.array_equal(bs, bo) and np.array_equal(lst, lo)


@pytest.mark.gpu
def test_page_locked_host_images_and_staging_slots(pkg, ob, oracle, gpu):
    """Host images from vh_host_alloc memory (one transfer per camera) and from
    pageable memory give the same matches over a run long enough to cycle both
    staging slots and all three ring slots."""
    W, H, S, T = 320, 160, 3, 6
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, T, disparity=5 + s, blur=4, seed=60 + s) for s in range(S)]
    ga = pkg.StreamGroup(S, pkg.Params.default())
    gb = pkg.StreamGroup(S, pkg.Params.default())
    L = pkg.pinned_empty((S, H, dims[2])); R = pkg.pinned_empty((S, H, dims[2]))
    po = ob.Params.default()
    F = [[[oracle.compute_features(po, im, dims)[1] for im in seqs[s][t]] for t in range(T)] for s in range(S)]
    for t in range(T):
        for s in range(S):
            L[s], R[s] = seqs[s][t]
        ga.pushBack(L, R, dims, False)
        L[...] = 0; R[...] = 0  # borrowed for the call only: the engine must not read them afterwards
        gb.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), np.stack([seqs[s][t][1] for s in range(S)]), dims, False)
        if t == 0:
            continue
        ga.matchFeatures(pkg.METHOD_QUAD); gb.matchFeatures(pkg.METHOD_QUAD)
        for s in range(S):
            want = oracle.matching(po, dims, 2, F[s][t - 1][0], F[s][t - 1][1], F[s][t][0], F[s][t][1])
            assert len(want) > 50
            assert ga.getMatches(s).tobytes() == want.tobytes() and gb.getMatches(s).tobytes() == want.tobytes()
    ga.close(); gb.close()


@pytest.mark.gpu
def test_dims_change_starts_a_new_sequence(pkg, ob, oracle, gpu):
    """A pushBack with different dims re-allocates and empties the ring buffer
    (include/viso_hip.h); the handle then behaves like a fresh one at the new size."""
    po = ob.Params.default()
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    for (W, H, seed) in ((320, 160, 5), (402, 131, 6), (320, 160, 7)):
        dims = [W, H, pkg.synth.bytes_per_line(W)]
        seq = pkg.synth.stereo_sequence(W, H, 2, disparity=6, blur=4, seed=seed)
        F = [[oracle.compute_features(po, im, dims)[1] for im in pair] for pair in seq]
        m.pushBack(seq[0][0], seq[0][1], dims, False)
        m.matchFeatures(pkg.METHOD_QUAD)
        assert len(m.getMatches()) == 0  # nothing carried over from the previous size
        assert np.array_equal(m.getFeatures(pkg.SET_1C), F[0][0])
        m.pushBack(seq[1][0], seq[1][1], dims, False)
        m.matchFeatures(pkg.METHOD_QUAD)
        want = oracle.matching(po, dims, 2, F[0][0], F[0][1], F[1][0], F[1][1])
        assert len(want) > 50 and m.getMatches().tobytes() == want.tobytes()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_flow_search_wide_key_path(mode, gpu):
    """The searches build (SAD << 16 | class-relative position) keys on v_sad_hi_u8 when a class
    holds fewer than 2^16 - 64 candidates, (SAD << 19 | position) up to 2^19 - 64 and 64-bit keys
    beyond.  The second and third encodings are forced here (VH_FLOW_WIDE_KEYS=1 / 2, read once
    per process, hence the subprocess) and must pass the same parity cases."""
    import subprocess, sys
    env = dict(os.environ, VH_FLOW_WIDE_KEYS=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "golden or random_configs or tie_break or ring_buffer"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_flow_search_tested_loop_path(gpu):
    """VH_FLOW_TESTED=1 selects the flow loop with the per-pair accept tests (three test classes
    per bin) instead of the speculative one: same results."""
    import subprocess, sys
    env = dict(os.environ, VH_FLOW_TESTED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "golden or random_configs or tie_break or ring_buffer or kitti"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_search_loop_policy_follows_the_data(pkg, ob, oracle, gpu, monkeypatch):
    """engine.hip: choose_loop.  Frames whose features have partners keep the speculative search loops
    (few winners fall outside their window); independent noise images -- no feature has a partner -- push
    the observed share of second searches far over the 6.5 % threshold and the tested loops take over;
    partnered frames bring the speculative ones back.  The match lists equal the oracle's in every phase."""
    monkeypatch.delenv("VH_FLOW_TESTED", raising=False)
    W, H, S = 512, 192, 3
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    p, po = pkg.Params.default(), ob.Params.default()
    rng = np.random.default_rng(5)
    clean = [[pkg.synth.frame(W, H, 2 * (t % 8), t % 4, 4, 6, 900 + s) for s in range(S)] for t in range(24)]
    noisy = [[rng.integers(0, 256, (H, dims[2]), dtype=np.uint8) for s in range(S)] for t in range(8)]

    def run(g, frames, check_every):
        prev = None
        for t, fr in enumerate(frames):
            g.pushBack(np.stack(fr), None, dims, False)
            if prev is not None or t > 0:
                g.matchFeatures(pkg.METHOD_FLOW)
                if t % check_every == 0:
                    for s in range(S):
                        fp, fc = feats_for(oracle, po, dims, [prev[s], fr[s]])
                        assert g.getMatches(s).tobytes() == oracle.matching(po, dims, 0, m1p=fp, m1c=fc).tobytes()
                else:
                    g.getMatches(0)  # completes the step, so that its statistics are visible to the next one
            prev = fr

    g = pkg.StreamGroup(S, p)
    run(g, clean[:6], 3)
    spec, rate = g.searchStats()
    assert spec and 0 <= rate < 0.055, (spec, rate)
    run(g, noisy, 4)
    spec, rate = g.searchStats()
    assert not spec and rate > 0.065, (spec, rate)
    run(g, clean[6:], 6)  # the 16th launch probes the speculative form and finds it cheap again
    spec, rate = g.searchStats()
    assert spec and rate < 0.055, (spec, rate)
    g.close()


@pytest.mark.gpu
def test_index_invariants_on_the_checking_build(pkg, gpu):
    """libviso_hip_check.so (-DVH_CHECK) verifies on the device every index the shipped kernels
    use unclamped -- row index -> bin position -> record, stage slots, winner positions
    (csrc/vh_dev.h: VH_CHECK_RANGE) -- and aborts the process on the first violation.  The parity
    cases that exercise those indices, the truncated-set case (features > capacity) among them,
    must pass on it: same results, no violation.  Every other test of this file runs on it (round 3: a subset)."""
    import subprocess, sys
    assert os.path.exists(pkg.CHECK_LIB_PATH), "build() makes it"
    env = dict(os.environ, VISO_HIP_LIB=pkg.CHECK_LIB_PATH)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "not checking_build"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "VH_CHECK" not in r.stderr


@pytest.mark.gpu
def test_quad_matching_with_motion_prior(pkg, ob, oracle, gpu):
    """vh_match_features with a Tr_delta.  The reference tree ignores the argument; the part it pins is
    findMatch's prediction term (test_find_match_prior_term).  The composition -- stock libviso2 predicts the position
    in the current right image from the (1p, 2p) pair and Tr_delta and searches hop 2 of the quad circle around it --
    is [upstream-recollection], restated in oracle/viso_oracle.c: vo_matching_quad_prior; the GPU is bit-exact against
    that, for a single matcher and for a group with a different motion per stream.  Without a Tr_delta nothing changes.
    PARITY UNPINNED: this is an oracle-vs-device test, not a reference fixture -- the reference has no code to generate one."""
    W, H, S = 480, 200, 2
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    K = dict(f=400.0, cu=W / 2.0, cv=H / 2.0, base=0.5)
    seqs = [pkg.synth.stereo_sequence(W, H, 2, disparity=7 + s, blur=4, seed=500 + s) for s in range(S)]
    po = ob.Params.default(**K)
    F = [[[oracle.compute_features(po, im, dims)[1] for im in seqs[s][t]] for t in range(2)] for s in range(S)]

    def motion(tz, ry, tx):
        T = np.eye(4)
        T[0, 0] = np.cos(ry); T[0, 2] = np.sin(ry); T[2, 0] = -np.sin(ry); T[2, 2] = np.cos(ry)
        T[0, 3] = tx; T[2, 3] = tz
        return T
    trs = [motion(-0.4, 0.01, 0.05), motion(1.5, -0.08, -0.6)]  # a plausible step forward; a prediction far off the truth
    plain = [oracle.matching(po, dims, 2, F[s][0][0], F[s][0][1], F[s][1][0], F[s][1][1]) for s in range(S)]
    want = [[oracle.matching_quad_prior(po, dims, tr, F[s][0][0], F[s][0][1], F[s][1][0], F[s][1][1]) for tr in trs] for s in range(S)]
    assert any(want[s][k].tobytes() != plain[s].tobytes() for s in range(S) for k in range(2))  # the prior changes something
    assert all(len(want[s][k]) > 50 for s in range(S) for k in range(2))
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    for t in range(2):
        m.pushBack(seqs[0][t][0], seqs[0][t][1], dims, False)
    with pytest.raises(pkg.VisoHipError) as ex:
        m.matchFeatures(pkg.METHOD_QUAD, Tr_delta=trs[0])  # no intrinsics yet
    assert ex.value.code == pkg.VH_ERR_STATE
    m.setIntrinsics(K["f"], K["cu"], K["cv"], K["base"])
    for k in (0, 1, 0):
        m.matchFeatures(pkg.METHOD_QUAD, Tr_delta=trs[k])
        assert m.getMatches().tobytes() == want[0][k].tobytes(), k
        m.matchFeatures(pkg.METHOD_QUAD)  # and without: the plain circle
        assert m.getMatches().tobytes() == plain[0].tobytes()
    m.matchFeatures(pkg.METHOD_FLOW, Tr_delta=trs[1])  # other methods take no prior
    assert m.getMatches().tobytes() == oracle.matching(po, dims, 0, m1p=F[0][0][0], m1c=F[0][1][0]).tobytes()
    m.close()
    g = pkg.StreamGroup(S, pkg.Params.default(**K))
    for t in range(2):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), np.stack([seqs[s][t][1] for s in range(S)]), dims, False)
    g.matchFeaturesPrior(pkg.METHOD_QUAD, np.stack([trs[1], trs[0]]))
    assert g.getMatches(0).tobytes() == want[0][1].tobytes() and g.getMatches(1).tobytes() == want[1][0].tobytes()
    g.close()


@pytest.mark.gpu
def test_failed_mask_allocation_leaves_the_group_usable(pkg, ob, oracle, gpu):
    """The flow method allocates its pixel mask on first use.  If that allocation fails, the call fails BEFORE anything
    of the step is queued -- the table buffers, the emission's chunk counters and the re-search counters keep their
    state -- and the following matches (quad, then flow once memory is there) equal the oracle's."""
    W, H, S = 320, 160, 2
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, 3, disparity=5 + s, blur=4, seed=90 + s) for s in range(S)]
    po = ob.Params.default()
    F = [[[oracle.compute_features(po, im, dims)[1] for im in seqs[s][t]] for t in range(3)] for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    for t in range(2):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), np.stack([seqs[s][t][1] for s in range(S)]), dims, False)
    g.matchFeatures(pkg.METHOD_QUAD)  # one ordinary launch first: the counters are in use
    g.debugFailNextAlloc()
    with pytest.raises(pkg.VisoHipError) as ex:
        g.matchFeatures(pkg.METHOD_FLOW)
    assert ex.value.code == pkg.VH_ERR_HIP
    for rep in range(3):  # both table buffers come round
        g.matchFeatures(pkg.METHOD_QUAD)
        for s in range(S):
            assert g.getMatches(s).tobytes() == oracle.matching(po, dims, 2, F[s][0][0], F[s][0][1], F[s][1][0], F[s][1][1]).tobytes()
    g.matchFeatures(pkg.METHOD_FLOW)
    for s in range(S):
        assert g.getMatches(s).tobytes() == oracle.matching(po, dims, 0, m1p=F[s][0][0], m1c=F[s][1][0]).tobytes()
    g.close()


@pytest.mark.gpu
def test_group_get_matches_all(pkg, ob, oracle, gpu):
    """vh_group_get_matches_all == per-stream vh_group_get_matches, also after the
    host-side outlier vote and into a page-locked buffer."""
    W, H, S = 320, 160, 4
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, 2, disparity=4 + s, blur=4, seed=80 + s) for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    out, counts = g.getMatchesAll(cap_per_stream=8)
    assert list(counts) == [0] * S  # nothing pushed yet
    for t in range(2):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), np.stack([seqs[s][t][1] for s in range(S)]), dims, False)
    g.matchFeatures(pkg.METHOD_QUAD)
    per = [g.getMatches(s) for s in range(S)]
    buf = pkg.pinned_empty((S, max(len(p) for p in per) + 3), pkg.P_MATCH_DTYPE)
    out, counts = g.getMatchesAll(out=buf)
    for s in range(S):
        assert counts[s] == len(per[s]) > 50 and out[s, :counts[s]].tobytes() == per[s].tobytes()
    with pytest.raises(pkg.VisoHipError) as e:  # too small: true counts still reported through the exception path
        g.getMatchesAll(cap_per_stream=10)
    assert e.value.code == pkg.VH_ERR_CAPACITY
    g.removeOutliers(2)
    out, counts = g.getMatchesAll(out=buf)
    for s in range(S):
        want = g.getMatches(s)
        assert counts[s] == len(want) and out[s, :counts[s]].tobytes() == want.tobytes()
    g.close()


@pytest.mark.gpu
def test_group_async_download(pkg, ob, oracle, gpu):
    """vh_group_download_matches_async: the lists of step t land in page-locked
    memory while step t+1 is already issued, and equal the synchronous result."""
    W, H, S, T = 320, 160, 3, 5
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, T, disparity=4 + s, blur=4, seed=90 + s) for s in range(S)]
    po = ob.Params.default()
    F = [[[oracle.compute_features(po, im, dims)[1] for im in seqs[s][t]] for t in range(T)] for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    bufs = [pkg.pinned_empty((S, 4096), pkg.P_MATCH_DTYPE) for _ in range(2)]
    cnts = [pkg.pinned_empty((S,), np.int32) for _ in range(2)]
    with pytest.raises(pkg.VisoHipError):
        g.downloadMatchesAsync(bufs[0], cnts[0])  # nothing matched yet
    pending = None
    for t in range(T):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), np.stack([seqs[s][t][1] for s in range(S)]), dims, False)
        if t == 0:
            continue
        g.matchFeatures(pkg.METHOD_QUAD)
        g.waitDownload()
        if pending is not None:
            tt, b = pending
            for s in range(S):
                want = oracle.matching(po, dims, 2, F[s][tt - 1][0], F[s][tt - 1][1], F[s][tt][0], F[s][tt][1])
                assert cnts[b][s] == len(want) > 50 and bufs[b][s, :len(want)].tobytes() == want.tobytes()
        g.downloadMatchesAsync(bufs[t % 2], cnts[t % 2])
        pending = (t, t % 2)
    g.waitDownload()
    tt, b = pending
    for s in range(S):
        want = oracle.matching(po, dims, 2, F[s][tt - 1][0], F[s][tt - 1][1], F[s][tt][0], F[s][tt][1])
        assert cnts[b][s] == len(want) and bufs[b][s, :len(want)].tobytes() == want.tobytes()
    g.close()


@pytest.mark.gpu
def test_row_pitch_beyond_the_envelope_is_refused(pkg, gpu):
    """emit_features addresses patch rows with 32-bit byte offsets (strides < 2^24, images <= 2^28
    bytes): a larger pitch is VH_ERR_UNSUPPORTED, not silently truncated offsets."""
    m = pkg.Matcher(pkg.Params.default(), outlier_removal=False)
    img = np.zeros((8, 64), np.uint8)
    for dims in ([64, 8, 1 << 24], [64, 8192, 1 << 16]):
        d = (C.c_int32 * 3)(*dims)
        rc = pkg._lib().vh_push_back(m._h, img.ctypes.data_as(C.c_void_p), None, d, 0)  # refused before any byte is read
        assert rc == pkg.VH_ERR_UNSUPPORTED, dims
    m.close()


# ------------------------------------------------ round 2: the bench's own entry points, stream ordering, overflow
@pytest.mark.gpu
@pytest.mark.parametrize("producer", ["side", "default"])
def test_group_push_back_device_orders_after_producer_stream(producer, gpu):
    """vh_group_push_back_device + vh_group_set_stream, the path bench.py times, driven from
    torch tensors produced on a torch stream: tests/stream_order_case.py (a process of its own,
    because torch must load its HIP runtime before libviso_hip.so is mapped -- as in bench.py --
    for the two to share one runtime and hence stream handles)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stream_order_case.py"), producer],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "stream-order ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_feature_capacity_overflow_is_reported_on_the_match_path(pkg, ob, oracle, gpu):
    """max_features below the feature count: the sets are truncated, and every way of
    fetching the matches says VH_ERR_CAPACITY instead of silently returning a list
    matched on truncated sets."""
    W, H, S = 320, 160, 2
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seq = pkg.synth.stereo_sequence(W, H, 2, disparity=5, blur=4, seed=7)
    n_full = len(oracle.compute_features(ob.Params.default(), seq[1][0], dims)[1])
    assert n_full > 400
    m = pkg.Matcher(pkg.Params.default(), max_features=256, max_matches=4096, outlier_removal=False)
    for l, r in seq:
        m.pushBack(l, r, dims, False)
    for method in (0, 1, 2):
        m.matchFeatures(method)
        n = C.c_int32(0)
        buf = np.zeros(4096, pkg.P_MATCH_DTYPE)
        rc = pkg._lib().vh_get_matches(m._h, buf.ctypes.data_as(C.c_void_p), 4096, C.byref(n))
        assert rc == pkg.VH_ERR_CAPACITY and 0 <= n.value <= 256
        with pytest.raises(pkg.VisoHipError) as e:
            m.getMatches()
        assert e.value.code == pkg.VH_ERR_CAPACITY
    m.close()
    g = pkg.StreamGroup(S, pkg.Params.default(), max_features=256, max_matches=4096)
    for l, r in seq:
        g.pushBack(np.stack([l] * S), np.stack([r] * S), dims, False)
    g.matchFeatures(pkg.METHOD_QUAD)
    with pytest.raises(pkg.VisoHipError) as e:
        g.getMatchesAll(cap_per_stream=4096)
    assert e.value.code == pkg.VH_ERR_CAPACITY
    out = pkg.pinned_empty((S, 4096), pkg.P_MATCH_DTYPE); cnt = pkg.pinned_empty((S,), np.int32)
    g.downloadMatchesAsync(out, cnt)
    with pytest.raises(pkg.VisoHipError) as e:
        g.waitDownload()
    assert e.value.code == pkg.VH_ERR_CAPACITY
    g.close()
    # a capacity that fits reports nothing
    m = pkg.Matcher(pkg.Params.default(), max_features=n_full + 64, max_matches=4096, outlier_removal=False)
    for l, r in seq:
        m.pushBack(l, r, dims, False)
    m.matchFeatures(2)
    assert len(m.getMatches()) > 50
    m.close()


@pytest.mark.gpu
def test_4k_stereo_quad_small_bins(pkg, ob, oracle, gpu):
    """configs[4] as BASELINE states it: a 3840x2160 STEREO pair, nms_n=3, match_binsize=25,
    quad match.  Features exact vs the oracle on all four images, both stereo-pass tables
    exact (vh_match_all(flow=False): 1p->2p and 2c->1c), the flow passes and the chain through
    the size-independent properties (circle closure re-derived with the oracle's findMatch)."""
    W, H = 3840, 2160
    dims = [W, H, 3840]
    over = {"nms_n": 3, "match_binsize": 25}
    p, po = pkg.Params.default(**over), ob.Params.default(**over)
    seq = pkg.synth.stereo_sequence(W, H, 2, disparity=14, seed=5)
    m = pkg.Matcher(p, outlier_removal=False)
    for l, r in seq:
        m.pushBack(l, r, dims, False)
    m.matchFeatures(pkg.METHOD_QUAD)
    got = m.getMatches()
    f = [m.getFeatures(k) for k in range(4)]
    m.close()
    want_f = feats_for(oracle, po, dims, (seq[0][0], seq[0][1], seq[1][0], seq[1][1]))
    for a, b in zip(f, want_f):
        assert np.array_equal(a, b)
    assert min(len(x) for x in f) > 100000 and len(got) > 80000
    assert np.array_equal(pkg.match_all(p, dims, f[0], f[1], flow=False), oracle.match_all(po, dims, f[0], f[1], flow=False))
    assert np.array_equal(pkg.match_all(p, dims, f[3], f[2], flow=False), oracle.match_all(po, dims, f[3], f[2], flow=False))
    want = oracle.matching(po, dims, 2, *want_f)  # the whole quad p_match list, byte for byte (~3 s of oracle time)
    assert len(want) == len(got) and got.tobytes() == want.tobytes()
    _check_match_properties(pkg, oracle, po, dims, f, got, 2, np.random.default_rng(4), 50)
    ok = (got["u1p"] - got["u2p"] == 14) & (got["u1c"] - got["u2c"] == 14)
    assert ok.mean() > 0.9


@pytest.mark.gpu
def test_4k_dense_maxima_beyond_the_old_envelope(pkg, ob, oracle, gpu):
    """configs[4], "dense maxima": a 3840x2160 frame pair of the blur-1 / gain-4 texture at
    nms_n = 2 -- more than 2^19 features per image (round 1 refused this with VH_ERR_CAPACITY).
    Features exact on both frames, the stereo-type table and a sample of the flow table exact
    vs the oracle, flow matches through the size-independent properties."""
    W, H = 3840, 2160
    dims = [W, H, 3840]
    over = {"nms_n": 2, "match_binsize": 25, "match_radius": 60}
    p, po = pkg.Params.default(**over), ob.Params.default(**over)
    Ip = pkg.synth.frame(W, H, 0, 0, 1, 4, 9); Ic = pkg.synth.frame(W, H, 3, 1, 1, 4, 9)
    m = pkg.Matcher(p, outlier_removal=False)
    m.pushBack(Ip, None, dims, False)
    m.pushBack(Ic, None, dims, False)
    m.matchFeatures(pkg.METHOD_FLOW)
    got = m.getMatches()
    fp, fc = m.getFeatures(pkg.SET_1P), m.getFeatures(pkg.SET_1C)
    m.close()
    assert len(fc) > (1 << 19)
    assert np.array_equal(fp, oracle.compute_features(po, Ip, dims)[1])
    assert np.array_equal(fc, oracle.compute_features(po, Ic, dims)[1])
    assert np.array_equal(pkg.match_all(p, dims, fc, fp, flow=False), oracle.match_all(po, dims, fc, fp, flow=False))
    f = [fp, np.zeros((0, 12), np.int32), fc, np.zeros((0, 12), np.int32)]
    want = oracle.matching(po, dims, 0, fp, f[1], fc, f[3])  # > 10^6 flow matches, byte for byte (~12 s of oracle time)
    assert len(want) == len(got) and got.tobytes() == want.tobytes()
    _check_match_properties(pkg, oracle, po, dims, f, got, 0, np.random.default_rng(8), 50)
    assert len(got) > 300000
    pix = got["v1c"].astype(np.int64) * W + got["u1c"].astype(np.int64)
    assert len(np.unique(pix)) == len(pix)
