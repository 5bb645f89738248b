"""CPU suite, part 1: the oracle (oracle/viso_oracle.c) is pinned against
 (a) the golden vectors generated from the reference's own code
     (oracle/gen_golden.py -> tests/golden/),
 (b) the SURVEY Appendix-B known answers, and
 (c) where oracle/_ref is built, the reference itself on fresh inputs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden(name, pkg, ob, oracle):
    p, dims, Ip, Ic, z = load_golden(name, pkg, ob.Params)
    m1p, m2p = oracle.compute_features(p, Ip, dims)
    m1c, m2c, du, dv = oracle.compute_features(p, Ic, dims, planes=True)
    assert np.array_equal(m2p, z["max2p"])
    assert np.array_equal(m2c, z["max2c"])
    assert np.array_equal(m1p, z["max1p"])
    assert np.array_equal(m1c, z["max1c"])
    assert oracle.fnv(np.ascontiguousarray(du[2:-2, 2:du.shape[1] - 16])) == int(z["du_interior_fnv"])
    assert oracle.fnv(np.ascontiguousarray(dv[2:-2, 2:dv.shape[1] - 16])) == int(z["dv_interior_fnv"])
    bs, lst = oracle.create_index(p, m2c, dims)
    assert np.array_equal(bs, z["bin_start"]) and np.array_equal(lst, z["bin_list"])
    assert np.array_equal(oracle.match_all(p, dims, m2c, m2p, flow=True), z["fwd"])
    pm = oracle.matching(p, dims, 0, m1p=m2p, m1c=m2c)
    assert pm.tobytes() == z["p_match"].tobytes()


@pytest.mark.parametrize("case", ["kitti_1241x376", "seq_1024x284"])
def test_oracle_known_answers(case, pkg, ob, oracle):
    """Counts and FNV-1a-64 hashes at the SURVEY Appendix-B sizes."""
    z = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    W, H, blur, gain, seed, dx, dy = [int(v) for v in z[case + "__gen"]]
    p = ob.Params.default()
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    Ip = pkg.synth.frame(W, H, 0, 0, blur, gain, seed)
    Ic = pkg.synth.frame(W, H, dx, dy, blur, gain, seed)
    _, m2p = oracle.compute_features(p, Ip, dims)
    _, m2c = oracle.compute_features(p, Ic, dims)
    assert len(m2p) == int(z[case + "__n2p"]) and len(m2c) == int(z[case + "__n2c"])
    assert oracle.fnv(m2p) == int(z[case + "__fnv_max2p"])
    assert oracle.fnv(m2c) == int(z[case + "__fnv_max2c"])
    assert np.array_equal(np.bincount(m2c[:, 3], minlength=4), z[case + "__class_hist"])
    pm = oracle.matching(p, dims, 0, m1p=m2p, m1c=m2c)
    assert len(pm) == int(z[case + "__n_match"])
    assert oracle.fnv(pm) == int(z[case + "__fnv_p_match"])
    assert pm[:32].tobytes() == z[case + "__head_p_match"].tobytes()


def test_appendix_b_literals(pkg, ob, oracle):
    """The hashes SURVEY.md Appendix B quotes, spelled out."""
    Ip = pkg.synth.frame(1241, 376, 0, 0)
    Ic = pkg.synth.frame(1241, 376, 5, 1)
    assert oracle.fnv(Ip) == 0xA43EA807A194B4DE and oracle.fnv(Ic) == 0xDD3B764F6C9922A1
    p = ob.Params.default()
    _, m2p = oracle.compute_features(p, Ip, [1241, 376, 1248])
    _, m2c = oracle.compute_features(p, Ic, [1241, 376, 1248])
    assert (len(m2p), len(m2c)) == (9514, 9526)
    assert oracle.fnv(m2p) == 0x10E2361988E416D6 and oracle.fnv(m2c) == 0xFC8DB1A05468A332
    pm = oracle.matching(p, [1241, 376, 1248], 0, m1p=m2p, m1c=m2c)
    assert len(pm) == 9092 and oracle.fnv(pm) == 0x8FCED7460901C09D
    true_shift = int(np.sum((pm["u1p"] - pm["u1c"] == 5) & (pm["v1p"] - pm["v1c"] == 1)))
    assert true_shift == 9069


def test_find_match_prior_term(pkg, ob, oracle):
    """findMatch's optional 4*||(u2,v2)-(u_,v_)|| term (src/matcher.cpp:257-262)."""
    p, dims, Ip, Ic, z = load_golden("small_default", pkg, ob.Params)
    g = np.load(os.path.join(GOLDEN, "find_match_prior.npz"))
    got = oracle.match_all_prior(p, dims, z["max2c"], z["max2p"], float(g["u_"]), float(g["v_"]))
    assert np.array_equal(got, g["best"])


def test_bucket_features_golden(pkg, ob, oracle):
    """bucketFeatures + LFSR shuffle (src/matcher.cpp:113-187)."""
    g = np.load(os.path.join(GOLDEN, "bucket_1024x284.npz"))
    p = ob.Params.default()
    dims = [1024, 284, 1024]
    _, m2p = oracle.compute_features(p, pkg.synth.frame(1024, 284, 0, 0), dims)
    _, m2c = oracle.compute_features(p, pkg.synth.frame(1024, 284, 5, 1), dims)
    pm = oracle.matching(p, dims, 0, m1p=m2p, m1c=m2c)
    for key in g.files:
        _, mf, bw, bh = key.split("_")
        out = oracle.bucket_features(pm, int(mf), float(bw), float(bh))
        assert out.tobytes() == g[key].tobytes(), key


def test_oracle_vs_reference_fresh_inputs(pkg, ob, oracle, reference):
    """Where the reference build exists: compare on inputs no fixture holds."""
    rng = np.random.default_rng(7)
    for trial in range(4):
        W = int(rng.integers(150, 420)); H = int(rng.integers(100, 260))
        over = {"nms_n": int(rng.integers(1, 5)), "nms_tau": int(rng.integers(10, 80)),
                "match_binsize": int(rng.integers(15, 90)), "match_radius": int(rng.integers(20, 250)),
                "half_resolution": int(trial % 2), "multi_stage": int(trial // 2)}
        p = ob.Params.default(**over)
        dims = [W, H, pkg.synth.bytes_per_line(W)]
        blur, gain, seed = int(rng.integers(2, 7)), int(rng.integers(1, 3)), int(rng.integers(1, 1000))
        Ip = pkg.synth.frame(W, H, 0, 0, blur, gain, seed)
        Ic = pkg.synth.frame(W, H, int(rng.integers(0, 6)), int(rng.integers(0, 4)), blur, gain, seed)
        a = oracle.compute_features(p, Ip, dims); b = reference.compute_features(p, Ip, dims)
        c = oracle.compute_features(p, Ic, dims); d = reference.compute_features(p, Ic, dims)
        for x, y in zip(a + c, b + d):
            assert np.array_equal(x, y), over
        assert len(a[1]) > 50
        assert oracle.matching(p, dims, 0, m1p=a[1], m1c=c[1]).tobytes() == \
            reference.matching_flow(p, dims, b[1], d[1]).tobytes(), over
        fo = oracle.filters(Ic); fr = reference.filters(Ic)
        for x, y in zip(fo, fr):  # valid interior only
            assert np.array_equal(x[3:H - 3, 3:W - 3], y[3:H - 3, 3:W - 3])


def test_stereo_quad_spec_properties(pkg, ob, oracle):
    """Stereo/quad compositions are NOT in the reference (parity unpinned):
    check the properties their definition implies."""
    W, H = 320, 160
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    p = ob.Params.default()
    seq = pkg.synth.stereo_sequence(W, H, 2, disparity=6, blur=4)
    f = [oracle.compute_features(p, img, dims)[1] for img in (seq[0][0], seq[0][1], seq[1][0], seq[1][1])]
    st = oracle.matching(p, dims, 1, m1c=f[2], m2c=f[3])
    assert len(st) > 100 and np.all(st["u1c"] >= st["u2c"]) and np.all(st["i1p"] == -1)
    assert np.all(np.abs(st["v1c"] - st["v2c"]) <= p.match_disp_tolerance)
    assert np.all(np.diff(st["i1c"]) > 0)
    assert np.mean(st["u1c"] - st["u2c"] == 6) > 0.9  # the generator's true disparity
    fwd = oracle.match_all(p, dims, f[2], f[3], flow=False)
    bwd = oracle.match_all(p, dims, f[3], f[2], flow=False)
    closed = np.nonzero(bwd[fwd] == np.arange(len(fwd)))[0]
    closed = closed[f[2][closed, 0] >= f[3][fwd[closed], 0]]
    assert np.array_equal(st["i1c"], closed)
    qd = oracle.matching(p, dims, 2, *f)
    assert len(qd) > 100 and np.all(np.diff(qd["i1p"]) > 0)
    assert np.all(qd["u1p"] >= qd["u2p"]) and np.all(qd["u1c"] >= qd["u2c"])
