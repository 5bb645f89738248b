"""stereo egomotion (VisualOdometryStereo::estimateMotion, src/viso_stereo.cpp:54-157).
CPU part: the plain-C restatement (oracle/viso_egomotion.c) is pinned bit for bit to the reference's
own code compiled in oracle/_ref, samples drawn as the reference draws them (rand() after srand(0))."""
import numpy as np
import pytest

from egomotion_scene import scene

CASES = [(400, 1, 0.25, 0.0), (60, 2, 0.4, 0.3), (1500, 3, 0.1, 0.5), (6, 4, 0.0, 0.0), (9, 5, 0.5, 0.0), (250, 6, 0.7, 0.2)]


def ego_params(ob, **kw):
    return ob.EgoParams.default(f=645.24, cu=635.96, cv=194.13, base=0.5707, **kw)


@pytest.mark.parametrize("n,seed,outliers,noise", CASES)
def test_oracle_equals_reference_estimate_motion(n, seed, outliers, noise, ob, oracle, reference):
    pm, tr_true = scene(ob.P_MATCH_DTYPE, n, seed, outliers=outliers, noise=noise)
    for kw in ({}, {"reweighting": 0}, {"ransac_iters": 50, "inlier_threshold": 1.0}):
        e = ego_params(ob, **kw)
        samples = oracle.draw_samples(len(pm), e.ransac_iters)
        ok_o, tr_o, inl_o = oracle.estimate_motion_stereo(e, pm, samples)
        ok_r, tr_r, inl_r = reference.estimate_motion_stereo(e, pm)
        assert ok_o == ok_r and np.array_equal(inl_o, inl_r), (n, seed, kw)
        assert tr_o.tobytes() == tr_r.tobytes(), (tr_o, tr_r)
        if ok_o and outliers <= 0.25 and n >= 400:
            assert np.allclose(tr_o, tr_true, atol=0.05), (tr_o, tr_true)  # the estimate is the scene's motion


def test_too_few_matches_and_degenerate(ob, oracle, reference):
    e = ego_params(ob)
    pm, _ = scene(ob.P_MATCH_DTYPE, 5, 9)
    assert oracle.estimate_motion_stereo(e, pm, oracle.draw_samples(5, 200))[0] is False
    assert reference.estimate_motion_stereo(e, pm)[0] is False
    pm, _ = scene(ob.P_MATCH_DTYPE, 30, 10)
    pm[:] = pm[0]  # all matches identical: singular normal equations in every hypothesis
    ok_o, tr_o, inl_o = oracle.estimate_motion_stereo(e, pm, oracle.draw_samples(30, 200))
    ok_r, tr_r, inl_r = reference.estimate_motion_stereo(e, pm)
    assert ok_o == ok_r and np.array_equal(inl_o, inl_r) and tr_o.tobytes() == tr_r.tobytes()


def _golden():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "egomotion.npz"))


def _golden_case(z, name, ob):
    pm = np.ascontiguousarray(z[name + "__pm"]).view(ob.P_MATCH_DTYPE).reshape(-1)
    g = z[name + "__ego"]
    e = ob.EgoParams(ransac_iters=int(g[0]), reweighting=int(g[1]), inlier_threshold=g[2], f=g[3], cu=g[4], cv=g[5], base=g[6])
    return pm, e, z[name + "__samples"], bool(z[name + "__ok"]), z[name + "__tr"], z[name + "__inliers"]


GOLDEN_CASES = ["s400", "s60_noisy", "s1500", "s250_hard", "s400_plain", "s300_tight", "s9"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_estimate_motion_golden(name, ob, oracle):
    """The restatement against vectors the reference's own estimateMotion produced (oracle/gen_golden_ego.py):
    what pins it on the GPU box, where the reference build is absent."""
    pm, e, samples, ok, tr, inl = _golden_case(_golden(), name, ob)
    ok_o, tr_o, inl_o = oracle.estimate_motion_stereo(e, pm, samples)
    assert ok_o == ok and np.array_equal(inl_o, inl) and tr_o.tobytes() == tr.tobytes()


def _rand3(ob, ego, n_sets):
    """rand() after srand(0), one fresh sequence per list (as one fresh VisualOdometryStereo per list would draw)."""
    r = ob.glibc_rand_after_srand0(3 * ego.ransac_iters).reshape(ego.ransac_iters, 3)
    return np.stack([r] * n_sets)


def _close(tr, want):
    return np.allclose(tr, want, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_gpu_estimate_motion_golden_batch(pkg, ob, oracle, gpu):
    """vh_estimate_motion_stereo, all golden scenes in ONE batched launch: inlier sets exact, tr within
    1e-9 relative of the reference's (the refinement sums its normal equations in parallel)."""
    z = _golden()
    cases = [_golden_case(z, n, ob) for n in GOLDEN_CASES]
    for sel in ([0, 1, 2, 3, 6], [4], [5]):  # one parameter set per launch
        e = cases[sel[0]][1]
        ge = pkg.EgoParams(ransac_iters=e.ransac_iters, reweighting=e.reweighting, inlier_threshold=e.inlier_threshold, f=e.f, cu=e.cu, cv=e.cv, base=e.base)
        tr, ok, inl = pkg.estimate_motion_stereo(ge, [cases[i][0] for i in sel], _rand3(ob, e, len(sel)))
        for k, i in enumerate(sel):
            pm, _, samples, ok_w, tr_w, inl_w = cases[i]
            assert np.array_equal(oracle.draw_samples(len(pm), e.ransac_iters), samples)  # the kernel draws from the same rand() values
            assert ok[k] == ok_w and np.array_equal(inl[k], inl_w), GOLDEN_CASES[i]
            assert _close(tr[k], tr_w), (GOLDEN_CASES[i], tr[k], tr_w)


@pytest.mark.gpu
def test_gpu_estimate_motion_random_scenes_vs_oracle(pkg, ob, oracle, gpu):
    rng = np.random.default_rng(5)
    e = ego_params(ob, ransac_iters=300)
    ge = pkg.EgoParams(ransac_iters=300, reweighting=1, inlier_threshold=2.0, f=e.f, cu=e.cu, cv=e.cv, base=e.base)
    lists = []
    for s in range(24):
        n = int(rng.integers(3, 900))
        trs = (rng.normal(0, 0.01), rng.normal(0, 0.02), rng.normal(0, 0.005), rng.normal(0, 0.05), rng.normal(0, 0.02), -abs(rng.normal(0.8, 0.4)))
        lists.append(scene(ob.P_MATCH_DTYPE, n, 100 + s, tr=trs, outliers=float(rng.uniform(0, 0.6)), noise=float(rng.uniform(0, 0.6)))[0])
    raw = rng.integers(0, 2 ** 31 - 1, (len(lists), 300, 3)).astype(np.int32)  # any rand()-like stream
    tr, ok, inl = pkg.estimate_motion_stereo(ge, lists, raw)
    n_ok = 0
    for s, pm in enumerate(lists):
        ok_o, tr_o, inl_o = oracle.estimate_motion_stereo(e, pm, oracle.draw_samples(len(pm), 300, raw[s].reshape(-1)))
        assert ok[s] == ok_o and np.array_equal(inl[s], inl_o), s
        assert _close(tr[s], tr_o), (s, tr[s], tr_o)
        n_ok += ok_o
    assert n_ok >= 12


@pytest.mark.gpu
def test_gpu_group_estimate_motion_on_device_matches(pkg, ob, oracle, gpu):
    """vh_group_estimate_motion: RANSAC straight on the device-resident quad match lists of a stream group."""
    S, W, H = 3, 480, 200
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, 2, disparity=6 + s, blur=4, seed=200 + s) for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    for t in range(2):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), np.stack([seqs[s][t][1] for s in range(S)]), dims, False)
    g.matchFeatures(pkg.METHOD_FLOW)
    ge = pkg.EgoParams.default(f=400.0, cu=W / 2, cv=H / 2, base=0.5)
    raw = np.random.default_rng(1).integers(0, 2 ** 31 - 1, (S, 200, 3)).astype(np.int32)
    with pytest.raises(pkg.VisoHipError) as ex:
        g.estimateMotion(ge, raw)  # flow matches carry no disparity
    assert ex.value.code == pkg.VH_ERR_STATE
    g.matchFeatures(pkg.METHOD_QUAD)
    tr, ok, ninl = g.estimateMotion(ge, raw)
    e = ob.EgoParams.default(f=400.0, cu=W / 2, cv=H / 2, base=0.5)
    for s in range(S):
        pm = g.getMatches(s)
        assert len(pm) > 100
        ok_o, tr_o, inl_o = oracle.estimate_motion_stereo(e, pm, oracle.draw_samples(len(pm), 200, raw[s].reshape(-1)))
        assert ok[s] == ok_o and ninl[s] == len(inl_o) and _close(tr[s], tr_o), (s, tr[s], tr_o)
    g.close()


@pytest.mark.gpu
def test_gpu_estimate_motion_rejects_hostile_inputs(pkg, ob, oracle, gpu):
    """Public C-ABI inputs reach the device validated: rand3 values with the sign bit set index like
    their low 31 bits (never a negative match index), a negative first offset is refused."""
    import ctypes as C
    e = ego_params(ob, ransac_iters=64)
    ge = pkg.EgoParams(ransac_iters=64, reweighting=1, inlier_threshold=2.0, f=e.f, cu=e.cu, cv=e.cv, base=e.base)
    pm = scene(ob.P_MATCH_DTYPE, 200, 77, outliers=0.2, noise=0.2)[0]
    raw = np.random.default_rng(9).integers(0, 2 ** 31 - 1, (1, 64, 3)).astype(np.int32)
    neg = (raw | np.int32(-2 ** 31)).astype(np.int32)  # every value negative
    tr_a, ok_a, inl_a = pkg.estimate_motion_stereo(ge, [pm], raw)
    tr_b, ok_b, inl_b = pkg.estimate_motion_stereo(ge, [pm], neg)
    assert ok_a[0] == ok_b[0] and np.array_equal(inl_a[0], inl_b[0]) and tr_a.tobytes() == tr_b.tobytes()
    offsets = np.array([-1, len(pm) - 1], np.int32)
    tr = np.zeros(6); ok = np.zeros(1, np.int32); ninl = np.zeros(1, np.int32)
    rc = pkg._lib().vh_estimate_motion_stereo(C.byref(ge), 0, 1, pm.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p),
                                              raw.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p), ok.ctypes.data_as(C.c_void_p),
                                              ninl.ctypes.data_as(C.c_void_p), None)
    assert rc == pkg.VH_ERR_INVALID_ARG
