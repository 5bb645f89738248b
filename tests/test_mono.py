"""SURVEY 8(f-4), mono half: VisualOdometryMono::estimateMotion (src/viso_mono.cpp:41-160) over
Matrix::svd (src/matrix.cpp:579-802).  CPU part: the plain-C restatement (oracle/viso_mono.c) is
pinned bit for bit to the reference's own code compiled in oracle/_ref, samples drawn as the
reference draws them (getRandomSample(N,8) on rand() after srand(0)), and to golden vectors the
reference produced (tests/golden/mono.npz, oracle/gen_golden_mono.py).  GPU part: the batched HIP
estimator -- inlier sets exact, tr within 1e-9 relative."""
import os

import numpy as np
import pytest

from egomotion_scene import mono_scene

CASES = [(400, 1, 0.2, 0.0), (200, 2, 0.3, 0.3), (1500, 3, 0.1, 0.5), (12, 4, 0.0, 0.0), (9, 5, 0.0, 0.0), (300, 6, 0.5, 0.2)]


def mono_params(ob, **kw):
    kw.setdefault("height", 1.65)
    return ob.MonoParams.default(f=645.24, cu=635.96, cv=194.13, **kw)


@pytest.mark.parametrize("shape", [(3, 3), (4, 4), (8, 9), (9, 9), (33, 9), (700, 9), (5, 3), (2, 6)])
def test_oracle_svd_equals_reference(shape, ob, oracle, reference):
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    for trial in range(6):
        a = rng.normal(size=shape)
        if trial == 1:
            a[-1] = a[0] * 2 - a[1] if shape[0] > 2 else a[0]    # rank deficient
        if trial == 2:
            a = np.round(a * 2)                                  # exact zeros, equal singular values on the way
        if trial == 3:
            a[:, 0] = 0                                          # a zero column: `scale == 0` branches
        Uo, Wo, Vo = oracle.svd(a)
        Ur, Wr, Vr = reference.svd(a)
        assert Uo.tobytes() == Ur.tobytes() and Wo.tobytes() == Wr.tobytes() and Vo.tobytes() == Vr.tobytes(), (shape, trial)
        if trial == 0 and shape[0] >= shape[1]:
            assert np.allclose(Uo[:, :shape[1]] @ np.diag(Wo) @ Vo.T, a, atol=1e-9)  # it is a decomposition of a


@pytest.mark.parametrize("n,seed,outliers,noise", CASES)
def test_oracle_equals_reference_estimate_motion_mono(n, seed, outliers, noise, ob, oracle, reference):
    pm, tr_true = mono_scene(ob.P_MATCH_DTYPE, n, seed, outliers=outliers, noise=noise)
    for kw in ({}, {"ransac_iters": 300}, {"pitch": -0.03, "inlier_threshold": 0.00002}, {"motion_threshold": 20.0}):
        e = mono_params(ob, **kw)
        samples = oracle.draw_samples_n(len(pm), 8, e.ransac_iters) if n >= 10 else np.zeros((e.ransac_iters, 8), np.int32)
        ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, pm, samples)
        ok_r, tr_r, inl_r = reference.estimate_motion_mono(e, pm)
        assert ok_o == ok_r and np.array_equal(inl_o, inl_r), (n, seed, kw)
        assert tr_o.tobytes() == tr_r.tobytes(), (tr_o, tr_r)
        if ok_o and not kw and n >= 400 and noise == 0:
            # the estimate is the scene's motion (the scale comes from the road plane: looser)
            assert np.allclose(tr_o[:3], tr_true[:3], atol=0.003) and np.allclose(tr_o[3:], tr_true[3:], atol=0.15), (tr_o, tr_true)


def _golden():
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "mono.npz"))


GOLDEN_CASES = ["m400", "m200_noisy", "m1500", "m300_hard", "m400_pitch", "m250_few_iters", "m12", "m9"]


def _golden_case(z, name, ob):
    pm = np.ascontiguousarray(z[name + "__pm"]).view(ob.P_MATCH_DTYPE).reshape(-1)
    g = z[name + "__mono"]
    e = ob.MonoParams.default(ransac_iters=int(g[0]), inlier_threshold=g[1], motion_threshold=g[2], height=g[3], pitch=g[4], f=g[5], cu=g[6], cv=g[7])
    return pm, e, z[name + "__samples"].astype(np.int32), bool(z[name + "__ok"]), z[name + "__tr"], z[name + "__inliers"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_estimate_motion_mono_golden(name, ob, oracle):
    """The restatement against vectors the reference's own estimateMotion produced: what pins it on
    the GPU box, where the reference build is absent."""
    pm, e, samples, ok, tr, inl = _golden_case(_golden(), name, ob)
    ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, pm, samples)
    assert ok_o == ok and np.array_equal(inl_o, inl) and tr_o.tobytes() == tr.tobytes()


def test_oracle_svd_golden(ob, oracle):
    z = _golden()
    k = 0
    while f"svd{k}__a" in z:
        U, W, V = oracle.svd(z[f"svd{k}__a"])
        assert U.tobytes() == z[f"svd{k}__U"].tobytes() and W.tobytes() == z[f"svd{k}__W"].tobytes() and V.tobytes() == z[f"svd{k}__V"].tobytes(), k
        k += 1
    assert k >= 7


def test_static_svd_header_equals_the_oracle(ob, oracle, tmp_path):
    """csrc/svd_static.h (what the mono kernels factorize 3x3 / 4x4 systems with, in registers) compiled for the
    host: bit-identical to the oracle's Matrix::svd restatement on 7 500 matrices; the U-less variant returns the same
    column up to sign and the same w; and the property mono_hyp's fast path rests on -- the rank-2 projection of -F is
    exactly minus that of F unless the decomposition raises its flag -- holds on 120 000 3x3 inputs, degenerate
    families included, with no random matrix flagged (tests/cpp/svd_static_check.cpp)."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "svd_static_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "svd_static_check.cpp"),
                           ob.ORACLE_SO, "-lm", "-Wl,-rpath," + os.path.dirname(ob.ORACLE_SO), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("0 mismatching") == 3 and "0 unflagged failures" in r.stdout, r.stdout + r.stderr


def _rand8(ob, iters, n_sets):
    r = ob.glibc_rand_after_srand0(8 * iters).reshape(iters, 8)
    return np.stack([r] * n_sets)


def _close(tr, want):
    return np.allclose(tr, want, rtol=1e-9, atol=1e-12)


def _gpu_params(pkg, e):
    return pkg.MonoParams.default(ransac_iters=e.ransac_iters, inlier_threshold=e.inlier_threshold, motion_threshold=e.motion_threshold,
                                  height=e.height, pitch=e.pitch, f=e.f, cu=e.cu, cv=e.cv)


@pytest.mark.gpu
def test_gpu_estimate_motion_mono_golden_batch(pkg, ob, oracle, gpu):
    """vh_estimate_motion_mono, the golden scenes of one parameter set in ONE batched launch: inlier sets
    exact, tr within 1e-9 relative of the reference's (device exp / asin / cos in the last steps)."""
    z = _golden()
    cases = [_golden_case(z, n, ob) for n in GOLDEN_CASES]
    for sel in ([0, 1, 2, 3, 6, 7], [4], [5]):
        e = cases[sel[0]][1]
        tr, ok, inl = pkg.estimate_motion_mono(_gpu_params(pkg, e), [cases[i][0] for i in sel], _rand8(ob, e.ransac_iters, len(sel)))
        for k, i in enumerate(sel):
            pm, _, samples, ok_w, tr_w, inl_w = cases[i]
            if len(pm) >= 10:
                assert np.array_equal(oracle.draw_samples_n(len(pm), 8, e.ransac_iters), samples)  # the kernel draws from the same rand() values
            assert ok[k] == ok_w and np.array_equal(inl[k], inl_w), GOLDEN_CASES[i]
            assert _close(tr[k], tr_w), (GOLDEN_CASES[i], tr[k], tr_w)


@pytest.mark.gpu
def test_gpu_estimate_motion_mono_random_scenes_vs_oracle(pkg, ob, oracle, gpu):
    rng = np.random.default_rng(6)
    e = mono_params(ob, ransac_iters=500)
    lists = []
    for s in range(20):
        n = int(rng.integers(5, 1200)) if s else 3000  # one list beyond 8 rows per lane of the cooperative SVD
        trs = (rng.normal(0, 0.004), rng.normal(0, 0.02), rng.normal(0, 0.003), rng.normal(0, 0.05), rng.normal(0, 0.02), -abs(rng.normal(0.8, 0.3)))
        lists.append(mono_scene(ob.P_MATCH_DTYPE, n, 300 + s, tr=trs, outliers=float(rng.uniform(0, 0.5)), noise=float(rng.uniform(0, 0.5)))[0])
    raw = rng.integers(0, 2 ** 31 - 1, (len(lists), 500, 8)).astype(np.int32)
    tr, ok, inl = pkg.estimate_motion_mono(_gpu_params(pkg, e), lists, raw)
    n_ok = 0
    for s, pm in enumerate(lists):
        samples = oracle.draw_samples_n(len(pm), 8, 500, raw[s].reshape(-1)) if len(pm) >= 10 else np.zeros((500, 8), np.int32)
        ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, pm, samples)
        assert ok[s] == ok_o and np.array_equal(inl[s], inl_o), s
        assert _close(tr[s], tr_o), (s, tr[s], tr_o)
        n_ok += ok_o
    assert n_ok >= 10


@pytest.mark.gpu
def test_gpu_group_estimate_motion_mono_on_device_matches(pkg, ob, oracle, gpu):
    """vh_group_estimate_motion_mono: the estimator straight on the device-resident flow match lists of a stream group."""
    S, W, H = 3, 480, 200
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, 2, disparity=6 + s, blur=4, seed=210 + s) for s in range(S)]
    g = pkg.StreamGroup(S, pkg.Params.default())
    for t in range(2):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), None, dims, False)
    e = ob.MonoParams.default(ransac_iters=400, height=1.65, f=400.0, cu=W / 2, cv=H / 2)
    raw = np.random.default_rng(2).integers(0, 2 ** 31 - 1, (S, 400, 8)).astype(np.int32)
    g.matchFeatures(pkg.METHOD_FLOW)
    tr, ok, ninl = g.estimateMotionMono(_gpu_params(pkg, e), raw)
    for s in range(S):
        pm = g.getMatches(s)
        assert len(pm) > 100
        ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, pm, oracle.draw_samples_n(len(pm), 8, 400, raw[s].reshape(-1)))
        assert ok[s] == ok_o and ninl[s] == len(inl_o) and _close(tr[s], tr_o), (s, tr[s], tr_o)
    g.close()


@pytest.mark.gpu
def test_gpu_device_post_stage_with_the_mono_estimator(pkg, ob, oracle, gpu):
    """vh_group_post_begin_device / _finish_device with the monocular estimator as the last stage: the vote, the bucketing
    and the pose of VisualOdometryMono::process (src/viso_mono.cpp:34-37) entirely on the GPU, two steps per batch, per
    stream equal to the oracle's chain on the same flow matches; also without an estimator (lists only)."""
    import ctypes as C
    S, W, H, T = 3, 480, 200, 5
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, T, disparity=6 + s, blur=4, seed=410 + s) for s in range(S)]
    po = ob.Params.default()
    F = [[oracle.compute_features(po, seqs[s][t][0], dims)[1] for t in range(T)] for s in range(S)]
    e = ob.MonoParams.default(ransac_iters=300, height=1.65, f=400.0, cu=W / 2, cv=H / 2)
    ge = _gpu_params(pkg, e)
    raw = np.random.default_rng(4).integers(0, 2 ** 31 - 1, (T, S, 300, 8)).astype(np.int32)

    def chain(t, s):
        pm, _ = oracle.remove_outliers(oracle.matching(po, dims, 0, m1p=F[s][t - 1], m1c=F[s][t]))
        q = pm.copy()
        n = oracle.lib.vo_bucket_features(q.ctypes.data_as(C.c_void_p), len(q), 2, C.c_float(50), C.c_float(50))
        return q[:n].copy()

    g = pkg.StreamGroup(S, pkg.Params.default())
    g.postDeviceConfig(2, 2, 64)
    for t in range(T):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), None, dims, False)
        if t == 0:
            continue
        g.matchFeatures(pkg.METHOD_FLOW)
        g.postBeginDevice(8192, 2, 50.0, 50.0, mono=ge, rand8=raw[t], want_lists=True)
    for t in range(1, T):  # the oldest first; the last batch holds one step only and is launched by its finish call
        got = g.postFinishDevice(T - 1 - t, want_lists=True)
        for s in range(S):
            q = chain(t, s)
            ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, q, oracle.draw_samples_n(len(q), 8, 300, raw[t, s].reshape(-1)))
            assert len(q) > 20 and got["lists"][s].tobytes() == q.tobytes(), (t, s)
            assert got["ok"][s] == ok_o and got["n_inliers"][s] == len(inl_o) and _close(got["tr"][s], tr_o), (t, s)
    # no estimator: the bucketed lists alone
    g.postBeginDevice(8192, 3, 40.0, 30.0, want_lists=True)
    got = g.postFinishDevice(0, want_lists=True, estimator=False)
    for s in range(S):
        pm, _ = oracle.remove_outliers(oracle.matching(po, dims, 0, m1p=F[s][T - 2], m1c=F[s][T - 1]))
        assert got["lists"][s].tobytes() == oracle.bucket_features(pm, 3, 40, 30).tobytes()
    g.close()


@pytest.mark.gpu
def test_gpu_pipelined_post_stage_with_the_mono_estimator(pkg, ob, oracle, gpu):
    """vh_group_post_begin / vh_group_post_finish_mono -- VisualOdometryMono::process after the matching
    (src/viso_mono.cpp:34-37) for a stream group: removeOutliers -> bucketFeatures(2, 50, 50) -> the monocular
    estimateMotion of step t finished while step t+1 is already issued, per stream equal to the oracle's chain on the
    same flow matches: bucketed lists bit for bit, success flags and inlier counts exact, tr to 1e-9."""
    import ctypes as C
    S, W, H, T = 3, 480, 200, 4
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    seqs = [pkg.synth.stereo_sequence(W, H, T, disparity=6 + s, blur=4, seed=410 + s) for s in range(S)]
    po = ob.Params.default()
    F = [[oracle.compute_features(po, seqs[s][t][0], dims)[1] for t in range(T)] for s in range(S)]
    e = ob.MonoParams.default(ransac_iters=300, height=1.65, f=400.0, cu=W / 2, cv=H / 2)
    ge = _gpu_params(pkg, e)
    raw = np.random.default_rng(4).integers(0, 2 ** 31 - 1, (T, S, 300, 8)).astype(np.int32)
    g = pkg.StreamGroup(S, pkg.Params.default())

    def check(t, got):
        for s in range(S):
            pm = oracle.matching(po, dims, 0, m1p=F[s][t - 1], m1c=F[s][t])
            pm, _ = oracle.remove_outliers(pm)
            q = pm.copy()
            n = oracle.lib.vo_bucket_features(q.ctypes.data_as(C.c_void_p), len(q), 2, C.c_float(50), C.c_float(50))
            q = q[:n].copy()
            ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, q, oracle.draw_samples_n(len(q), 8, 300, raw[t, s].reshape(-1)))
            assert len(q) > 20 and got["lists"][s].tobytes() == q.tobytes(), (t, s)
            assert got["ok"][s] == ok_o and got["n_inliers"][s] == len(inl_o), (t, s)
            assert _close(got["tr"][s], tr_o), (t, s, got["tr"][s], tr_o)

    for t in range(T):
        g.pushBack(np.stack([seqs[s][t][0] for s in range(S)]), None, dims, False)
        if t == 0:
            continue
        g.matchFeatures(pkg.METHOD_FLOW)
        g.postBegin(8192)
        if t >= 2:  # step t is in flight on the GPU; finish step t-1
            check(t - 1, g.postFinish(1, 2, 50.0, 50.0, host_threads=2, mono=ge, rand8=raw[t - 1]))
    check(T - 1, g.postFinish(0, 2, 50.0, 50.0, host_threads=2, mono=ge, rand8=raw[T - 1]))
    g.close()
    # quad lists carry the left camera's flow as well (stereo lists do not: refused)
    g = pkg.StreamGroup(1, pkg.Params.default())
    Fq = [[oracle.compute_features(po, im, dims)[1] for im in seqs[0][t]] for t in range(2)]
    for t in range(2):
        g.pushBack(seqs[0][t][0][None], seqs[0][t][1][None], dims, False)
    g.matchFeatures(pkg.METHOD_QUAD)
    g.postBegin(8192)
    got = g.postFinish(0, 2, 50.0, 50.0, host_threads=1, mono=ge, rand8=raw[0][:1])
    pm, _ = oracle.remove_outliers(oracle.matching(po, dims, 2, Fq[0][0], Fq[0][1], Fq[1][0], Fq[1][1]))
    q = pm.copy()
    n = oracle.lib.vo_bucket_features(q.ctypes.data_as(C.c_void_p), len(q), 2, C.c_float(50), C.c_float(50))
    q = q[:n].copy()
    ok_o, tr_o, inl_o = oracle.estimate_motion_mono(e, q, oracle.draw_samples_n(len(q), 8, 300, raw[0, 0].reshape(-1)))
    assert got["lists"][0].tobytes() == q.tobytes() and got["ok"][0] == ok_o and got["n_inliers"][0] == len(inl_o) and _close(got["tr"][0], tr_o)
    g.matchFeatures(pkg.METHOD_STEREO)
    g.postBegin(8192)
    with pytest.raises(pkg.VisoHipError) as ex:
        g.postFinish(0, 2, 50.0, 50.0, host_threads=1, mono=ge, rand8=raw[0][:1])
    assert ex.value.code == pkg.VH_ERR_STATE
    g.close()


@pytest.mark.gpu
def test_gpu_mono_signed_recount_path(gpu):
    """mono_hyp counts inliers with +-F (no U in the 8x9 decomposition) and hands a list to the signed kernel behind it
    when a hypothesis raises svd_static's mirror-image flag -- noise-free scenes do, image data does not.
    VH_MONO_SIGNED=1 sends every list down that second path (read once per process, hence the subprocess): the same
    GPU cases must pass on it."""
    import subprocess, sys
    env = dict(os.environ, VH_MONO_SIGNED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "not signed_recount and not pipelined"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
