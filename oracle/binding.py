"""ctypes doors onto the oracle -- TEST INFRASTRUCTURE ONLY.

`Oracle` wraps oracle/libviso_oracle.so (own plain-C restatement, travels to
the GPU box).  `Reference` wraps oracle/_ref/libviso_ref.so (the reference's own
sources compiled in-container by `make -C oracle _ref`; the prebuilt file
travels, the sources do not).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libviso_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libviso_ref.so")
REFERENCE_ROOT = "/root/reference"


class Params(C.Structure):
    """POD mirror of Matcher::parameters (reference src/matcher.h:45-72)."""
    _fields_ = [(n, C.c_int32) for n in (
        "nms_n", "nms_tau", "match_binsize", "match_radius", "match_disp_tolerance",
        "outlier_disp_tolerance", "outlier_flow_tolerance", "multi_stage",
        "half_resolution", "refinement")] + [(n, C.c_double) for n in ("f", "cu", "cv", "base")]

    @classmethod
    def default(cls, **kw):
        p = cls(nms_n=2, nms_tau=50, match_binsize=50, match_radius=200,
                match_disp_tolerance=2, outlier_disp_tolerance=5, outlier_flow_tolerance=5,
                multi_stage=0, half_resolution=0, refinement=0)
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p


class EgoParams(C.Structure):
    """VisualOdometryStereo::parameters + calibration (reference src/viso_stereo.h:31-43, src/viso.h:41-50)."""
    _fields_ = [("ransac_iters", C.c_int32), ("reweighting", C.c_int32), ("inlier_threshold", C.c_double),
                ("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double), ("base", C.c_double)]

    @classmethod
    def default(cls, **kw):
        e = cls(ransac_iters=200, reweighting=1, inlier_threshold=2.0, f=1.0, cu=0.0, cv=0.0, base=1.0)
        for k, v in kw.items():
            if not hasattr(e, k):
                raise AttributeError(k)
            setattr(e, k, v)
        return e


class MonoParams(C.Structure):
    """VisualOdometryMono::parameters + calibration (reference src/viso_mono.h:32-46, src/viso.h:41-50)."""
    _fields_ = [("ransac_iters", C.c_int32), ("pad_", C.c_int32), ("inlier_threshold", C.c_double), ("motion_threshold", C.c_double),
                ("height", C.c_double), ("pitch", C.c_double), ("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double)]

    @classmethod
    def default(cls, **kw):
        e = cls(ransac_iters=2000, pad_=0, inlier_threshold=0.00001, motion_threshold=100.0, height=1.0, pitch=0.0, f=1.0, cu=0.0, cv=0.0)
        for k, v in kw.items():
            if not hasattr(e, k):
                raise AttributeError(k)
            setattr(e, k, v)
        return e


def glibc_rand_after_srand0(count: int) -> np.ndarray:
    """The first `count` values of rand() after srand(0): what the reference's
    VisualOdometry constructor seeds (src/viso.cpp:35) and getRandomSample consumes."""
    libc = C.CDLL("libc.so.6")
    libc.srand(0)
    libc.rand.restype = C.c_int
    return np.array([libc.rand() for _ in range(count)], np.int32)


P_MATCH_DTYPE = np.dtype([
    ("u1p", "<f4"), ("v1p", "<f4"), ("i1p", "<i4"),
    ("u2p", "<f4"), ("v2p", "<f4"), ("i2p", "<i4"),
    ("u1c", "<f4"), ("v1c", "<f4"), ("i1c", "<i4"),
    ("u2c", "<f4"), ("v2c", "<f4"), ("i2c", "<i4")])
assert P_MATCH_DTYPE.itemsize == 48


def build(ref: bool = True) -> None:
    """Compile the C restatement and, when the reference tree is present, the
    reference harness.  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        subprocess.check_call(["make", "-s", "-C", HERE, "_ref"])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dims(dims):
    return (C.c_int32 * 3)(*[int(d) for d in dims])


def _feat(m):
    m = np.ascontiguousarray(m, dtype=np.int32).reshape(-1, 12)
    return m, m.shape[0]


def bin_counts(dims, binsize):
    ubn = -(-int(dims[0]) // int(binsize))
    vbn = -(-int(dims[1]) // int(binsize))
    return ubn, vbn


class Oracle:
    """Own CPU restatement (oracle/viso_oracle.c)."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.vo_fnv1a64.restype = C.c_uint64
        self.lib.vo_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        self.lib.vo_bucket_features.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float]
        self.lib.vo_find_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_double, C.c_double]

    def fnv(self, a) -> int:
        a = np.ascontiguousarray(a)
        return int(self.lib.vo_fnv1a64(_ptr(a), C.c_uint64(a.nbytes)))

    def filters(self, img):
        h, bpl = img.shape
        img = np.ascontiguousarray(img, dtype=np.uint8)
        du = np.empty((h, bpl), np.uint8); dv = np.empty((h, bpl), np.uint8)
        f1 = np.empty((h, bpl), np.int16); f2 = np.empty((h, bpl), np.int16)
        self.lib.vo_filters(_ptr(img), C.c_int32(bpl), C.c_int32(h), _ptr(du), _ptr(dv), _ptr(f1), _ptr(f2))
        return du, dv, f1, f2

    def half_resolution(self, img, dims):
        dh = (C.c_int32 * 3)()
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self.lib.vo_half_resolution(_ptr(img), _dims(dims), dh, None)
        out = np.zeros((dh[1], dh[2]), np.uint8)
        self.lib.vo_half_resolution(_ptr(img), _dims(dims), dh, _ptr(out))
        return out, list(dh)

    def compute_features(self, params, img, dims, cap=None, planes=False):
        """-> (max1 [n1,12], max2 [n2,12][, du, dv])"""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if cap is None:
            cap = 4 * (dims[0] // (params.nms_n + 1) + 1) * (dims[1] // (params.nms_n + 1) + 1)
        m1 = np.zeros((cap, 12), np.int32); m2 = np.zeros((cap, 12), np.int32)
        n1 = C.c_int32(0); n2 = C.c_int32(0)
        du = dv = None
        if planes:
            if params.half_resolution:
                _, dh = self.half_resolution(img, dims)
                shape = (dh[1], dh[2])
            else:
                shape = (dims[1], dims[2])
            du = np.zeros(shape, np.uint8); dv = np.zeros(shape, np.uint8)
        rc = self.lib.vo_compute_features(C.byref(params), _ptr(img), _dims(dims), _ptr(m1), C.c_int32(cap),
                                          C.byref(n1), _ptr(m2), C.c_int32(cap), C.byref(n2), _ptr(du), _ptr(dv))
        if rc != 0:
            raise ValueError("vo_compute_features: bad dims")
        r = (m1[:min(n1.value, cap)].copy(), m2[:min(n2.value, cap)].copy())
        return r + (du, dv) if planes else r

    def create_index(self, params, m, dims):
        m, n = _feat(m)
        ubn, vbn = bin_counts(dims, params.match_binsize)
        bs = np.zeros(4 * ubn * vbn + 1, np.int32); lst = np.zeros(max(n, 1), np.int32)
        self.lib.vo_create_index(_ptr(m), C.c_int32(n), C.c_int32(params.match_binsize), C.c_int32(ubn),
                                 C.c_int32(vbn), _ptr(bs), _ptr(lst))
        return bs, lst[:n]

    def match_all(self, params, dims, m1, m2, flow=True):
        m1, n1 = _feat(m1); m2, n2 = _feat(m2)
        best = np.zeros(max(n1, 1), np.int32)
        self.lib.vo_match_all(C.byref(params), _dims(dims), _ptr(m1), C.c_int32(n1), _ptr(m2), C.c_int32(n2),
                              C.c_int32(1 if flow else 0), _ptr(best))
        return best[:n1]

    def match_all_prior(self, params, dims, m1, m2, u_, v_, flow=True):
        """findMatch with the u_,v_ distance term, for every query."""
        m1, n1 = _feat(m1); m2, n2 = _feat(m2)
        bs, lst = self.create_index(params, m2, dims)
        ubn, vbn = bin_counts(dims, params.match_binsize)
        self.lib.vo_find_match.restype = C.c_int32
        out = np.zeros(n1, np.int32)
        for i in range(n1):
            out[i] = self.lib.vo_find_match(C.byref(params), _ptr(m1), i, _ptr(m2), _ptr(bs), _ptr(lst),
                                            ubn, vbn, 1 if flow else 0, float(u_), float(v_))
        return out

    def matching(self, params, dims, method, m1p=None, m2p=None, m1c=None, m2c=None, cap=None):
        z = np.zeros((0, 12), np.int32)
        sets = [_feat(z if m is None else m) for m in (m1p, m2p, m1c, m2c)]
        if cap is None:
            cap = max(s[1] for s in sets) + 1
        out = np.zeros(cap, P_MATCH_DTYPE); n = C.c_int32(0)
        rc = self.lib.vo_matching(C.byref(params), _dims(dims), C.c_int32(method),
                                  _ptr(sets[0][0]), C.c_int32(sets[0][1]), _ptr(sets[1][0]), C.c_int32(sets[1][1]),
                                  _ptr(sets[2][0]), C.c_int32(sets[2][1]), _ptr(sets[3][0]), C.c_int32(sets[3][1]),
                                  _ptr(out), C.c_int32(cap), C.byref(n))
        if rc != 0:
            raise ValueError("vo_matching: invalid method")
        return out[:min(n.value, cap)].copy()

    def matching_quad_prior(self, params, dims, tr16, m1p, m2p, m1c, m2c):
        """Quad matching with the motion prior Tr_delta (row-major 4x4) on hop 2 [upstream-recollection]."""
        sets = [_feat(m) for m in (m1p, m2p, m1c, m2c)]
        cap = max(s[1] for s in sets) + 1
        tr = np.ascontiguousarray(tr16, np.float64).reshape(16)
        out = np.zeros(cap, P_MATCH_DTYPE); n = C.c_int32(0)
        rc = self.lib.vo_matching_quad_prior(C.byref(params), _dims(dims), _ptr(tr),
                                             _ptr(sets[0][0]), C.c_int32(sets[0][1]), _ptr(sets[1][0]), C.c_int32(sets[1][1]),
                                             _ptr(sets[2][0]), C.c_int32(sets[2][1]), _ptr(sets[3][0]), C.c_int32(sets[3][1]),
                                             _ptr(out), C.c_int32(cap), C.byref(n))
        assert rc == 0
        return out[:min(n.value, cap)].copy()

    def bucket_features(self, pm, max_features, bw, bh):
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE).copy()
        n = self.lib.vo_bucket_features(_ptr(pm), C.c_int32(len(pm)), C.c_int32(max_features),
                                        C.c_float(bw), C.c_float(bh))
        return pm[:n].copy()

    def delaunay(self, xy):
        """-> (triangle corners [t,3], deepest flip stack)"""
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        cap = 6 * len(xy) + 16
        tri = np.zeros(cap, np.int32); depth = C.c_int32(0)
        self.lib.vo_delaunay.restype = C.c_int32
        n = self.lib.vo_delaunay(_ptr(xy), C.c_int32(len(xy)), _ptr(tri), C.c_int32(cap), C.byref(depth))
        return tri[:n].reshape(-1, 3).copy(), depth.value

    def remove_outliers(self, pm):
        """-> (kept matches, deepest flip stack)"""
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE).copy()
        depth = C.c_int32(0)
        n = self.lib.vo_remove_outliers(_ptr(pm), C.c_int32(len(pm)), C.byref(depth))
        return pm[:n].copy(), depth.value


    # ---- stereo egomotion -------------------------------------------------
    def draw_samples(self, n_matches: int, iters: int, rand_values=None) -> np.ndarray:
        """VisualOdometry::getRandomSample(N,3) x iters (src/viso.cpp:86-106) -> [iters,3] int32;
        rand_values defaults to glibc's rand() sequence after srand(0)."""
        r = glibc_rand_after_srand0(3 * iters) if rand_values is None else np.ascontiguousarray(rand_values, np.int32)
        out = np.zeros((iters, 3), np.int32)
        self.lib.vo_draw_samples(C.c_int32(n_matches), C.c_int32(iters), _ptr(r), _ptr(out))
        return out

    def estimate_motion_stereo(self, ego, pm, samples):
        """VisualOdometryStereo::estimateMotion (src/viso_stereo.cpp:54-157) -> (ok, tr[6], inlier indices)."""
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE)
        samples = np.ascontiguousarray(samples, np.int32)
        tr = np.zeros(6, np.float64); inl = np.zeros(max(len(pm), 1), np.int32); n = C.c_int32(0)
        self.lib.vo_estimate_motion_stereo.restype = C.c_int32
        ok = self.lib.vo_estimate_motion_stereo(C.byref(ego), _ptr(pm), C.c_int32(len(pm)), _ptr(samples), _ptr(tr), _ptr(inl), C.byref(n))
        return bool(ok), tr, inl[:n.value].copy()


    # ---- monocular egomotion ----------------------------------------------
    def draw_samples_n(self, n_matches: int, num: int, iters: int, rand_values=None) -> np.ndarray:
        """VisualOdometry::getRandomSample(N,num) x iters (src/viso.cpp:86-106) -> [iters,num] int32."""
        r = glibc_rand_after_srand0(num * iters) if rand_values is None else np.ascontiguousarray(rand_values, np.int32)
        assert r.size >= num * iters
        out = np.zeros((iters, num), np.int32)
        self.lib.vo_draw_samples_n(C.c_int32(n_matches), C.c_int32(num), C.c_int32(iters), _ptr(r), _ptr(out))
        return out

    def svd(self, a):
        """Matrix::svd (src/matrix.cpp:579-802) -> (U2 [m,m], W [min(m,n)], V [n,n])."""
        a = np.ascontiguousarray(a, np.float64)
        m, n = a.shape
        U = np.zeros((m, m)); W = np.zeros(min(m, n)); V = np.zeros((n, n))
        self.lib.vo_svd(_ptr(a), C.c_int32(m), C.c_int32(n), _ptr(U), _ptr(W), _ptr(V))
        return U, W, V

    def estimate_motion_mono(self, mono, pm, samples):
        """VisualOdometryMono::estimateMotion (src/viso_mono.cpp:41-160) -> (ok, tr[6], inlier indices)."""
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE)
        samples = np.ascontiguousarray(samples, np.int32)
        assert samples.shape == (mono.ransac_iters, 8)
        tr = np.zeros(6, np.float64); inl = np.zeros(max(len(pm), 1), np.int32); n = C.c_int32(0)
        self.lib.vo_estimate_motion_mono.restype = C.c_int32
        ok = self.lib.vo_estimate_motion_mono(C.byref(mono), _ptr(pm), C.c_int32(len(pm)), _ptr(samples), _ptr(tr), _ptr(inl), C.byref(n))
        return bool(ok), tr, inl[:n.value].copy()


class Reference:
    """The reference's own CPU/SSE code (oracle/_ref/libviso_ref.so)."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle _ref` where /root/reference exists)")
        self.lib = C.CDLL(path)
        assert self.lib.ref_sizeof_p_match() == 48
        self.lib.ref_bucket_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float]
        self.lib.ref_match_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_double, C.c_double, C.c_void_p]

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def filters(self, img):
        h, bpl = img.shape
        img = np.ascontiguousarray(img, dtype=np.uint8)
        du = np.empty((h, bpl), np.uint8); dv = np.empty((h, bpl), np.uint8)
        f1 = np.empty((h, bpl), np.int16); f2 = np.empty((h, bpl), np.int16)
        self.lib.ref_filters(_ptr(img), C.c_int32(bpl), C.c_int32(h), _ptr(du), _ptr(dv), _ptr(f1), _ptr(f2))
        return du, dv, f1, f2

    def half_resolution(self, params, img, dims):
        dh = (C.c_int32 * 3)()
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self.lib.ref_half_resolution(C.byref(params), _ptr(img), _dims(dims), dh, None)
        out = np.zeros((dh[1], dh[2]), np.uint8)
        self.lib.ref_half_resolution(C.byref(params), _ptr(img), _dims(dims), dh, _ptr(out))
        return out, list(dh)

    def compute_features(self, params, img, dims, cap=None, planes=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if cap is None:
            cap = 4 * (dims[0] // (params.nms_n + 1) + 1) * (dims[1] // (params.nms_n + 1) + 1)
        m1 = np.zeros((cap, 12), np.int32); m2 = np.zeros((cap, 12), np.int32)
        n1 = C.c_int32(0); n2 = C.c_int32(0)
        du = dv = None
        if planes:
            w2 = dims[0] // 2
            shape = (dims[1] // 2, w2 + 15 - (w2 - 1) % 16) if params.half_resolution else (dims[1], dims[2])
            du = np.zeros(shape, np.uint8); dv = np.zeros(shape, np.uint8)
        self.lib.ref_compute_features(C.byref(params), _ptr(img), _dims(dims), _ptr(m1), C.c_int32(cap), C.byref(n1),
                                      _ptr(m2), C.c_int32(cap), C.byref(n2), _ptr(du), _ptr(dv))
        r = (m1[:min(n1.value, cap)].copy(), m2[:min(n2.value, cap)].copy())
        return r + (du, dv) if planes else r

    def create_index(self, params, m, dims):
        m, n = _feat(m)
        ubn, vbn = bin_counts(dims, params.match_binsize)
        bs = np.zeros(4 * ubn * vbn + 1, np.int32); lst = np.zeros(max(n, 1), np.int32)
        self.lib.ref_create_index(C.byref(params), _ptr(m), C.c_int32(n), C.c_int32(ubn), C.c_int32(vbn),
                                  _ptr(bs), _ptr(lst))
        return bs, lst[:n]

    def match_all(self, params, dims, m1, m2, u_=-1.0, v_=-1.0):
        m1, n1 = _feat(m1); m2, n2 = _feat(m2)
        best = np.zeros(max(n1, 1), np.int32)
        self.lib.ref_match_all(C.byref(params), _dims(dims), _ptr(m1), n1, _ptr(m2), n2,
                               float(u_), float(v_), _ptr(best))
        return best[:n1]

    def matching_flow(self, params, dims, m1p, m1c):
        m1p, n1p = _feat(m1p); m1c, n1c = _feat(m1c)
        cap = n1c + 1
        out = np.zeros(cap, P_MATCH_DTYPE); n = C.c_int32(0)
        rc = self.lib.ref_matching_flow(C.byref(params), _dims(dims), _ptr(m1p), C.c_int32(n1p), _ptr(m1c),
                                        C.c_int32(n1c), _ptr(out), C.c_int32(cap), C.byref(n))
        assert rc == 0
        return out[:n.value].copy()

    def bucket_features(self, params, pm, max_features, bw, bh):
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE).copy()
        n = self.lib.ref_bucket_features(C.byref(params), _ptr(pm), len(pm), max_features, float(bw), float(bh))
        assert n >= 0
        return pm[:n].copy()

    def delaunay(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        cap = 6 * len(xy) + 16
        tri = np.zeros(cap, np.int32)
        n = self.lib.ref_delaunay(_ptr(xy), C.c_int32(len(xy)), _ptr(tri), C.c_int32(cap))
        assert n >= 0, n
        return tri[:n].reshape(-1, 3).copy()

    def remove_outliers(self, pm):
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE).copy()
        n = self.lib.ref_remove_outliers(_ptr(pm), C.c_int32(len(pm)))
        assert n >= 0, n
        return pm[:n].copy()

    def estimate_motion_stereo(self, ego, pm):
        """The reference's VisualOdometryStereo::estimateMotion on a fresh object (srand(0))."""
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE)
        tr = np.zeros(6, np.float64); inl = np.zeros(max(len(pm), 1), np.int32); n = C.c_int32(0)
        self.lib.ref_estimate_motion_stereo.restype = C.c_int32
        ok = self.lib.ref_estimate_motion_stereo(C.byref(ego), _ptr(pm), C.c_int32(len(pm)), _ptr(tr), _ptr(inl), C.byref(n))
        return bool(ok), tr, inl[:n.value].copy()

    def estimate_motion_mono(self, mono, pm):
        """The reference's VisualOdometryMono::estimateMotion on a fresh object (srand(0))."""
        pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE)
        tr = np.zeros(6, np.float64); inl = np.zeros(max(len(pm), 1), np.int32); n = C.c_int32(0)
        self.lib.ref_estimate_motion_mono.restype = C.c_int32
        ok = self.lib.ref_estimate_motion_mono(C.byref(mono), _ptr(pm), C.c_int32(len(pm)), _ptr(tr), _ptr(inl), C.byref(n))
        return bool(ok), tr, inl[:n.value].copy()

    def svd(self, a):
        """The reference's Matrix::svd."""
        a = np.ascontiguousarray(a, np.float64)
        m, n = a.shape
        U = np.zeros((m, m)); W = np.zeros(min(m, n)); V = np.zeros((n, n))
        self.lib.ref_svd(_ptr(a), C.c_int32(m), C.c_int32(n), _ptr(U), _ptr(W), _ptr(V))
        return U, W, V
