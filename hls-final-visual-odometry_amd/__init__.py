"""viso-hip: MI355X-native feature detection + matching behind the libviso2
`Matcher` surface of Chang-Tun-Yu/HLS-final-Visual-Odometry.

This package is the thin Python host mirror over the C ABI of
`libviso_hip.so` (include/viso_hip.h).  All compute happens in the hand-written
HIP kernels under csrc/; there is no CPU or PyTorch fallback: importing works
without a GPU (so the library and its exported symbols can be checked), any
compute call without a usable GPU raises `VisoHipError(VH_ERR_NO_DEVICE)`, and a
missing shared library raises at import.

The directory name contains hyphens, so load it with
`__graft_entry__.load_package()` (importlib under the name
`hls_final_visual_odometry_amd`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import synth  # noqa: F401  (synthetic KITTI-shaped frames)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VISO_HIP_LIB") or os.path.join(HERE, "libviso_hip.so")  # override: experiment builds
CHECK_LIB_PATH = os.path.join(HERE, "libviso_hip_check.so")  # the -DVH_CHECK build (tests only)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))



def source_sha256() -> str:
    """sha256 over what determines the code object: every source, header and the Makefile under csrc/ plus the public
    header (names and contents, sorted).  Build-independent -- the `.so` hashes differently on every rebuild -- so a
    profile (profiles/traffic_latest.json) can say which code it was taken on (bench.py: roofline.traffic)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
             if f.endswith((".hip", ".h", ".hpp", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(INCLUDE, "viso_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


VH_OK = 0
VH_ERR_INVALID_ARG, VH_ERR_NO_DEVICE, VH_ERR_HIP, VH_ERR_CAPACITY, VH_ERR_UNSUPPORTED, VH_ERR_STATE = -1, -2, -3, -4, -5, -6
SET_1P, SET_2P, SET_1C, SET_2C = 0, 1, 2, 3
METHOD_FLOW, METHOD_STEREO, METHOD_QUAD = 0, 1, 2

#: every symbol include/viso_hip.h declares (checked by the CPU test-suite)
ABI_SYMBOLS = (
    "vh_abi_version", "vh_device_count", "vh_error_string", "vh_last_error", "vh_default_params",
    "vh_create", "vh_create_ex", "vh_destroy", "vh_set_intrinsics", "vh_push_back", "vh_push_back_device",
    "vh_match_features", "vh_remove_outliers", "vh_remove_outliers_pm", "vh_remove_outliers_device", "vh_bucket_features", "vh_get_matches", "vh_get_features", "vh_synchronize",
    "vh_set_stream", "vh_clear_stream", "vh_stream_wait_images", "vh_host_alloc", "vh_host_free", "vh_compute_features", "vh_filters", "vh_create_index", "vh_match_all", "vh_match_all_prior", "vh_match",
    "vh_group_create", "vh_group_destroy", "vh_group_streams", "vh_group_device_bytes", "vh_group_push_back_device",
    "vh_group_push_back", "vh_group_match_features", "vh_group_match_features_prior", "vh_group_remove_outliers", "vh_group_get_matches", "vh_group_get_matches_all", "vh_group_download_matches_async", "vh_group_wait_download", "vh_group_get_features",
    "vh_group_get_counts", "vh_group_synchronize", "vh_group_set_stream", "vh_group_clear_stream", "vh_group_stream_wait_images", "vh_group_profile_enable",
    "vh_group_profile_read", "vh_group_profile_reset", "vh_group_debug_fail_next_alloc", "vh_debug_vote_stack_slots",
    "vh_default_ego_params", "vh_estimate_motion_stereo", "vh_group_estimate_motion", "vh_group_search_stats",
    "vh_default_mono_params", "vh_estimate_motion_mono", "vh_group_estimate_motion_mono",
    "vh_group_post_begin", "vh_group_post_finish", "vh_group_post_finish_mono",
    "vh_group_post_device_config", "vh_group_post_begin_device", "vh_group_post_finish_device",
)


class Params(C.Structure):
    """POD mirror of Matcher::parameters (reference src/matcher.h:45-72)."""
    _fields_ = [(n, C.c_int32) for n in (
        "nms_n", "nms_tau", "match_binsize", "match_radius", "match_disp_tolerance",
        "outlier_disp_tolerance", "outlier_flow_tolerance", "multi_stage",
        "half_resolution", "refinement")] + [(n, C.c_double) for n in ("f", "cu", "cv", "base")]

    @classmethod
    def default(cls, **kw):
        """Matcher::parameters() defaults (reference src/matcher.h:60-71)."""
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
    _fields_ = [("ransac_iters", C.c_int32), ("reserved_", C.c_int32), ("inlier_threshold", C.c_double), ("motion_threshold", C.c_double),
                ("height", C.c_double), ("pitch", C.c_double), ("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double)]

    @classmethod
    def default(cls, **kw):
        e = cls(ransac_iters=2000, reserved_=0, inlier_threshold=0.00001, motion_threshold=100.0, height=1.0, pitch=0.0, f=1.0, cu=0.0, cv=0.0)
        for k, v in kw.items():
            if not hasattr(e, k):
                raise AttributeError(k)
            setattr(e, k, v)
        return e


#: Matcher::p_match (reference src/matcher.h:89-104), 48 bytes
P_MATCH_DTYPE = np.dtype([
    ("u1p", "<f4"), ("v1p", "<f4"), ("i1p", "<i4"),
    ("u2p", "<f4"), ("v2p", "<f4"), ("i2p", "<i4"),
    ("u1c", "<f4"), ("v1c", "<f4"), ("i1c", "<i4"),
    ("u2c", "<f4"), ("v2c", "<f4"), ("i2c", "<i4")])


class VisoHipError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = _lib().vh_error_string(code).decode()
        last = _lib().vh_last_error().decode()
        super().__init__(f"{where}: {msg} ({code})" + (f" [{last}]" if last else ""))


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libviso_hip.so (in-tree), and the -DVH_CHECK
    variant libviso_hip_check.so (index invariants verified on the device, csrc/vh_dev.h) that
    tests/ run a part of the suite on."""
    cmd = ["make", "-C", CSRC, "-j4"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    subprocess.check_call(cmd + ["VARIANT=check"])
    return LIB_PATH


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the HIP path)")
        lib = C.CDLL(LIB_PATH)
        lib.vh_error_string.restype = C.c_char_p
        lib.vh_error_string.argtypes = [C.c_int32]
        lib.vh_last_error.restype = C.c_char_p
        vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
        sig = {
            "vh_create": [vp, i32, vp], "vh_create_ex": [vp, i32, i32, i32, vp], "vh_destroy": [vp],
            "vh_set_intrinsics": [vp, f64, f64, f64, f64],
            "vh_push_back": [vp, vp, vp, vp, i32], "vh_push_back_device": [vp, vp, vp, vp, i32],
            "vh_match_features": [vp, i32, vp], "vh_bucket_features": [vp, i32, f32, f32],
            "vh_get_matches": [vp, vp, i32, vp], "vh_get_features": [vp, i32, vp, i32, vp],
            "vh_synchronize": [vp], "vh_set_stream": [vp, vp], "vh_clear_stream": [vp], "vh_stream_wait_images": [vp, vp],
            "vh_remove_outliers": [vp], "vh_remove_outliers_pm": [vp, i32, vp], "vh_group_remove_outliers": [vp, i32],
            "vh_remove_outliers_device": [i32, i32, vp, i64, vp, i32, i32, f32, f32, vp, i32, vp, vp, vp],
            "vh_host_alloc": [i32, C.c_size_t, vp], "vh_host_free": [vp],
            "vh_compute_features": [vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp],
            "vh_filters": [i32, vp, i32, i32, vp, vp, vp, vp],
            "vh_create_index": [vp, i32, vp, vp, i32, vp, vp],
            "vh_match_all": [vp, i32, vp, vp, i32, vp, i32, i32, vp],
            "vh_match_all_prior": [vp, i32, vp, vp, i32, vp, i32, i32, f64, f64, vp],
            "vh_match": [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp],
            "vh_group_create": [vp, i32, i32, i32, i32, vp], "vh_group_destroy": [vp], "vh_group_streams": [vp],
            "vh_group_push_back_device": [vp, vp, vp, i64, vp, i32],
            "vh_group_push_back": [vp, vp, vp, i64, vp, i32],
            "vh_group_match_features": [vp, i32], "vh_group_match_features_prior": [vp, i32, vp], "vh_group_get_matches": [vp, i32, vp, i32, vp],
            "vh_group_get_matches_all": [vp, vp, i32, vp],
            "vh_group_download_matches_async": [vp, vp, i32, vp], "vh_group_wait_download": [vp],
            "vh_group_get_features": [vp, i32, i32, vp, i32, vp], "vh_group_get_counts": [vp, vp, vp],
            "vh_group_synchronize": [vp], "vh_group_set_stream": [vp, vp], "vh_group_clear_stream": [vp],
            "vh_group_stream_wait_images": [vp, vp],
            "vh_group_profile_enable": [vp, i32], "vh_group_profile_read": [vp, C.c_char_p, vp, vp],
            "vh_group_profile_reset": [vp], "vh_group_debug_fail_next_alloc": [vp], "vh_debug_vote_stack_slots": [i32],
            "vh_estimate_motion_stereo": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp],
            "vh_group_estimate_motion": [vp, vp, vp, vp, vp, vp],
            "vh_estimate_motion_mono": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp],
            "vh_group_estimate_motion_mono": [vp, vp, vp, vp, vp, vp],
            "vh_group_post_begin": [vp, i32],
            "vh_group_post_finish": [vp, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp],
            "vh_group_post_finish_mono": [vp, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp],
            "vh_group_post_device_config": [vp, i32, i32, i32],
            "vh_group_post_begin_device": [vp, i32, i32, f32, f32, vp, vp, vp, vp, i32],
            "vh_group_post_finish_device": [vp, i32, vp, vp, vp, vp, i32, vp],
            "vh_group_search_stats": [vp, vp, vp],
        }
        for name, args in sig.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = None if name.endswith("destroy") else i32
        lib.vh_group_device_bytes.argtypes = [vp]
        lib.vh_group_device_bytes.restype = i64
        _LIB = lib
    return _LIB


def _check(rc: int, where: str, allow=()):
    if rc != VH_OK and rc not in allow:
        raise VisoHipError(rc, where)
    return rc


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dims(dims):
    return (C.c_int32 * 3)(*[int(d) for d in dims])


def _feat(m):
    if m is None:
        return np.zeros((0, 12), np.int32), 0
    m = np.ascontiguousarray(m, dtype=np.int32).reshape(-1, 12)
    return m, m.shape[0]


def device_count() -> int:
    """Visible HIP devices (0 when there is none)."""
    return max(0, _lib().vh_device_count())


def pinned_empty(shape, dtype=np.uint8, device: int = 0) -> np.ndarray:
    """numpy array over page-locked host memory (vh_host_alloc): image buffers whose
    upload in pushBack runs at PCIe rate.  Freed when the array is collected."""
    import weakref
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = C.c_void_p()
    _check(_lib().vh_host_alloc(int(device), C.c_size_t(max(n, 1)), C.byref(ptr)), "vh_host_alloc")
    buf = (C.c_uint8 * max(n, 1)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, _lib().vh_host_free, C.c_void_p(ptr.value))
    return arr


def abi_version() -> int:
    return _lib().vh_abi_version()


# --------------------------------------------------------------------------- one stream
class Matcher:
    """Host mirror of the reference's `Matcher` public surface
    (src/matcher.h:75-143): pushBack / matchFeatures / bucketFeatures /
    getMatches, same argument meaning, plus getFeatures for the parity checks."""

    def __init__(self, param: Params | None = None, device: int = 0, max_features: int = 0,
                 max_matches: int = 0, outlier_removal: bool = True):
        self.param = param if param is not None else Params.default()
        # matchFeatures ends with removeOutliers as the reference's does (src/matcher.cpp:108);
        # False gives the bare Matcher::matching result
        self.outlier_removal = bool(outlier_removal)
        h = C.c_void_p()
        _check(_lib().vh_create_ex(C.byref(self.param), device, max_features, max_matches, C.byref(h)), "vh_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib().vh_destroy(self._h)
            self._h = None

    __del__ = close

    def setIntrinsics(self, f, cu, cv, base):
        _check(_lib().vh_set_intrinsics(self._h, f, cu, cv, base), "vh_set_intrinsics")

    def pushBack(self, I1, I2=None, dims=None, replace=False):
        """I1/I2: (H, bpl) uint8 numpy arrays (host) -- Matcher::pushBack (src/matcher.cpp:51-91).
        Like the reference, a dimension mismatch is reported and the call returns False."""
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        if I2 is not None:
            I2 = np.ascontiguousarray(I2, dtype=np.uint8)
        if dims is None:
            dims = [I1.shape[1], I1.shape[0], I1.shape[1]]
        rc = _lib().vh_push_back(self._h, _ptr(I1), _ptr(I2), _dims(dims), 1 if replace else 0)
        if rc == VH_ERR_INVALID_ARG:
            print("ERROR: Image dimension mismatch!")
            return False
        _check(rc, "vh_push_back")
        return True

    def pushBackDevice(self, ptr1: int, ptr2: int | None, dims, replace=False):
        _check(_lib().vh_push_back_device(self._h, C.c_void_p(ptr1), C.c_void_p(ptr2) if ptr2 else None,
                                          _dims(dims), 1 if replace else 0), "vh_push_back_device")

    def matchFeatures(self, method: int, Tr_delta=None):
        tr = None
        if Tr_delta is not None:
            tr = np.ascontiguousarray(Tr_delta, dtype=np.float64).reshape(16)
        _check(_lib().vh_match_features(self._h, int(method), _ptr(tr)), "vh_match_features")
        if self.outlier_removal:
            self.removeOutliers()

    def removeOutliers(self):
        """removeOutliers (src/remove_outliers.cpp:4-94), which the reference's matchFeatures
        runs right after matching (src/matcher.cpp:108); host side, flow and quad matches."""
        _check(_lib().vh_remove_outliers(self._h), "vh_remove_outliers")

    def bucketFeatures(self, max_features: int, bucket_width: float, bucket_height: float):
        _check(_lib().vh_bucket_features(self._h, int(max_features), float(bucket_width), float(bucket_height)),
               "vh_bucket_features")

    def getMatches(self) -> np.ndarray:
        n = C.c_int32(0)
        rc = _check(_lib().vh_get_matches(self._h, None, 0, C.byref(n)), "vh_get_matches", allow=(VH_ERR_CAPACITY,))
        out = np.zeros(n.value, P_MATCH_DTYPE)
        if n.value:
            _check(_lib().vh_get_matches(self._h, _ptr(out), n.value, C.byref(n)), "vh_get_matches")
        elif rc != VH_OK:  # an empty list from truncated feature sets is still an error
            raise VisoHipError(rc, "vh_get_matches")
        return out

    def getFeatures(self, which: int) -> np.ndarray:
        n = C.c_int32(0)
        _check(_lib().vh_get_features(self._h, which, None, 0, C.byref(n)), "vh_get_features", allow=(VH_ERR_CAPACITY,))
        out = np.zeros((n.value, 12), np.int32)
        if n.value:
            _check(_lib().vh_get_features(self._h, which, _ptr(out), n.value, C.byref(n)), "vh_get_features")
        return out

    def synchronize(self):
        _check(_lib().vh_synchronize(self._h), "vh_synchronize")

    def setStream(self, hip_stream: int | None):
        """Order every pushBack after `hip_stream` (a hipStream_t handle; 0 is the legacy
        default stream, a stream like any other); None removes the ordering."""
        if hip_stream is None:
            _check(_lib().vh_clear_stream(self._h), "vh_clear_stream")
        else:
            _check(_lib().vh_set_stream(self._h, C.c_void_p(int(hip_stream))), "vh_set_stream")

    def streamWaitImages(self, hip_stream: int):
        """Make `hip_stream` wait (device side) until the last pushBackDevice's images were consumed."""
        _check(_lib().vh_stream_wait_images(self._h, C.c_void_p(int(hip_stream))), "vh_stream_wait_images")


# ------------------------------------------------------------------ S streams in lock step
class StreamGroup:
    """S independent camera streams stepped together (vh_group_*): the
    multi-stream configuration, one sequence per stream, no exchange between
    streams.  Images are device pointers (e.g. torch `tensor.data_ptr()`)."""

    def __init__(self, n_streams: int, param: Params | None = None, device: int = 0,
                 max_features: int = 0, max_matches: int = 0):
        self.param = param if param is not None else Params.default()
        self.S = int(n_streams)
        h = C.c_void_p()
        _check(_lib().vh_group_create(C.byref(self.param), device, self.S, max_features, max_matches, C.byref(h)),
               "vh_group_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib().vh_group_destroy(self._h)
            self._h = None

    __del__ = close

    def pushBackDevice(self, ptr1: int, ptr2: int | None, stride_bytes: int, dims, replace=False):
        _check(_lib().vh_group_push_back_device(self._h, C.c_void_p(ptr1), C.c_void_p(ptr2) if ptr2 else None,
                                                int(stride_bytes), _dims(dims), 1 if replace else 0),
               "vh_group_push_back_device")

    def pushBack(self, I1, I2=None, dims=None, replace=False):
        """I1/I2: (S, H, bpl) uint8 numpy arrays."""
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        assert I1.ndim == 3 and I1.shape[0] == self.S
        if I2 is not None:
            I2 = np.ascontiguousarray(I2, dtype=np.uint8)
        if dims is None:
            dims = [I1.shape[2], I1.shape[1], I1.shape[2]]
        _check(_lib().vh_group_push_back(self._h, _ptr(I1), _ptr(I2), I1.shape[1] * I1.shape[2], _dims(dims),
                                         1 if replace else 0), "vh_group_push_back")

    def matchFeatures(self, method: int):
        _check(_lib().vh_group_match_features(self._h, int(method)), "vh_group_match_features")

    def deviceBytes(self) -> int:
        """Device memory held by the group (after the first pushBack)."""
        return int(_lib().vh_group_device_bytes(self._h))

    def removeOutliers(self, host_threads: int = 0):
        """Matcher.removeOutliers for every stream, on `host_threads` host workers (0: all)."""
        _check(_lib().vh_group_remove_outliers(self._h, int(host_threads)), "vh_group_remove_outliers")

    def matchFeaturesPrior(self, method: int, Tr_delta):
        """matchFeatures with a motion prior per stream: Tr_delta [S, 4, 4] (vh_group_match_features_prior)."""
        tr = np.ascontiguousarray(Tr_delta, dtype=np.float64).reshape(self.S, 16)
        _check(_lib().vh_group_match_features_prior(self._h, int(method), _ptr(tr)), "vh_group_match_features_prior")

    def getMatches(self, stream: int) -> np.ndarray:
        n = C.c_int32(0)
        rc = _check(_lib().vh_group_get_matches(self._h, stream, None, 0, C.byref(n)), "vh_group_get_matches",
                    allow=(VH_ERR_CAPACITY,))
        out = np.zeros(n.value, P_MATCH_DTYPE)
        if n.value:
            _check(_lib().vh_group_get_matches(self._h, stream, _ptr(out), n.value, C.byref(n)), "vh_group_get_matches")
        elif rc != VH_OK:
            raise VisoHipError(rc, "vh_group_get_matches")
        return out

    def getFeatures(self, stream: int, which: int) -> np.ndarray:
        n = C.c_int32(0)
        _check(_lib().vh_group_get_features(self._h, stream, which, None, 0, C.byref(n)), "vh_group_get_features",
               allow=(VH_ERR_CAPACITY,))
        out = np.zeros((n.value, 12), np.int32)
        if n.value:
            _check(_lib().vh_group_get_features(self._h, stream, which, _ptr(out), n.value, C.byref(n)),
                   "vh_group_get_features")
        return out

    def getMatchesAll(self, out: np.ndarray | None = None, cap_per_stream: int | None = None):
        """-> (records [S, cap_per_stream] P_MATCH_DTYPE, counts [S]); one wait for the whole group.
        Pass `out` (e.g. from pinned_empty) to reuse a buffer."""
        if out is None:
            if cap_per_stream is None:
                cap_per_stream = int(self.getCounts()[1].max(initial=0))
            out = np.zeros((self.S, max(cap_per_stream, 1)), P_MATCH_DTYPE)
        assert out.dtype == P_MATCH_DTYPE and out.ndim == 2 and out.shape[0] == self.S and out.flags.c_contiguous
        counts = np.zeros(self.S, np.int32)
        _check(_lib().vh_group_get_matches_all(self._h, _ptr(out), out.shape[1], _ptr(counts)), "vh_group_get_matches_all")
        return out, counts

    def downloadMatchesAsync(self, out: np.ndarray, counts: np.ndarray):
        """Start the asynchronous download of all streams' match lists into page-locked
        `out` [S, cap] / `counts` [S] (pinned_empty); waitDownload() before reading them."""
        assert out.dtype == P_MATCH_DTYPE and out.ndim == 2 and out.shape[0] == self.S and out.flags.c_contiguous
        assert counts.dtype == np.int32 and counts.shape == (self.S,) and counts.flags.c_contiguous
        _check(_lib().vh_group_download_matches_async(self._h, _ptr(out), out.shape[1], _ptr(counts)),
               "vh_group_download_matches_async")

    def waitDownload(self):
        _check(_lib().vh_group_wait_download(self._h), "vh_group_wait_download")

    def getCounts(self):
        nf = np.zeros((self.S, 4), np.int32)
        nm = np.zeros(self.S, np.int32)
        _check(_lib().vh_group_get_counts(self._h, _ptr(nf), _ptr(nm)), "vh_group_get_counts")
        return nf, nm

    def synchronize(self):
        _check(_lib().vh_group_synchronize(self._h), "vh_group_synchronize")

    def setStream(self, hip_stream: int | None):
        """Order every pushBack after `hip_stream` (a hipStream_t handle; 0 is the legacy
        default stream, a stream like any other); None removes the ordering."""
        if hip_stream is None:
            _check(_lib().vh_group_clear_stream(self._h), "vh_group_clear_stream")
        else:
            _check(_lib().vh_group_set_stream(self._h, C.c_void_p(int(hip_stream))), "vh_group_set_stream")

    def streamWaitImages(self, hip_stream: int):
        """Make `hip_stream` wait (device side) until the last pushBackDevice's images were consumed."""
        _check(_lib().vh_group_stream_wait_images(self._h, C.c_void_p(int(hip_stream))), "vh_group_stream_wait_images")

    def estimateMotion(self, ego: "EgoParams", rand3):
        """VisualOdometryStereo::estimateMotion (reference src/viso_stereo.cpp:54-157) on every stream's
        device-resident quad matches; rand3 [S, ransac_iters, 3] int32 rand() values -> (tr [S,6], ok [S], n_inliers [S])."""
        rand3 = np.ascontiguousarray(rand3, np.int32)
        assert rand3.shape == (self.S, ego.ransac_iters, 3)
        tr = np.zeros((self.S, 6), np.float64); ok = np.zeros(self.S, np.int32); ninl = np.zeros(self.S, np.int32)
        _check(_lib().vh_group_estimate_motion(self._h, C.byref(ego), _ptr(rand3), _ptr(tr), _ptr(ok), _ptr(ninl)), "vh_group_estimate_motion")
        return tr, ok.astype(bool), ninl

    def postBegin(self, cap_per_stream: int):
        """Start the download of this step's match lists into an internal page-locked slot (vh_group_post_begin)."""
        _check(_lib().vh_group_post_begin(self._h, int(cap_per_stream)), "vh_group_post_begin")

    def postFinish(self, age: int, max_features: int, bucket_width: float, bucket_height: float, host_threads: int = 0,
                   ego: "EgoParams" = None, rand3=None, want_lists: bool = True, list_cap: int = 0, mono: "MonoParams" = None, rand8=None):
        """removeOutliers + bucketFeatures (+ the stereo estimateMotion when `ego` is given, the monocular one when
        `mono` is) of the step begun `age` begins ago (vh_group_post_finish / vh_group_post_finish_mono)
        -> dict(tr, ok, n_inliers, lists, host_ms)."""
        S = self.S
        tr = np.zeros((S, 6), np.float64); ok = np.zeros(S, np.int32); ninl = np.zeros(S, np.int32)
        counts = np.zeros(S, np.int32); ms = C.c_double(0.0)
        out = None
        if want_lists:
            list_cap = int(list_cap) if list_cap else 4096
            out = np.zeros((S, list_cap), P_MATCH_DTYPE)
        r3 = None
        if ego is not None:
            r3 = np.ascontiguousarray(rand3, np.int32)
            assert r3.shape == (S, ego.ransac_iters, 3)
        if mono is not None:
            assert ego is None
            r8 = np.ascontiguousarray(rand8, np.int32)
            assert r8.shape == (S, mono.ransac_iters, 8)
            _check(_lib().vh_group_post_finish_mono(self._h, int(age), int(max_features), C.c_float(bucket_width), C.c_float(bucket_height),
                                                    int(host_threads), C.byref(mono), _ptr(r8), _ptr(tr), _ptr(ok),
                                                    _ptr(ninl), _ptr(out), int(list_cap), _ptr(counts), C.byref(ms)), "vh_group_post_finish_mono")
        else:
            _check(_lib().vh_group_post_finish(self._h, int(age), int(max_features), C.c_float(bucket_width), C.c_float(bucket_height),
                                               int(host_threads), C.byref(ego) if ego is not None else None, _ptr(r3), _ptr(tr), _ptr(ok),
                                               _ptr(ninl), _ptr(out), int(list_cap), _ptr(counts), C.byref(ms)), "vh_group_post_finish")
        lists = [out[s, :counts[s]].copy() for s in range(S)] if want_lists else None
        return {"tr": tr, "ok": ok.astype(bool), "n_inliers": ninl, "lists": lists, "counts": counts, "host_ms": ms.value}

    def postDeviceConfig(self, steps_per_batch: int = 64, batches: int = 3, lanes_per_wave: int = 16):
        """Shape of the device post stage's pipeline (vh_group_post_device_config); the defaults are the library's own
        (64 steps per batch, 3 batches, 16 lists per wave: the throughput-optimal shape, up to a second of latency)."""
        _check(_lib().vh_group_post_device_config(self._h, int(steps_per_batch), int(batches), int(lanes_per_wave)), "vh_group_post_device_config")

    def postBeginDevice(self, cap_per_stream: int, max_features: int, bucket_width: float, bucket_height: float,
                        ego: "EgoParams" = None, rand3=None, mono: "MonoParams" = None, rand8=None, want_lists: bool = False):
        """This step's match lists enter the device post stage: removeOutliers -> bucketFeatures -> estimateMotion, all on
        the GPU (vh_group_post_begin_device)."""
        r3 = r8 = None
        if ego is not None:
            r3 = np.ascontiguousarray(rand3, np.int32)
            assert r3.shape == (self.S, ego.ransac_iters, 3)
        if mono is not None:
            r8 = np.ascontiguousarray(rand8, np.int32)
            assert r8.shape == (self.S, mono.ransac_iters, 8)
        _check(_lib().vh_group_post_begin_device(self._h, int(cap_per_stream), int(max_features), C.c_float(bucket_width), C.c_float(bucket_height),
                                                 C.byref(ego) if ego is not None else None, _ptr(r3),
                                                 C.byref(mono) if mono is not None else None, _ptr(r8), 1 if want_lists else 0),
               "vh_group_post_begin_device")

    def postFinishDevice(self, age: int, want_lists: bool = False, list_cap: int = 4096, estimator: bool = True, strict: bool = True):
        """Results of the step begun `age` begins ago (vh_group_post_finish_device) -> dict(tr, ok, n_inliers, lists, counts).
        strict=False: a refused list does not raise; its stream reports counts = -1 and the dict carries the code as "rc"."""
        S = self.S
        tr = np.zeros((S, 6), np.float64); ok = np.zeros(S, np.int32); ninl = np.zeros(S, np.int32); counts = np.zeros(S, np.int32)
        out = np.zeros((S, int(list_cap)), P_MATCH_DTYPE) if want_lists else None
        rc = _lib().vh_group_post_finish_device(self._h, int(age), _ptr(tr) if estimator else None, _ptr(ok) if estimator else None,
                                                _ptr(ninl) if estimator else None, _ptr(out), int(list_cap) if want_lists else 0, _ptr(counts))
        if strict or rc in (VH_ERR_INVALID_ARG, VH_ERR_STATE, VH_ERR_HIP, VH_ERR_NO_DEVICE):
            _check(rc, "vh_group_post_finish_device")
        lists = [out[s, :max(int(counts[s]), 0)].copy() for s in range(S)] if want_lists else None
        return {"tr": tr, "ok": ok.astype(bool), "n_inliers": ninl, "lists": lists, "counts": counts, "rc": rc}

    def estimateMotionMono(self, mono: "MonoParams", rand8):
        """VisualOdometryMono::estimateMotion (reference src/viso_mono.cpp:41-160) on every stream's device-resident
        flow (or quad) matches; rand8 [S, ransac_iters, 8] int32 rand() values -> (tr [S,6], ok [S], n_inliers [S])."""
        rand8 = np.ascontiguousarray(rand8, np.int32)
        assert rand8.shape == (self.S, mono.ransac_iters, 8)
        tr = np.zeros((self.S, 6), np.float64); ok = np.zeros(self.S, np.int32); ninl = np.zeros(self.S, np.int32)
        _check(_lib().vh_group_estimate_motion_mono(self._h, C.byref(mono), _ptr(rand8), _ptr(tr), _ptr(ok), _ptr(ninl)), "vh_group_estimate_motion_mono")
        return tr, ok.astype(bool), ninl

    def searchStats(self):
        """-> (speculative loops in use?, last observed share of re-searched queries or -1)."""
        sp = C.c_int32(0); rate = C.c_double(-1.0)
        _check(_lib().vh_group_search_stats(self._h, C.byref(sp), C.byref(rate)), "vh_group_search_stats")
        return bool(sp.value), rate.value

    def debugFailNextAlloc(self):
        """Test hook: the group's next device allocation fails once."""
        _check(_lib().vh_group_debug_fail_next_alloc(self._h), "vh_group_debug_fail_next_alloc")

    def profileEnable(self, on: bool = True):
        _check(_lib().vh_group_profile_enable(self._h, 1 if on else 0), "vh_group_profile_enable")

    def profileReset(self):
        _check(_lib().vh_group_profile_reset(self._h), "vh_group_profile_reset")

    def profileRead(self, name: str):
        ms = C.c_double(0)
        n = C.c_int64(0)
        _check(_lib().vh_group_profile_read(self._h, name.encode(), C.byref(ms), C.byref(n)), "vh_group_profile_read")
        return ms.value, n.value


# ------------------------------------------------------------------ stateless primitives
def compute_features(param: Params, img, dims, device: int = 0, planes: bool = False, cap: int | None = None):
    """Matcher::computeFeatures (reference src/matcher.cpp:585-672) ->
    (max1 [n1,12], max2 [n2,12][, I_du, I_dv])."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if cap is None:
        cap = 4 * (dims[0] // (param.nms_n + 1) + 1) * (dims[1] // (param.nms_n + 1) + 1)
    m1 = np.zeros((cap, 12), np.int32)
    m2 = np.zeros((cap, 12), np.int32)
    n1, n2 = C.c_int32(0), C.c_int32(0)
    du = dv = None
    if planes:
        if param.half_resolution:
            w2 = dims[0] // 2
            shape = (dims[1] // 2, w2 + 15 - (w2 - 1) % 16)
        else:
            shape = (dims[1], dims[2])
        du = np.zeros(shape, np.uint8)
        dv = np.zeros(shape, np.uint8)
    _check(_lib().vh_compute_features(C.byref(param), device, _ptr(img), _dims(dims), _ptr(m1), cap, C.byref(n1),
                                      _ptr(m2), cap, C.byref(n2), _ptr(du), _ptr(dv)), "vh_compute_features")
    r = (m1[:n1.value].copy(), m2[:n2.value].copy())
    return r + (du, dv) if planes else r


def filters(img, device: int = 0):
    """sobel5x5 / blob5x5 / checkerboard5x5 (reference src/filter.h:80-96) on
    the valid interior -> (du, dv, f1, f2)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, bpl = img.shape
    du = np.empty((h, bpl), np.uint8); dv = np.empty((h, bpl), np.uint8)
    f1 = np.empty((h, bpl), np.int16); f2 = np.empty((h, bpl), np.int16)
    _check(_lib().vh_filters(device, _ptr(img), bpl, h, _ptr(du), _ptr(dv), _ptr(f1), _ptr(f2)), "vh_filters")
    return du, dv, f1, f2


def create_index(param: Params, dims, m, device: int = 0):
    """Matcher::createIndexVector (reference src/matcher.cpp:194-214) as CSR."""
    m, n = _feat(m)
    ubn = -(-int(dims[0]) // param.match_binsize)
    vbn = -(-int(dims[1]) // param.match_binsize)
    bs = np.zeros(4 * ubn * vbn + 1, np.int32)
    lst = np.zeros(max(n, 1), np.int32)
    _check(_lib().vh_create_index(C.byref(param), device, _dims(dims), _ptr(m), n, _ptr(bs), _ptr(lst)), "vh_create_index")
    return bs, lst[:n]


def match_all(param: Params, dims, m1, m2, flow: bool = True, device: int = 0):
    """Matcher::findMatch (reference src/matcher.cpp:216-272) for every query."""
    m1, n1 = _feat(m1)
    m2, n2 = _feat(m2)
    best = np.zeros(max(n1, 1), np.int32)
    _check(_lib().vh_match_all(C.byref(param), device, _dims(dims), _ptr(m1), n1, _ptr(m2), n2, 1 if flow else 0,
                               _ptr(best)), "vh_match_all")
    return best[:n1]


def match_all_prior(param: Params, dims, m1, m2, u_: float, v_: float, flow: bool = True, device: int = 0):
    """Matcher::findMatch with the u_,v_ prediction term (reference src/matcher.cpp:257-262)."""
    m1, n1 = _feat(m1)
    m2, n2 = _feat(m2)
    best = np.zeros(max(n1, 1), np.int32)
    _check(_lib().vh_match_all_prior(C.byref(param), device, _dims(dims), _ptr(m1), n1, _ptr(m2), n2,
                                     1 if flow else 0, float(u_), float(v_), _ptr(best)), "vh_match_all_prior")
    return best[:n1]


def estimate_motion_stereo(ego: EgoParams, match_lists, rand3, device: int = 0):
    """VisualOdometryStereo::estimateMotion (reference src/viso_stereo.cpp:54-157), batched over
    `match_lists` (a list of p_match arrays); rand3 [n_sets, ransac_iters, 3] int32 rand() values.
    -> (tr [n,6], ok [n] bool, [inlier index arrays])."""
    lists = [np.ascontiguousarray(m, dtype=P_MATCH_DTYPE) for m in match_lists]
    n = len(lists)
    offsets = np.zeros(n + 1, np.int32)
    offsets[1:] = np.cumsum([len(m) for m in lists])
    pm = np.concatenate(lists) if offsets[-1] else np.zeros(0, P_MATCH_DTYPE)
    rand3 = np.ascontiguousarray(rand3, np.int32)
    assert rand3.shape == (n, ego.ransac_iters, 3)
    tr = np.zeros((n, 6), np.float64); ok = np.zeros(n, np.int32); ninl = np.zeros(n, np.int32)
    inl = np.zeros(max(int(offsets[-1]), 1), np.int32)
    _check(_lib().vh_estimate_motion_stereo(C.byref(ego), device, n, _ptr(pm), _ptr(offsets), _ptr(rand3), _ptr(tr), _ptr(ok),
                                            _ptr(ninl), _ptr(inl)), "vh_estimate_motion_stereo")
    return tr, ok.astype(bool), [inl[offsets[s]:offsets[s] + ninl[s]].copy() for s in range(n)]


def estimate_motion_mono(mono: MonoParams, match_lists, rand8, device: int = 0):
    """VisualOdometryMono::estimateMotion (reference src/viso_mono.cpp:41-160), batched over `match_lists`;
    rand8 [n_sets, ransac_iters, 8] int32 rand() values.  -> (tr [n,6], ok [n] bool, [inlier index arrays])."""
    lists = [np.ascontiguousarray(m, dtype=P_MATCH_DTYPE) for m in match_lists]
    n = len(lists)
    offsets = np.zeros(n + 1, np.int32)
    offsets[1:] = np.cumsum([len(m) for m in lists])
    pm = np.concatenate(lists) if offsets[-1] else np.zeros(0, P_MATCH_DTYPE)
    rand8 = np.ascontiguousarray(rand8, np.int32)
    assert rand8.shape == (n, mono.ransac_iters, 8)
    tr = np.zeros((n, 6), np.float64); ok = np.zeros(n, np.int32); ninl = np.zeros(n, np.int32)
    inl = np.zeros(max(int(offsets[-1]), 1), np.int32)
    _check(_lib().vh_estimate_motion_mono(C.byref(mono), device, n, _ptr(pm), _ptr(offsets), _ptr(rand8), _ptr(tr), _ptr(ok),
                                          _ptr(ninl), _ptr(inl)), "vh_estimate_motion_mono")
    return tr, ok.astype(bool), [inl[offsets[s]:offsets[s] + ninl[s]].copy() for s in range(n)]


def remove_outliers(pm) -> np.ndarray:
    """removeOutliers (reference src/remove_outliers.cpp:4-94) on p_match records; host only."""
    pm = np.ascontiguousarray(pm, dtype=P_MATCH_DTYPE).copy()
    n = C.c_int32(0)
    _check(_lib().vh_remove_outliers_pm(_ptr(pm), len(pm), C.byref(n)), "vh_remove_outliers_pm")
    return pm[:n.value].copy()


def remove_outliers_device(lists, lanes_per_wave: int = 1, max_features: int = 0, bucket_width: float = 50.0, bucket_height: float = 50.0,
                           device: int = 0, out_cap: int = 0, strict: bool = True):
    """removeOutliers (and bucketFeatures when max_features >= 1) of several match lists at once on the GPU
    (vh_remove_outliers_device) -> (lists, triangles per list, sweep kernel ms).  strict=False: an error code does not
    raise; the healthy lists are returned (a refused list comes back empty) with the code as a fourth element."""
    lists = [np.ascontiguousarray(m, dtype=P_MATCH_DTYPE) for m in lists]
    n = len(lists)
    stride = max([len(m) for m in lists] + [1])
    pm = np.zeros((n, stride), P_MATCH_DTYPE)
    for l, m in enumerate(lists):
        pm[l, :len(m)] = m
    counts = np.array([len(m) for m in lists], np.int32)
    out_cap = int(out_cap) if out_cap else stride
    out = np.zeros((n, out_cap), P_MATCH_DTYPE)
    oc = np.zeros(n, np.int32); ntri = np.zeros(n, np.int32); ms = C.c_float(0.0)
    rc = _lib().vh_remove_outliers_device(device, n, _ptr(pm), stride, _ptr(counts), int(lanes_per_wave), int(max_features),
                                          C.c_float(bucket_width), C.c_float(bucket_height), _ptr(out), out_cap, _ptr(oc), _ptr(ntri),
                                          C.byref(ms))
    if strict:
        _check(rc, "vh_remove_outliers_device")
        return [out[l, :oc[l]].copy() for l in range(n)], ntri, ms.value
    return [out[l, :min(int(oc[l]), out_cap)].copy() for l in range(n)], ntri, ms.value, rc


def match(param: Params, dims, method: int, m1p=None, m2p=None, m1c=None, m2c=None, device: int = 0, cap=None):
    """Matcher::matching (reference src/matcher.cpp:274-344) on given feature arrays."""
    sets = [_feat(m) for m in (m1p, m2p, m1c, m2c)]
    if cap is None:
        cap = max(s[1] for s in sets) + 1
    out = np.zeros(cap, P_MATCH_DTYPE)
    n = C.c_int32(0)
    _check(_lib().vh_match(C.byref(param), device, _dims(dims), int(method),
                           _ptr(sets[0][0]), sets[0][1], _ptr(sets[1][0]), sets[1][1],
                           _ptr(sets[2][0]), sets[2][1], _ptr(sets[3][0]), sets[3][1],
                           _ptr(out), cap, C.byref(n)), "vh_match")
    return out[:n.value].copy()
