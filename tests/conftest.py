import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes host mirror over libviso_hip.so)."""
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def ob():
    """oracle.binding (test infrastructure)."""
    b = entry.load_oracle()
    if not os.path.exists(b.ORACLE_SO):
        b.build(ref=False)
    return b


@pytest.fixture(scope="session")
def oracle(ob):
    return ob.Oracle()


@pytest.fixture(scope="session")
def reference(ob):
    """The reference's own code (oracle/_ref); absent -> skip."""
    if not ob.Reference.available():
        if os.path.isdir(os.path.join(ob.REFERENCE_ROOT, "src")):
            ob.build(ref=True)
        else:
            pytest.skip("oracle/_ref/libviso_ref.so not built (needs /root/reference)")
    return ob.Reference()


@pytest.fixture(scope="session")
def gpu(pkg):
    if pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the HIP path has no CPU fallback)")
    return 0


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and (f.startswith("small_") or f.startswith("dense_")))


def load_golden(name, pkg, params_cls):
    """-> (params, dims, Iprev, Icur, arrays) with the images re-generated."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    W, H, blur, gain, seed, dx, dy = [int(v) for v in z["gen"]]
    over = {str(k): int(v) for k, v in z["params"]}
    p = params_cls.default(**over)
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    Ip = pkg.synth.frame(W, H, 0, 0, blur, gain, seed)
    Ic = pkg.synth.frame(W, H, dx, dy, blur, gain, seed)
    return p, dims, Ip, Ic, z
