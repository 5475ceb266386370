import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ssdr-al_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "libssdr_al_emu.so")
GPU_LIB = os.path.join(PKG, "libssdr_al.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


_GPU = None


def _have_gpu():
    """A HIP device the product library can initialise (no framework involved: a box without torch must not silently skip the
    parity tests).  Without libssdr_al.so the answer comes from the device nodes, so a missing build FAILS the gpu tests."""
    global _GPU
    if _GPU is None:
        if os.path.exists(GPU_LIB):
            import ctypes
            try:
                lib = ctypes.CDLL(GPU_LIB)
                lib.ssdr_init.argtypes = [ctypes.c_int]
                _GPU = lib.ssdr_init(0) == 0
            except OSError:
                _GPU = False
        else:
            _GPU = os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)
    return _GPU


@pytest.fixture(scope="session")
def emu_lib():
    """CPU logic-test build of the HIP sources (tests/hipemu) — test infrastructure, never the product."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc"), "emu"])
    return EMU_LIB


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Runs a parity test twice: against the CPU logic build (not gpu) and against the gfx950 build (gpu)."""
    from ssdr_al import _lib
    if request.param == "emu":
        path = request.getfixturevalue("emu_lib")
    else:
        if not _have_gpu():
            pytest.skip("no GPU")
        path = GPU_LIB
        assert os.path.exists(path), "libssdr_al.so missing: run __graft_entry__.build()"
    _lib.use(path)
    yield request.param
    _lib.use(None)


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.c()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    bad = bits(a) != bits(b)
    assert not bad.any(), "%s: %d of %d elements differ, first at %s" % (what, bad.sum(), bad.size, np.argwhere(bad)[0])
