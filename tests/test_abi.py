"""The C-ABI library loads and exports every symbol include/ssdr_al.h declares; without a GPU the product
fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest

from conftest import GPU_LIB, ROOT, _have_gpu


def _declared():
    hdr = open(os.path.join(ROOT, "include", "ssdr_al.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ssdr_\w+)\s*\(", hdr)))


def test_product_library_exports_every_declared_symbol():
    assert os.path.exists(GPU_LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(GPU_LIB)
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_emu_library_exports_the_same_abi(emu_lib):
    lib = ctypes.CDLL(emu_lib)
    assert not [n for n in _declared() if not hasattr(lib, n)]


def test_no_cpu_fallback_without_device():
    if _have_gpu():
        pytest.skip("GPU present")
    import numpy as np
    from ssdr_al import _lib, knn
    _lib.use(GPU_LIB)
    try:
        with pytest.raises(_lib.SsdrError) as e:
            knn.knn(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), 1)
        assert e.value.status == 2 and "no CPU fallback" in str(e.value)
    finally:
        _lib.use(None)


def test_missing_library_is_an_error(tmp_path):
    from ssdr_al import _lib
    _lib.use(str(tmp_path / "nope.so"))
    try:
        with pytest.raises(_lib.SsdrError):
            _lib.lib()
    finally:
        _lib.use(None)


def test_fork_after_init_is_refused_and_fork_before_init_works(emu_lib):
    """SURVEY 8b "Threading": the reference calls knn_search from forked DataLoader workers.  Loading the library never touches HIP;
    a child forked BEFORE the first op initialises its own context, a child forked AFTER it gets a clear error (no hang)."""
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from ssdr_al import _lib, knn
_lib.use(%r)
p = np.random.default_rng(0).random((200, 3), dtype=np.float32)
def child(expect_ok):
    pid = os.fork()
    if pid == 0:
        try:
            knn.knn(p, p, 4)
            os._exit(0 if expect_ok else 3)
        except _lib.SsdrError as e:
            os._exit(3 if expect_ok else (0 if "fork" in str(e) else 4))
    return os.waitpid(pid, 0)[1]
assert child(True) == 0, "a child forked before the first op must be able to use the library"
knn.knn(p, p, 4)                       # the parent initialises now
assert child(False) == 0, "a child forked after the parent's init must be refused with the fork message"
print("ok")
"""
    from conftest import PKG, ROOT
    r = subprocess.run([sys.executable, "-c", code % (ROOT, PKG, emu_lib)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
