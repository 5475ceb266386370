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
