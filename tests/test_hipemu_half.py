"""The CPU logic build's half-precision conversions (tests/hipemu/hip/hip_runtime.h: g++ 11 has no _Float16 in C++) against NumPy's float16:
every half-precision value, every midpoint between two neighbours (round to nearest even), subnormals, overflow.  The chamfer screening
(csrc/select_chamfer.hip) cuts its coordinates into half-precision pieces; on the CPU build these functions stand in for v_cvt_f16_f32 / v_cvt_f32_f16."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_half_conversions_match_numpy(tmp_path):
    header = open(os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h")).read()
    a = header.index("static inline unsigned short hipemu_f32_to_f16")
    b = header.index("// 32x32x16 f16 MFMA")
    src = ("#include <cstring>\n"
           "static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }\n"
           "static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }\n" + header[a:b] +
           'extern "C" void conv(const float* x, unsigned short* o, float* back, int n) {\n'
           "    for (int i = 0; i < n; ++i) { o[i] = hipemu_f32_to_f16(x[i]); back[i] = hipemu_f16_to_f32(o[i]); }\n}\n")
    cpp, so = tmp_path / "half.cpp", tmp_path / "half.so"
    cpp.write_text(src)
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", str(cpp), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    rng = np.random.default_rng(0)
    halves = np.arange(0, 0x7bff, dtype=np.uint16).view(np.float16).astype(np.float32)            # every finite non-negative half
    mid = ((halves[:-1].astype(np.float64) + halves[1:].astype(np.float64)) / 2).astype(np.float32)  # exact in float32: the ties
    x = np.concatenate([rng.normal(0, 1, 100000), rng.normal(0, 1e-5, 100000), rng.normal(0, 1e-7, 50000), rng.normal(0, 3e4, 50000),
                        [0.0, -0.0, 65504, 65519.9, 65520, 70000, -65520, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.99e-8, 1e-9],
                        mid, -mid, halves, -halves]).astype(np.float32)
    out = np.empty(x.size, np.uint16); back = np.empty(x.size, np.float32)
    lib.conv(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), back.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.size))
    with np.errstate(over="ignore"):
        want = x.astype(np.float16)
    assert np.array_equal(out, want.view(np.uint16))
    assert np.array_equal(back, want.astype(np.float32))
