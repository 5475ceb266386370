"""AddressSanitizer / UBSan runs of the CPU builds (the C oracle and the CPU logic build of the HIP sources).
GPU sanitizers are not available on the pool; these are the CPU builds only."""
import os
import subprocess
import sys

from conftest import PKG, ROOT


def test_cpu_builds_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(PKG, "csrc"), "emu-san"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_san.so"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, OMP_NUM_THREADS="2", SSDR_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_san.so"),
               # the CPU stand-in runs work-items as ucontext fibers (ASan cannot follow swapcontext stacks: no fake stacks, no leak pass)
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_san_worker.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert "sanitized pass ok" in r.stdout
