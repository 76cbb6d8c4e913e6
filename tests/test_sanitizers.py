"""`make check-asan`: the host C — lv2_*.c, mtr_setup.c, oracle/mtr_oracle.c — rebuilt with
-fsanitize=address,undefined and the CPU suite run against it.  This test IS that target (40 s); it skips inside
the instrumented run itself and where the compiler ships no sanitizer runtime."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_make_check_asan():
    if os.environ.get("MTR_ORACLE_SO") or os.environ.get("MTR_PLUGIN_SO"):
        pytest.skip("already inside the instrumented run")
    if not shutil.which("gcc") or not shutil.which("make"):
        pytest.skip("no gcc / make")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("this gcc has no libasan")
    out = subprocess.run(["make", "-C", ROOT, "check-asan"], capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
