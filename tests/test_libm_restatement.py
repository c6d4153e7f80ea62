"""csrc/swd_libm.h (the device's exp / log1p, used by the quaternary decoder) against the host C library, bit for
bit.  The reference's bp4_osd calls glibc's exp and log1p; its goldens were recorded on an FMA-capable x86-64 with
glibc 2.35, whose exp has an FMA build -- the restatement follows that build, so the comparison needs such a host."""
import ctypes as C
import os
import platform
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return platform.machine() == "x86_64" and " fma " in f.read().replace("\n", " ")
    except OSError:
        return False


@pytest.fixture(scope="module")
def shim():
    if not _host_has_fma():
        pytest.skip("host C library would select its non-FMA exp here; the restatement follows the FMA build")
    d = tempfile.mkdtemp(prefix="swd_libm_")
    so = os.path.join(d, "libm_check.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "slidingwindowdecoder_amd", "csrc"),
                           os.path.join(ROOT, "tests", "csrc", "libm_check.c"), "-o", so, "-lm"])
    L = C.CDLL(so)
    for f in (L.swd_check_exp, L.swd_check_log1p):
        f.restype = C.c_long
        f.argtypes = [C.c_long, C.c_uint64, C.POINTER(C.c_double)]
    return L


def test_exp_table_is_current():
    """the committed table equals what scripts/gen_exp_table.py derives"""
    path = os.path.join(ROOT, "slidingwindowdecoder_amd", "csrc", "swd_exp_table.h")
    with tempfile.TemporaryDirectory(prefix="swd_exp_") as d:  # never touches the tracked header (its mtime drives make)
        fresh = os.path.join(d, "swd_exp_table.h")
        subprocess.check_call(["python3", os.path.join(ROOT, "scripts", "gen_exp_table.py"), fresh], stdout=subprocess.DEVNULL)
        assert open(fresh).read() == open(path).read()


def test_exp_matches_host_libm(shim):
    x = C.c_double(0.0)
    bad = shim.swd_check_exp(3_000_000, 12345, C.byref(x))
    assert bad == 0, f"{bad} arguments differ, first {x.value!r}"


def test_log1p_matches_host_libm(shim):
    x = C.c_double(0.0)
    bad = shim.swd_check_log1p(3_000_000, 999, C.byref(x))
    assert bad == 0, f"{bad} arguments differ, first {x.value!r}"
