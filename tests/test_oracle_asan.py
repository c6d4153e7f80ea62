"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not
available on this pool).  The reference has leaks and reads of uninitialised memory exactly where the oracle
restates it (src/include/mod2sparse_extra.cpp:8, bpgd.cpp:357-358); the oracle must be clean on the golden
subsets that reach every decoder: osd_window (OSD-0 / CS / E, shortening ties), the guessing decoders, the [[144]]
sliding trace, the rank-deficient window and bp4_osd incl. the SHYPS matrices."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_subset_under_asan_ubsan():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libswd_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=libasan, SWD_ORACLE_SO=so,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",  # the interpreter's own exit-time leaks are not ours
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    sel = ("test_bb72_osd_window or test_bb72_guessing or test_bb144_rank_deficient_inconsistent or test_bb144_gdg_trace "
           "or test_bp4_shyps_oracle_matches_reference or test_bp4_camel_decode_oracle_matches_reference or test_bb72_full_history_values")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle_golden.py",
                        "tests/test_oracle_bp4.py", "-k", sel], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout
