"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU).  oracle/san_check.c runs the stream,
LWE, SSP and SNARK layers end to end at a small instance for both moduli; any sanitizer report aborts it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("compiler has no sanitizer runtime: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([os.path.join(ROOT, "oracle", "_build", "san_check")], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "san_check ok" in run.stdout
