"""GPU: the reference-signature C shim (libmfuoco_gpu_debug.so) driven by a C program that restates the assertions of
the reference's own test programs (c-lwe-snarks_amd/host/test_shim.c): same function names and call sequences."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_shim_properties():
    exe = os.path.join(ROOT, "c-lwe-snarks_amd", "host", "test_shim")
    if not os.path.exists(exe):
        pytest.fail("host/test_shim has not been built (make -C c-lwe-snarks_amd shim); the shim needs gmp.h at build time")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "snark ok" in r.stdout
