"""CPU, build container only: the reference's UNMODIFIED driver sources (compiled where they lie, with the reference's own
headers) link against libmfuoco_gpu*.so -- the drop-in claim of INTEGRATION.md.  Nothing is run (no GPU here) and nothing
is copied; skipped where /root/reference is absent (the GPU box).  <flint/nmod_poly.h> resolves to the shim's
layout-compatible header because FLINT is not installed in this image."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
PKG = os.path.join(ROOT, "c-lwe-snarks_amd")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists("/opt/conda/include/gmp.h"),
                                reason="reference sources or gmp.h not present")


@pytest.mark.parametrize("prog,ndebug", [("benchmark_snark", True), ("benchmark_lwe", True), ("benchmark_eval", True), ("test_snark", False), ("test_lwe", False),
                                         ("test_entropy", False), ("test_ssp", False), ("test_aes", False)])
def test_reference_driver_links_unchanged(tmp_path, prog, ndebug):
    lib = "mfuoco_gpu" if ndebug else "mfuoco_gpu_debug"
    if not os.path.exists(os.path.join(PKG, f"lib{lib}.so")):
        pytest.skip("shim not built")
    out = tmp_path / prog
    cmd = ["gcc", "-std=c11", "-w", "-DNDEBUG" if ndebug else "-UNDEBUG", "-O1", f"-I{REF}", f"-I{PKG}/host/include/mfuoco",
           "-idirafter", "/opt/conda/include", os.path.join(REF, prog + ".c"), "-o", str(out), f"-L{PKG}", f"-l{lib}", "-lmfhip",
           "/usr/lib/x86_64-linux-gnu/libgmp.so.10", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # every reference-library symbol the driver needs is provided by the shim (none by the reference's .c files)
    nm = subprocess.run(["nm", "-u", str(out)], capture_output=True, text=True).stdout
    assert shutil.which("nm") and ("prover" in nm or "regev_encrypt2" in nm or "aesctr_prg" in nm or "random_ssp" in nm or "aesctr_init" in nm)
    if prog == "benchmark_eval":  # src/benchmark_eval.c:30-86: D encryptions, then ONE eval_poly over the mapped ./coeffs -- all three from the shim
        assert "eval_poly" in nm and "regev_encrypt2" in nm and "ct_export" in nm
