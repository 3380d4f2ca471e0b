"""CPU: libmfhip.so loads and exports every entry point include/mfhip.h declares (no compute without a GPU), and the
Python binding refuses to work without a HIP device instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    import c_lwe_snarks_amd as mf

    lib = mf.load_library()
    hdr = open(os.path.join(ROOT, "include", "mfhip.h")).read()
    declared = sorted(set(re.findall(r"\b(mfh_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mfhip.h but not exported by libmfhip.so"
    assert set(mf.EXPORTS) <= set(declared)


def test_no_cpu_fallback():
    import torch

    import c_lwe_snarks_amd as mf

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(mf.MfhError):
        mf.Context(mf.DEBUG, 0)
    # and the C ABI itself refuses
    lib = mf.load_library()
    h = ctypes.c_void_p()
    cp = mf._CParams(1470, 736, 256, 64)
    assert lib.mfh_ctx_create(ctypes.byref(h), 0, ctypes.byref(cp)) != 0


def test_product_does_not_link_the_oracle():
    import subprocess

    out = subprocess.run(["ldd", os.path.join(ROOT, "c-lwe-snarks_amd", "libmfhip.so")], capture_output=True, text=True).stdout
    assert "mf_oracle" not in out and "mfref" not in out and "libgmp" not in out
    for fn in os.listdir(os.path.join(ROOT, "c-lwe-snarks_amd", "csrc")):
        src = open(os.path.join(ROOT, "c-lwe-snarks_amd", "csrc", fn)).read()
        assert "mf_oracle" not in src and "oracle/" not in src


def test_dist_header_symbols_are_exported():
    """libmfuoco_gpu_dist.so (C entry points for N GPUs, RCCL called directly) exports what host/include/mfuoco/mfuoco_dist.h declares
    and really links librccl"""
    import subprocess

    hdr = open(os.path.join(ROOT, "c-lwe-snarks_amd", "host", "include", "mfuoco", "mfuoco_dist.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only
    declared = sorted(set(re.findall(r"\b(mfuoco_[a-z0-9_]+)\s*\(", hdr)))
    assert "mfuoco_prover_batch_sharded" in declared and "mfuoco_comm_create" in declared
    for so in ("libmfuoco_gpu_dist.so", "libmfuoco_gpu_dist_debug.so"):
        path = os.path.join(ROOT, "c-lwe-snarks_amd", so)
        if not os.path.exists(path):
            pytest.fail(f"{so} has not been built (make -C c-lwe-snarks_amd dist)")
        syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        for name in declared:
            assert re.search(rf"\bT {name}\b", syms), f"{name} not exported by {so}"
        und = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
        for name in ("ncclReduceScatter", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclAllReduce", "ncclCommInitRank", "mfh_prove_batch_partial"):
            assert name in und, f"{so} does not call {name}"
        assert "librccl" in subprocess.run(["ldd", path], capture_output=True, text=True).stdout
        # the product library carries the RCCL transport only: the host-shared-memory rehearsal transport of the tests (host/mfuoco_dist_rehearsal.c) is
        # compiled into the test drivers, so no rehearsal symbol, no shm_open / mmap and no spin-wait scaffolding is in the .so
        allsyms = subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout
        assert "rehearsal" not in allsyms and "shm_open" not in allsyms and "mmap" not in allsyms
        assert "rehearsal" not in subprocess.run(["strings", path], capture_output=True, text=True).stdout.lower()


def test_shim_header_symbols_are_exported():
    """libmfuoco_gpu.so / libmfuoco_gpu_debug.so export every FUNCTION host/include/mfuoco/mangiafuoco_api.h declares: the reference's names (src/aes.h, src/entropy.h,
    src/lwe.h, src/ssp.h, src/snark.h) and the additions (batch prover / verifier / decryption, files, device and resident-CRS controls); static inline helpers of the
    header (regev_encrypt, mpz_dotp, rng_gen, rand_modp) are not symbols"""
    import subprocess

    hdr = open(os.path.join(ROOT, "c-lwe-snarks_amd", "host", "include", "mfuoco", "mangiafuoco_api.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"^\s*#.*?$", "", hdr, flags=re.M)          # macros
    hdr = re.sub(r"static inline[^{;]*\{[^}]*\}", "", hdr)  # inline helpers
    declared = sorted(set(re.findall(r"\b([a-z][a-z0-9_]+)\s*\([^;{]*\)\s*;", hdr)) | set(re.findall(r"\(\*([a-z][a-z0-9_]+)\([^;{]*\)\)\[", hdr)))  # (also `T (*f(args))[N];`)
    for must in ("prover", "verifier", "setup", "eval_poly", "regev_encrypt2", "regev_decrypt", "aesctr_prg", "rng_seek", "mpz2_urandomb", "ct_smudge", "random_ssp",
                 "mfuoco_prover_batch", "mfuoco_gpu_set_resident_crs", "mfuoco_gpu_device", "mfuoco_crs_map", "mfuoco_rows_map"):
        assert must in declared, must
    for so in ("libmfuoco_gpu.so", "libmfuoco_gpu_debug.so"):
        path = os.path.join(ROOT, "c-lwe-snarks_amd", so)
        if not os.path.exists(path):
            pytest.fail(f"{so} has not been built (make -C c-lwe-snarks_amd shim)")
        syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        missing = [n for n in declared if not re.search(rf"\bT {n}\b", syms)]
        assert not missing, (so, missing)
