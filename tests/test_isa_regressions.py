"""Guards on the generated gfx950 ISA of the dominant kernel (no GPU needed: hipcc cross-compiles).

The AES inner loop of k_eval must form a lookup address with exactly one VALU instruction (v_perm_b32 or v_bitop3_b32):
that holds only while the T-table is the kernel's first LDS object (LDS address 0).  When it was not, the compiler added
one v_add_u32 per lookup (203 per block) and the kernel ran 19 % slower -- see DESIGN.md section 4.1."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "mfhip.s"
    src = os.path.join(ROOT, "c-lwe-snarks_amd", "csrc", "mfhip.hip")
    subprocess.check_call([HIPCC, "-O3", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "c-lwe-snarks_amd", "csrc"), "-S", "--cuda-device-only", "-o", str(out), src],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernel(asm, mangled_prefix):
    m = re.search(r"^(%s\w*):.*?s_endpgm" % re.escape(mangled_prefix), asm, re.S | re.M)
    assert m, f"kernel {mangled_prefix} not found in the assembly"
    return m.group(0)


def _hottest_loop(body):
    """opcodes of the single-block inner loop (label ... backward branch to it) that holds the table lookups and is shortest: the AES block loop"""
    lines = body.splitlines()
    best = []
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if not m:
            continue
        for k in range(i + 1, len(lines)):
            if re.search(r"s_cbranch_\w+\s+%s\b" % re.escape(m.group(1)), lines[k]):
                cand = [x.split()[0] for x in lines[i + 1:k] if x.strip() and not x.strip().startswith(";")]
                # the innermost loop that holds a whole block's lookups
                if cand.count("ds_read_b32") >= 150 and (not best or len(cand) < len(best)):
                    best = cand
                break
    return best


@pytest.mark.parametrize("kernel", ["_Z6k_evalILi736ELi2E", "_Z6k_evalILi736ELi1E", "_Z9k_encryptILi736E", "_Z8k_expandILi736E"])
def test_aes_inner_loop_has_no_per_lookup_address_add(asm, kernel):
    ops = _hottest_loop(_kernel(asm, kernel))
    n = {k: ops.count(k) for k in ("ds_read_b32", "v_add_u32_e32", "v_perm_b32", "v_bitop3_b32", "v_alignbit_b32")}
    assert 190 <= n["ds_read_b32"] <= 205, n          # 5 shortcut + 11 x 16 + 16 last-round lookups (+ the span constants' b32)
    assert n["v_add_u32_e32"] <= 16, n                 # loop counters and tile addresses only
    assert n["v_perm_b32"] + n["v_bitop3_b32"] <= 270, n


def test_aes_kernels_do_not_spill_in_the_loop(asm):
    body = _kernel(asm, "_Z6k_evalILi736ELi2E")
    ops = _hottest_loop(body)
    assert not any(o.startswith("scratch_") for o in ops)
