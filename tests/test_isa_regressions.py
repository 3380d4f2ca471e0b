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


# ---- round 2: guards on the waits and spills that cost measurable time when they regressed -------------------
def _asm_of(tmp_path_factory, name):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_" + name) / (name + ".s")
    src = os.path.join(ROOT, "c-lwe-snarks_amd", "csrc", name + ".hip")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "c-lwe-snarks_amd", "csrc"), "-S", "--cuda-device-only", "-o", str(out), src],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


@pytest.fixture(scope="module")
def asm_evalmm(tmp_path_factory):
    return _asm_of(tmp_path_factory, "evalmm")


def _loops(body):
    """(label, [instruction lines]) of every single-block loop of a kernel"""
    lines = body.splitlines()
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if not m:
            continue
        for k in range(i + 1, len(lines)):
            if re.match(r"^\.LBB\d+_\d+:", lines[k]):
                break
            if re.search(r"s_cbranch_\w+\s+%s\b" % re.escape(m.group(1)), lines[k]):
                out.append((m.group(1), [x.strip() for x in lines[i + 1:k] if x.strip() and not x.strip().startswith(";")]))
                break
    return out


def _mfma_loop(body, min_mfma):
    best = max((l for _, l in _loops(body)), key=lambda l: sum("v_mfma" in x for x in l), default=[])
    assert sum("v_mfma" in x for x in best) >= min_mfma, "the kernel's MFMA loop was not found"
    return best


def test_witness_gemm_loads_are_awaited_four_steps_after_issue(asm_evalmm):
    """k_witness_mm8q: per step a wave issues one SSP fragment load and one bit fragment load, both consumed four steps later, so the
    only vector-memory waits of the loop are s_waitcnt vmcnt(6) / vmcnt(7) (two loads per step x 3 younger steps, + the step's own);
    smaller counts mean the compiler fell back to awaiting the stream a step or two after its issue (0.76 ms per 248 statements whatever
    the prefetch depth).  And nothing of the loop lives in scratch."""
    loop = _mfma_loop(_kernel(asm_evalmm, "_ZN12_GLOBAL__N_114k_witness_mm8qE"), 32)
    waits = [int(m.group(1)) for x in loop for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", x)] if m]
    assert waits and min(waits) >= 6, waits
    assert not any(x.startswith("scratch_") for x in loop)


def test_streaming_gemm_waits_one_stage_behind(asm_evalmm):
    """k_mmstream: the A fragments of stage s + 1 and the digit fragments of stage s + 2 are awaited one stage (16 younger loads minus the
    k-step's own) after their issue: vmcnt(12) / vmcnt(13), never a single-digit count inside the stage loop"""
    loop = _mfma_loop(_kernel(asm_evalmm, "_ZN12_GLOBAL__N_110k_mmstreamE"), 128)
    waits = [int(m.group(1)) for x in loop for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", x)] if m]
    assert waits and min(waits) >= 12, waits
    assert not any(x.startswith("scratch_") for x in loop)


def test_persistent_streaming_gemm_keeps_the_same_waits(asm_evalmm):
    """k_mmstream_p (the persistent one-workgroup-per-CU grid, round 4): the item loop around the stage loop must not change what the stage loop
    waits for.  (The scheduler places the first two MFMAs of k-step 1 between the loads of k-step 0 here, so the same two waits read vmcnt(10) after
    one load and vmcnt(11) after three instead of vmcnt(12) after three and four: the same loads of the previous stage are awaited.)"""
    loop = _mfma_loop(_kernel(asm_evalmm, "_ZN12_GLOBAL__N_112k_mmstream_pE"), 128)
    waits = [int(m.group(1)) for x in loop for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", x)] if m]
    assert waits and min(waits) >= 10, waits
    assert not any(x.startswith("scratch_") for x in loop)


def test_one_wave_per_simd_gemm_keeps_its_accumulators_in_accvgprs(asm_evalmm):
    """k_mmstream_w: 256 accumulators pinned in AccVGPRs by inline-asm MFMAs -- no v_accvgpr move and no scratch in the stage loop (the compiler's own placement
    made 600 moves per stage and spilled); memory operations dealt one per fragment (no two global loads back to back); ONE s_barrier per stage, behind a wait for
    the wave's ds_writes only (lgkmcnt(4): the four fragment reads issued after them stay in flight); s_waitcnt merged per group (19, was 91)"""
    body = _kernel(asm_evalmm, "_ZN12_GLOBAL__N_112k_mmstream_wE")
    loop = _mfma_loop(body, 256)
    assert sum("v_mfma_i32_16x16x64_i8 a[" in x for x in loop) == 256  # destination in AccVGPRs
    assert not any(x.startswith(("v_accvgpr", "scratch_")) for x in loop)
    assert sum(x.startswith("s_barrier") for x in loop) == 1
    k = next(i for i, x in enumerate(loop) if x.startswith("s_barrier"))
    assert re.search(r"s_waitcnt lgkmcnt\(4\)", loop[k - 1]), loop[k - 3:k + 1]
    assert sum(x.startswith("s_waitcnt") for x in loop) <= 24
    assert not any(x.startswith(("s_load", "s_buffer_load")) for x in loop)  # (scalar loads share lgkmcnt and return out of order: the hand-written lgkmcnt(4) assumes none)
    # between the k-step's last ds_write and the barrier exactly the four fragment reads of group 9 are issued
    w = max(i for i, x in enumerate(loop[:k]) if x.startswith("ds_write"))
    assert sum(x.startswith("ds_read") for x in loop[w:k]) == 4 and not any(x.startswith("ds_write") for x in loop[w + 1:k])
    mem = [x.split()[0] for x in loop if x.startswith(("global_load", "ds_write", "ds_read", "v_mfma"))]
    assert not any(a.startswith("global_load") and b.startswith("global_load") for a, b in zip(mem, mem[1:]))


def test_expansion_kernel_has_no_scratch(tmp_path_factory):
    """k_expand_mm at 64 VGPRs (8 waves per SIMD): the lane offset of a piece's store is recomputed per piece; kept live it was spilled and
    reloaded before every store behind s_waitcnt vmcnt(0)"""
    asm = _asm_of(tmp_path_factory, "expandmm")
    for k in ("_ZN12_GLOBAL__N_111k_expand_mmILi736E", "_ZN12_GLOBAL__N_111k_expand_mmILi1472E"):
        assert "scratch_" not in _kernel(asm, k), k


def test_ntt_block_kernel_reads_its_twiddles_from_lds(tmp_path_factory):
    """k_ntt_lds_mul8: 8 point loads (+ the cached transform of the other factor) and the twiddle staging are its only global loads -- the
    88 twiddle reads per thread come from LDS (as global gathers they bound the kernel: 0.43 against 0.27 ms per 248 x 3 transforms)"""
    body = _kernel(_asm_of(tmp_path_factory, "poly"), "_ZN12_GLOBAL__N_114k_ntt_lds_mul8E")
    assert len(re.findall(r"^\s+global_load_dword\b", body, re.M)) <= 20
