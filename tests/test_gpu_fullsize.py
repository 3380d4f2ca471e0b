"""GPU, BASELINE.json's full sizes (NDEBUG default instance: D = 2^15, M = 21845, 87 381 CRS rows, 11.8 GB of stream):
size-independent properties instead of a full oracle run (the oracle needs ~2 minutes per full-size proof).

  * acceptance: setup -> prover -> the four verifier equations on GPU-decrypted values (src/snark.c:192-250);
    a flipped witness bit is rejected;
  * linearity: eval(c0) + eval(c1) == eval(c0 + c1) over a whole region (exact, mod 2^704);
  * the sharded prover (3 shares) and the resident-CRS regime give the byte-identical proof;
  * sampled windows of the 11.8 GB stream against the oracle; sampled CRS rows decrypt to s^i / alpha*s^i;
  * sampled rows of the full-size batched encryption against the oracle.
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
SEED = bytes((91 * i + 7) & 0xFF for i in range(40))


@pytest.fixture(scope="module")
def world(gpu_ctx_factory):
    import torch

    import bench
    import c_lwe_snarks_amd as mf

    p = mf.DEFAULT
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 424242)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    return dict(mf=mf, bench=bench, torch=torch, ctx=ctx, p=p, inst=inst, d_crs=d_crs)


def _entropy(seed):
    rng = np.random.default_rng(seed)
    return int(rng.integers(0, ol.P, dtype=np.uint64)), rng.integers(0, 256, size=400, dtype=np.uint8).tobytes(), bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist())


def test_stream_windows_and_crs_rows(world, oracle):
    W = world
    ctx, p, inst = W["ctx"], W["p"], W["inst"]
    end = (2 * p.d + p.m) * p.ctr_ct
    for off in (0, p.ctr_as - 100, p.ctr_bt, p.ctr_bv + 12345 * p.ctr_ct + 4, end - 4096):
        assert ctx.to_host(ctx.keystream(off, 4096)).tobytes() == oracle.keystream(SEED, off, 4096)
    unit = ctx.to_device(np.ones(1, dtype=np.uint32))
    for region_off, row0, i, expect in [(p.ctr_s, 0, 0, 1), (p.ctr_s, 0, 31000, pow(inst["s"], 31000, ol.P)),
                                        (p.ctr_as, p.d, p.d - 1, inst["alpha"] * pow(inst["s"], p.d - 1, ol.P) % ol.P)]:
        ct, _ = ctx.eval_rows(region_off + i * p.ctr_ct, 1, W["d_crs"][(row0 + i) * p.ctb:], unit)
        assert int(ctx.to_host(ctx.decrypt(inst["sk"], ct, 1), np.uint32)[0]) == expect
    # sampled rows of the 87 381-row batched encryption against the oracle's regev_encrypt2
    sk = ctx.to_host(inst["sk"], np.uint64).reshape(p.n, p.L)
    err = ctx.to_host(inst["err"], np.uint64).reshape(-1, p.L)
    msg = ctx.to_host(ctx.setup_messages(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"]), np.uint32)
    crs = ctx.to_host(W["d_crs"])
    for i in (0, 1, 40001, 2 * p.d, 2 * p.d + p.m - 1):
        r = oracle.rng(SEED, i * p.ctr_ct)
        exp = oracle.ct_export(p, oracle.encrypt(p, r, sk, int(msg[i]), err[i]))
        assert crs[i * p.ctb:(i + 1) * p.ctb].tobytes() == exp


def test_full_size_proof_is_accepted_and_sound(world):
    W = world
    ctx, inst = W["ctx"], W["inst"]
    delta, mags, signs = _entropy(1)
    proof = ctx.prove(W["d_crs"], inst["d_ssp"], inst["bits"], delta, mags, signs)
    assert W["bench"].verify_on_gpu(W["mf"], ctx, inst, proof)
    bad = bytearray(inst["bits"])
    bad[100] ^= 0x10
    proof_bad = ctx.prove(W["d_crs"], inst["d_ssp"], bytes(bad), delta, mags, signs)
    assert not W["bench"].verify_on_gpu(W["mf"], ctx, inst, proof_bad)
    W["proof"] = proof.clone()
    W["entropy"] = (delta, mags, signs)


def test_sharded_and_resident_give_identical_proof(world):
    W = world
    ctx, p, inst, torch = W["ctx"], W["p"], W["inst"], W["torch"]
    delta, mags, signs = W["entropy"]
    lanes = None
    for r in range(3):
        part = ctx.prove_partial(W["d_crs"], inst["d_ssp"], inst["bits"], delta, r, 3)
        ln = ctx.ct_to_lanes(part, 5).clone()
        lanes = ln if lanes is None else lanes + ln
    proof = ctx.ct_from_lanes(lanes, 5)
    ctx.prove_finish(proof, mags, signs)
    assert torch.equal(proof, W["proof"])
    image = ctx.crs_expand(0, 2 * p.d + p.m, W["d_crs"])
    ctx.set_resident(image)
    try:
        proof_r = ctx.prove(W["d_crs"], inst["d_ssp"], inst["bits"], delta, mags, signs)
    finally:
        ctx.set_resident(None)
    assert torch.equal(proof_r, W["proof"])


def test_linearity_over_a_whole_region(world):
    W = world
    ctx, p = W["ctx"], W["p"]
    rng = np.random.default_rng(3)
    c0 = rng.integers(0, 1 << 31, size=p.d, dtype=np.uint64)
    c1 = rng.integers(0, 1 << 31, size=p.d, dtype=np.uint64)
    d0, d1, d01 = (ctx.to_device(c.astype(np.uint32)) for c in (c0, c1, c0 + c1))
    c8 = W["d_crs"][p.d * p.ctb:]
    e0, e1 = ctx.eval_rows(p.ctr_as, p.d, c8, d0, d1)
    e01, _ = ctx.eval_rows(p.ctr_as, p.d, c8, d01)
    assert W["torch"].equal(ctx.ct_add(e0, e1), e01)


@pytest.mark.parametrize("logq", [736, 1472])
def test_2pow20_constraints_instance(gpu_ctx_factory, logq):
    """BASELINE configs 4/5 in the constraint dimension: D = 2^20 (2 097 216 CRS rows = 284 GB / 567 GB of public stream,
    polynomial products of length 2^21), with M = 64 wires so that the dense SSP still fits (the reference's 2^20 instance
    would need a 5.9 TB SSP and cannot be run at all).  Properties: the device verifier accepts the proof, rejects a proof for a
    flipped witness bit and a tampered ciphertext, and the 3-way sharded proof is byte-identical."""
    import torch

    import bench
    import c_lwe_snarks_amd as mf

    p = mf.Params(logq=logq, d=1 << 20, m=64)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 99 + logq)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    delta, mags, signs = _entropy(logq)
    proof = ctx.prove(d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs)
    ver = lambda pr: int(ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], pr, 1))[0])
    assert ver(proof) == 1
    bad_bits = bytearray(inst["bits"])
    bad_bits[3] ^= 2
    assert ver(ctx.prove(d_crs, inst["d_ssp"], bytes(bad_bits), delta, mags, signs)) == 0
    tampered = proof.clone()
    tampered[(3 * p.ct_limbs + p.n * p.L) * 8] ^= 1  # low byte of b in v_w
    assert ver(tampered) == 0
    lanes = None
    for r in range(3):
        ln = ctx.ct_to_lanes(ctx.prove_partial(d_crs, inst["d_ssp"], inst["bits"], delta, r, 3), 5).clone()
        lanes = ln if lanes is None else lanes + ln
    shard = ctx.ct_from_lanes(lanes, 5)
    ctx.prove_finish(shard, mags, signs)
    assert torch.equal(shard, proof)
    ctx.close()


def test_default_size_proof_matches_oracle_hashes(gpu_ctx_factory):
    """Bit-exactness at the FULL NDEBUG default size against the oracle's complete prover run (tests/golden/default_size_proof.json,
    generated once on the CPU by tests/golden/make_default_size_golden.py): t, the witness polynomial, h = (v^2-1)/t and all five
    proof ciphertexts, before and after smudging."""
    import hashlib
    import importlib.util
    import json

    import c_lwe_snarks_amd as mf

    gdir = os.path.join(ROOT, "tests", "golden")
    gold_path = os.path.join(gdir, "default_size_proof.json")
    if not os.path.exists(gold_path):
        pytest.fail("tests/golden/default_size_proof.json is missing")
    gold = json.load(open(gold_path))
    spec = importlib.util.spec_from_file_location("mk_gold", os.path.join(gdir, "make_default_size_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    p = mf.DEFAULT
    I = mk.instance(p)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(I["seed"])
    sha = lambda t: hashlib.sha256(ctx.to_host(t).tobytes()).hexdigest()
    d_t = ctx.ssp_prg_make_t(gold["prg_seed"], I["bits"])
    assert sha(d_t) == gold["t_sha256"]
    ctx.ssp_set_prg(gold["prg_seed"], d_t)
    ctx.ssp_prepare(None)
    d_w = ctx.witness_poly(None, I["bits"], I["delta"])
    assert sha(d_w) == gold["w_sha256"]
    # h = (v^2 - 1) / t with v = w + v_0 (src/snark.c:161-169): the polynomial step on its own against the oracle's h
    d_v = ctx.poly_add(d_w, ctx.ssp_prg_fill(gold["prg_seed"], 1, 1), p.d)
    assert sha(ctx.poly_h(d_v)) == gold["h_sha256"]
    d_crs = ctx.to_device(I["c8"])
    pre = ctx.prove_partial(d_crs, None, I["bits"], I["delta"], 0, 1)
    names = ["h", "hat_h", "hat_v", "v_w", "b_w"]
    nb = p.ct_limbs * 8
    for k, nme in enumerate(names):
        assert sha(pre[k * nb:(k + 1) * nb]) == gold["pre_smudge_sha256"][nme], nme
    proof = ctx.prove(d_crs, None, I["bits"], I["delta"], I["mags"], I["signs"])
    for k, nme in enumerate(names):
        assert sha(proof[k * nb:(k + 1) * nb]) == gold["proof_sha256"][nme], nme
    assert sha(proof) == gold["proof_all_sha256"]
    # the same proof from the dense image of the same SSP (generator mode == dense mode at full size)
    import torch

    dense = torch.cat([d_t, ctx.ssp_prg_fill(gold["prg_seed"], 1, p.m + 2)])
    assert sha(ctx.prove(d_crs, dense, I["bits"], I["delta"], I["mags"], I["signs"])) == gold["proof_all_sha256"]
    ctx.close()
