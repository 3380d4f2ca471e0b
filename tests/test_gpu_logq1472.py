"""GPU parity for the doubled-modulus parameter set of BASELINE config 5 (log q = 1472, CT_BYTES = 184, 23 limbs,
effective modulus 2^1472).  The reference cannot run it (`#error "Not implemented"`, src/lwe.h:119-121): parity is
against the oracle, whose restatement generalises modq to "keep floor(logq/64) limbs" and is pinned at 736."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SEED = bytes((5 * i + 1) & 0xFF for i in range(40))


@pytest.fixture(scope="module")
def mf():
    import c_lwe_snarks_amd as m

    return m


@pytest.fixture(scope="module")
def setup(gpu_ctx_factory, mf):
    p = mf.Params(logq=1472, d=64, m=16)
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    return c, p


def test_sample_and_eval(setup, oracle):
    ctx, p = setup
    assert (p.L, p.K, p.ctb) == (23, 23, 184)
    got = ctx.to_host(ctx.sample_rows(p.ctr_ct * 3, 2), np.uint64).reshape(2, p.n, p.L)
    assert np.array_equal(got, oracle.sample_rows(p, SEED, p.ctr_ct * 3, 2))
    rng = np.random.default_rng(0)
    nrows = 21
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = [rng.integers(0, ol.P, size=nrows, dtype=np.uint64) for _ in range(2)]
    co[0][4] = 0
    r0, r1 = ctx.eval_rows(p.ctr_as, nrows, ctx.to_device(c8), ctx.to_device(co[0].astype(np.uint32)), ctx.to_device(co[1].astype(np.uint32)))
    for r, c in zip((r0, r1), co):
        exp = oracle.eval_poly(p, SEED, p.ctr_as, c8.tobytes(), c)
        assert np.array_equal(ctx.to_host(r, np.uint64).reshape(exp.shape), exp)


def test_encrypt_decrypt(setup, oracle):
    ctx, p = setup
    rng = np.random.default_rng(1)
    nrows = 5
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    msg = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
    err = ol.rand_values(rng, nrows, p.L, 559)
    d_sk = ctx.to_device(sk)
    c8 = ctx.encrypt_rows(0, nrows, d_sk, ctx.to_device(msg.astype(np.uint32)), ctx.to_device(err))
    r = oracle.rng(SEED, 0)
    exp = b"".join(oracle.ct_export(p, oracle.encrypt(p, r, sk, int(msg[i]), err[i])) for i in range(nrows))
    assert ctx.to_host(c8).tobytes() == exp
    for i in range(nrows):
        unit = np.zeros(nrows, dtype=np.uint32)
        unit[i] = 1
        ct, _ = ctx.eval_rows(0, nrows, c8, ctx.to_device(unit))
        assert int(ctx.to_host(ctx.decrypt(d_sk, ct, 1), np.uint32)[0]) == int(msg[i])


def test_full_snark_at_logq1472(setup, oracle):
    """setup -> prover -> verifier at log q = 1472 (D = 64, M = 16): every proof element equals the oracle's, the oracle's
    verifier and the device verifier accept, and the resident-CRS layout (11 full planes + 1 half plane) gives the same proof."""
    ctx, p = setup
    rng = np.random.default_rng(77)
    bits = rng.bytes((p.m + 7) // 8)
    tape = rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8)
    ssp = oracle.ssp_from_tape(p, tape, bits)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    crs = oracle.setup(p, SEED, ssp, alpha, beta, s, sk, etape)
    stream_order = np.concatenate([crs["s"], crs["as_"], crs["t"], crs["v"][: (p.m - 1) * p.ctb]])
    d_ssp = ctx.ssp_upload(ssp)
    ctx.ssp_prepare(d_ssp)
    d_sk = ctx.to_device(sk)
    d_crs = ctx.setup(d_ssp, alpha, beta, s, d_sk, ctx.to_device(etape))
    assert np.array_equal(ctx.to_host(d_crs), stream_order)
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=400, dtype=np.uint8).tobytes()
    signs = bytes([0, 1, 0, 1, 1])
    t5 = b"".join(mags[80 * k: 80 * k + 80] + signs[k: k + 1] for k in range(5))
    ref = oracle.prover(p, crs, ssp, bits, delta, t5, 80)
    proof = ctx.prove(d_crs, d_ssp, bits, delta, mags, signs)
    got = ctx.to_host(proof, np.uint64).reshape(5, p.n + 1, p.L)
    assert np.array_equal(got, ref["proof"])
    assert oracle.verifier(p, ssp, alpha, beta, s, sk, got)
    assert int(ctx.to_host(ctx.verify(d_ssp, alpha, beta, s, d_sk, proof, 1))[0]) == 1
    image = ctx.crs_expand(0, 2 * p.d + p.m, d_crs)
    assert image.numel() == (2 * p.d + p.m) * ctx.resident_row_bytes() and ctx.resident_row_bytes() == 1472 * 46 * 4
    ctx.set_resident(image)
    try:
        again = ctx.to_host(ctx.prove(d_crs, d_ssp, bits, delta, mags, signs)).copy()
    finally:
        ctx.set_resident(None)
    assert np.array_equal(again, ctx.to_host(proof))
