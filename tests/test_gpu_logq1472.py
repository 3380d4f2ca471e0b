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


@pytest.mark.parametrize("path,nrows,off", [(1, 5, 0), (2, 5, 0), (2, 37, 8 + 16 * 77)])  # 2 = matrix-core kernel: 12 column tiles, >= 3 column chunks
def test_encrypt_decrypt(setup, oracle, path, nrows, off):
    ctx, p = setup
    rng = np.random.default_rng(1)
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    msg = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
    err = ol.rand_values(rng, nrows, p.L, 559)
    d_sk = ctx.to_device(sk)
    ctx.set_encrypt_path(path)
    try:
        c8 = ctx.encrypt_rows(off, nrows, d_sk, ctx.to_device(msg.astype(np.uint32)), ctx.to_device(err))
    finally:
        ctx.set_encrypt_path(0)
    if off:
        r = oracle.rng(SEED, off)
        exp = b"".join(oracle.ct_export(p, oracle.encrypt(p, r, sk, int(msg[i]), err[i])) for i in range(nrows))
        assert ctx.to_host(c8).tobytes() == exp
        return
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


@pytest.mark.parametrize("nrows,nvec,cb", [(21, 3, 4), (300, 40, 4), (64, 63, 4), (70, 200, 1)])
def test_matrix_core_eval_at_1472(setup, oracle, nrows, nvec, cb):
    """mfh_eval_rows_multi at logq = 1472 (k_evalmm16<0,1472>: one coordinate = 184 bytes = 11.5 row tiles per column tile) against the
    VALU path and the oracle."""
    ctx, p = setup
    rng = np.random.default_rng(nrows + nvec)
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = rng.integers(0, 256 if cb == 1 else ol.P, size=(nvec, nrows), dtype=np.uint64).astype(np.uint32)
    co[0, :] = 0
    co[1 % nvec, :] = 0xFF if cb == 1 else 0xFFFFFFFA
    off = p.ctr_as + 3 * p.ctr_ct
    d_c8 = ctx.to_device(c8)
    got = ctx.to_host(ctx.eval_rows_multi(off, nrows, d_c8, ctx.to_device(co), nvec, coeff_bytes=cb), np.uint64).reshape(nvec, p.n + 1, p.L)
    for v in sorted({0, 1 % nvec, nvec // 2, nvec - 1}):
        ref, _ = ctx.eval_rows(off, nrows, d_c8, ctx.to_device(co[v]))
        assert np.array_equal(got[v], ctx.to_host(ref, np.uint64).reshape(p.n + 1, p.L)), f"vector {v}"
    exp = oracle.eval_poly(p, SEED, off, c8.tobytes(), co[nvec - 1].astype(np.uint64))
    assert np.array_equal(got[nvec - 1], exp.reshape(p.n + 1, p.L))


def test_batch_prover_at_1472(gpu_ctx_factory, mf, oracle):
    """mfh_prove_batch at logq = 1472: transient image (default), per-group regeneration and resident image, against mfh_prove proof by proof"""
    p = mf.Params(logq=1472, d=128, m=24)
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    rng = np.random.default_rng(1472)
    nbytes = (p.m + 7) // 8
    wit = rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()
    ssp = oracle.ssp_from_tape(p, rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8), wit)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    d_ssp = c.ssp_upload(ssp)
    c.ssp_prepare(d_ssp)
    d_crs = c.setup(d_ssp, alpha, beta, s, c.to_device(sk), c.to_device(etape))
    nb = 34
    stmts = [wit if b % 2 == 0 else rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for b in range(nb)]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nb, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
    signs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nb)]
    got = c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs), np.uint64).reshape(nb, -1).copy()
    for b in (0, 1, 17, 33):
        one = c.to_host(c.prove(d_crs, d_ssp, stmts[b], deltas[b], mags[b], signs[b]), np.uint64).reshape(-1)
        assert np.array_equal(got[b], one), f"proof {b}"
    c.set_batch_image(False)  # (the call above expanded the CRS into its transient image; now every group regenerates the keystream)
    assert np.array_equal(c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs), np.uint64).reshape(nb, -1), got)
    c.set_batch_image(True)
    image = c.crs_expand_mm(d_crs)
    c.set_resident_mm(image)
    try:
        res = c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs), np.uint64).reshape(nb, -1)
    finally:
        c.set_resident_mm(None)
    assert np.array_equal(res, got)
    ok = c.to_host(c.verify(d_ssp, alpha, beta, s, c.to_device(sk), c.to_device(got), nb))
    assert [bool(x) for x in ok] == [b % 2 == 0 for b in range(nb)]
