"""mfh_eval_rows_multi (matrix-core eval_poly for many coefficient vectors, csrc/evalmm.hip) against mfh_eval_rows (the VALU path,
itself pinned to the oracle in test_gpu_parity.py) and against the oracle directly: bit-exact."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SEED = bytes((5 * i + 9) & 0xFF for i in range(40))


@pytest.fixture(scope="module")
def ctx(gpu_ctx_factory):
    import c_lwe_snarks_amd as mf

    c = gpu_ctx_factory(mf.DEBUG)
    c.set_seed(SEED)
    return c


def _vectors(rng, nvec, nrows, kind):
    co = rng.integers(0, ol.P, size=(nvec, nrows), dtype=np.uint64).astype(np.uint32)
    if kind == "edges":
        co[0, :] = 0                      # an all-zero vector
        co[1 % nvec, :] = 0xFFFFFFFA      # p - 1 everywhere: every 7-bit digit at its maximum pattern
        co[2 % nvec, ::2] = 1             # 0/1 vectors like b_w's witness bits
        co[2 % nvec, 1::2] = 0
    return co


@pytest.mark.parametrize("nrows,nvec,kind,off_rows", [
    (1, 1, "rand", 0), (5, 2, "rand", 3), (128, 3, "edges", 0), (129, 12, "rand", 7), (300, 13, "edges", 1), (1100, 25, "rand", 2),
])
def test_multi_equals_single_vector_path(ctx, nrows, nvec, kind, off_rows):
    p = ctx.params
    rng = np.random.default_rng(nrows * 100 + nvec)
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = _vectors(rng, nvec, nrows, kind)
    off = p.ctr_as + off_rows * p.ctr_ct  # odd row offsets start mid-block (a row is 8 mod 16 bytes)
    d_c8 = ctx.to_device(c8)
    got = ctx.to_host(ctx.eval_rows_multi(off, nrows, d_c8, ctx.to_device(co), nvec), np.uint64).reshape(nvec, p.n + 1, p.L)
    for v in range(nvec):
        ref, _ = ctx.eval_rows(off, nrows, d_c8, ctx.to_device(co[v]))
        assert np.array_equal(got[v], ctx.to_host(ref, np.uint64).reshape(p.n + 1, p.L)), f"vector {v}"


def test_multi_matches_oracle_and_accumulates(ctx, oracle):
    p = ctx.params
    rng = np.random.default_rng(77)
    nrows, nvec = 40, 4
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = _vectors(rng, nvec, nrows, "rand")
    d_c8, d_co = ctx.to_device(c8), ctx.to_device(co)
    out = ctx.eval_rows_multi(p.ctr_s, nrows, d_c8, d_co, nvec)
    got = ctx.to_host(out, np.uint64).reshape(nvec, p.n + 1, p.L).copy()
    for v in range(nvec):
        exp = oracle.eval_poly(p, SEED, p.ctr_s, c8.tobytes(), co[v].astype(np.uint64))
        assert np.array_equal(got[v], exp.reshape(p.n + 1, p.L))
    # accumulate = 1 adds onto the previous value (mod 2^704): twice the sum
    ctx.eval_rows_multi(p.ctr_s, nrows, d_c8, d_co, nvec, out=out, accumulate=True)
    twice = ctx.to_host(out, np.uint64).reshape(nvec, p.n + 1, p.L)
    for v in range(nvec):
        a = ctx.to_device(got[v])
        assert np.array_equal(twice[v], ctx.to_host(ctx.ct_add(a, a), np.uint64).reshape(p.n + 1, p.L))


def test_multi_argument_checks(ctx):
    import c_lwe_snarks_amd as mf

    p = ctx.params
    with pytest.raises(mf.MfhError):
        ctx.eval_rows_multi(0, 4, ctx.zeros(4 * p.ctb), ctx.zeros(26 * 4 * 4), 26)
    out = ctx.eval_rows_multi(0, 0, ctx.zeros(16), ctx.zeros(16), 2)  # no rows: zero ciphertexts
    assert not ctx.to_host(out).any()
