"""GPU parity tests: the HIP path (through the C ABI of include/mfhip.h) against the CPU oracle, bit-exact.

Sizes here are ones the oracle finishes in seconds; the properties the reference's own tests pin
(src/test_entropy.c, src/test_lwe.c) are restated against the C ABI.  Full-size checks live in
test_gpu_fullsize.py.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SEED = bytes(range(40))
SEED2 = bytes((7 * i + 3) & 0xFF for i in range(40))


@pytest.fixture(scope="module")
def mf():
    import c_lwe_snarks_amd as m

    return m


@pytest.fixture(scope="module")
def ctx(gpu_ctx_factory, mf):
    c = gpu_ctx_factory(mf.DEBUG)
    c.set_seed(SEED)
    return c


# ---------------------------------------------------------------- L0/L1: keystream, seek, sampler
@pytest.mark.parametrize("off,n", [(0, 48), (0, 16), (92, 8), (5, 1), (15, 2), (16, 4096), (135240, 135240),
                                    (8863223880, 92 * 7), ((1 << 36) + 3, 1000), (12, 92 * 1470 + 5), (0, 1 << 20)])
def test_keystream_matches_oracle(ctx, oracle, off, n):
    got = ctx.to_host(ctx.keystream(off, n)).tobytes()
    assert got == oracle.keystream(SEED, off, n)


def test_keystream_kat(ctx):
    # seed bytes 0..39, verified there against `openssl enc -aes-256-ecb`
    assert ctx.to_host(ctx.keystream(0, 16)).tobytes().hex() == "8477f45516027713a26a881ae67882bf"
    assert ctx.to_host(ctx.keystream(92, 8)).tobytes().hex() == "ed29f6cae21f9e67"


def test_keystream_seek_and_chunking(ctx):
    # src/test_entropy.c:111-156: bulk == chunked generation; seek(512) == reading past 512 bytes
    bulk = ctx.to_host(ctx.keystream(0, 92 * 1470 * 3)).tobytes()
    for k in (0, 1, 1469, 1470, 2 * 1470 + 17):
        assert ctx.to_host(ctx.keystream(92 * k, 92)).tobytes() == bulk[92 * k: 92 * k + 92]
    assert ctx.to_host(ctx.keystream(512, 8)).tobytes() == bulk[512:520]


def test_other_seed(gpu_ctx_factory, mf, oracle):
    c = gpu_ctx_factory(mf.DEBUG)
    c.set_seed(SEED2)
    assert c.to_host(c.keystream(1234567, 333)).tobytes() == oracle.keystream(SEED2, 1234567, 333)


@pytest.mark.parametrize("off,nrows", [(0, 1), (135240, 2), (8863223880, 1)])
def test_sample_rows(ctx, oracle, mf, off, nrows):
    p = mf.DEBUG
    got = ctx.to_host(ctx.sample_rows(off, nrows), np.uint64).reshape(nrows, p.n, p.L)
    exp = oracle.sample_rows(p, SEED, off, nrows)
    assert np.array_equal(got, exp)


# ---------------------------------------------------------------- L2: ciphertext algebra
def _rand_ct(rng, p, count=1, bits=None):
    bits = p.logq if bits is None else bits
    return ol.rand_values(rng, count * (p.n + 1), p.L, bits).reshape(count, p.n + 1, p.L)


def test_ct_elementwise(ctx, oracle, mf):
    p = mf.DEBUG
    rng = np.random.default_rng(1)
    a, b = _rand_ct(rng, p)[0], _rand_ct(rng, p)[0]
    da, db = ctx.to_device(a), ctx.to_device(b)
    assert np.array_equal(ctx.to_host(ctx.ct_add(da, db), np.uint64).reshape(a.shape), oracle.ct_add(p, a, b))
    for x in (0, 1, 5, 0xFFFFFFFA):
        assert np.array_equal(ctx.to_host(ctx.ct_mul_ui(da, x), np.uint64).reshape(a.shape), oracle.ct_mul_ui(p, a, x))
        r0 = oracle.modq(p, b[0])  # accumulators are always reduced values
        bb = b.copy()
        bb[:, p.K:] = 0
        drop = ctx.to_device(bb)
        ctx.ct_addmul_ui(drop, da, x)
        assert np.array_equal(ctx.to_host(drop, np.uint64).reshape(a.shape), oracle.ct_addmul_ui(p, bb, a, x))
    # modq quirk: effective modulus is 2^704, not 2^736: 2^720 + 5 -> 5
    v = np.zeros_like(a)
    v[0] = ol.int_to_limbs((1 << 720) + 5, p.L)
    got = ctx.to_host(ctx.ct_mul_ui(ctx.to_device(v), 1), np.uint64).reshape(a.shape)
    assert ol.limbs_to_int(got[0]) == 5


def test_ct_mul_ui_rejects_scalar_ge_p(ctx, mf):
    p = mf.DEBUG
    d = ctx.zeros(p.ct_limbs * 8)
    with pytest.raises(mf.MfhError):
        ctx.ct_mul_ui(d, 0xFFFFFFFB)  # the reference asserts b < GAMMA_P (src/lwe.c:133)


@pytest.mark.parametrize("path", [0, 1])  # 0 = tile kernel (k_eval), 1 = wave-autonomous kernel (k_eval_w)
@pytest.mark.parametrize("nrows,nacc,off_kind", [(1, 1, "s"), (2, 1, "s"), (37, 2, "s"), (100, 1, "as"), (9, 2, "bv"), (64, 2, "odd"), (700, 2, "as")])
def test_eval_rows_matches_oracle(ctx, oracle, mf, nrows, nacc, off_kind, path):
    ctx.set_eval_path(path)
    try:
        _eval_rows_case(ctx, oracle, mf, nrows, nacc, off_kind)
    finally:
        ctx.set_eval_path(0)


def _eval_rows_case(ctx, oracle, mf, nrows, nacc, off_kind):
    p = mf.DEBUG
    off = {"s": p.ctr_s, "as": p.ctr_as, "bv": p.ctr_bv, "odd": p.ctr_ct * 3 + 0}[off_kind]
    rng = np.random.default_rng(nrows * 10 + nacc)
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    coeffs = [rng.integers(0, ol.P, size=nrows, dtype=np.uint64) for _ in range(nacc)]
    if nrows > 4:
        coeffs[0][1] = 0
        coeffs[0][3] = 0
        if nacc > 1:
            coeffs[1][3] = 0  # row 3 has all-zero coefficients: skipped on the GPU, expanded-and-ignored in the reference
            coeffs[1][2] = ol.P - 1
    d_c8 = ctx.to_device(c8)
    d_co = [ctx.to_device(c.astype(np.uint32)) for c in coeffs]
    r0, r1 = ctx.eval_rows(off, nrows, d_c8, d_co[0], d_co[1] if nacc > 1 else None)
    exp0 = oracle.eval_poly(p, SEED, off, c8.tobytes(), coeffs[0])
    assert np.array_equal(ctx.to_host(r0, np.uint64).reshape(exp0.shape), exp0)
    if nacc > 1:
        exp1 = oracle.eval_poly(p, SEED, off, c8.tobytes(), coeffs[1])
        assert np.array_equal(ctx.to_host(r1, np.uint64).reshape(exp1.shape), exp1)


def test_eval_rows_accumulates_like_reference(ctx, oracle, mf):
    # eval_poly accumulates into rop (src/lwe.c:183): two half calls == one full call
    p = mf.DEBUG
    rng = np.random.default_rng(5)
    nrows = 20
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
    d_c8, d_co = ctx.to_device(c8), ctx.to_device(co.astype(np.uint32))
    full, _ = ctx.eval_rows(0, nrows, d_c8, d_co)
    half, _ = ctx.eval_rows(0, 10, d_c8, d_co)
    ctx.eval_rows(10 * p.ctr_ct, 10, d_c8[10 * p.ctb:], d_co[40:], rop0=half, accumulate=True)
    assert np.array_equal(ctx.to_host(full), ctx.to_host(half))
    assert np.array_equal(ctx.to_host(full, np.uint64).reshape(p.n + 1, p.L), oracle.eval_poly(p, SEED, 0, c8.tobytes(), co))


def test_eval_rows_empty(ctx, mf):
    p = mf.DEBUG
    r0, _ = ctx.eval_rows(0, 0, ctx.zeros(16), ctx.zeros(16))
    assert not ctx.to_host(r0).any()


@pytest.mark.parametrize("path", [1, 2])  # 1 = VALU kernel (k_encrypt), 2 = matrix-core kernel (k_encrypt_mm: <sk, a> as a Toeplitz int8 GEMM)
@pytest.mark.parametrize("nrows,off_kind", [(1, "s"), (3, "s"), (5, "bv"), (33, "as"), (70, "odd8")])
def test_encrypt_rows_matches_oracle(ctx, oracle, mf, nrows, off_kind, path):
    p = mf.DEBUG
    # "odd8": a stream offset that is 8 mod 16 and crosses a 256-block counter span inside the rows (the counter-mode shortcut's refresh)
    off = {"s": p.ctr_s, "as": p.ctr_as, "bv": p.ctr_bv, "odd8": 3 * p.ctr_ct + 8 * 1001}[off_kind]
    rng = np.random.default_rng(nrows)
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    if nrows == 3:  # extreme key digits: all-ones values (every balanced digit -1 after the first), zero, and 0x80.. / 0x7f.. bytes
        sk[0] = ol.int_to_limbs((1 << p.logq) - 1, p.L)
        sk[1] = 0
        sk[2] = ol.int_to_limbs(int.from_bytes(b"\x80" * 92, "little"), p.L)
        sk[3] = ol.int_to_limbs(int.from_bytes(b"\x7f" * 92, "little"), p.L)
    msg = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
    err = ol.rand_values(rng, nrows, p.L, 559)  # GAMMA_LOG_SIGMA + 3 bits (src/lwe.c:62)
    ctx.set_encrypt_path(path)
    try:
        got = ctx.to_host(ctx.encrypt_rows(off, nrows, ctx.to_device(sk), ctx.to_device(msg.astype(np.uint32)), ctx.to_device(err)))
    finally:
        ctx.set_encrypt_path(0)
    r = oracle.rng(SEED, off)
    exp = b"".join(oracle.ct_export(p, oracle.encrypt(p, r, sk, int(msg[i]), err[i])) for i in range(nrows))
    assert got.tobytes() == exp


def test_encrypt_paths_agree_on_a_large_batch(ctx, mf):
    """4200 rows (two parities, nine workgroup pairs, many column chunks; the size from which the default picks the matrix-core kernel): the
    matrix-core kernel == the VALU kernel, byte for byte; an offset that is not a multiple of 8 is refused by the forced matrix-core path
    and served by the VALU kernel otherwise"""
    p = mf.DEBUG
    rng = np.random.default_rng(600)
    nrows = 4200
    d_sk = ctx.to_device(ol.rand_values(rng, p.n, p.L, p.logq))
    d_msg = ctx.to_device(rng.integers(0, ol.P, size=nrows, dtype=np.uint64).astype(np.uint32))
    d_err = ctx.to_device(ol.rand_values(rng, nrows, p.L, 559))
    out = {}
    for path in (1, 2, 0):
        ctx.set_encrypt_path(path)
        try:
            out[path] = ctx.to_host(ctx.encrypt_rows(p.ctr_as, nrows, d_sk, d_msg, d_err)).copy()
        finally:
            ctx.set_encrypt_path(0)
    assert np.array_equal(out[1], out[2]) and np.array_equal(out[0], out[2])
    ctx.set_encrypt_path(2)
    try:
        with pytest.raises(mf.MfhError):
            ctx.encrypt_rows(4, 40, d_sk, d_msg, d_err)
    finally:
        ctx.set_encrypt_path(0)
    ctx.encrypt_rows(4, 40, d_sk, d_msg, d_err)  # auto: falls back to the VALU kernel


def test_encrypt_decrypt_roundtrip_and_homomorphism(ctx, oracle, mf):
    # src/test_lwe.c:74-95 (dec(enc(m)) == m) and :105-181 (eval_poly of unit coefficients decrypts to sum m)
    p = mf.DEBUG
    rng = np.random.default_rng(11)
    nrows = 100
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    msg = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
    err = ol.rand_values(rng, nrows, p.L, 559)
    d_sk = ctx.to_device(sk)
    c8 = ctx.encrypt_rows(0, nrows, d_sk, ctx.to_device(msg.astype(np.uint32)), ctx.to_device(err))
    ones = ctx.to_device(np.ones(nrows, dtype=np.uint32))
    ev, _ = ctx.eval_rows(0, nrows, c8, ones)
    got = int(ctx.to_host(ctx.decrypt(d_sk, ev, 1), np.uint32)[0])
    assert got == int(msg.sum() % ol.P)
    # single-row: import + decrypt each of the first 5
    for i in range(5):
        unit = np.zeros(nrows, dtype=np.uint32)
        unit[i] = 1
        ct, _ = ctx.eval_rows(0, nrows, c8, ctx.to_device(unit))
        assert int(ctx.to_host(ctx.decrypt(d_sk, ct, 1), np.uint32)[0]) == int(msg[i])
        # and the oracle agrees on the decryption of the GPU ciphertext
        assert oracle.decrypt(p, sk, ctx.to_host(ct, np.uint64).reshape(p.n + 1, p.L)) == int(msg[i])


def test_decrypt_matches_oracle_on_unreduced_b(ctx, oracle, mf):
    p = mf.DEBUG
    rng = np.random.default_rng(3)
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    cts = _rand_ct(rng, p, count=3)  # b up to 736 bits, as after a raw ct_import (src/lwe.c:125)
    got = ctx.to_host(ctx.decrypt(ctx.to_device(sk), ctx.to_device(cts), 3), np.uint32)
    for i in range(3):
        assert int(got[i]) == oracle.decrypt(p, sk, cts[i])


@pytest.mark.parametrize("logq", [736, 1472])
@pytest.mark.parametrize("length,rop_kind", [(1, "zero"), (1, "full"), (7, "full"), (257, "full"), (1470, "zero"), (1470, "full")])
def test_add_dotp_matches_oracle(gpu_ctx_factory, oracle, mf, logq, length, rop_kind):
    """mpz_add_dotp (src/lwe.c:20-28): rop + sum_j a[j] b[j] accumulated unreduced, one modq at the end.  Operands are full
    logq-bit values (as sk and freshly sampled a_j are), rop either zero or an unreduced logq-bit value (its bits above 2^(64K) must
    vanish); len 1 (no reduction tree), 257 (one lane of the 256 takes two terms), 1470 (= GAMMA_N, what regev_encrypt2 / regev_decrypt pass)."""
    p = mf.Params(logq=logq, d=64, m=16)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(1000 * logq + length)
    a = ol.rand_values(rng, length, p.L, p.logq)
    b = ol.rand_values(rng, length, p.L, p.logq)
    if length >= 7:  # edge operands: all-ones (longest carry chains) and zero
        a[0] = ol.int_to_limbs((1 << p.logq) - 1, p.L)
        b[0] = a[0]
        a[1] = 0
        b[2] = ol.int_to_limbs(1, p.L)
    rop = np.zeros(p.L, dtype=np.uint64) if rop_kind == "zero" else ol.rand_values(rng, 1, p.L, p.logq)[0]
    exp = oracle.add_dotp(p, rop, a, b)
    # independent check of the oracle's value itself: Python integers
    want = (ol.limbs_to_int(rop) + sum(ol.limbs_to_int(a[j]) * ol.limbs_to_int(b[j]) for j in range(length))) % (1 << (64 * p.K))
    assert ol.limbs_to_int(exp) == want
    d_rop = c.to_device(rop)
    c.add_dotp(d_rop, c.to_device(a), c.to_device(b), length)
    got = c.to_host(d_rop, np.uint64)
    assert np.array_equal(got, exp)


def test_smudge(ctx, oracle, mf):
    # src/test_lwe.c:183-205: smudging preserves decryption; and bit-exact against the oracle
    p = mf.DEBUG
    rng = np.random.default_rng(4)
    cts = _rand_ct(rng, p, count=4, bits=704)
    cts[:, :, p.K:] = 0
    mags = rng.integers(0, 256, size=4 * 80, dtype=np.uint8)
    signs = bytes([0, 1, 1, 0])
    d = ctx.to_device(cts)
    ctx.ct_smudge(d, 4, mags.tobytes(), 80, signs)
    got = ctx.to_host(d, np.uint64).reshape(cts.shape)
    for i in range(4):
        exp, neg = oracle.ct_smudge(p, cts[i], mags[80 * i: 80 * i + 80].tobytes(), signs[i])
        assert not neg
        assert np.array_equal(got[i], exp)


# ---------------------------------------------------------------- L3: witness polynomial
def test_witness_poly(ctx, oracle, mf):
    p = mf.DEBUG
    rng = np.random.default_rng(6)
    ssp = rng.integers(0, 1 << 63, size=(p.m + 3) * p.d, dtype=np.uint64)  # arbitrary u64: import reduces mod p
    bits = rng.integers(0, 256, size=(p.m + 7) // 8, dtype=np.uint8).tobytes()
    delta = 0xDEADBEE
    d_ssp = ctx.ssp_upload(ssp)
    got = ctx.to_host(ctx.witness_poly(d_ssp, bits, delta), np.uint32)
    red = (ssp % np.uint64(ol.P)).reshape(p.m + 3, p.d)
    assert np.array_equal(ctx.to_host(d_ssp, np.uint32).reshape(p.m + 3, p.d), red.astype(np.uint32))
    w = (red[0].astype(object) * delta) % ol.P
    for i in range(1, p.m):
        if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
            w = (w + red[i + 1].astype(object)) % ol.P
    assert np.array_equal(got.astype(np.uint64), np.array(w, dtype=np.uint64))


def test_ssp_upload_staged_on_host_threads(gpu_ctx_factory, mf):
    """mfh_ssp_upload of an image large enough for its threaded path (from 16 chunks of 4 MB: eight host threads stage every eighth chunk through their own pinned pair and
    stream, the reduction mod p in the same stream): 210 MB of arbitrary uint64 values -- nmod_poly_import reduces them (src/ssp.c:28-34) -- a slot range that starts and ends
    inside chunks, then the whole image again; checked against numpy's `% p`, untouched slots must stay untouched"""
    import torch

    p = mf.Params(d=1 << 15, m=800)
    ctx = gpu_ctx_factory(p)
    rng = np.random.default_rng(99)
    slots = p.m + 3
    ssp = rng.integers(0, 1 << 63, size=slots * p.d, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=slots * p.d, dtype=np.uint64)  # all 64 bits
    red = (ssp % np.uint64(ol.P)).astype(np.uint32).reshape(slots, p.d)
    d_ssp = torch.full((slots * p.d * 4,), 0xA5, dtype=torch.uint8, device=ctx.device)
    first, n = 37, 613  # 160 MB: starts and ends mid-chunk
    ctx.ssp_upload(ssp, d_ssp, first, n)
    got = ctx.to_host(d_ssp, np.uint32).reshape(slots, p.d)
    assert np.array_equal(got[first:first + n], red[first:first + n])
    assert (got[:first] == 0xA5A5A5A5).all() and (got[first + n:] == 0xA5A5A5A5).all()
    ctx.ssp_upload(ssp, d_ssp)
    assert np.array_equal(ctx.to_host(d_ssp, np.uint32).reshape(slots, p.d), red)
    # ... and a small range (one copy on the caller's stream)
    d_ssp[: 3 * p.d * 4] = 0
    ctx.ssp_upload(ssp, d_ssp, 0, 3)
    assert np.array_equal(ctx.to_host(d_ssp, np.uint32).reshape(slots, p.d), red)
    ctx.close()


# ---------------------------------------------------------------- BASELINE configs 4/5: stream offsets of a 2^20-constraint CRS
@pytest.mark.parametrize("logq", [736, 1472])
def test_rows_at_2pow20_scale_offsets(gpu_ctx_factory, oracle, mf, logq):
    """D = 2^20, M = 699 050 (378 GB of stream at logq 736, 757 GB at 1472): rows addressed deep inside the AS / BV
    regions (byte offsets > 2^38) must equal the oracle's.  The reference cannot run these sizes at all."""
    p = mf.Params(logq=logq, d=1 << 20, m=699050)
    c = gpu_ctx_factory(p)
    c.set_seed(SEED2)
    rng = np.random.default_rng(logq)
    for off in (p.ctr_as + 7 * p.ctr_ct, p.ctr_bv + 699000 * p.ctr_ct):
        assert off > (1 << 37)
        nrows = 6
        c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
        co = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
        r0, _ = c.eval_rows(off, nrows, c.to_device(c8), c.to_device(co.astype(np.uint32)))
        exp = oracle.eval_poly(p, SEED2, off, c8.tobytes(), co)
        assert np.array_equal(c.to_host(r0, np.uint64).reshape(exp.shape), exp)
        got = c.to_host(c.sample_rows(off, 1), np.uint64).reshape(1, p.n, p.L)
        assert np.array_equal(got, oracle.sample_rows(p, SEED2, off, 1))


def test_eval_over_a_2pow20_row_region_sparse(gpu_ctx_factory, oracle, mf):
    """A whole S region of the 2^20-constraint CRS (1 048 576 rows) in one call: all but a handful of coefficients are zero, so
    the active-row compaction must pick exactly those rows at their (far apart) stream offsets."""
    p = mf.Params(d=1 << 20, m=699050)
    c = gpu_ctx_factory(p)
    c.set_seed(SEED2)
    rng = np.random.default_rng(9)
    nrows = p.d
    rows = sorted(int(x) for x in rng.choice(nrows, size=5, replace=False)) + [nrows - 1]
    co = np.zeros(nrows, dtype=np.uint32)
    vals = rng.integers(1, ol.P, size=len(rows), dtype=np.uint64)
    co[rows] = vals.astype(np.uint32)
    c8 = c.to_device(rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8))
    r0, _ = c.eval_rows(p.ctr_s, nrows, c8, c.to_device(co))
    exp = np.zeros((p.n + 1, p.L), dtype=np.uint64)
    c8h = c.to_host(c8)
    for r, v in zip(rows, vals):
        exp = oracle.eval_poly(p, SEED2, r * p.ctr_ct, c8h[r * p.ctb:(r + 1) * p.ctb].tobytes(), np.array([v], dtype=np.uint64), rop=exp)
    assert np.array_equal(c.to_host(r0, np.uint64).reshape(exp.shape), exp)


# ---------------------------------------------------------------- error behaviour of the C ABI
def test_error_paths(gpu_ctx_factory, mf):
    import ctypes

    lib = mf.load_library()
    h = ctypes.c_void_p()
    bad = mf._CParams(1470, 700, 256, 64)  # the reference has `#error "Not implemented"` for any other GAMMA_LOGQ
    assert lib.mfh_ctx_create(ctypes.byref(h), 0, ctypes.byref(bad)) == -4
    assert lib.mfh_ctx_create(ctypes.byref(h), 99, ctypes.byref(mf._CParams(1470, 736, 256, 64))) == -2
    c = gpu_ctx_factory(mf.DEBUG)  # no seed set yet
    with pytest.raises(mf.MfhError, match="mfh_set_seed"):
        c.keystream(0, 16)
    c.set_seed(SEED)
    with pytest.raises(mf.MfhError):
        c.witness_poly(c.zeros((mf.DEBUG.m + 3) * mf.DEBUG.d * 4), bytes(8), 0xFFFFFFFB)  # delta must be < p
    with pytest.raises(mf.MfhError):
        c.poly_h(c.zeros(mf.DEBUG.d * 4))  # no SSP prepared
    assert lib.mfh_eval_rows(c._h, 0, 4, None, None, None, None, None, 0) == -1


def test_degenerate_inputs(ctx, mf):
    """all-zero coefficient vectors (every row skipped) and an all-zero witness (empty row selection)"""
    p = mf.DEBUG
    nrows = 33
    c8 = ctx.zeros(nrows * p.ctb)
    z = ctx.zeros(nrows * 4)
    r0, r1 = ctx.eval_rows(0, nrows, c8, z, z)
    assert not ctx.to_host(r0).any() and not ctx.to_host(r1).any()
    keep = ctx.to_device(np.arange(p.ct_limbs, dtype=np.uint64) & np.uint64(0xFFFF))
    before = ctx.to_host(keep).copy()
    ctx.eval_rows(0, nrows, c8, z, rop0=keep, accumulate=True)
    got = ctx.to_host(keep, np.uint64).reshape(p.n + 1, p.L)
    exp = before.view(np.uint64).reshape(p.n + 1, p.L).copy()
    exp[:, p.K:] = 0  # accumulate re-applies modq
    assert np.array_equal(got, exp)
    rng = np.random.default_rng(12)
    ssp = rng.integers(0, ol.P, size=(p.m + 3) * p.d, dtype=np.uint64)
    d_ssp = ctx.ssp_upload(ssp)
    w = ctx.to_host(ctx.witness_poly(d_ssp, bytes((p.m + 7) // 8), 3), np.uint32)
    assert np.array_equal(w.astype(np.uint64), ssp[: p.d] * np.uint64(3) % np.uint64(ol.P))


# ---------------------------------------------------------------- regev_decrypt as a batch on the matrix cores (encmm.hip)
@pytest.mark.parametrize("count", [1, 16, 37, 300])  # one lane row; a whole wave tile; a ragged one; more than one 256-row workgroup
@pytest.mark.parametrize("logq", [736, 1472])
def test_decrypt_matrix_core_path_matches_oracle(gpu_ctx_factory, oracle, mf, logq, count):
    """mfh_decrypt with <a, sk> as a Toeplitz int8 GEMM (k_decrypt_mm) against the oracle's regev_decrypt (src/lwe.c:105-111) on ciphertexts with
    unreduced b (as after a raw ct_import), all-ones / zero values and a key with extreme balanced digits; counts that are multiples of neither
    the 32-row wave tile nor the 256-row workgroup; both paths against each other"""
    p = mf.Params(logq=logq, d=64, m=16)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(logq)
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    sk[0] = ol.int_to_limbs((1 << p.logq) - 1, p.L)      # digits 0xff ... : every balanced digit carries
    sk[1] = ol.int_to_limbs(int("80" * p.ctb, 16), p.L)  # digits 0x80: the most negative balanced digit
    sk[2] = 0
    cts = _rand_ct(rng, p, count=count)
    cts[0, :p.n] = ol.int_to_limbs((1 << p.logq) - 1, p.L)
    if count > 2:
        cts[1, :p.n] = 0
        cts[2, 5] = ol.int_to_limbs((1 << (64 * p.L)) - 1, p.L)  # bits above logq set in memory (an unreduced value): only the low 64K bits may matter
    d_sk, d_ct = c.to_device(sk), c.to_device(cts)
    c.set_decrypt_path(2)
    try:
        got = c.to_host(c.decrypt(d_sk, d_ct, count), np.uint32)
    finally:
        c.set_decrypt_path(0)
    c.set_decrypt_path(1)
    try:
        valu = c.to_host(c.decrypt(d_sk, d_ct, count), np.uint32)
    finally:
        c.set_decrypt_path(0)
    assert np.array_equal(got, valu), "the two kernels disagree"
    for i in (range(count) if count <= 40 else list(range(8)) + [255, 256, 257, count - 1]):
        want = oracle.decrypt(p, sk, cts[i])
        assert int(got[i]) == want, f"ciphertext {i}: matrix-core path"


def test_decrypt_paths_agree_on_a_batch_and_recover_the_messages(gpu_ctx_factory, mf):
    """4200 real encryptions (rows of the public stream): the full ciphertexts through k_decrypt (VALU) and k_decrypt_mm (default from 4096 on),
    and their seed-compressed form through mfh_decrypt_rows (a regenerated from the stream): all three return the messages"""
    import torch

    p = mf.DEBUG
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    rng = np.random.default_rng(4200)
    B = 4200
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    msg = rng.integers(0, ol.P, size=B, dtype=np.uint64).astype(np.uint32)
    err = ol.rand_values(rng, B, p.L, 559)
    off = 8 * 12345
    d_sk = c.to_device(sk)
    c8 = c.encrypt_rows(off, B, d_sk, c.to_device(msg), c.to_device(err))
    cts = torch.zeros((B, p.n + 1, p.L), dtype=torch.int64, device=c.device)
    cts[:, : p.n] = c.sample_rows(off, B).view(torch.int64).view(B, p.n, p.L)
    bpad = torch.zeros((B, p.L * 8), dtype=torch.uint8, device=c.device)
    bpad[:, : p.ctb] = c8.view(B, p.ctb)
    cts[:, p.n] = bpad.view(torch.int64)
    flat = cts.view(torch.uint8).reshape(-1)
    want = torch.from_numpy(msg.view(np.int32)).to(c.device)
    auto = c.decrypt(d_sk, flat, B).view(torch.int32)
    c.set_decrypt_path(1)
    try:
        valu = c.decrypt(d_sk, flat, B).view(torch.int32)
    finally:
        c.set_decrypt_path(0)
    rows = c.decrypt_rows(off, B, d_sk, c8).view(torch.int32)
    assert torch.equal(auto, want) and torch.equal(valu, want) and torch.equal(rows, want)
    with pytest.raises(mf.MfhError):
        c.decrypt_rows(off + 4, B, d_sk, c8)  # rows must start at byte 0 or 8 of an AES block


def test_digest128_is_a_function_of_every_byte(gpu_ctx_factory):
    """mfh_digest128 (the cache key under which the host shim keeps an expanded CRS): equal buffers give equal digests, any single changed byte -- in the body,
    in the last word, in a tail of 1 - 3 bytes -- changes both halves, and so does moving a word"""
    import torch

    import c_lwe_snarks_amd as mf

    ctx = gpu_ctx_factory(mf.DEBUG)
    g = torch.Generator(device="cuda").manual_seed(7)
    for nbytes in (8039052, 4096, 92, 7, 5, 4, 1):
        buf = torch.randint(0, 256, (nbytes + 8,), dtype=torch.uint8, device="cuda", generator=g)[:nbytes]
        d0 = ctx.digest128(buf)
        assert d0 == ctx.digest128(buf.clone()) and d0 != (0, 0)
        for pos in {0, nbytes // 2, nbytes - 1}:
            b2 = buf.clone()
            b2[pos] ^= 0x40
            d1 = ctx.digest128(b2)
            assert d1[0] != d0[0] and d1[1] != d0[1], (nbytes, pos)
        if nbytes >= 4096:
            b3 = buf.clone()
            b3[0:4], b3[8:12] = buf[8:12].clone(), buf[0:4].clone()  # two words swapped: a position-independent sum would not notice
            assert ctx.digest128(b3) != d0
    assert ctx.digest128(torch.zeros(16, dtype=torch.uint8, device="cuda"), 0) == (0, 0)
    with pytest.raises(mf.MfhError):
        ctx.digest128(torch.zeros(16, dtype=torch.uint8, device="cuda")[1:], 8)  # not 4-byte aligned
