"""GPU parity for the SNARK layer: polynomial step, setup(), prover() against the oracle, bit-exact, at the
reference's debug parameters (D=256, M=64: what src/test_snark.c runs) plus the properties that test pins."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SEED = bytes((11 * i + 5) & 0xFF for i in range(40))


@pytest.fixture(scope="module")
def mf():
    import c_lwe_snarks_amd as m

    return m


@pytest.fixture(scope="module")
def ctx(gpu_ctx_factory, mf):
    c = gpu_ctx_factory(mf.DEBUG)
    c.set_seed(SEED)
    return c


def _u32(ctx, arr):
    return ctx.to_device(np.ascontiguousarray(arr, dtype=np.uint32))


# ------------------------------------------------------------------ polynomial arithmetic
@pytest.mark.parametrize("la,lb", [(1, 1), (2, 3), (256, 256), (100, 511), (1000, 1)])
def test_poly_mul(ctx, la, lb):
    rng = np.random.default_rng(la * 1000 + lb)
    a = rng.integers(0, ol.P, size=la, dtype=np.uint64)
    b = rng.integers(0, ol.P, size=lb, dtype=np.uint64)
    a[0] = ol.P - 1
    b[-1] = ol.P - 1
    got = ctx.to_host(ctx.poly_mul(_u32(ctx, a), la, _u32(ctx, b), lb), np.uint32)
    exp = [0] * (la + lb - 1)
    for i, x in enumerate(a.tolist()):
        for j, y in enumerate(b.tolist()):
            exp[i + j] = (exp[i + j] + x * y) % ol.P
    assert got.tolist() == exp


@pytest.mark.parametrize("case", ["dense", "low_deg_t", "small_v", "valid_ssp", "constant_t"])
def test_poly_h_matches_oracle(ctx, oracle, mf, case):
    p = mf.DEBUG
    rng = np.random.default_rng(["dense", "low_deg_t", "small_v", "valid_ssp", "constant_t"].index(case) + 4800)  # (not hash(case): that changes from run to run)
    v = rng.integers(0, ol.P, size=p.d, dtype=np.uint64)
    t = rng.integers(0, ol.P, size=p.d, dtype=np.uint64)
    if case == "low_deg_t":
        t[p.d - 5:] = 0  # deg t = d-6: the quotient has more than d coefficients; the first d are kept
    if case == "small_v":
        v[10:] = 0  # deg(v^2-1) < deg t: quotient 0
    if case == "constant_t":
        t[1:] = 0  # deg t = 0: the only case in which the "- 1" of v^2 - 1 reaches the quotient (its coefficient 0)
    if case == "valid_ssp":
        bits = rng.integers(0, 256, size=(p.m + 7) // 8, dtype=np.uint8).tobytes()
        tape = rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8)
        ssp = oracle.ssp_from_tape(p, tape, bits).reshape(p.m + 3, p.d)
        t = ssp[0].copy()
        v = ssp[1].copy()
        for i in range(1, p.m):
            if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
                v = (v + ssp[i + 1]) % np.uint64(ol.P)
        assert oracle.poly_divides(v, t)
    ctx.poly_prepare_t(_u32(ctx, t))
    got = ctx.to_host(ctx.poly_h(_u32(ctx, v)), np.uint32).astype(np.uint64)
    assert np.array_equal(got, oracle.poly_h(v, t))


def test_poly_square_and_hat_paths_at_2048_point_blocks(gpu_ctx_factory, oracle, mf):
    """Transforms of 2^11 points or more run their low 11 stages in registers (k_ntt_lds_mul8: squaring and cached-transform products):
    a 1500-coefficient square against exact integer convolution, and h = (v^2 - 1) / t at d = 2048 against the oracle."""
    p = mf.Params(logq=736, d=2048, m=64)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(2048)
    a = rng.integers(0, ol.P, size=1500, dtype=np.uint64)
    a[0] = a[-1] = ol.P - 1
    d_a = _u32(c, a)
    got = c.to_host(c.poly_mul(d_a, 1500, d_a, 1500), np.uint32)
    exp = np.convolve(a.astype(object), a.astype(object)) % ol.P
    assert got.tolist() == [int(x) for x in exp]
    v = rng.integers(0, ol.P, size=p.d, dtype=np.uint64)
    t = rng.integers(0, ol.P, size=p.d, dtype=np.uint64)
    c.poly_prepare_t(_u32(c, t))
    got = c.to_host(c.poly_h(_u32(c, v)), np.uint32).astype(np.uint64)
    assert np.array_equal(got, oracle.poly_h(v, t))


@pytest.mark.parametrize("d,m", [(256, 64), (1152, 40), (2048, 24), (4096, 16)])
def test_exact_division_path_is_checked_and_falls_back(gpu_ctx_factory, oracle, mf, d, m):
    """A batch of h = (v^2 - 1) / t goes through two CYCLIC products of length N = 2^ceil(log2 d) (the exact quotient of a valid witness is determined modulo x^N - 1),
    is checked on the device and recomputed by Euclidean division when one statement does not divide (src/snark.c:166-169 computes nmod_poly_div whatever the witness):
    (i) valid witnesses only: equal to the oracle's quotients, no statement recomputed; (ii) the same batch with two statements whose v does not belong to the SSP:
    equal to the oracle's (Euclidean) quotients, those two counted; (iii) the same with the path switched off.  d = 1152: N = 2048 > d; 2048 / 4096: the register
    kernels with one and two passes above the 2048-point blocks."""
    p = mf.Params(logq=736, d=d, m=m)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(d)
    nb = 9
    tape = rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8)
    base_bits = rng.integers(0, 256, size=(p.m + 7) // 8, dtype=np.uint8).tobytes()
    ssp = oracle.ssp_from_tape(p, tape, base_bits).reshape(p.m + 3, p.d)
    t = ssp[0].copy()

    def v_of(bits):
        v = ssp[1].copy()
        for i in range(1, p.m):
            if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
                v = (v + ssp[i + 1]) % np.uint64(ol.P)
        return v

    v_ok = v_of(base_bits)
    assert oracle.poly_divides(v_ok, t)
    # valid statements: +- v (the same square), and v with t-multiples added is no longer of degree < d, so vary by sign and by delta t instead: (v + delta t)^2 - 1 = v^2 - 1 + t (...)
    vs = []
    for k in range(nb):
        delta = int(rng.integers(0, ol.P, dtype=np.uint64))
        vs.append((v_ok.astype(object) + delta * t.astype(object)) % ol.P)
    V = np.array(vs, dtype=np.uint64)
    for v in V:
        assert oracle.poly_divides(v, t)
    exp = np.stack([oracle.poly_h(v, t) for v in V])
    c.poly_prepare_t(_u32(c, t))
    c.set_poly_exact(2)  # every batch tries the path (the default backs off after a failed check: below)
    assert c.poly_exact_fallbacks() == 0  # (an exact path exists for this t, nothing has failed)
    got = c.to_host(c.poly_h_many(_u32(c, V.reshape(-1)), nb), np.uint32).astype(np.uint64).reshape(nb, d)
    assert np.array_equal(got, exp)
    assert c.poly_exact_fallbacks() == 0
    # two statements of the batch do not divide
    other = bytearray(base_bits)
    other[0] ^= 0x14
    V2 = V.copy()
    V2[2] = v_of(bytes(other))
    V2[7, d // 2] = (V2[7, d // 2] + np.uint64(1)) % np.uint64(ol.P)
    assert not oracle.poly_divides(V2[2], t) and not oracle.poly_divides(V2[7], t)
    exp2 = np.stack([oracle.poly_h(v, t) for v in V2])
    got2 = c.to_host(c.poly_h_many(_u32(c, V2.reshape(-1)), nb), np.uint32).astype(np.uint64).reshape(nb, d)
    assert np.array_equal(got2, exp2)
    assert c.poly_exact_fallbacks() == 2
    # ... and a valid batch behind it is again exact (the flag is per batch)
    got = c.to_host(c.poly_h_many(_u32(c, V.reshape(-1)), nb), np.uint32).astype(np.uint64).reshape(nb, d)
    assert np.array_equal(got, exp) and c.poly_exact_fallbacks() == 0
    c.set_poly_exact(0)
    got3 = c.to_host(c.poly_h_many(_u32(c, V2.reshape(-1)), nb), np.uint32).astype(np.uint64).reshape(nb, d)
    assert np.array_equal(got3, exp2) and c.poly_exact_fallbacks() == 0
    # the default mode: a batch in which the check fails sends the batches behind it down the Euclidean path alone (64 of them), so that a caller whose statements
    # do not satisfy the SSP does not pay for both -- seen here by the check not running at all: the same two statements are no longer counted
    c.set_poly_exact(1)
    d_v2 = _u32(c, V2.reshape(-1))
    for k in range(3):
        got4 = c.to_host(c.poly_h_many(d_v2, nb), np.uint32).astype(np.uint64).reshape(nb, d)
        assert np.array_equal(got4, exp2)
        assert c.poly_exact_fallbacks() == (2 if k == 0 else 0)  # (the call waits for the stream: the next batch finds the mark)
    got = c.to_host(c.poly_h_many(_u32(c, V.reshape(-1)), nb), np.uint32).astype(np.uint64).reshape(nb, d)
    assert np.array_equal(got, exp)
    c.set_poly_exact(1)  # setting a mode forgets the mark
    got4 = c.to_host(c.poly_h_many(d_v2, nb), np.uint32).astype(np.uint64).reshape(nb, d)
    assert np.array_equal(got4, exp2) and c.poly_exact_fallbacks() == 2


def test_exact_division_path_is_not_offered_for_a_short_t(gpu_ctx_factory, oracle, mf):
    """deg t < d - 1: the quotient has more than d coefficients, nothing is determined modulo x^N - 1 -- Euclidean division only"""
    p = mf.Params(logq=736, d=256, m=64)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(77)
    t = rng.integers(0, ol.P, size=p.d, dtype=np.uint64)
    t[p.d - 3:] = 0
    V = rng.integers(0, ol.P, size=(6, p.d), dtype=np.uint64)
    c.poly_prepare_t(_u32(c, t))
    assert c.poly_exact_fallbacks() == -1
    got = c.to_host(c.poly_h_many(_u32(c, V.reshape(-1)), 6), np.uint32).astype(np.uint64).reshape(6, p.d)
    assert np.array_equal(got, np.stack([oracle.poly_h(v, t) for v in V]))


def test_poly_prepare_rejects_zero_t(ctx, mf):
    with pytest.raises(mf.MfhError):
        ctx.poly_prepare_t(ctx.zeros(mf.DEBUG.d * 4))


# ------------------------------------------------------------------ a full debug-size instance
@pytest.fixture(scope="module")
def instance(ctx, oracle, mf):
    p = mf.DEBUG
    rng = np.random.default_rng(2024)
    bits = bytearray(rng.integers(0, 256, size=(p.m + 7) // 8, dtype=np.uint8).tobytes())
    tape = rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8)
    ssp = oracle.ssp_from_tape(p, tape, bytes(bits))
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    crs = oracle.setup(p, SEED, ssp, alpha, beta, s, sk, etape)
    d_ssp = ctx.ssp_upload(ssp)
    ctx.ssp_prepare(d_ssp)
    return dict(p=p, bits=bytes(bits), ssp=ssp, alpha=alpha, beta=beta, s=s, sk=sk, etape=etape, crs=crs, d_ssp=d_ssp)


def _crs_stream_order(p, crs):
    return np.concatenate([crs["s"], crs["as_"], crs["t"], crs["v"][: (p.m - 1) * p.ctb]])


def test_setup_messages(ctx, oracle, instance):
    I = instance
    p = I["p"]
    got = ctx.to_host(ctx.setup_messages(I["d_ssp"], I["alpha"], I["beta"], I["s"]), np.uint32)
    ssp = I["ssp"].reshape(p.m + 3, p.d)
    exp = []
    x = 1
    for _ in range(p.d):
        exp.append(x)
        x = x * I["s"] % ol.P
    exp += [e * I["alpha"] % ol.P for e in exp[: p.d]]
    exp.append(oracle.poly_eval(ssp[0], I["s"]) * I["beta"] % ol.P)
    exp += [oracle.poly_eval(ssp[i + 1], I["s"]) * I["beta"] % ol.P for i in range(1, p.m)]
    assert got.tolist() == exp


def test_setup_matches_oracle(ctx, instance):
    I = instance
    p = I["p"]
    d_crs = ctx.setup(I["d_ssp"], I["alpha"], I["beta"], I["s"], ctx.to_device(I["sk"]), ctx.to_device(I["etape"]))
    got = ctx.to_host(d_crs)
    assert np.array_equal(got, _crs_stream_order(p, I["crs"]))
    I["d_crs"] = d_crs


def test_setup_image_leaves_the_rows_the_prover_streams(ctx, oracle, instance):
    """mfh_setup_image (SURVEY 8(f)1's by-product clause): the CRS is the oracle's, the row image is what mfh_crs_expand writes for that CRS, and a proof over it
    (k_mac_resident) is bit for bit the oracle's prover() -- no a-vector regenerated by the first proof under the new CRS"""
    I = instance
    p = I["p"]
    d_crs, rows = ctx.setup_image(I["d_ssp"], I["alpha"], I["beta"], I["s"], ctx.to_device(I["sk"]), ctx.to_device(I["etape"]))
    assert np.array_equal(ctx.to_host(d_crs), _crs_stream_order(p, I["crs"]))
    assert np.array_equal(ctx.to_host(rows), ctx.to_host(ctx.crs_expand(0, 2 * p.d + p.m, d_crs)))
    # the a parts depend on the seed alone: expanded without the ciphertexts (in two row slices, as the shim's setup() does beside its SSP upload) and the b column
    # filled in afterwards, the image is the same bytes
    nrows_all, rb, cut = 2 * p.d + p.m, ctx.resident_row_bytes(), 301
    img2 = ctx.zeros(nrows_all * rb)
    ctx.crs_expand(0, cut, None, out=img2)
    ctx.crs_expand(cut * p.ctr_ct, nrows_all - cut, None, out=img2[cut * rb:])
    assert not np.array_equal(ctx.to_host(img2), ctx.to_host(rows))
    ctx.crs_image_set_b(img2, 0, cut, d_crs)
    ctx.crs_image_set_b(img2, cut, nrows_all - cut, d_crs[cut * p.ctb:])
    assert np.array_equal(ctx.to_host(img2), ctx.to_host(rows))
    rng = np.random.default_rng(5)
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes()
    signs = bytes([0, 1, 1, 0, 1])
    tape = b"".join(mags[80 * k: 80 * k + 80] + signs[k: k + 1] for k in range(5))
    ctx.set_resident(rows)
    try:
        got = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], I["bits"], delta, mags, signs), np.uint64).reshape(5, p.n + 1, p.L)
    finally:
        ctx.set_resident(None)
    ref = oracle.prover(p, I["crs"], I["ssp"], I["bits"], delta, tape, 80, want_pre=False)
    for k, name in enumerate(("h", "hat_h", "hat_v", "v_w", "b_w")):
        assert np.array_equal(got[k], ref["proof"][k]), name


def test_crs_structure_properties(ctx, instance):
    # src/test_snark.c:35-70: dec(s[0]) = 1, dec(as[0]) = alpha, alpha*dec(s[i]) = dec(as[i]) for i = 1, D-1
    I = instance
    p = I["p"]
    d_crs = I["d_crs"] if "d_crs" in I else ctx.to_device(_crs_stream_order(p, I["crs"]))
    d_sk = ctx.to_device(I["sk"])

    def dec_row(region_off, region_row0, i):
        unit = np.zeros(1, dtype=np.uint32) + 1
        ct, _ = ctx.eval_rows(region_off + i * p.ctr_ct, 1, d_crs[(region_row0 + i) * p.ctb:], ctx.to_device(unit))
        return int(ctx.to_host(ctx.decrypt(d_sk, ct, 1), np.uint32)[0])

    assert dec_row(p.ctr_s, 0, 0) == 1
    assert dec_row(p.ctr_as, p.d, 0) == I["alpha"]
    for i in (1, p.d - 1):
        assert dec_row(p.ctr_s, 0, i) * I["alpha"] % ol.P == dec_row(p.ctr_as, p.d, i)


def test_prover_matches_oracle_and_verifies(ctx, oracle, instance):
    I = instance
    p = I["p"]
    rng = np.random.default_rng(7)
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes()
    signs = bytes([1, 0, 1, 0, 1])
    tape = b"".join(mags[80 * k: 80 * k + 80] + signs[k: k + 1] for k in range(5))
    ref = oracle.prover(p, I["crs"], I["ssp"], I["bits"], delta, tape, 80)
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    got = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], I["bits"], delta, mags, signs), np.uint64).reshape(5, p.n + 1, p.L)
    names = ["h", "hat_h", "hat_v", "v_w", "b_w"]
    for k in range(5):
        assert np.array_equal(got[k], ref["proof"][k]), f"proof element {names[k]} differs from the oracle"
    # acceptance: the reference-semantics verifier (oracle) accepts the GPU proof (src/test_snark.c:105-107)
    assert oracle.verifier(p, I["ssp"], I["alpha"], I["beta"], I["s"], I["sk"], got)
    # src/test_snark.c:81-89: alpha*dec(h) == dec(hat_h), 0 < dec(h) < p  -- decrypted on the GPU
    dec = ctx.to_host(ctx.decrypt(ctx.to_device(I["sk"]), ctx.to_device(got), 5), np.uint32)
    assert 0 < int(dec[0]) < ol.P and int(dec[0]) * I["alpha"] % ol.P == int(dec[1])
    # a wrong witness bit must be rejected
    bad = bytearray(I["bits"])
    bad[0] ^= 1
    got_bad = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], bytes(bad), delta, mags, signs), np.uint64).reshape(5, p.n + 1, p.L)
    assert not oracle.verifier(p, I["ssp"], I["alpha"], I["beta"], I["s"], I["sk"], got_bad)


def test_sharded_prover_equals_single(ctx, oracle, instance, mf):
    """partial proofs over row shares, summed as uint64 lanes, equal the single-GPU proof bit for bit.
    (world = 3 emulated on one GPU: the all-reduce is a plain tensor sum here; tests/test_dist_cpu.py runs the real
    collective with gloo.)"""
    from c_lwe_snarks_amd import dist as mfdist

    I = instance
    p = I["p"]
    rng = np.random.default_rng(8)
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes()
    signs = bytes([0, 1, 1, 0, 0])
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    single = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], I["bits"], delta, mags, signs)).copy()
    one = ctx.to_host(mfdist.prove_sharded(ctx, d_crs, I["d_ssp"], I["bits"], delta, mags, signs, 0, 1)).copy()
    assert np.array_equal(single, one)
    world = 3
    lanes = None
    for r in range(world):
        part = ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, r, world)
        ln = ctx.ct_to_lanes(part, 5).clone()
        lanes = ln if lanes is None else lanes + ln
    proof = ctx.ct_from_lanes(lanes, 5)
    ctx.prove_finish(proof, mags, signs)
    assert np.array_equal(ctx.to_host(proof), single)
    # the CPU restatements used by the gloo test agree with the device kernels
    part = ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, 1, world)
    limbs = ctx.to_host(part, np.uint64).reshape(5, p.n + 1, p.L)
    assert np.array_equal(mfdist.lanes_from_limbs_cpu(limbs, p.K).reshape(-1), ctx.to_host(ctx.ct_to_lanes(part, 5), np.int64))
    assert np.array_equal(mfdist.limbs_from_lanes_cpu(ctx.to_host(lanes, np.int64).reshape(5, p.n + 1, p.lanes), p.L, p.K).reshape(-1),
                          ctx.to_host(ctx.ct_from_lanes(lanes, 5), np.uint64))


def test_resident_crs_matches_regenerated(ctx, oracle, instance, mf):
    """SURVEY 8(d) second regime: rows expanded once into the streaming layout give the same eval_poly results and the
    same proof as regenerating the keystream."""
    I = instance
    p = I["p"]
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    nrows_all = 2 * p.d + p.m
    image = ctx.crs_expand(0, nrows_all, d_crs)
    assert image.numel() == nrows_all * ctx.resident_row_bytes()
    rng = np.random.default_rng(21)
    for first, nrows in [(0, 37), (p.d, 64), (2 * p.d, p.m), (5, 1)]:
        co = [rng.integers(0, ol.P, size=nrows, dtype=np.uint64) for _ in range(2)]
        co[0][0] = 0
        if nrows > 3:
            co[1][3] = 0
            co[0][3] = 0
        d_co = [ctx.to_device(c.astype(np.uint32)) for c in co]
        r0, r1 = ctx.eval_rows_resident(image, first, nrows, d_co[0], d_co[1])
        e0, e1 = ctx.eval_rows(first * p.ctr_ct, nrows, d_crs[first * p.ctb:], d_co[0], d_co[1])
        assert np.array_equal(ctx.to_host(r0), ctx.to_host(e0)) and np.array_equal(ctx.to_host(r1), ctx.to_host(e1))
        exp = oracle.eval_poly(p, SEED, first * p.ctr_ct, ctx.to_host(d_crs)[first * p.ctb:(first + nrows) * p.ctb].tobytes(), co[0])
        assert np.array_equal(ctx.to_host(r0, np.uint64).reshape(exp.shape), exp)
        s0, _ = ctx.eval_rows_resident(image, first, nrows, d_co[1])
        assert np.array_equal(ctx.to_host(s0), ctx.to_host(r1))
    delta = 777
    mags = rng.integers(0, 256, size=400, dtype=np.uint8).tobytes()
    signs = bytes([1, 1, 0, 0, 1])
    regen = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], I["bits"], delta, mags, signs)).copy()
    ctx.set_resident(image)
    try:
        res = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], I["bits"], delta, mags, signs)).copy()
    finally:
        ctx.set_resident(None)
    assert np.array_equal(regen, res)


def test_sharded_witness_lanes(ctx, instance):
    """the second exchange of the sharded prover: per-rank witness lanes sum to the replicated witness polynomial"""
    I = instance
    p = I["p"]
    delta = 4242
    world = 3
    lanes = None
    for r in range(world):
        ln = ctx.witness_lanes(I["d_ssp"], I["bits"], r, world).clone()
        lanes = ln if lanes is None else lanes + ln
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    total = None
    for r in range(world):
        part = ctx.prove_partial_w(d_crs, I["d_ssp"], I["bits"], delta, r, world, lanes)
        ln = ctx.ct_to_lanes(part, 5).clone()
        total = ln if total is None else total + ln
    proof = ctx.ct_from_lanes(total, 5)
    ref = ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, 0, 1)
    assert np.array_equal(ctx.to_host(proof), ctx.to_host(ref))


def test_device_verifier_agrees_with_oracle(ctx, oracle, instance):
    I = instance
    p = I["p"]
    rng = np.random.default_rng(31)
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    proofs = []
    expect = []
    for k in range(4):
        bits = bytearray(I["bits"])
        if k % 2:
            bits[k] ^= 4  # invalid witness
        delta = int(rng.integers(0, ol.P, dtype=np.uint64))
        pr = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], bytes(bits), delta, rng.integers(0, 256, size=400, dtype=np.uint8).tobytes(), bytes(5)),
                         np.uint64).reshape(5, p.n + 1, p.L).copy()
        if k == 2:
            pr[4, p.n, 0] ^= np.uint64(2)  # tamper with b_w
        proofs.append(pr)
        expect.append(oracle.verifier(p, I["ssp"], I["alpha"], I["beta"], I["s"], I["sk"], pr))
    ok = ctx.to_host(ctx.verify(I["d_ssp"], I["alpha"], I["beta"], I["s"], ctx.to_device(I["sk"]), ctx.to_device(np.stack(proofs)), 4))
    assert [bool(x) for x in ok] == expect == [True, False, False, False]


def test_sharded_resident_image(ctx, instance):
    """each rank keeps only its share of the expanded CRS (S share | AS share | BT+BV share) and streams it"""
    I = instance
    p = I["p"]
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    delta = 99
    ref = ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, 0, 1)
    world = 3
    total = None
    rows = 0
    for r in range(world):
        image = ctx.crs_expand_share(d_crs, r, world)
        rows += image.numel() // ctx.resident_row_bytes()
        ctx.set_resident_share(image, r, world)
        try:
            part = ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, r, world)
            with pytest.raises(Exception):
                ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, (r + 1) % world, world)  # image of another rank
        finally:
            ctx.set_resident_share(None, 0, 1)
        ln = ctx.ct_to_lanes(part, 5).clone()
        total = ln if total is None else total + ln
    assert rows == 2 * p.d + p.m
    assert np.array_equal(ctx.to_host(ctx.ct_from_lanes(total, 5)), ctx.to_host(ref))


@pytest.mark.parametrize("nres", [0, 1, 255, 300, 512, 513, 540])
def test_partially_resident_crs(ctx, instance, nres):
    """rows [0, nres) streamed from the expanded image, the rest regenerated from the seed: same proof for every split point
    (inside S, at the S/AS boundary, inside AS, at the AS/BT boundary, inside BV)"""
    I = instance
    p = I["p"]
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    delta = 5150
    ref = ctx.to_host(ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, 0, 1)).copy()
    image = ctx.crs_expand(0, max(nres, 1), d_crs)
    ctx.set_resident_prefix(image, nres)
    try:
        got = ctx.to_host(ctx.prove_partial(d_crs, I["d_ssp"], I["bits"], delta, 0, 1)).copy()
    finally:
        ctx.set_resident(None)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_prover_scheduling_modes_and_zero_coefficients(ctx, oracle, instance, mode):
    """Every queueing mode of mfh_set_overlap gives the oracle's proof, also when whole coefficient vectors are zero: with
    delta = 0 and an all-zero witness w = 0, so the S region is evaluated with an all-zero first vector on the path that
    skips the row compaction (dense hint), and b_w's coefficient vector is all zero on the path that compacts."""
    I = instance
    p = I["p"]
    zero_bits = bytes(len(I["bits"]))
    mags = bytes(range(80)) * 5
    signs = bytes([0, 1, 0, 1, 0])
    tape = b"".join(mags[80 * k: 80 * k + 80] + signs[k: k + 1] for k in range(5))
    d_crs = ctx.to_device(_crs_stream_order(p, I["crs"]))
    ctx.set_overlap(mode)
    try:
        for bits, delta in ((zero_bits, 0), (I["bits"], 12345)):
            ref = oracle.prover(p, I["crs"], I["ssp"], bits, delta, tape, 80)
            got = ctx.to_host(ctx.prove(d_crs, I["d_ssp"], bits, delta, mags, signs), np.uint64).reshape(5, p.n + 1, p.L)
            for k in range(5):
                assert np.array_equal(got[k], ref["proof"][k]), (mode, delta, k)
    finally:
        ctx.set_overlap(1)
