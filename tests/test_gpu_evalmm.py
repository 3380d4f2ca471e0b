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
        co[1 % nvec, :] = 0xFFFFFFFA      # p - 1 everywhere
        co[3 % nvec, :] = 0xFFFFFFFF      # every byte 255 (the ABI takes any uint32)
        co[4 % nvec, :] = 0x80808080      # every offset digit exactly 0
        co[2 % nvec, ::2] = 1             # 0/1 vectors like b_w's witness bits
        co[2 % nvec, 1::2] = 0
    return co


@pytest.mark.parametrize("nrows,nvec,kind,off_rows", [
    (1, 1, "rand", 0), (5, 2, "rand", 3), (128, 3, "edges", 0), (129, 12, "rand", 7), (300, 13, "edges", 1), (1100, 31, "rand", 2), (64, 16, "edges", 0),
    (3, 32, "rand", 0), (257, 40, "edges", 1), (700, 63, "rand", 3),  # > 31 vectors: the 256-column kernel (k_evalmm16)
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


def test_multi_one_byte_coefficients(ctx):
    """coeff_bytes = 1: up to 127 vectors with coefficients < 256 (b_w's witness bits), one digit column each"""
    p = ctx.params
    rng = np.random.default_rng(11)
    nrows, nvec = 300, 200
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = rng.integers(0, 2, size=(nvec, nrows), dtype=np.uint32)
    co[0, :] = 255
    co[1, :] = 0
    co[2, :] = rng.integers(0, 256, size=nrows, dtype=np.uint32)
    d_c8 = ctx.to_device(c8)
    off = p.ctr_bt
    got = ctx.to_host(ctx.eval_rows_multi(off, nrows, d_c8, ctx.to_device(co), nvec, coeff_bytes=1), np.uint64).reshape(nvec, p.n + 1, p.L)
    for v in (0, 1, 2, 3, 33, 69, 127, 128, 199):
        ref, _ = ctx.eval_rows(off, nrows, d_c8, ctx.to_device(co[v]))
        assert np.array_equal(got[v], ctx.to_host(ref, np.uint64).reshape(p.n + 1, p.L)), f"vector {v}"


def test_multi_argument_checks(ctx):
    import c_lwe_snarks_amd as mf

    p = ctx.params
    with pytest.raises(mf.MfhError):
        ctx.eval_rows_multi(0, 4, ctx.zeros(4 * p.ctb), ctx.zeros(64 * 4 * 4), 64)
    with pytest.raises(mf.MfhError):  # the 256-column kernel needs row segments at byte 0 or 8 of an AES block
        ctx.eval_rows_multi(4, 4, ctx.zeros(4 * p.ctb), ctx.zeros(40 * 4 * 4), 40)
    out = ctx.eval_rows_multi(0, 0, ctx.zeros(16), ctx.zeros(16), 2)  # no rows: zero ciphertexts
    assert not ctx.to_host(out).any()


@pytest.mark.parametrize("nproofs,transient_image", [(1, True), (5, True), (16, True), (33, True), (33, False), (126, False), (250, True), (250, False), (500, True)])
def test_prove_batch_equals_single_proofs(gpu_ctx_factory, oracle, nproofs, transient_image):
    """mfh_prove_batch: proof b of the batch == mfh_prove(statement b), bit for bit (S / AS groups of 63 - 64 coefficient vectors on the 256-column kernel when the image is streamed, of 31 proofs when it is regenerated, smaller groups on the 128-column one, BT+BV super-groups of 255 (128-column kernel up to 127 proofs, 256-column kernel above): 33 = one
    full + one partial S / AS group, 250 = two BT+BV groups, 500 = three, so the double-buffered w / h / v scratch of the chain is reused); the first proof is also checked against the oracle's prover and every proof is accepted by the device verifier.
    transient_image: calls with more than 31 proofs expand the CRS once per call into a scratch image and stream it for every group
    (the default), or (mfh_set_batch_image 0) regenerate the keystream per group."""
    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    c.set_batch_image(transient_image)
    rng = np.random.default_rng(4242)
    nbytes = (p.m + 7) // 8
    wits = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(nproofs)]
    tape = rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8)
    # one SSP must be satisfied by every witness of the batch: random_ssp builds it around ONE witness, so use the same satisfying
    # witness with different deltas / smudging for the accepted proofs, and flipped witnesses for the rest (those must be rejected)
    ssp = oracle.ssp_from_tape(p, tape, wits[0])
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    crs = oracle.setup(p, SEED, ssp, alpha, beta, s, sk, etape)
    d_ssp = c.ssp_upload(ssp)
    c.ssp_prepare(d_ssp)
    d_crs = c.to_device(np.concatenate([crs["s"], crs["as_"], crs["t"], crs["v"][: (p.m - 1) * p.ctb]]))
    valid = [b % 3 != 2 for b in range(nproofs)]
    stmts = [wits[0] if valid[b] else wits[b] for b in range(nproofs)]
    if nproofs > 2:
        assert stmts[2] != wits[0]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nproofs, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nproofs)]
    signs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nproofs)]
    got = c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs), np.uint64).reshape(nproofs, 5, p.n + 1, p.L).copy()
    for b in range(nproofs):
        one = c.to_host(c.prove(d_crs, d_ssp, stmts[b], deltas[b], mags[b], signs[b]), np.uint64).reshape(5, p.n + 1, p.L)
        assert np.array_equal(got[b], one), f"proof {b} of the batch differs from the single-proof path"
    stape = b"".join(mags[0][80 * k: 80 * k + 80] + signs[0][k: k + 1] for k in range(5))
    ref = oracle.prover(p, crs, ssp, stmts[0], deltas[0], stape, 80)
    assert np.array_equal(got[0], np.stack(ref["proof"]))
    ok = c.to_host(c.verify(d_ssp, alpha, beta, s, c.to_device(sk), c.to_device(got), nproofs))
    assert [bool(x) for x in ok] == valid


def test_batch_prover_from_the_matrix_core_crs_image(gpu_ctx_factory, oracle):
    """Second regime of the batch prover: the CRS expanded once into an MFMA A-fragment image (mfh_crs_expand_mm); with the image
    registered mfh_eval_rows_multi / mfh_prove_batch stream it from HBM (k_mmstream) and give the bytes of the regenerate path."""
    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    rng = np.random.default_rng(515)
    nbytes = (p.m + 7) // 8
    wit = rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()
    ssp = oracle.ssp_from_tape(p, rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8), wit)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    d_ssp = c.ssp_upload(ssp)
    c.ssp_prepare(d_ssp)
    d_crs = c.setup(d_ssp, alpha, beta, s, c.to_device(sk), c.to_device(etape))
    nb = 35
    stmts = [wit if b % 2 == 0 else rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for b in range(nb)]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nb, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
    signs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nb)]
    c.set_batch_image(False)  # every group regenerates the keystream
    regen = c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs)).copy()
    c.set_batch_image(True)   # the call expands the CRS into its own transient image
    assert np.array_equal(c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs)), regen)
    image = c.crs_expand_mm(d_crs)
    assert image.numel() == int(c.lib.mfh_crs_mm_image_bytes(c._h)) == 3 * (736 * 11) * 4 * 1024  # row tiles x k-steps x 1 KiB
    c.set_resident_mm(image)
    try:
        res = c.to_host(c.prove_batch(d_crs, d_ssp, stmts, deltas, mags, signs))
        assert np.array_equal(res, regen)
        # a region that is not one of the image's three (other row count) still regenerates, and agrees with the VALU path
        co = rng.integers(0, ol.P, size=(2, 10), dtype=np.uint64).astype(np.uint32)
        got = c.to_host(c.eval_rows_multi(p.ctr_as, 10, d_crs[p.d * p.ctb:], c.to_device(co), 2), np.uint64).reshape(2, p.n + 1, p.L)
        r0, r1 = c.eval_rows(p.ctr_as, 10, d_crs[p.d * p.ctb:], c.to_device(co[0]), c.to_device(co[1]))
        assert np.array_equal(got[0], c.to_host(r0, np.uint64).reshape(p.n + 1, p.L))
        assert np.array_equal(got[1], c.to_host(r1, np.uint64).reshape(p.n + 1, p.L))
        # the whole AS region with 3 vectors: from the image == from the seed
        co3 = rng.integers(0, ol.P, size=(3, p.d), dtype=np.uint64).astype(np.uint32)
        a = c.to_host(c.eval_rows_multi(p.ctr_as, p.d, d_crs[p.d * p.ctb:], c.to_device(co3), 3)).copy()
    finally:
        c.set_resident_mm(None)
    b = c.to_host(c.eval_rows_multi(p.ctr_as, p.d, d_crs[p.d * p.ctb:], c.to_device(co3), 3))
    assert np.array_equal(a, b)
    ok = c.to_host(c.verify(d_ssp, alpha, beta, s, c.to_device(sk), c.to_device(res), nb))
    assert [bool(x) for x in ok] == [b % 2 == 0 for b in range(nb)]


@pytest.mark.parametrize("nstmt", [1, 7, 32, 33, 64, 65, 124, 128, 129, 200, 248, 256])
def test_witness_pass_on_the_matrix_cores(gpu_ctx_factory, nstmt):
    """mfh_witness_poly_mm (bits x SSP bytes as a GEMM, one read of the SSP) and mfh_witness_poly_multi (VALU, 12 at a time) give
    mfh_witness_poly's polynomials, also for all-zero / all-one witnesses and edge SSP values (0, p - 1)."""
    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(900 + nstmt)
    ssp = rng.integers(0, ol.P, size=(p.m + 3) * p.d, dtype=np.uint64)
    ssp[: 3 * p.d: 2] = ol.P - 1
    ssp[1: 3 * p.d: 2] = 0
    ssp[5 * p.d: 6 * p.d] = ol.P - 1
    d_ssp = c.ssp_upload(ssp)
    nbytes = (p.m + 7) // 8
    wits = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(nstmt)]
    wits[0] = bytes(nbytes)
    if nstmt > 1:
        wits[1] = b"\\xff" * nbytes
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nstmt, dtype=np.uint64)]
    deltas[0] = 0
    got_mm = c.to_host(c.witness_poly_many(d_ssp, wits, deltas, mm=True), np.uint32).reshape(nstmt, p.d)
    for b in range(nstmt):
        ref = c.to_host(c.witness_poly(d_ssp, wits[b], deltas[b]), np.uint32)
        assert np.array_equal(got_mm[b], ref), f"statement {b}"
    if nstmt <= 12:
        got_v = c.to_host(c.witness_poly_many(d_ssp, wits, deltas, mm=False), np.uint32).reshape(nstmt, p.d)
        assert np.array_equal(got_v, got_mm)


@pytest.mark.parametrize("nstmt,d,m", [(248, 256, 1100), (130, 384, 1000), (256, 128, 777), (124, 256, 1100), (248, 128, 2300), (129, 128, 4000)])
def test_witness_pass_row_steps_and_chunks(gpu_ctx_factory, nstmt, d, m):
    """The same at sizes where a row chunk has several 4-step rounds and a 1..3-step tail (the 256-statement pass keeps its bit fragments
    in an LDS ring two steps ahead and its SSP fragments four steps ahead; up to 64 row steps -- or at d >= 2^14 -- it has one row chunk and
    finishes in the kernel, else it leaves chunk partials to k_witness_mm_finish), against the VALU form 12 statements at a time."""
    import c_lwe_snarks_amd as mf

    p = mf.Params(d=d, m=m)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(7000 + nstmt + m)
    ssp = rng.integers(0, ol.P, size=(p.m + 3) * p.d, dtype=np.uint64)
    ssp[7 * p.d: 8 * p.d] = ol.P - 1
    d_ssp = c.ssp_upload(ssp)
    nbytes = (p.m + 7) // 8
    wits = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(nstmt)]
    wits[nstmt - 1] = b"\xff" * nbytes
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nstmt, dtype=np.uint64)]
    got_mm = c.to_host(c.witness_poly_many(d_ssp, wits, deltas, mm=True), np.uint32).reshape(nstmt, p.d)
    for b0 in range(0, nstmt, 12):
        b1 = min(nstmt, b0 + 12)
        ref = c.to_host(c.witness_poly_many(d_ssp, wits[b0:b1], deltas[b0:b1], mm=False), np.uint32).reshape(b1 - b0, p.d)
        assert np.array_equal(got_mm[b0:b1], ref), f"statements {b0}..{b1}"
    one = c.to_host(c.witness_poly(d_ssp, wits[nstmt - 1], deltas[nstmt - 1]), np.uint32)
    assert np.array_equal(got_mm[nstmt - 1], one)


@pytest.mark.parametrize("logq,d,m", [(736, 256, 64), (736, 1152, 1000), (1472, 128, 24), (736, 320, 70)])
def test_expansion_kernels_write_the_same_image(gpu_ctx_factory, logq, d, m):
    """mfh_crs_expand_mm through the barrier-free writer (k_expand_mm: lane = row, 16 x 16 byte transposition on the matrix cores) and
    through the LDS-tile writer (k_evalmm16<MODE 1>): the same bytes for every row tile, byte position and row that exists, in all three
    regions (S and AS start at byte 0 / 8 of an AES block by row parity; BT+BV has m rows: not a multiple of 64), and for a rank's shares.
    (Row tiles past the b coordinate and rows past the region are never read back and are left to each kernel.)"""
    import c_lwe_snarks_amd as mf

    p = mf.Params(logq=logq, d=d, m=m)
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    rng = np.random.default_rng(d + m)
    d_crs = c.to_device(rng.integers(0, 256, size=(2 * p.d + p.m) * p.ctb, dtype=np.uint8))
    mt_per_tile, ct = (11, 2) if logq == 736 else (12, 1)
    mtiles = (p.n + 1 + ct - 1) // ct * mt_per_tile
    for rank, world in ((0, 1), (1, 3)):
        imgs = []
        for path in (1, 0):
            c.set_expand_path(path)
            try:
                imgs.append(c.to_host(c.crs_expand_mm_share(d_crs, rank, world)).copy())
            finally:
                c.set_expand_path(0)
        old, new = imgs
        assert old.size == new.size
        base = 0
        for total in (p.d, p.d, p.m):
            rows = total * (rank + 1) // world - total * rank // world
            KS = (rows + 255) // 256 * 4
            nb = mtiles * KS * 1024
            a = old[base:base + nb].reshape(mtiles, KS, 4, 16, 16)  # [row tile][k-step][row group g4][byte position c16][row e]
            b = new[base:base + nb].reshape(mtiles, KS, 4, 16, 16)
            base += nb
            row_of = (64 * np.arange(KS)[:, None, None] + 16 * np.arange(4)[None, :, None] + np.arange(16)[None, None, :])  # [ks][g4][e]
            rmask = (row_of < rows)[None, :, :, None, :]
            if logq == 736:
                pos_ok = np.ones((mtiles, 16), dtype=bool)
                last = 11 * ((p.n + 1) // 2) + 5 if (p.n + 1) % 2 else None  # the tile holding b's bytes 80..87
                full = 22 * (p.n // 4) + 11 * ((p.n % 4) // 2) + 5  # row tiles completely filled: keystream coordinates + 80 bytes of b
                pos_ok[full + 1:] = False
                pos_ok[full, 8:] = False
                assert last is None or last == full
            else:
                pos_ok = np.ones((mtiles, 16), dtype=bool)
                pos_ok[11::12, 8:] = False  # a coordinate's 12th row tile holds 8 bytes
            mask = rmask & pos_ok[:, None, None, :, None]
            assert np.array_equal(a[np.broadcast_to(mask, a.shape)], b[np.broadcast_to(mask, b.shape)]), (rank, world, total)
    c.close()


@pytest.mark.parametrize("nrows,chunk_rows,nvec,cb", [(2 * 131071, 0, 3, 4), (2 * 131071, 0, 40, 4), (1100, 500, 3, 4)])
def test_row_chunks_stay_within_the_int32_bound(ctx, nrows, chunk_rows, nvec, cb):
    """a chunk of <= 131 071 rows rounded UP to whole 256-row stages could reach 131 072 (nrows = 2 x 131 071: two chunks of
    131 072).  The limit is now rounded DOWN to whole stages before dividing (three chunks here; 500 -> chunks of 256); both the
    128-column (3 vectors) and the 256-column kernel (40 vectors), compared with the VALU path (k_eval) over the same rows."""
    p = ctx.params
    rng = np.random.default_rng(nrows + nvec)
    c8 = ctx.to_device(rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8))
    co = rng.integers(0, ol.P, size=(nvec, nrows), dtype=np.uint64).astype(np.uint32)
    co[0, :] = 0xFFFFFFFA  # p - 1 everywhere: the largest digits
    ctx.set_mm_chunk_rows(chunk_rows)
    try:
        got = ctx.to_host(ctx.eval_rows_multi(p.ctr_s, nrows, c8, ctx.to_device(co), nvec, coeff_bytes=cb), np.uint64).reshape(nvec, p.n + 1, p.L)
    finally:
        ctx.set_mm_chunk_rows(0)
    for v in (0, 1, nvec - 1):
        ref, _ = ctx.eval_rows(p.ctr_s, nrows, c8, ctx.to_device(co[v]))
        assert np.array_equal(got[v], ctx.to_host(ref, np.uint64).reshape(p.n + 1, p.L)), f"vector {v}"


def test_tuning_setters_check_their_ranges(ctx):
    import c_lwe_snarks_amd as mf

    for bad in (65, 1000):
        with pytest.raises(mf.MfhError):
            ctx.set_encrypt_chunks(bad)
    for bad in (1, 31, 257):
        with pytest.raises(mf.MfhError):
            ctx.set_witness_per(bad)
    ctx.set_encrypt_chunks(8)
    ctx.set_encrypt_chunks(0)
    ctx.set_witness_per(124)
    ctx.set_witness_per(0)
