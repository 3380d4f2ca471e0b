"""GPU: the generator-defined SSP (BASELINE configs 4/5).  Small sizes: generator mode == dense mode on the
materialised image == the oracle.  Full config shapes (D = 2^20 constraints, M = 699 050 wires; log q = 736 and 1472), which the
reference cannot run at all: setup -> prover -> device verifier, acceptance / rejection properties."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SEED = bytes((17 * i + 9) & 0xFF for i in range(40))
PRG_SEED = 0x0123456789ABCDEF


@pytest.fixture(scope="module")
def mf():
    import c_lwe_snarks_amd as m

    return m


def test_generator_mode_equals_dense_mode_and_oracle(gpu_ctx_factory, oracle, mf):
    p = mf.DEBUG
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    rng = np.random.default_rng(5)
    bits = rng.bytes((p.m + 7) // 8)
    d_t = ctx.ssp_prg_make_t(PRG_SEED, bits)
    ctx.ssp_set_prg(PRG_SEED, d_t)
    import torch

    dense = torch.cat([d_t, ctx.ssp_prg_fill(PRG_SEED, 1, p.m + 2)])  # slot 0 = t, slots 1.. = generator
    host = ctx.to_host(dense, np.uint32).astype(np.uint64)  # the oracle's (and the reference's) uint64 layout
    assert (host < ol.P).all()
    ssp2 = host.reshape(p.m + 3, p.d)
    v = ssp2[1].copy()
    for i in range(1, p.m):
        if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
            v = (v + ssp2[i + 1]) % np.uint64(ol.P)
    assert oracle.poly_divides(v, ssp2[0])  # make_t built a valid instance
    delta = 31337
    assert torch.equal(ctx.witness_poly(None, bits, delta), ctx.witness_poly(dense, bits, delta))
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    assert torch.equal(ctx.setup_messages(None, alpha, beta, s), ctx.setup_messages(dense, alpha, beta, s))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    d_sk, d_err = ctx.to_device(sk), ctx.to_device(etape)
    ctx.ssp_prepare(None)
    d_crs = ctx.setup(None, alpha, beta, s, d_sk, d_err)
    crs = oracle.setup(p, SEED, host, alpha, beta, s, sk, etape)
    assert np.array_equal(ctx.to_host(d_crs), np.concatenate([crs["s"], crs["as_"], crs["t"], crs["v"][: (p.m - 1) * p.ctb]]))
    mags = rng.integers(0, 256, size=400, dtype=np.uint8).tobytes()
    signs = bytes([1, 0, 0, 1, 0])
    proof = ctx.prove(d_crs, None, bits, delta, mags, signs)
    t5 = b"".join(mags[80 * k: 80 * k + 80] + signs[k: k + 1] for k in range(5))
    ref = oracle.prover(p, crs, host, bits, delta, t5, 80)
    assert np.array_equal(ctx.to_host(proof, np.uint64).reshape(5, p.n + 1, p.L), ref["proof"])
    assert int(ctx.to_host(ctx.verify(None, alpha, beta, s, d_sk, proof, 1))[0]) == 1
    assert int(ctx.to_host(ctx.verify(dense, alpha, beta, s, d_sk, proof, 1))[0]) == 1
    # sharded witness lanes in generator mode
    lanes = sum(ctx.witness_lanes(None, bits, r, 3).clone() for r in range(3))
    part = ctx.prove_partial_w(d_crs, None, bits, delta, 0, 1, lanes)
    ctx.prove_finish(part, mags, signs)
    assert torch.equal(part, proof)
    # the batch prover in generator mode: 14 statements (the witness pass generates every selected row once per 12) == the dense SSP
    nb = 14
    stmts = [bits if b % 3 else rng.bytes((p.m + 7) // 8) for b in range(nb)]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nb, dtype=np.uint64)]
    bm = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
    bs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nb)]
    ctx.ssp_prepare(None)
    gen = ctx.prove_batch(d_crs, None, stmts, deltas, bm, bs).clone()
    ctx.ssp_prepare(dense)
    assert torch.equal(ctx.prove_batch(d_crs, dense, stmts, deltas, bm, bs), gen)
    ctx.ssp_prepare(None)
    assert torch.equal(gen.view(nb, -1)[1], ctx.prove(d_crs, None, stmts[1], deltas[1], bm[1], bs[1]))


@pytest.mark.parametrize("logq", [736, 1472])
def test_full_2pow20_config_shape(gpu_ctx_factory, mf, logq):
    """BASELINE config 4 (logq 736) / 5 (logq 1472): D = 1 048 576, M = 699 050: 2 796 202 CRS rows, 378 / 757 GB of public
    stream regenerated per proof, 3.7e11 generator-defined SSP coefficients per witness polynomial."""
    import torch

    p = mf.Params(logq=logq, d=1 << 20, m=699050)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    rng = np.random.default_rng(logq)
    bits = rng.bytes((p.m + 7) // 8)
    d_t = ctx.ssp_prg_make_t(PRG_SEED, bits)
    ctx.ssp_set_prg(PRG_SEED, d_t)
    ctx.ssp_prepare(None)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    g = torch.Generator(device=ctx.device)
    g.manual_seed(logq)
    sk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (p.n, p.L), dtype=torch.int64, device=ctx.device, generator=g)
    if p.logq - 64 * (p.L - 1) < 64:
        sk[:, p.L - 1] &= (1 << (p.logq - 64 * (p.L - 1))) - 1
    rows = 2 * p.d + p.m
    err = torch.randint(-(2 ** 63), 2 ** 63 - 1, (rows, p.L), dtype=torch.int64, device=ctx.device, generator=g)
    err[:, 8] &= (1 << 47) - 1
    err[:, 9:] = 0
    d_sk, d_err = sk.view(torch.uint8).reshape(-1), err.view(torch.uint8).reshape(-1)
    d_crs = ctx.setup(None, alpha, beta, s, d_sk, d_err)
    del err, d_err
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=400, dtype=np.uint8).tobytes()
    signs = bytes([0, 1, 1, 0, 1])
    torch.cuda.synchronize()
    import time

    t0 = time.perf_counter()
    proof = ctx.prove(d_crs, None, bits, delta, mags, signs)
    torch.cuda.synchronize()
    print(f"\n[config D=2^20 M=699050 logq={logq}] one proof on one MI355X: {time.perf_counter() - t0:.3f} s")
    assert int(ctx.to_host(ctx.verify(None, alpha, beta, s, d_sk, proof, 1))[0]) == 1
    bad = bytearray(bits)
    bad[1000] ^= 8
    assert int(ctx.to_host(ctx.verify(None, alpha, beta, s, d_sk, ctx.prove(d_crs, None, bytes(bad), delta, mags, signs), 1))[0]) == 0
    ctx.close()


def test_config4_row_sharded_batch_emulated_on_one_gpu(gpu_ctx_factory, mf):
    """BASELINE config 4 at its full shape through the ROW-SHARDED batch prover, the 8 ranks emulated one after the other on one GPU:
    every rank expands its own 45 GB share of the matrix-core image (350 K rows; a 2^20 / 8 = 131 072-row S / AS share is one row more
    than an int32 accumulator holds: two row chunks), streams it for a group of 31 statements and contributes uint64 lanes; the summed
    proofs must equal mfh_prove_batch's (which regenerates the keystream: the 363 GB image fits no single GPU) bit for bit, verify for
    the satisfying witnesses and fail for the others."""
    import time

    import torch

    from c_lwe_snarks_amd import dist as mfdist

    p = mf.Params(d=1 << 20, m=699050)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    rng = np.random.default_rng(4)
    bits0 = rng.bytes((p.m + 7) // 8)
    d_t = ctx.ssp_prg_make_t(PRG_SEED, bits0)
    ctx.ssp_set_prg(PRG_SEED, d_t)
    ctx.ssp_prepare(None)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    g = torch.Generator(device=ctx.device)
    g.manual_seed(4)
    sk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (p.n, p.L), dtype=torch.int64, device=ctx.device, generator=g)
    sk[:, p.L - 1] &= (1 << (p.logq - 64 * (p.L - 1))) - 1
    rows = 2 * p.d + p.m
    err = torch.randint(-(2 ** 63), 2 ** 63 - 1, (rows, p.L), dtype=torch.int64, device=ctx.device, generator=g)
    err[:, 8] &= (1 << 47) - 1
    err[:, 9:] = 0
    d_sk = sk.view(torch.uint8).reshape(-1)
    d_crs = ctx.setup(None, alpha, beta, s, d_sk, err.view(torch.uint8).reshape(-1))
    del err
    nb, world = 31, 8
    stmts = [bits0 if b % 2 == 0 else rng.bytes(len(bits0)) for b in range(nb)]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nb, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
    signs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nb)]
    want = ctx.prove_batch(d_crs, None, stmts, deltas, mags, signs).clone()
    per, owned = mfdist.statement_shares(nb, world)
    shares = mfdist.row_shares(p.d, world)
    # the chain as dist.prove_batch_sharded runs it for a generator-defined SSP: every rank its coefficient range of w of ALL statements
    # (1 / 8 of the generation of the selected rows), the ranges handed to the statement owners, who finish the chain
    w_all = torch.empty((nb, p.d), dtype=torch.int32, device=ctx.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for lo, hi in shares:
        w_all[:, lo:hi] = ctx.batch_witness_cols(None, stmts, deltas, lo, hi - lo)
    torch.cuda.synchronize()
    t_wit = time.perf_counter() - t0
    t0 = time.perf_counter()
    whv = []
    for a, b in owned:
        x = torch.empty((3, b - a, p.d), dtype=torch.int32, device=ctx.device)
        x[0] = w_all[a:b]
        whv.append(ctx.batch_chain_from_w(None, x))
    torch.cuda.synchronize()
    t_fin = time.perf_counter() - t0
    del w_all
    one = ctx.batch_chain(None, stmts[:2], deltas[:2])
    assert torch.equal(one, whv[0][:, :2])  # (the per-owner chain gives the same polynomials)
    lps = 5 * (p.n + 1) * p.lanes
    total = torch.zeros(per * world * lps, dtype=torch.int64, device=ctx.device)
    image = ctx.empty(max(int(ctx.lib.mfh_crs_mm_share_bytes(ctx._h, r, world)) for r in range(world)))
    t_exp = t_rows = 0.0
    for r in range(world):
        lo, hi = shares[r]
        cs = hi - lo
        recv = torch.cat([w[:, :, lo:hi].permute(1, 0, 2).reshape(-1) for w in whv])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.crs_expand_mm_share(d_crs, r, world, out=image)
        torch.cuda.synchronize()
        t_exp += time.perf_counter() - t0
        ctx.set_resident_mm_share(image, r, world)
        try:
            t0 = time.perf_counter()
            partial = ctx.prove_batch_partial(d_crs, r, world, stmts, recv, recv[cs:], recv[2 * cs:], 3 * cs)
            torch.cuda.synchronize()
            t_rows += time.perf_counter() - t0
        finally:
            ctx.set_resident_mm_share(None, 0, 1)
        lanes = torch.zeros_like(total)
        ctx.ct_to_lanes(partial, nb * 5, out=lanes)
        total += lanes
        del recv, partial, lanes
    got = []
    for r, (a, b) in enumerate(owned):
        if b > a:
            proofs = ctx.ct_from_lanes(total[r * per * lps:(r * per + (b - a)) * lps], (b - a) * 5)
            ctx.prove_batch_finish(d_crs, deltas[a:b], mags[a:b], signs[a:b], proofs)
            got.append(proofs)
    got = torch.cat(got)
    print(f"\n[config 4, row-sharded batch, 8 ranks emulated on one GPU, {nb} statements] per rank: share image {image.numel() / 1e9:.1f} GB expanded in "
          f"{t_exp / world * 1e3:.0f} ms, row work {t_rows / world * 1e3:.0f} ms, its coefficient range of w of all statements "
          f"{t_wit / world * 1e3:.0f} ms; the rest of the chains of all statements {t_fin * 1e3:.0f} ms")
    assert torch.equal(got, want)
    ok = ctx.to_host(ctx.verify(None, alpha, beta, s, d_sk, got, nb))
    assert [bool(x) for x in ok] == [b % 2 == 0 for b in range(nb)]
    del image
    ctx.close()


@pytest.mark.parametrize("d,m,nstmt", [(256, 70, 1), (256, 70, 33), (384, 1000, 65), (128, 333, 124), (256, 70, 129), (128, 1100, 248), (256, 2300, 256), (128, 4001, 200)])
def test_generator_defined_witness_pass_on_the_matrix_cores(gpu_ctx_factory, mf, d, m, nstmt):
    """mfh_witness_poly_mm with d_ssp = NULL: the B fragments of the bits x SSP-bytes GEMM are generated in the kernel (k_witness_mm_prg)
    instead of loaded; above 128 statements by k_witness_mm8q_prg, whose four waves per coefficient tile share the hashes through LDS
    (chunk partials + finishing kernel).  Same polynomials as the single-statement VALU pass over the generator (mfh_witness_poly) and as the GEMM over the
    dense image of the same SSP, also for all-zero / all-one witnesses."""
    import torch

    p = mf.Params(d=d, m=m)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(d + m + nstmt)
    nbytes = (p.m + 7) // 8
    wits = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(nstmt)]
    wits[0] = bytes(nbytes)
    if nstmt > 1:
        wits[1] = b"\xff" * nbytes
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nstmt, dtype=np.uint64)]
    d_t = c.ssp_prg_make_t(PRG_SEED, wits[-1])
    c.ssp_set_prg(PRG_SEED, d_t)
    got = c.to_host(c.witness_poly_many(None, wits, deltas, mm=True), np.uint32).reshape(nstmt, p.d)
    for b in range(nstmt):
        assert np.array_equal(got[b], c.to_host(c.witness_poly(None, wits[b], deltas[b]), np.uint32)), f"statement {b}"
    dense = torch.cat([d_t, c.ssp_prg_fill(PRG_SEED, 1, p.m + 2)])
    assert np.array_equal(got, c.to_host(c.witness_poly_many(dense, wits, deltas, mm=True), np.uint32).reshape(nstmt, p.d))
    c.close()


@pytest.mark.parametrize("d,m,nstmt,dense", [(512, 300, 40, False), (512, 2300, 250, False), (384, 1000, 130, True), (1024, 700, 31, True)])
def test_witness_pass_by_coefficient_range(gpu_ctx_factory, mf, d, m, nstmt, dense):
    """mfh_batch_witness_cols: the coefficients [col0, col0 + ncols) of w of every statement (what a rank of the row-sharded prover computes
    -- 1 / world of the rows' generation or read), for ranges that tile the polynomial, equal mfh_batch_chain's w restricted to the range;
    mfh_batch_chain_from_w on the assembled w gives the same h and v.  Generator-defined and dense SSP, all kernels of the pass (32 .. 256
    statements per pass, one row chunk finished in the kernel / several chunks)."""
    import torch

    p = mf.Params(d=d, m=m)
    c = gpu_ctx_factory(p)
    rng = np.random.default_rng(d + m + nstmt)
    nbytes = (p.m + 7) // 8
    wits = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(nstmt)]
    wits[0] = bytes(nbytes)
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nstmt, dtype=np.uint64)]
    d_t = c.ssp_prg_make_t(PRG_SEED, wits[-1])
    c.ssp_set_prg(PRG_SEED, d_t)
    d_ssp = torch.cat([d_t, c.ssp_prg_fill(PRG_SEED, 1, p.m + 2)]) if dense else None
    c.ssp_prepare(d_ssp)
    want = c.batch_chain(d_ssp, wits, deltas)
    cuts = [0, 128, 128, d - 128, d]  # (an empty range included)
    whv = torch.empty_like(want)
    for a, b in zip(cuts[:-1], cuts[1:]):
        got = c.batch_witness_cols(d_ssp, wits, deltas, a, b - a)
        assert torch.equal(got, want[0][:, a:b]), f"coefficients {a}..{b}"
        whv[0][:, a:b] = got
    c.batch_chain_from_w(d_ssp, whv)
    assert torch.equal(whv, want)
    with pytest.raises(mf.MfhError):
        c.batch_witness_cols(d_ssp, wits, deltas, 64, 128)  # a range must start at a multiple of 128
    c.close()
