"""The headline path (mfh_prove_batch: k_expand_mm image writer -> k_mmstream -> k_evalmm_finish, witness GEMM, batched
polynomial step) under ORACLE parity at sizes that exercise its loops, which the DEBUG-size tests of test_gpu_evalmm.py do not:

  * the NDEBUG default size (D = 2^15, M = 21845): 128 stages of 256 rows per k_mmstream launch; proof 0 of a 130-statement batch is
    the instance of tests/golden/make_default_size_golden.py and must reproduce the oracle's SHA-256s of all five ciphertexts;
  * intermediate sizes with D and M chosen so that the row counts are multiples of neither 256 nor 64, with several row chunks forced
    (mfh_set_mm_chunk_rows): k_mmstream's double-buffer swap, one-stage-ahead prefetch, `ulast` clamp, KS > 4 image addressing and
    nchunks > 1 all run, compared with the oracle's prover (src/snark.c:117-190 restated) and with the single-proof VALU path.
"""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = bytes((13 * i + 5) & 0xFF for i in range(40))
NAMES = ["h", "hat_h", "hat_v", "v_w", "b_w"]


def _statements(rng, p, n, valid_bits=None):
    nbytes = (p.m + 7) // 8
    bits = [valid_bits if (valid_bits is not None and b % 3 != 2) else rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for b in range(n)]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=n, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(n)]
    signs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(n)]
    return bits, deltas, mags, signs


@pytest.fixture(scope="module")
def golden():
    gdir = os.path.join(ROOT, "tests", "golden")
    gold = json.load(open(os.path.join(gdir, "default_size_proof.json")))
    spec = importlib.util.spec_from_file_location("mk_gold", os.path.join(gdir, "make_default_size_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return gold, mk


@pytest.mark.parametrize("mode", ["transient", "resident", "regenerate"])
def test_default_size_batch_proof0_matches_oracle_hashes(gpu_ctx_factory, golden, mode):
    """130 statements at mf.DEFAULT through mfh_prove_batch; statement 0 is the golden instance (oracle prover run, 111 s of CPU):
    its five ciphertexts must hash to tests/golden/default_size_proof.json.  transient: the call expands the CRS into its own image
    and streams it (k_mmstream, 4 groups per launch); resident: the image registered by the caller; regenerate: every group runs AES
    again (k_evalmm16<MODE 0>, set_batch_image(False)) -- that mode on the generator-defined SSP (VALU witness pass), the other two
    on the dense image of the same SSP (witness pass as a GEMM, 124 + 6 statements)."""
    import torch

    import c_lwe_snarks_amd as mf

    gold, mk = golden
    p = mf.DEFAULT
    I = mk.instance(p)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(I["seed"])
    d_t = ctx.ssp_prg_make_t(gold["prg_seed"], I["bits"])
    if mode == "regenerate":
        ctx.ssp_set_prg(gold["prg_seed"], d_t)
        d_ssp = None
    else:
        d_ssp = torch.cat([d_t, ctx.ssp_prg_fill(gold["prg_seed"], 1, p.m + 2)])
    ctx.ssp_prepare(d_ssp)
    d_crs = ctx.to_device(I["c8"])
    nb = 130
    rng = np.random.default_rng(77)
    bits, deltas, mags, signs = _statements(rng, p, nb)
    bits[0], deltas[0], mags[0], signs[0] = I["bits"], I["delta"], I["mags"], I["signs"]
    ctx.set_batch_image(mode != "regenerate")
    image = None
    if mode == "resident":
        image = ctx.crs_expand_mm(d_crs)
        ctx.set_resident_mm(image)
    try:
        out = ctx.prove_batch(d_crs, d_ssp, bits, deltas, mags, signs)
    finally:
        if image is not None:
            ctx.set_resident_mm(None)
    ctx.sync()
    nbytes = p.ct_limbs * 8
    proofs = out.view(nb, 5 * nbytes)
    sha = lambda t: hashlib.sha256(ctx.to_host(t).tobytes()).hexdigest()
    for k, nme in enumerate(NAMES):
        assert sha(proofs[0][k * nbytes:(k + 1) * nbytes]) == gold["proof_sha256"][nme], nme
    assert sha(proofs[0]) == gold["proof_all_sha256"]
    # other positions of the batch against the single-proof path (itself pinned to the same golden in test_gpu_fullsize.py):
    # last of the first group, first of the second, last of the first witness GEMM, first of the second, last statement
    for b in (30, 31, 123, 124, nb - 1):
        one = ctx.prove(d_crs, d_ssp, bits[b], deltas[b], mags[b], signs[b])
        assert torch.equal(proofs[b], one), f"statement {b}"
    del image, out
    ctx.close()


@pytest.mark.parametrize("mode", ["transient", "resident", "one_wave_per_simd"])
def test_default_size_shipped_launch_shape_matches_oracle_hashes(gpu_ctx_factory, golden, mode):
    """The launch shape bench.py's headline runs, under the golden at the NDEBUG default size: 510 statements = two full super-groups of 255, each
    served by ONE persistent k_mmstream_p launch over the S and AS images (8 + 8 groups of 63 / 64 coefficient vectors, 128 stages of 256 rows,
    506 tile groups) and b_w of both super-groups by ONE k_mmstream_pb launch over the BT+BV image (the loops of src/snark.c:147-174).  Statement 0
    is the golden instance (tests/golden/default_size_proof.json: the oracle's complete prover, 111 s of CPU) and must reproduce its five SHA-256s;
    statements at the super-group boundary and the last one equal the single-proof path (pinned to the same golden in test_gpu_fullsize.py).
    transient: the call's own image; resident: the caller's; one_wave_per_simd: the k_mmstream_w body (mfh_set_mm_stream(1, 2, 0, 0)).  The timing
    records say which grid ran: every streaming launch of the call must have been the persistent one."""
    import torch

    import c_lwe_snarks_amd as mf

    gold, mk = golden
    p = mf.DEFAULT
    I = mk.instance(p)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(I["seed"])
    d_t = ctx.ssp_prg_make_t(gold["prg_seed"], I["bits"])
    d_ssp = torch.cat([d_t, ctx.ssp_prg_fill(gold["prg_seed"], 1, p.m + 2)])
    ctx.ssp_prepare(d_ssp)
    d_crs = ctx.to_device(I["c8"])
    nb = 510
    rng = np.random.default_rng(510)
    bits, deltas, mags, signs = _statements(rng, p, nb)
    bits[0], deltas[0], mags[0], signs[0] = I["bits"], I["delta"], I["mags"], I["signs"]
    image = None
    if mode == "resident":
        image = ctx.crs_expand_mm(d_crs)
        ctx.set_resident_mm(image)
    if mode == "one_wave_per_simd":
        ctx.set_mm_stream(1, 2, 0, 0)
    for k in ("mmstream_rounds_persistent", "mmstream_bw_persistent", "mmstream_rounds", "mmstream_bw", "evalmm_resident", "evalmm"):
        ctx.timing_drain(k)
    ctx.set_timing(True)
    try:
        out = ctx.prove_batch(d_crs, d_ssp, bits, deltas, mags, signs)
        ctx.sync()
    finally:
        ctx.set_timing(False)
        ctx.set_mm_stream(1, True, 0, 64)
        if image is not None:
            ctx.set_resident_mm(None)
    # what ran: 2 persistent S + AS launches of 16 groups over 32768 rows, 1 persistent b_w launch of 2 groups over 21845 rows, nothing else streamed or regenerated
    n_rounds, _, rows_rounds = ctx.timing_drain("mmstream_rounds_persistent")
    work_rounds = ctx.timing_work_rows()
    n_bw, _, rows_bw = ctx.timing_drain("mmstream_bw_persistent")
    work_bw = ctx.timing_work_rows()
    assert (n_rounds, rows_rounds, work_rounds) == (2, 2 * p.d, 2 * 16 * p.d)
    assert (n_bw, rows_bw, work_bw) == (1, p.m, 2 * p.m)
    assert ctx.timing_drain("mmstream_rounds")[0] == 0 and ctx.timing_drain("mmstream_bw")[0] == 0  # no launch of the non-persistent grid
    assert ctx.timing_drain("evalmm_resident")[0] == 0 and ctx.timing_drain("evalmm")[0] == 0        # no single-group launch, no per-group AES
    nbytes = p.ct_limbs * 8
    proofs = out.view(nb, 5 * nbytes)
    sha = lambda t: hashlib.sha256(ctx.to_host(t).tobytes()).hexdigest()
    for k, nme in enumerate(NAMES):
        assert sha(proofs[0][k * nbytes:(k + 1) * nbytes]) == gold["proof_sha256"][nme], nme
    assert sha(proofs[0]) == gold["proof_all_sha256"]
    for b in (254, 255, nb - 1):
        one = ctx.prove(d_crs, d_ssp, bits[b], deltas[b], mags[b], signs[b])
        assert torch.equal(proofs[b], one), f"statement {b}"
    del image, out
    ctx.close()


def _instance(mf, ctx, oracle, p, seed_int):
    """valid SSP (oracle's random_ssp restatement), GPU setup -> CRS; returns what the oracle's prover / verifier need"""
    rng = np.random.default_rng(seed_int)
    nbytes = (p.m + 7) // 8
    wit = rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()
    ssp = oracle.ssp_from_tape(p, rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8), wit)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    d_ssp = ctx.ssp_upload(ssp)
    ctx.ssp_prepare(d_ssp)
    d_crs = ctx.setup(d_ssp, alpha, beta, s, ctx.to_device(sk), ctx.to_device(etape))
    c8 = ctx.to_host(d_crs)
    cb = p.ctb
    crs = dict(seed=SEED, s=c8[: p.d * cb].copy(), as_=c8[p.d * cb: 2 * p.d * cb].copy(), t=c8[2 * p.d * cb: (2 * p.d + 1) * cb].copy(),
               v=np.concatenate([c8[(2 * p.d + 1) * cb:], np.zeros(cb, dtype=np.uint8)]))
    return dict(rng=rng, wit=wit, ssp=ssp, alpha=alpha, beta=beta, s=s, sk=sk, d_ssp=d_ssp, d_crs=d_crs, crs=crs)


@pytest.mark.parametrize("d,m,nproofs,chunk_rows,mode", [
    (1152, 1000, 125, 0, "transient"),       # 4.5 stages per S / AS launch, 3.9 per BT+BV launch; 4 groups per launch + 1 single
    (1152, 1000, 64, 512, "transient"),      # 3 row chunks of 512 / 512 / 128 rows (S, AS), 2 of 512 / 488 (BT+BV)
    (3968, 777, 40, 1024, "resident"),       # 15.5 stages, 4 chunks, caller-registered image
    (1152, 1000, 35, 512, "regenerate"),     # k_evalmm16<MODE 0> with several chunks
    (2176, 321, 33, 0, "regenerate"),        # 8.5 units of 256 rows in one chunk of the AES kernel
])
def test_prove_batch_multistage_against_oracle(gpu_ctx_factory, oracle, d, m, nproofs, chunk_rows, mode):
    import torch

    import c_lwe_snarks_amd as mf

    p = mf.Params(d=d, m=m)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    W = _instance(mf, ctx, oracle, p, 31 * d + m)
    bits, deltas, mags, signs = _statements(W["rng"], p, nproofs, valid_bits=W["wit"])
    valid = [b % 3 != 2 for b in range(nproofs)]
    ctx.set_mm_chunk_rows(chunk_rows)
    ctx.set_batch_image(mode != "regenerate")
    image = None
    if mode == "resident":
        image = ctx.crs_expand_mm(W["d_crs"])
        ctx.set_resident_mm(image)
    try:
        out = ctx.prove_batch(W["d_crs"], W["d_ssp"], bits, deltas, mags, signs)
        if image is not None:
            # mfh_eval_rows_multi straight through the image against the oracle's eval_poly: the whole AS region, 3 vectors
            co = W["rng"].integers(0, ol.P, size=(3, p.d), dtype=np.uint64).astype(np.uint32)
            got = ctx.to_host(ctx.eval_rows_multi(p.ctr_as, p.d, W["d_crs"][p.d * p.ctb:], ctx.to_device(co), 3), np.uint64).reshape(3, p.n + 1, p.L)
            for v in range(3):
                exp = oracle.eval_poly(p, SEED, p.ctr_as, W["crs"]["as_"].tobytes(), co[v].astype(np.uint64))
                assert np.array_equal(got[v], exp.reshape(p.n + 1, p.L)), f"eval_rows_multi through the image, vector {v}"
    finally:
        if image is not None:
            ctx.set_resident_mm(None)
        ctx.set_mm_chunk_rows(0)
    got = ctx.to_host(out, np.uint64).reshape(nproofs, 5, p.n + 1, p.L).copy()
    # the oracle's complete prover for a valid and an invalid statement (first group) and the last statement of the batch
    for b in (0, 2, nproofs - 1):
        stape = b"".join(mags[b][80 * k: 80 * k + 80] + signs[b][k: k + 1] for k in range(5))
        ref = oracle.prover(p, W["crs"], W["ssp"], bits[b], deltas[b], stape, 80)
        assert np.array_equal(got[b], np.stack(ref["proof"])), f"statement {b} differs from the oracle's prover"
    for b in range(nproofs):
        one = ctx.to_host(ctx.prove(W["d_crs"], W["d_ssp"], bits[b], deltas[b], mags[b], signs[b]), np.uint64).reshape(5, p.n + 1, p.L)
        assert np.array_equal(got[b], one), f"proof {b} of the batch differs from the single-proof path"
    ok = ctx.to_host(ctx.verify(W["d_ssp"], W["alpha"], W["beta"], W["s"], ctx.to_device(W["sk"]), ctx.to_device(got), nproofs))
    assert [bool(x) for x in ok] == valid
    assert oracle.verifier(p, W["ssp"], W["alpha"], W["beta"], W["s"], W["sk"], got[0])
    del image, out
    ctx.close()


def test_chunk_rows_knob_rejects_overflowing_values(gpu_ctx_factory):
    import c_lwe_snarks_amd as mf

    ctx = gpu_ctx_factory(mf.DEBUG)
    with pytest.raises(mf.MfhError):
        ctx.set_mm_chunk_rows(131072)  # 131072 rows of -128 x -128 overflow an int32 accumulator
    ctx.set_mm_chunk_rows(131071)
    ctx.set_mm_chunk_rows(0)


@pytest.mark.parametrize("ngl,merge", [(1, True), (2, True), (4, True), (3, True), (4, False), (1, False)])
def test_batch_launch_shapes_give_identical_proofs(gpu_ctx_factory, ngl, merge):
    """mfh_set_batch_launch: groups per streaming launch and S + AS in one launch or two change the launch geometry of k_mmstream (per-group
    image pointers, up to 16 groups in a grid), never the proofs.  250 statements = two super-groups, 9 groups: every shape has a ragged
    last round."""
    import torch

    import c_lwe_snarks_amd as mf
    import oracle_lib  # noqa: F401

    import sys
    sys.path.insert(0, ROOT)
    import bench

    p = mf.Params(d=1152, m=1000)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 99)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(5)
    nb = 250
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()  # default shape: 8 groups per region, merged
    ctx.set_batch_launch(ngl, merge)
    try:
        got = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
    finally:
        ctx.set_batch_launch(8, True)
    assert torch.equal(got, want)
    one = ctx.prove(d_crs, inst["d_ssp"], bits[249], deltas[249], mags[249], signs[249])
    assert torch.equal(want.view(nb, -1)[249], one)
    ctx.close()


@pytest.mark.parametrize("layout", [(0, False, 0), (1, False, 0), (0, True, 0), (0, True, 1), (1, True, 1), (1, True, 2), (1, 2, 0), (0, 2, 0)])
@pytest.mark.parametrize("nb,ngl,chunk_rows", [(250, 8, 0), (250, 4, 512), (130, 8, 0), (96, 4, 256), (33, 8, 0), (255, 2, 0)])
def test_streaming_launch_layouts_give_identical_proofs(gpu_ctx_factory, layout, nb, ngl, chunk_rows):
    """mfh_set_mm_stream: which workgroup takes which (group, tile group) of a streaming launch (map 0 | 1), one workgroup per item or a persistent
    one-workgroup-per-CU grid looping over its XCD's items, with or without the speed-only rendezvous of the sharers -- never the proofs.  Against the
    round-3 layout (map 0, one workgroup per item).  Group counts per region that divide 32 (8, 4, 2: map 1 and the persistent grid apply) and that do
    not (130 statements = 5 groups, 96 with 4 per launch = 3 + 1: the host falls back to map 0 / one workgroup per item); 2 - 5 row chunks per item
    (the persistent grid walks chunk by chunk); 255 statements with 2 groups per region = 4 launches per super-group."""
    import sys

    import torch

    import c_lwe_snarks_amd as mf

    sys.path.insert(0, ROOT)
    import bench

    # (the doubled modulus for one shape: 12 instead of 11 row tiles per column tile, 1104 instead of 506 tile groups -- 138 per XCD, no multiple of 4 or 8)
    p = mf.Params(d=1152, m=1000, logq=1472 if (nb, ngl) == (250, 8) and layout[1] else 736)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 99)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(nb)
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    ctx.set_batch_launch(ngl, True)
    ctx.set_mm_chunk_rows(chunk_rows)
    try:
        ctx.set_mm_stream(0, False, 0, 0)
        want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()
        ctx.set_mm_stream(layout[0], layout[1], layout[2], 16)
        got = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
    finally:
        ctx.set_mm_stream(1, True, 0, 64)
        ctx.set_batch_launch(8, True)
        ctx.set_mm_chunk_rows(0)
    assert torch.equal(got, want)
    one = ctx.prove(d_crs, inst["d_ssp"], bits[nb - 1], deltas[nb - 1], mags[nb - 1], signs[nb - 1])
    assert torch.equal(want.view(nb, -1)[nb - 1], one)
    ctx.close()


@pytest.mark.parametrize("width,early,nb,ngl", [(28, False, 1000, 8), (28, True, 1000, 8), (24, True, 1000, 8), (32, True, 1000, 8), (5, True, 1000, 8),
                                                 (32, True, 300, 8), (28, True, 810, 8), (32, True, 810, 2), (28, True, 300, 2)])
def test_narrow_persistent_grid_and_early_chain_give_identical_proofs(gpu_ctx_factory, width, early, nb, ngl):
    """mfh_set_mm_width: the persistent S / AS launch on `width` workgroups per XCD (each then strides over its XCD's items by `width`) and, with early,
    the chain of super-group k + 1 queued beside the row work of k, the epilogues and smudging of k on the side stream beside the row work of k + 1 (two halves
    of the digit / partial-product scratch).  1000 statements = 4 super-groups, so every hand-over happens twice; never the proofs.
    300 = 255 + 45 and 810 = 3 x 255 + 45 statements: the last super-group is SHORT and has an odd index, so its scratch is smaller than its predecessor's -- the two
    halves must still be one full super-group apart (round 5 placed the short one at ITS size, inside the half its predecessor's epilogues were reading), and the
    caller's stream must wait for super-group k - 2's epilogues before it reuses their half; with 2 groups per launch (4 launches per super-group) a super-group's
    scratch is cut differently again."""
    import sys

    import torch

    import c_lwe_snarks_amd as mf

    sys.path.insert(0, ROOT)
    import bench

    p = mf.Params(d=1152, m=1000)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 99)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(width)
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()
    ctx.set_mm_width(width, early)
    ctx.set_batch_launch(ngl, True)
    try:
        got = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
        ctx.sync()
        again = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)  # (scratch halves and events re-used by a second call)
        ctx.sync()
    finally:
        ctx.set_mm_width(32, False)
        ctx.set_batch_launch(8, True)
    assert torch.equal(got, want) and torch.equal(again, want)
    for b in [x for x in (0, 254, 255, 299, 509, 510, 765, nb - 1) if x < nb]:
        one = ctx.prove(d_crs, inst["d_ssp"], bits[b], deltas[b], mags[b], signs[b])
        assert torch.equal(want.view(nb, -1)[b], one), f"statement {b}"
    ctx.close()


@pytest.mark.parametrize("logq,nb,ngl,chunk_rows,persistent,bw", [(736, 300, 8, 0, True, True), (736, 300, 4, 512, True, False), (736, 130, 8, 256, False, True), (1472, 130, 8, 0, True, True),
                                                                    (1472, 40, 2, 512, False, False), (736, 600, 8, 0, 2, True)])
def test_packed_partial_products_give_identical_proofs(gpu_ctx_factory, logq, nb, ngl, chunk_rows, persistent, bw):
    """mfh_set_mm_pack: the streaming kernels recombine the partial products they hold before writing them (four byte positions per lane; for four-byte coefficient
    vectors also the vector's four digit columns, a DPP quad) and the epilogue applies the signedness corrections to the recombined sums -- against the int32 layout of
    rounds 1 - 5, and against mfh_prove (k_eval: no matrix core, no digits).  Several row chunks (records of several chunks summed in the epilogue), both moduli (22 / 46
    words per value, 11 / 12 row tiles per column tile), the persistent grid and one workgroup per item, b_w merged (several one-byte groups per launch) and per super-group
    (k_mmstream1, with the delta ct_t term in its epilogue), borrowed ones columns (8 groups per region: the lender's record).  persistent = 2: the one-wave-per-SIMD body
    writes int32 whatever the switch says, and its epilogue must follow."""
    import sys

    import torch

    import c_lwe_snarks_amd as mf

    sys.path.insert(0, ROOT)
    import bench

    p = mf.Params(d=1152, m=1000, logq=logq)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 99)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(nb + logq)
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    ctx.set_batch_launch(ngl, True)
    ctx.set_mm_chunk_rows(chunk_rows)
    ctx.set_mm_stream(1, persistent, 0, 64)
    ctx.set_batch_bw(bw)
    try:
        ctx.set_mm_pack(False)
        want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()
        ctx.set_mm_pack(True)
        got = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
        # ... and one evaluation from a registered image (k_mmstream1, four-byte vectors and one-byte columns)
        image = ctx.crs_expand_mm(d_crs)
        ctx.set_resident_mm(image)
        co = rng.integers(0, mf.P, size=(5, p.d), dtype=np.uint64).astype(np.uint32)
        cb = rng.integers(0, 256, size=(7, p.d), dtype=np.uint32)
        d_co, d_cb = ctx.to_device(co), ctx.to_device(cb)
        c8 = d_crs[: p.d * p.ctb]
        ev = {}
        for pack in (False, True):
            ctx.set_mm_pack(pack)
            ev[pack] = (ctx.eval_rows_multi(p.ctr_s, p.d, c8, d_co, 5).clone(), ctx.eval_rows_multi(p.ctr_s, p.d, c8, d_cb, 7, coeff_bytes=1).clone())
        ctx.set_resident_mm(None)
    finally:
        ctx.set_mm_pack(True)
        ctx.set_mm_stream(1, True, 0, 64)
        ctx.set_batch_launch(8, True)
        ctx.set_mm_chunk_rows(0)
        ctx.set_batch_bw(True)
    assert torch.equal(got, want)
    assert torch.equal(ev[True][0], ev[False][0]) and torch.equal(ev[True][1], ev[False][1])
    for which, vec, k in ((0, co, 3), (1, cb, 6)):
        one = ctx.eval_rows(p.ctr_s, p.d, c8, ctx.to_device(vec[k]))[0]
        assert torch.equal(ev[True][which].view(vec.shape[0], -1)[k], one.reshape(-1)), (which, k)
    for b in (0, nb // 2, nb - 1):
        single = ctx.prove(d_crs, inst["d_ssp"], bits[b], deltas[b], mags[b], signs[b])
        assert torch.equal(got.view(nb, -1)[b], single), f"statement {b}"
    ctx.close()


@pytest.mark.parametrize("nslabs", [0, 3])
def test_stream_wait_hands_over_finished_super_groups(gpu_ctx_factory, nslabs):
    """mfh_prove_batch_stream_wait: a stream of the caller's waits for statements [0, upto) of the call just queued and copies them out while the later super-groups
    still run (what the host shim's mfuoco_prover_batch does with its copy stream): what arrives is what the output holds when the whole call has finished"""
    import sys

    import torch

    import c_lwe_snarks_amd as mf

    sys.path.insert(0, ROOT)
    import bench

    p = mf.Params(d=1152, m=1000)
    ctx = gpu_ctx_factory(p)
    assert ctx.prove_batch_supergroup() == 0
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 99)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(3)
    nb = 700
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    ctx.sync()
    torch.cuda.synchronize()
    ctx.set_batch_slabs(nslabs)  # (3: the out-of-core form -- every slab touches every proof, all super-groups complete together at the end of the call)
    try:
        out = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)  # queued, not waited for
    finally:
        ctx.set_batch_slabs(0)
    sg = ctx.prove_batch_supergroup()
    assert sg == 255
    st = torch.cuda.Stream()
    rows = out.view(nb, -1)
    host = torch.empty(rows.shape, dtype=rows.dtype, pin_memory=True)
    for a in range(0, nb, sg):
        b = min(a + sg, nb)
        ctx.prove_batch_stream_wait(b, st)
        with torch.cuda.stream(st):
            host[a:b].copy_(rows[a:b], non_blocking=True)
    st.synchronize()
    ctx.sync()
    assert torch.equal(host, rows.cpu())
    with pytest.raises(mf.MfhError):
        ctx.prove_batch_stream_wait(nb + 1, st)
    with pytest.raises(mf.MfhError):
        ctx.prove_batch_stream_wait(0, st)
    for b_ in (0, 254, 255, nb - 1):
        assert torch.equal(rows[b_], ctx.prove(d_crs, inst["d_ssp"], bits[b_], deltas[b_], mags[b_], signs[b_]))
    ctx.close()


@pytest.mark.parametrize("nb,persistent", [(600, 1), (600, 0), (2300, 1), (300, 2)])
def test_bw_of_all_super_groups_in_one_launch_gives_identical_proofs(gpu_ctx_factory, nb, persistent):
    """mfh_set_batch_bw: b_w (src/snark.c:143-155) of all super-groups of a call in ONE streaming launch per up to 8 of them over the BT+BV image (default) against one
    launch per super-group (rounds 1 - 3): 600 statements = 3 super-groups (a group count that does not divide 32: the non-persistent fallback), 2300 = 10 super-groups =
    a launch of 8 and one of 2, 300 = 2 with the one-wave-per-SIMD body; delta ct_t lands in every statement's b_w either way"""
    import sys

    import torch

    import c_lwe_snarks_amd as mf

    sys.path.insert(0, ROOT)
    import bench

    p = mf.Params(d=512, m=300)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 41)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(nb)
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    try:
        ctx.set_mm_stream(1, persistent, 0, 0)
        ctx.set_batch_bw(False)
        want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()
        ctx.set_batch_bw(True)
        got = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
    finally:
        ctx.set_mm_stream(1, 1, 0, 64)
        ctx.set_batch_bw(True)
    assert torch.equal(got, want)
    for b in (0, 255, nb - 1):
        one = ctx.prove(d_crs, inst["d_ssp"], bits[b], deltas[b], mags[b], signs[b])
        assert torch.equal(got.view(nb, -1)[b], one)
    ctx.close()


@pytest.mark.parametrize("d,m,nb,nslabs,chunk_rows", [(256, 64, 40, 3, 0), (1152, 1000, 270, 5, 0), (1152, 1000, 64, 2, 256), (256, 10, 33, 12, 0)])
def test_row_slabs_give_identical_proofs(gpu_ctx_factory, d, m, nb, nslabs, chunk_rows):
    """mfh_set_batch_slabs: the out-of-core form of mfh_prove_batch (what a 2^20-constraint CRS needs on one GPU) -- the CRS rows cut into
    slabs, each slab's image expanded once and streamed for every group of the call, results accumulated mod 2^(64K), all chains first.
    Forced here at small sizes: 3 / 5 / 2 / 12 slabs (the last with fewer BT+BV rows than slabs), two super-groups, row chunks inside a
    slab; bit-identical to the in-core call, accepted / rejected as the witnesses demand."""
    import sys

    import torch

    import c_lwe_snarks_amd as mf

    sys.path.insert(0, ROOT)
    import bench

    p = mf.Params(d=d, m=m)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = bench.build_instance(mf, ctx, torch, p, 7 * d + m)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(nb)
    bits, deltas, mags, signs = _statements(rng, p, nb, valid_bits=inst["bits"])
    want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()
    ctx.set_batch_slabs(nslabs)
    ctx.set_mm_chunk_rows(chunk_rows)
    ctx.set_timing(True)
    try:
        got = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
    finally:
        ctx.set_timing(False)
        ctx.set_batch_slabs(0)
        ctx.set_mm_chunk_rows(0)
    assert ctx.timing_drain("expandmm")[0] > 0 and ctx.timing_drain("evalmm")[0] == 0  # slabs expanded and streamed, nothing regenerated per group
    assert torch.equal(got, want)
    ok = ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], got, nb))
    assert [bool(x) for x in ok] == [b % 3 != 2 for b in range(nb)]
    ctx.close()


def test_more_than_four_super_groups(gpu_ctx_factory, oracle):
    """1300 statements in one call: 6 super-groups of 255 (the staged host inputs -- witness bits, deltas, smudging terms -- of all
    super-groups travel in one copy, one staging area per super-group; the chain alternates between two w | h | v areas): proofs at the
    super-group boundaries equal the single-proof prover's, statement 0 the oracle's, and the verifier accepts exactly the valid ones."""
    import torch

    import c_lwe_snarks_amd as mf

    p = mf.Params(d=256, m=200)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst = _instance(mf, ctx, oracle, p, 31)
    nb = 1300
    bits, deltas, mags, signs = _statements(inst["rng"], p, nb, inst["wit"])
    out = ctx.prove_batch(inst["d_crs"], inst["d_ssp"], bits, deltas, mags, signs).view(nb, -1)
    d_sk = ctx.to_device(inst["sk"])
    ok = ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], d_sk, out.reshape(-1), nb))
    assert [bool(x) for x in ok] == [b % 3 != 2 for b in range(nb)]
    for b in (0, 247, 248, 254, 255, 256, 509, 510, 764, 765, 1019, 1020, 1274, 1275, nb - 1):
        assert torch.equal(out[b], ctx.prove(inst["d_crs"], inst["d_ssp"], bits[b], deltas[b], mags[b], signs[b])), f"statement {b}"
    for b in (0, 1240):
        stape = b"".join(mags[b][80 * k: 80 * k + 80] + signs[b][k: k + 1] for k in range(5))
        ref = oracle.prover(p, inst["crs"], inst["ssp"], bits[b], deltas[b], stape, 80)
        assert np.array_equal(ctx.to_host(out[b], np.uint64).reshape(5, p.n + 1, p.L), np.stack(ref["proof"])), f"statement {b} differs from the oracle's prover"


@pytest.mark.parametrize("nslabs", [0, 3])
def test_too_wide_smudge_is_rejected_before_any_gpu_work(gpu_ctx_factory, oracle, nslabs):
    """the row-slab regime of mfh_prove_batch (and mfh_prove_batch_finish) checked maglen only in the smudging at the very end,
    after d_proofs had been overwritten; now every regime rejects it up front and leaves the output untouched"""
    import torch

    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    W = _instance(mf, ctx, oracle, p, 4242)
    nb, maglen = 40, 88  # 88 + 4 > 88 surviving bytes
    rng = np.random.default_rng(5)
    bits = [W["wit"]] * nb
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nb, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=5 * maglen, dtype=np.uint8).tobytes() for _ in range(nb)]
    signs = [bytes(5)] * nb
    out = torch.full((nb * 5 * p.ct_limbs * 8,), 0x5A, dtype=torch.uint8, device=ctx.device)
    ctx.set_batch_slabs(nslabs)
    try:
        with pytest.raises(mf.MfhError, match="too wide"):
            ctx.prove_batch(W["d_crs"], W["d_ssp"], bits, deltas, mags, signs, maglen=maglen, out=out)
        assert bool((out == 0x5A).all())
        with pytest.raises(mf.MfhError, match="too wide"):
            ctx.prove_batch_finish(W["d_crs"], deltas, mags, signs, out, maglen=maglen)
        assert bool((out == 0x5A).all())
    finally:
        ctx.set_batch_slabs(0)
