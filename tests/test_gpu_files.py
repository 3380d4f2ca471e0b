"""CRS / SSP / proof images on disk drive the GPU path unchanged.  setup() on the GPU -> crs.mfuoco ->
mapped image -> prover() on the GPU == the oracle's proof from the oracle's CRS; the proof survives its file; the
image the GPU setup wrote is byte-identical to the image of the oracle's CRS."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SEED = bytes((7 * i + 3) & 0xFF for i in range(40))


def test_setup_prove_verify_through_files(gpu_ctx_factory, oracle, tmp_path):
    import c_lwe_snarks_amd as mf
    from c_lwe_snarks_amd import files as mff

    p = mf.DEBUG
    ctx = gpu_ctx_factory(p)
    rng = np.random.default_rng(99)
    bits = rng.integers(0, 256, size=(p.m + 7) // 8, dtype=np.uint8).tobytes()
    tape = rng.integers(0, 256, size=p.m * 8 * p.d, dtype=np.uint8)
    ssp = oracle.ssp_from_tape(p, tape, bits)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, p.n, p.L, p.logq)
    etape = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)

    # SSP through its image
    ssp_path = str(tmp_path / "ssp.mfuoco")
    mff.ssp_write(ssp_path, p, ssp)
    ssp_m = mff.ssp_map(ssp_path, p)
    d_ssp = ctx.ssp_upload(np.ascontiguousarray(ssp_m).reshape(-1))
    ctx.ssp_prepare(d_ssp)

    # setup on the GPU, CRS to its image; the oracle's CRS gives the same image
    ctx.set_seed(SEED)
    d_crs = ctx.setup(d_ssp, alpha, beta, s, ctx.to_device(sk), ctx.to_device(etape))
    crs_path = str(tmp_path / "crs.mfuoco")
    mff.crs_write(crs_path, p, SEED, ctx.to_host(d_crs))
    ref_crs = oracle.setup(p, SEED, ssp, alpha, beta, s, sk, etape)
    ref_path = str(tmp_path / "crs_oracle.mfuoco")
    mff.crs_write(ref_path, p, SEED, np.concatenate([ref_crs["s"], ref_crs["as_"], ref_crs["t"], ref_crs["v"][: (p.m - 1) * p.ctb]]))
    assert open(crs_path, "rb").read() == open(ref_path, "rb").read()

    # a fresh context knows nothing but the two images
    ctx2 = gpu_ctx_factory(p)
    seed, rows = mff.crs_map(crs_path, p)
    assert seed == SEED
    ctx2.set_seed(seed)
    d_ssp2 = ctx2.ssp_upload(np.ascontiguousarray(mff.ssp_map(ssp_path, p)).reshape(-1))
    ctx2.ssp_prepare(d_ssp2)
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=400, dtype=np.uint8).tobytes()
    signs = bytes([0, 1, 1, 0, 0])
    proof = ctx2.to_host(ctx2.prove(ctx2.to_device(np.ascontiguousarray(rows)), d_ssp2, bits, delta, mags, signs), np.uint64)
    stape = b"".join(mags[80 * k: 80 * k + 80] + signs[k: k + 1] for k in range(5))
    ref = oracle.prover(p, ref_crs, ssp, bits, delta, stape, 80)
    assert np.array_equal(proof.reshape(5, p.n + 1, p.L), np.stack(ref["proof"]))

    proof_path = str(tmp_path / "proof.mfuoco")
    mff.proof_write(proof_path, p, proof)
    back = mff.proof_read(proof_path, p)
    assert np.array_equal(back, proof.reshape(5, p.n + 1, p.L))
    ok = ctx2.to_host(ctx2.verify(d_ssp2, alpha, beta, s, ctx2.to_device(sk), ctx2.to_device(back), 1))
    assert bool(ok[0])
    assert oracle.verifier(p, ssp, alpha, beta, s, sk, back)
