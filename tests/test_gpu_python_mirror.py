"""GPU: setup(), prover() -- one statement and a batch -- and the device verifier against the Python-integer restatement of the reference's source
(tests/test_oracle_python_mirror.py: Mirror; mpz_t = int, modq = mod 2^704, polynomials as lists), WITHOUT the C oracle's LWE / SSP / SNARK layers in between: the only
thing the expected values take from oracle/ is the public AES-CTR keystream, the layer pinned to the reference's real aes.c + entropy.c.  A second route from the
reference's text to the bytes the HIP path must produce, at the reference's debug parameters (src/lwe.h:18-21)."""
import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_python_mirror import Mirror

pytestmark = pytest.mark.gpu
SEED = bytes((29 * i + 3) & 0xFF for i in range(40))


def _limbs(vals, L):
    return np.array([ol.int_to_limbs(v, L) for v in vals], dtype=np.uint64)


def test_setup_prover_batch_and_verifier_against_the_python_mirror(gpu_ctx_factory, oracle):
    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    c = gpu_ctx_factory(p)
    c.set_seed(SEED)
    mi = Mirror(oracle, p)
    rng = np.random.default_rng(31337)
    bits = rng.bytes((p.m + 7) // 8)
    tape = rng.integers(0, 1 << 63, size=p.m * p.d, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=p.m * p.d, dtype=np.uint64)
    t, v = mi.random_ssp(tape, bits)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk_l = ol.rand_values(rng, p.n, p.L, p.logq)
    err_l = ol.rand_values(rng, 2 * p.d + p.m, p.L, 559)
    sk, errs = [ol.limbs_to_int(r) for r in sk_l], [ol.limbs_to_int(r) for r in err_l]

    # the SSP in the reference's layout (src/ssp.h:6-9): slot 0 = t, slot i + 1 = v_i, 8-byte coefficients
    ssp = np.zeros((p.m + 3, p.d), dtype=np.uint64)
    ssp[0] = t
    for i in range(p.m):
        ssp[i + 1] = v[i]
    d_ssp = c.ssp_upload(ssp.reshape(-1))
    c.ssp_prepare(d_ssp)

    # setup(): the compressed CRS, byte for byte (src/snark.c:57-115)
    crs = mi.setup(SEED, t, v, alpha, beta, s, sk, errs)
    d_crs = c.setup(d_ssp, alpha, beta, s, c.to_device(sk_l), c.to_device(err_l))
    assert c.to_host(d_crs).tobytes() == crs["s"] + crs["as_"] + crs["t"] + crs["v"]

    # prover(): one statement with the satisfying witness, one with a witness that does not satisfy the SSP (src/snark.c:117-190 proves either)
    other = bytearray(bits)
    other[1] ^= 0x24
    stmts = []
    for k, b in enumerate([bits, bytes(other), bits, bits, bytes(other), bits]):
        delta = int(rng.integers(0, ol.P, dtype=np.uint64))
        smudges = [(rng.bytes(80), int(rng.integers(0, 2))) for _ in range(5)]
        stmts.append((b, delta, smudges))
    exp = []
    for b, delta, smudges in stmts:
        out = mi.prover(SEED, crs, t, v, b, delta, smudges)
        exp.append(np.stack([_limbs(ct, p.L) for ct in out["proof"]]))
    for k in (0, 1):
        b, delta, smudges = stmts[k]
        got = c.to_host(c.prove(d_crs, d_ssp, b, delta, b"".join(m for m, _ in smudges), bytes(sg for _, sg in smudges)), np.uint64).reshape(5, p.n + 1, p.L)
        assert np.array_equal(got, exp[k]), k
    # ... and the six as ONE batch (matrix-core path; polynomial step: exact division with two statements falling back to Euclidean division)
    c.set_poly_exact(2)
    got = c.to_host(c.prove_batch(d_crs, d_ssp, [x[0] for x in stmts], [x[1] for x in stmts], [b"".join(m for m, _ in x[2]) for x in stmts],
                                  [bytes(sg for _, sg in x[2]) for x in stmts]), np.uint64).reshape(6, 5, p.n + 1, p.L)
    assert np.array_equal(got, np.stack(exp))
    assert c.poly_exact_fallbacks() == 2
    # verifier(): the mirror's verdicts on the GPU's proofs, and the device verifier's
    ok = c.to_host(c.verify(d_ssp, alpha, beta, s, c.to_device(sk_l), c.to_device(got.reshape(-1)), 6))
    want = [True, False, True, True, False, True]
    assert [bool(int(x)) for x in ok] == want
    for k in (0, 1):
        proof = [[ol.limbs_to_int(r) for r in ct] for ct in got[k]]
        assert mi.verifier(t, v, alpha, beta, s, sk, proof) == want[k]
