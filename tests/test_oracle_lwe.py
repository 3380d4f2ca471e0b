"""CPU: the oracle's LWE / SSP / SNARK layers.  lwe.c, ssp.c and snark.c of the reference cannot be built here (FLINT is
absent), so these layers are pinned by (i) a second restatement that makes the same libgmp calls as the reference
(oracle/gmp_check.c; vectors in tests/golden/gmp_lwe.json and a live randomized cross-check), and (ii) the properties the
reference's own tests assert (src/test_lwe.c, src/test_ssp.c, src/test_snark.c), at the reference's debug parameters."""
import ctypes
import json
import os

import numpy as np
import pytest

import c_lwe_snarks_amd as mf
import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "gmp_lwe.json")))
GMP_SO = os.path.join(os.path.dirname(HERE), "oracle", "libmf_gmpcheck.so")
P = mf.DEBUG
L = P.L


def _h(x):
    return int(x, 16)


def test_modq_effective_modulus_is_2_704(oracle):
    # src/lwe.h:107-118 keeps 11 limbs, not 736 bits
    for c in GOLD["modq"]:
        x = _h(c["in"])
        got = ol.limbs_to_int(oracle.modq(P, ol.int_to_limbs(x, L)))
        assert got == _h(c["out"]) == x % (1 << 704)
    assert ol.limbs_to_int(oracle.modq(P, ol.int_to_limbs((1 << 720) + 5, L))) == 5


def _one(v):
    """embed one value as coordinate 0 of an otherwise zero ciphertext"""
    ct = np.zeros((P.n + 1, L), dtype=np.uint64)
    ct[0] = ol.int_to_limbs(v, L)
    return ct


def test_ct_ops_golden(oracle):
    for c in GOLD["addmul_ui"]:
        got = oracle.ct_addmul_ui(P, _one(_h(c["rop"])), _one(_h(c["a"])), c["b"])[0]
        assert ol.limbs_to_int(got) == _h(c["out"])
    for c in GOLD["mul_ui"]:
        assert ol.limbs_to_int(oracle.ct_mul_ui(P, _one(_h(c["a"])), c["b"])[0]) == _h(c["out"])
    for c in GOLD["add"]:
        assert ol.limbs_to_int(oracle.ct_add(P, _one(_h(c["a"])), _one(_h(c["b"])))[0]) == _h(c["out"])


def test_encrypt_decrypt_smudge_golden(oracle):
    for i, c in enumerate(GOLD["encrypt_b"]):
        n = c["n"]
        pp = mf.Params(n=n, d=P.d, m=P.m)
        a = np.array([ol.int_to_limbs(_h(x), L) for x in c["a"]])
        sk = np.array([ol.int_to_limbs(_h(x), L) for x in c["sk"]])
        b = np.zeros(L, dtype=np.uint64)
        b[:] = 0
        # b = e*p + <sk,a> + m through the oracle's primitives
        cp = oracle.cp(pp)
        e = ol.int_to_limbs(_h(c["e"]), L)
        tmp = np.zeros(L, dtype=np.uint64)
        exp_b = (_h(c["e"]) * ol.P + sum(_h(x) * _h(y) for x, y in zip(c["a"], c["sk"])) + c["m"]) % (1 << 704)
        assert exp_b == _h(c["b"])  # the GMP result is the plain integer formula mod 2^704
        oracle.lib.mfo_add_dotp(ctypes.byref(cp), tmp.ctypes.data_as(ctypes.c_void_p), sk.ctypes.data_as(ctypes.c_void_p),
                                a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n))
        assert (ol.limbs_to_int(tmp) + _h(c["e"]) * ol.P + c["m"]) % (1 << 704) == _h(c["b"])
        ct = np.concatenate([a, ol.int_to_limbs(_h(c["b"]), L)[None, :]])
        assert oracle.decrypt(pp, sk, ct) == GOLD["decrypt"][i]["m"] == c["m"]
    for c in GOLD["smudge"]:
        ct = np.zeros((P.n + 1, L), dtype=np.uint64)
        ct[P.n] = ol.int_to_limbs(_h(c["b"]), L)
        got, neg = oracle.ct_smudge(P, ct, bytes.fromhex(c["mag"]), c["sign"])
        assert ol.limbs_to_int(got[P.n]) == _h(c["out"]) and int(neg) == c["negative"]


@pytest.mark.skipif(not os.path.exists(GMP_SO), reason="libmf_gmpcheck.so not built (gmp.h missing)")
def test_randomized_against_libgmp(oracle):
    g = ctypes.CDLL(GMP_SO)
    g.gx_decrypt.restype = ctypes.c_uint64
    rng = np.random.default_rng(1)

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    n = 40
    pp = mf.Params(n=n, d=P.d, m=P.m)
    for _ in range(25):
        a = ol.rand_values(rng, n, L, 736)
        sk = ol.rand_values(rng, n, L, 736)
        e = ol.rand_values(rng, 1, L, 559)[0]
        m = int(rng.integers(0, ol.P, dtype=np.uint64))
        b = np.zeros(L, dtype=np.uint64)
        g.gx_encrypt_b(ptr(b), ptr(a), ptr(sk), ctypes.c_size_t(n), ctypes.c_uint64(m), ptr(e))
        tmp = np.zeros(L, dtype=np.uint64)
        cp = oracle.cp(pp)
        oracle.lib.mfo_add_dotp(ctypes.byref(cp), ptr(tmp), ptr(sk), ptr(a), ctypes.c_size_t(n))
        assert (ol.limbs_to_int(tmp) + ol.limbs_to_int(e) * ol.P + m) % (1 << 704) == ol.limbs_to_int(b)
        ct = np.concatenate([a, b[None, :]])
        assert oracle.decrypt(pp, sk, ct) == int(g.gx_decrypt(ptr(a), ptr(b), ptr(sk), ctypes.c_size_t(n))) == m
        # unreduced b (736 bits), as after a raw ct_import
        braw = ol.rand_values(rng, 1, L, 736)
        ct2 = np.concatenate([a, braw])
        assert oracle.decrypt(pp, sk, ct2) == int(g.gx_decrypt(ptr(a), ptr(braw[0]), ptr(sk), ctypes.c_size_t(n)))
        r = ol.rand_values(rng, 1, L, 704)[0]
        x = int(rng.integers(0, ol.P, dtype=np.uint64))
        r2 = r.copy()
        g.gx_addmul_ui(ptr(r2), ptr(a[0]), ctypes.c_uint64(x))
        assert np.array_equal(oracle.ct_addmul_ui(P, _one(ol.limbs_to_int(r)), _one(ol.limbs_to_int(a[0])), x)[0], r2)


# ---- properties asserted by the reference's tests, on the oracle --------------------------------------------------
def test_lwe_properties(oracle):
    rng = np.random.default_rng(2)
    seed = rng.bytes(40)
    sk = ol.rand_values(rng, P.n, L, P.logq)
    r, r2 = oracle.rng(seed), oracle.rng(seed)
    msgs, c8 = [], b""
    for i in range(12):
        m = int(rng.integers(0, ol.P, dtype=np.uint64))
        e = ol.rand_values(rng, 1, L, 559)[0]
        ct = oracle.encrypt(P, r, sk, m, e)
        assert oracle.decrypt(P, sk, ct) == m  # src/test_lwe.c:74-95
        buf = oracle.ct_export(P, ct)
        assert np.array_equal(oracle.ct_import(P, r2, buf), ct)  # :36-70 export -> import with a twin rng
        sm, neg = oracle.ct_smudge(P, ct, rng.bytes(80), int(rng.integers(0, 2)))
        assert not neg and oracle.decrypt(P, sk, sm) == m  # :183-205
        msgs.append(m)
        c8 += buf
    ev = oracle.eval_poly(P, seed, 0, c8, np.ones(12, dtype=np.uint64))  # :105-181
    assert oracle.decrypt(P, sk, ev) == sum(msgs) % ol.P


def test_ssp_properties(oracle):
    rng = np.random.default_rng(3)
    pp = mf.Params(d=64, m=20)
    bits = rng.bytes(3)
    tape = rng.integers(0, 256, size=pp.m * 8 * pp.d, dtype=np.uint8)
    ssp = oracle.ssp_from_tape(pp, tape, bits).reshape(pp.m + 3, pp.d)
    assert (ssp < ol.P).all()
    v = ssp[1].copy()
    for i in range(1, pp.m):
        if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
            v = (v + ssp[i + 1]) % np.uint64(ol.P)
    assert oracle.poly_divides(v, ssp[0])  # src/test_ssp.c:37-79
    v[3] = (v[3] + np.uint64(1)) % np.uint64(ol.P)
    assert not oracle.poly_divides(v, ssp[0])
    # quotient * t + 0 == v^2 - 1 at a random point
    v[3] = (v[3] + np.uint64(ol.P - 1)) % np.uint64(ol.P)
    h = oracle.poly_h(v, ssp[0])
    x = 123456789
    assert (oracle.poly_eval(h, x) * oracle.poly_eval(ssp[0], x) - oracle.poly_eval(v, x) ** 2 + 1) % ol.P == 0


def test_snark_properties_debug_params(oracle):
    rng = np.random.default_rng(4)
    seed = rng.bytes(40)
    bits = rng.bytes((P.m + 7) // 8)
    tape = rng.integers(0, 256, size=P.m * 8 * P.d, dtype=np.uint8)
    ssp = oracle.ssp_from_tape(P, tape, bits)
    alpha, beta, s = (int(x) for x in rng.integers(1, ol.P, size=3, dtype=np.uint64))
    sk = ol.rand_values(rng, P.n, L, P.logq)
    etape = ol.rand_values(rng, 2 * P.d + P.m, L, 559)
    crs = oracle.setup(P, seed, ssp, alpha, beta, s, sk, etape)
    # src/test_snark.c:35-70
    r = oracle.rng(seed, P.ctr_s)
    assert oracle.decrypt(P, sk, oracle.ct_import(P, r, crs["s"][: P.ctb].tobytes())) == 1
    r = oracle.rng(seed, P.ctr_as)
    assert oracle.decrypt(P, sk, oracle.ct_import(P, r, crs["as_"][: P.ctb].tobytes())) == alpha
    for i in (1, P.d - 1):
        r = oracle.rng(seed, P.ctr_s + i * P.ctr_ct)
        si = oracle.decrypt(P, sk, oracle.ct_import(P, r, crs["s"][i * P.ctb: (i + 1) * P.ctb].tobytes()))
        r = oracle.rng(seed, P.ctr_as + i * P.ctr_ct)
        asi = oracle.decrypt(P, sk, oracle.ct_import(P, r, crs["as_"][i * P.ctb: (i + 1) * P.ctb].tobytes()))
        assert si * alpha % ol.P == asi
    delta = int(rng.integers(0, ol.P, dtype=np.uint64))
    tape5 = b"".join(rng.bytes(80) + bytes([int(rng.integers(0, 2))]) for _ in range(5))
    out = oracle.prover(P, crs, ssp, bits, delta, tape5, 80)
    h_s = oracle.decrypt(P, sk, out["proof"][0])
    assert 0 < h_s < ol.P and h_s * alpha % ol.P == oracle.decrypt(P, sk, out["proof"][1])  # :81-89
    assert oracle.verifier(P, ssp, alpha, beta, s, sk, out["proof"])  # :105-107
    assert oracle.verifier(P, ssp, alpha, beta, s, sk, out["pre"])
    bad = out["proof"].copy()
    bad[3, P.n, 0] ^= np.uint64(1 << 40)
    assert not oracle.verifier(P, ssp, alpha, beta, s, sk, bad)
