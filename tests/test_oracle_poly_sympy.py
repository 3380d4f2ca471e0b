"""CPU: the oracle's restatement of the FLINT calls on the path -- nmod_poly_pow / sub / div (h = (v^2 - 1) / t, src/snark.c:166-169), nmod_poly_rem (the
divisibility src/test_ssp.c:37-79 asserts), nmod_poly_evaluate_nmod (src/snark.c:93-106,197-215) -- against an INDEPENDENT implementation of F_p[x]: sympy's
Poly(..., modulus = p).  FLINT is not in this image, so the reference's lwe.c / ssp.c / snark.c cannot be built and the oracle's L3 / L4 arithmetic is
"parity unpinned" in the task's sense; what this adds is that every result the reference takes from FLINT is a CANONICAL element of F_p[x], i.e.
determined by the mathematics and not by FLINT's algorithms, and a computer-algebra system that shares no code with the oracle computes the same canonical values.
(Not reference output, and not claimed as such.)"""
import numpy as np
import pytest

import oracle_lib as ol

sympy = pytest.importorskip("sympy")
P = ol.P


def _poly(coeffs, x):
    # sympy wants the highest coefficient first; modulus = p gives F_p[x] (symmetric representatives: reduced below)
    return sympy.Poly([int(c) for c in coeffs[::-1]], x, modulus=P)


def _canon(poly, n):
    c = [int(a) % P for a in poly.all_coeffs()[::-1]]
    return np.array((c + [0] * n)[:n], dtype=np.uint64)


@pytest.mark.parametrize("case", ["dense", "low_degree_t", "tiny_v", "monic_t", "valid_ssp"])
def test_quotient_remainder_and_evaluation_against_sympy(oracle, case):
    import c_lwe_snarks_amd as mf

    d = 96
    rng = np.random.default_rng(["dense", "low_degree_t", "tiny_v", "monic_t", "valid_ssp"].index(case) + 100)
    v = rng.integers(0, P, size=d, dtype=np.uint64)
    t = rng.integers(0, P, size=d, dtype=np.uint64)
    v[0], t[-1] = P - 1, P - 2  # extreme coefficients; deg t = d - 1
    if case == "low_degree_t":
        t[d - 7:] = 0          # the quotient has more than d coefficients: the first d are kept (the oracle and the GPU keep D of them)
        t[d - 8] = 3
    if case == "tiny_v":
        v[5:] = 0              # deg(v^2 - 1) < deg t: quotient 0
    if case == "monic_t":
        t[-1] = 1
    if case == "valid_ssp":    # t = v_0 + sum_{bits} v_i - 1 and v = that sum: t divides v^2 - 1 (src/ssp.c:37-77)
        pp = mf.Params(d=d, m=12)
        bits = rng.bytes(2)
        ssp = oracle.ssp_from_tape(pp, rng.integers(0, 256, size=pp.m * 8 * pp.d, dtype=np.uint8), bits).reshape(pp.m + 3, pp.d)
        t = ssp[0].copy()
        v = ssp[1].copy()
        for i in range(1, pp.m):
            if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
                v = (v + ssp[i + 1]) % np.uint64(P)
    x = sympy.symbols("x")
    fv, ft = _poly(v, x), _poly(t, x)
    num = fv * fv - sympy.Poly(1, x, modulus=P)
    q, r = sympy.div(num, ft)
    # h = the first d coefficients of the Euclidean quotient (nmod_poly_div; nmod_poly_get_coeff_ui beyond the length is 0)
    assert np.array_equal(oracle.poly_h(v, t), _canon(q, d)), case
    # divisibility = zero remainder (nmod_poly_rem)
    assert oracle.poly_divides(v, t) == bool(r.is_zero), case
    if case == "valid_ssp":
        assert r.is_zero
    # Horner evaluation (nmod_poly_evaluate_nmod)
    for pt in (0, 1, 2, 123456789, P - 1):
        assert oracle.poly_eval(v, pt) == int(fv.eval(pt)) % P
        assert oracle.poly_eval(t, pt) == int(ft.eval(pt)) % P
