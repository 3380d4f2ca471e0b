"""CPU: the algebra behind the polynomial step's exact-division path (c-lwe-snarks_amd/csrc/poly.hip, round 6), in plain Python integers and checked against the oracle's
Euclidean division -- (i) the norm recursion a^-1 = a(-x) [a(x) a(-x)]^-1 inverts a unit of F_p[x] / (x^N - 1), the bracket having even powers only; (ii) when
t | v^2 - 1 and deg t = d - 1 <= N - 1, the quotient nmod_poly_div returns (src/snark.c:166-169) is (v^2 - 1 mod x^N - 1) t^-1 mod x^N - 1; (iii) when t does not divide,
that product is NOT the quotient and the identity h t = v^2 - 1 fails at a random point -- which is what the device check tests before the result is used."""
import numpy as np
import pytest

import c_lwe_snarks_amd as mf
import oracle_lib as ol

P = ol.P


def cyc_mul(a, b, n):
    c = [0] * n
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                c[(i + j) % n] = (c[(i + j) % n] + x * y) % P
    return c


def conj(a):
    return [(-x) % P if i & 1 else x for i, x in enumerate(a)]


def cyc_inv(a):
    """a^-1 in F_p[x] / (x^n - 1), n = len(a) a power of two; None when a is not a unit"""
    n = len(a)
    if n == 1:
        return [pow(a[0], P - 2, P)] if a[0] % P else None
    ac = conj(a)
    norm = cyc_mul(a, ac, n)
    assert not any(norm[1::2])  # a(x) a(-x) is even: an element of the ring of half the length in x^2
    half = cyc_inv(norm[0::2])
    if half is None:
        return None
    spread = [0] * n
    spread[0::2] = half
    return cyc_mul(ac, spread, n)


def evaluate(f, x):
    r = 0
    for c in reversed(f):
        r = (r * x + c) % P
    return r


@pytest.mark.parametrize("d,m", [(64, 12), (48, 9), (33, 7)])
def test_exact_quotient_is_the_cyclic_product(oracle, d, m):
    p = mf.Params(d=d, m=m)
    rng = np.random.default_rng(d)
    bits = rng.bytes((m + 7) // 8)
    tape = rng.integers(0, 256, size=m * 8 * d, dtype=np.uint8)
    ssp = oracle.ssp_from_tape(p, tape, bits).reshape(m + 3, d)
    t = [int(x) for x in ssp[0]]
    assert t[-1] != 0
    v = ssp[1].copy()
    for i in range(1, m):
        if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
            v = (v + ssp[i + 1]) % np.uint64(P)
    delta = int(rng.integers(1, P, dtype=np.uint64))
    v = np.array([(int(a) + delta * b) % P for a, b in zip(v, t)], dtype=np.uint64)  # v + delta t still satisfies t | v^2 - 1
    assert oracle.poly_divides(v, ssp[0])
    N = 1 << (d - 1).bit_length()
    tinv = cyc_inv(t + [0] * (N - d))
    assert tinv is not None and cyc_mul(t + [0] * (N - d), tinv, N) == [1] + [0] * (N - 1)
    vv = [int(x) for x in v] + [0] * (N - d)
    sq = cyc_mul(vv, vv, N)
    sq[0] = (sq[0] - 1) % P
    h = cyc_mul(sq, tinv, N)
    assert h[:d] == [int(x) for x in oracle.poly_h(v, ssp[0])] and not any(h[d:])
    for r in (2, 12345, P - 2):
        assert (evaluate(h, r) * evaluate(t, r) - evaluate(vv, r) ** 2 + 1) % P == 0
    # a witness that does not satisfy the SSP: the cyclic product is not the quotient, and the check sees it
    v2 = v.copy()
    v2[d // 3] = (v2[d // 3] + np.uint64(1)) % np.uint64(P)
    assert not oracle.poly_divides(v2, ssp[0])
    vv2 = [int(x) for x in v2] + [0] * (N - d)
    sq2 = cyc_mul(vv2, vv2, N)
    sq2[0] = (sq2[0] - 1) % P
    h2 = cyc_mul(sq2, tinv, N)
    assert h2[:d] != [int(x) for x in oracle.poly_h(v2, ssp[0])]
    assert any((evaluate(h2[:d], r) * evaluate(t, r) - evaluate(vv2, r) ** 2 + 1) % P for r in (2, 12345, P - 2))


def test_a_factor_of_x_n_minus_1_has_no_inverse():
    # t = x - 1 divides x^N - 1: the recursion ends in the scalar t(1) = 0 and the path is not offered
    assert cyc_inv([P - 1, 1, 0, 0, 0, 0, 0, 0]) is None
    # x + 1 as well (t(-1) = 0 one level down)
    assert cyc_inv([1, 1, 0, 0]) is None
