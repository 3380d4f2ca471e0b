"""CPU: the oracle's LWE / SSP / SNARK layers (oracle/mf_oracle.c, L2 - L4) against a SECOND restatement that shares no code with it: the reference's
random_ssp / setup / prover / verifier (src/ssp.c:37-77, src/snark.c:57-250) and the lwe.c primitives under them (src/lwe.c:20-28, 62-160) written down again in plain
Python integers and lists -- mpz_t = int, modq = "mod 2^704" (src/lwe.h:107-118), nmod_poly_t = list of ints mod p with schoolbook multiplication and long division.
The only thing taken from the oracle is the public keystream (L0 / L1: that layer is pinned to the reference's real aes.c + entropy.c, tests/test_oracle_stream.py).
lwe.c / ssp.c / snark.c cannot be built here (FLINT is absent), so this is not reference output and not claimed as such: it is a second reading of the
same source, in another language and another number representation (arbitrary-precision ints against u128 limb loops), that must agree with the first one bit for bit
on the CRS, on w and h, on the proof before and after smudging, and on the verifier's verdict -- a transcription slip in either one shows up as a difference."""
import numpy as np
import pytest

import c_lwe_snarks_amd as mf
import oracle_lib as ol

PP = ol.P  # GAMMA_P = 2^32 - 5


class Mirror:
    def __init__(self, oracle, p):
        self.o, self.p = oracle, p
        self.q = 1 << (64 * (p.logq // 64))  # modq keeps logq / 64 limbs (src/lwe.h:107-118: SIZ(a) = pos)

    # ---- src/entropy.c:11-26,47-62 over the public stream ------------------------------------------------------------------------
    def sample_a(self, seed, off):
        """mpz2_urandommv(c, rng, GAMMA_LOGQ, GAMMA_N): n values of logq / 8 stream bytes each, little endian; returns (a, offset behind them)"""
        p = self.p
        buf = self.o.keystream(seed, off, p.n * p.ctb)
        return [int.from_bytes(buf[j * p.ctb:(j + 1) * p.ctb], "little") for j in range(p.n)], off + p.n * p.ctb

    def modq(self, x):
        assert x >= 0  # (src/lwe.h:109)
        return x % self.q

    # ---- src/lwe.c -------------------------------------------------------------------------------------------------------------------
    def encrypt(self, seed, off, sk, m, e):
        """regev_encrypt2 (src/lwe.c:78-98): c[N] = e p; a sampled; c[N] = modq(c[N] + <sk, a>); c[N] = modq(c[N] + m)"""
        assert m < PP
        b = e * PP
        a, off = self.sample_a(seed, off)
        b = self.modq(b + sum(x * y for x, y in zip(sk, a)))
        return a + [self.modq(b + m)], off

    def decrypt(self, sk, ct):
        """regev_decrypt (src/lwe.c:105-111): (b - modq(<a, sk>)) mod p, b as it stands"""
        n = self.p.n
        return (ct[n] - self.modq(sum(x * y for x, y in zip(ct[:n], sk)))) % PP

    def ct_export(self, ct):
        return ct[self.p.n].to_bytes(self.p.ctb, "little")  # (src/lwe.c:115-119)

    def ct_import(self, seed, off, buf):
        a, off = self.sample_a(seed, off)  # (src/lwe.c:122-126: b imported whole, not reduced)
        return a + [int.from_bytes(bytes(buf[:self.p.ctb]), "little")], off

    def ct_mul_ui(self, a, x):
        return [self.modq(v * x) for v in a]  # (src/lwe.c:131-139)

    def ct_addmul_ui(self, rop, a, x):
        return [self.modq(r + v * x) for r, v in zip(rop, a)]  # (src/lwe.c:141-149)

    def ct_add(self, a, b):
        return [self.modq(x + y) for x, y in zip(a, b)]  # (src/lwe.c:151-157)

    def ct_smudge(self, ct, mag, sign):
        """ct_smudge (src/lwe.c:62-75): +- u p on the b coordinate, u = the tape's 640 bits"""
        u = int.from_bytes(mag, "little")
        u = -u if sign & 1 else u
        out = list(ct)
        out[self.p.n] = self.modq(ct[self.p.n] + u * PP)  # (asserts b + u p >= 0 as the reference's debug build does)
        return out

    def eval_poly(self, seed, off, c8, coeff):
        """eval_poly (src/lwe.c:177-187) from a zero accumulator: sum_i coeff[i] * import(c8[i])"""
        p = self.p
        rop = [0] * (p.n + 1)
        for i in range(p.d):
            ct, off = self.ct_import(seed, off, c8[i * p.ctb:(i + 1) * p.ctb])
            rop = self.ct_addmul_ui(rop, ct, coeff[i] if i < len(coeff) else 0)
        return rop

    # ---- F_p[x] ----------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def poly_eval(f, x):
        r = 0
        for c in reversed(f):
            r = (r * x + c) % PP
        return r

    @staticmethod
    def poly_div(num, den):
        """Euclidean quotient (nmod_poly_div), coefficients low to high"""
        den = list(den)
        while den and not den[-1]:
            den.pop()
        num = list(num)
        inv = pow(den[-1], PP - 2, PP)
        q = [0] * max(len(num) - len(den) + 1, 1)
        for i in range(len(num) - 1, len(den) - 2, -1):
            c = num[i] * inv % PP
            q[i - len(den) + 1] = c
            if c:
                for j, dj in enumerate(den):
                    num[i - len(den) + 1 + j] = (num[i - len(den) + 1 + j] - c * dj) % PP
        return q

    # ---- src/ssp.c:37-77 -------------------------------------------------------------------------------------------------------------
    def random_ssp(self, tape, bits):
        """tape: m * d little-endian uint64 draws; returns [t, v_0 .. v_{m-1}] as lists mod p"""
        p = self.p
        v = [[int(x) % PP for x in tape[i * p.d:(i + 1) * p.d]] for i in range(p.m)]
        t = list(v[0])
        for i in range(1, p.m):
            if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
                t = [(x + y) % PP for x, y in zip(t, v[i])]
        t[0] = (t[0] - 1) % PP
        return t, v

    # ---- src/snark.c -----------------------------------------------------------------------------------------------------------------
    def setup(self, seed, t, v, alpha, beta, s, sk, errs):
        p = self.p
        off, ei = 0, 0
        crs = dict(s=b"", as_=b"", v=b"", t=b"")
        cur = 1
        for _ in range(p.d):  # :75-82
            ct, off = self.encrypt(seed, off, sk, cur, errs[ei])
            ei += 1
            crs["s"] += self.ct_export(ct)
            cur = cur * s % PP
        cur = alpha
        for _ in range(p.d):  # :84-91
            ct, off = self.encrypt(seed, off, sk, cur, errs[ei])
            ei += 1
            crs["as_"] += self.ct_export(ct)
            cur = cur * s % PP
        ct, off = self.encrypt(seed, off, sk, self.poly_eval(t, s) * beta % PP, errs[ei])  # :97-101
        ei += 1
        crs["t"] = self.ct_export(ct)
        for i in range(1, p.m):  # :104-110
            ct, off = self.encrypt(seed, off, sk, self.poly_eval(v[i], s) * beta % PP, errs[ei])
            ei += 1
            crs["v"] += self.ct_export(ct)
        return crs

    def prover(self, seed, crs, t, v, bits, delta, smudges):
        p = self.p
        w = [c * delta % PP for c in t]  # :141
        off = p.ctr_bt  # :143-145
        b_w, off = self.ct_import(seed, off, crs["t"])
        b_w = self.ct_mul_ui(b_w, delta)
        for i in range(1, p.m):  # :147-155 (the stream advances over every row, selected or not)
            ct, off = self.ct_import(seed, off, crs["v"][(i - 1) * p.ctb:i * p.ctb])
            if (bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
                w = [(x + y) % PP for x, y in zip(w, v[i])]
                b_w = self.ct_add(b_w, ct)
        w_out = list(w)
        v_w = self.eval_poly(seed, p.ctr_s, crs["s"], w)  # :157-158
        w = [(x + y) % PP for x, y in zip(w, v[0])]  # :161-164
        hat_v = self.eval_poly(seed, p.ctr_as, crs["as_"], w)
        sq = [0] * (2 * p.d - 1)  # :166-169
        for i, x in enumerate(w):
            if x:
                for j, y in enumerate(w):
                    sq[i + j] = (sq[i + j] + x * y) % PP
        sq[0] = (sq[0] - 1) % PP
        h = self.poly_div(sq, t)
        pi_h = self.eval_poly(seed, p.ctr_s, crs["s"], h)  # :171-174
        hat_h = self.eval_poly(seed, p.ctr_as, crs["as_"], h)
        pre = [pi_h, hat_h, hat_v, v_w, b_w]
        (m0, s0), (m1, s1), (m2, s2), (m3, s3), (m4, s4) = smudges  # :185-189: h, hat_h, hat_v, v_w, v_w again; b_w never
        post = [self.ct_smudge(pi_h, m0, s0), self.ct_smudge(hat_h, m1, s1), self.ct_smudge(hat_v, m2, s2),
                self.ct_smudge(self.ct_smudge(v_w, m3, s3), m4, s4), b_w]
        return dict(pre=pre, proof=post, w=w_out, h=(h + [0] * p.d)[:p.d])

    def verifier(self, t, v, alpha, beta, s, sk, proof):
        h_s, hath_s, hatv_s, w_s, b_s = (self.decrypt(sk, c) for c in proof)  # :206-210
        t_s = self.poly_eval(t, s)
        v_s = (self.poly_eval(v[0], s) + w_s) % PP
        if h_s * alpha % PP != hath_s or v_s * alpha % PP != hatv_s:  # eq-pke :220-225
            return False
        if (v_s * v_s - 1 - h_s * t_s) % PP:  # eq-div :227-231
            return False
        if w_s * beta % PP != b_s:  # eq-lin :233-235
            return False
        # test-error :238-241: test = ceil(-modq(<b_w, sk>) / p) <= 0, so SIZ(test) <= 0 < 80 -- never rejects
        test = -(self.modq(sum(x * y for x, y in zip(proof[4][:self.p.n], sk))) // PP)
        return test <= 0


def _ints(arr):
    return [ol.limbs_to_int(r) for r in arr]


def _cts(arr):
    return [_ints(ct) for ct in arr]


@pytest.mark.parametrize("n,d,m,case", [(1470, 256, 64, 0), (96, 128, 40, 1), (33, 64, 9, 2)])
def test_python_integer_mirror_agrees_with_the_oracle(oracle, n, d, m, case):
    """(1470, 256, 64) are the reference's debug parameters (src/lwe.h:18-21); the smaller ones add odd shapes: m - 1 not a multiple of 8, an n whose a-vectors end off
    an AES block boundary at every row"""
    p = mf.Params(n=n, d=d, m=m)
    mi = Mirror(oracle, p)
    rng = np.random.default_rng(900 + case)
    seed = rng.bytes(40)
    bits = rng.bytes((m + 7) // 8)
    tape = rng.integers(0, 1 << 63, size=m * d, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=m * d, dtype=np.uint64)  # full 64-bit draws
    alpha, beta, s = (int(x) for x in rng.integers(1, PP, size=3, dtype=np.uint64))
    sk_l = ol.rand_values(rng, n, p.L, p.logq)
    err_l = ol.rand_values(rng, 2 * d + m, p.L, 559)  # errdist_uniform: GAMMA_LOG_SIGMA + 3 bits (src/lwe.c:57-60)
    delta = int(rng.integers(0, PP, dtype=np.uint64))
    smudges = [(rng.bytes(80), int(rng.integers(0, 2))) for _ in range(5)]
    if case == 1:
        smudges[3], smudges[4] = (smudges[3][0], 0), (smudges[4][0], 1)  # v_w smudged up, then down

    # SSP: both from the same draws
    ssp_o = oracle.ssp_from_tape(p, tape.view(np.uint8), bits)
    t, v = mi.random_ssp(tape, bits)
    so = ssp_o.reshape(m + 3, d)
    assert [int(x) for x in so[0]] == t
    for i in range(m):
        assert [int(x) for x in so[i + 1]] == v[i]

    # setup
    crs_o = oracle.setup(p, seed, ssp_o, alpha, beta, s, sk_l, err_l)
    sk, errs = _ints(sk_l), _ints(err_l)
    crs = mi.setup(seed, t, v, alpha, beta, s, sk, errs)
    assert crs["s"] == crs_o["s"].tobytes() and crs["as_"] == crs_o["as_"].tobytes() and crs["t"] == crs_o["t"].tobytes()
    assert crs["v"] == crs_o["v"].tobytes()[:(m - 1) * p.ctb]

    # prover
    tape5 = b"".join(mg + bytes([sg]) for mg, sg in smudges)
    out_o = oracle.prover(p, crs_o, ssp_o, bits, delta, tape5, 80)
    out = mi.prover(seed, crs, t, v, bits, delta, smudges)
    assert [int(x) for x in out_o["w"]] == out["w"]
    assert [int(x) for x in out_o["h"]] == out["h"]
    assert _cts(out_o["pre"]) == out["pre"]
    assert _cts(out_o["proof"]) == out["proof"]

    # verifier: same verdicts, accepted and rejected
    assert mi.verifier(t, v, alpha, beta, s, sk, out["proof"]) and oracle.verifier(p, ssp_o, alpha, beta, s, sk_l, out_o["proof"])
    bad_o = out_o["proof"].copy()
    bad_o[2, n, 0] ^= np.uint64(1 << 33)
    bad = [list(c) for c in out["proof"]]
    bad[2][n] ^= 1 << 33
    assert not mi.verifier(t, v, alpha, beta, s, sk, bad) and not oracle.verifier(p, ssp_o, alpha, beta, s, sk_l, bad_o)
    for k in range(5):
        assert mi.decrypt(sk, out["proof"][k]) == oracle.decrypt(p, sk_l, out_o["proof"][k])
