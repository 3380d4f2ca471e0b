"""ctypes/numpy wrapper around oracle/libmf_oracle.so (the CPU restatement) for the tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this: the oracle is the
checker, never the product.  Values are numpy uint64 arrays of L limbs; ciphertexts are (n+1, L).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFB


class CParams(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint32), ("logq", ctypes.c_uint32), ("d", ctypes.c_uint32), ("m", ctypes.c_uint32)]


class CRng(ctypes.Structure):
    _fields_ = [("rk", ctypes.c_uint32 * 60), ("nonce", ctypes.c_uint64), ("ctr", ctypes.c_uint64),
                ("remb", ctypes.c_uint8 * 16), ("rem", ctypes.c_size_t)]


class CCrs(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_void_p), ("s", ctypes.c_void_p), ("as_", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("t", ctypes.c_void_p)]


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libmf_oracle.so")
    src = os.path.join(ORACLE_DIR, "mf_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)
    return so


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


class Oracle:
    def __init__(self):
        self.lib = ctypes.CDLL(build_oracle())
        L = self.lib
        L.mfo_decrypt.restype = ctypes.c_uint64
        L.mfo_poly_eval.restype = ctypes.c_uint64
        L.mfo_ct_smudge.restype = ctypes.c_int
        L.mfo_verifier.restype = ctypes.c_int
        L.mfo_poly_divides.restype = ctypes.c_int
        L.mfo_bench_eval_rows.restype = ctypes.c_uint64
        L.mfo_bench_encrypt.restype = ctypes.c_uint64
        L.mfo_bench_decrypt.restype = ctypes.c_uint64

    # ---- params ----
    @staticmethod
    def cp(p):
        return CParams(p.n, p.logq, p.d, p.m)

    # ---- stream ----
    def keystream(self, seed: bytes, off: int, n: int) -> bytes:
        buf = ctypes.create_string_buffer(n)
        self.lib.mfo_keystream(ctypes.c_char_p(bytes(seed)), ctypes.c_uint64(off), buf, ctypes.c_size_t(n))
        return buf.raw

    def rng(self, seed: bytes, off: int = 0):
        r = CRng()
        self.lib.mfo_rng_init(ctypes.byref(r), ctypes.c_char_p(bytes(seed)))
        if off:
            self.lib.mfo_rng_seek(ctypes.byref(r), ctypes.c_uint64(off))
        return r

    def rng_seek(self, r, off):
        self.lib.mfo_rng_seek(ctypes.byref(r), ctypes.c_uint64(off))

    def rng_gen(self, r, n) -> bytes:
        buf = ctypes.create_string_buffer(n)
        self.lib.mfo_rng_gen(ctypes.byref(r), buf, ctypes.c_size_t(n))
        return buf.raw

    def urandomb(self, r, nbits) -> np.ndarray:
        out = np.zeros((nbits + 63) // 64, dtype=np.uint64)
        self.lib.mfo_urandomb(_p(out), ctypes.byref(r), ctypes.c_size_t(nbits))
        return out

    # ---- lwe ----
    def add_dotp(self, p, rop, a, b):
        """mpz_add_dotp (src/lwe.c:20-28): rop + sum_j a[j] b[j], modq once at the end"""
        rop = np.ascontiguousarray(rop, dtype=np.uint64).copy()
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        cp = self.cp(p)
        self.lib.mfo_add_dotp(ctypes.byref(cp), _p(rop), _p(a), _p(b), ctypes.c_size_t(a.shape[0]))
        return rop

    def sample_rows(self, p, seed, off, nrows):
        r = self.rng(seed, off)
        out = np.zeros((nrows, p.n + 1, p.L), dtype=np.uint64)
        cp = self.cp(p)
        for i in range(nrows):
            self.lib.mfo_sample_a(ctypes.byref(cp), _p(out[i]), ctypes.byref(r))
        return out[:, : p.n, :].copy()

    def modq(self, p, v):
        v = np.ascontiguousarray(v, dtype=np.uint64).copy()
        cp = self.cp(p)
        self.lib.mfo_modq(ctypes.byref(cp), _p(v))
        return v

    def encrypt(self, p, r, sk, m, e):
        ct = np.zeros((p.n + 1, p.L), dtype=np.uint64)
        cp = self.cp(p)
        e = np.ascontiguousarray(e, dtype=np.uint64)
        self.lib.mfo_encrypt(ctypes.byref(cp), _p(ct), ctypes.byref(r), _p(sk), ctypes.c_uint64(int(m)), _p(e))
        return ct

    def decrypt(self, p, sk, ct) -> int:
        cp = self.cp(p)
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        return int(self.lib.mfo_decrypt(ctypes.byref(cp), _p(sk), _p(ct)))

    def ct_export(self, p, ct) -> bytes:
        buf = ctypes.create_string_buffer(p.ctb)
        cp = self.cp(p)
        self.lib.mfo_ct_export(ctypes.byref(cp), buf, _p(np.ascontiguousarray(ct)))
        return buf.raw

    def ct_import(self, p, r, buf: bytes):
        ct = np.zeros((p.n + 1, p.L), dtype=np.uint64)
        cp = self.cp(p)
        self.lib.mfo_ct_import(ctypes.byref(cp), _p(ct), ctypes.byref(r), ctypes.c_char_p(bytes(buf)))
        return ct

    def ct_mul_ui(self, p, a, x):
        rop = np.zeros_like(a)
        cp = self.cp(p)
        self.lib.mfo_ct_mul_ui(ctypes.byref(cp), _p(rop), _p(np.ascontiguousarray(a)), ctypes.c_uint64(x))
        return rop

    def ct_addmul_ui(self, p, rop, a, x):
        rop = np.ascontiguousarray(rop).copy()
        cp = self.cp(p)
        self.lib.mfo_ct_addmul_ui(ctypes.byref(cp), _p(rop), _p(np.ascontiguousarray(a)), ctypes.c_uint64(x))
        return rop

    def ct_add(self, p, a, b):
        rop = np.zeros_like(a)
        cp = self.cp(p)
        self.lib.mfo_ct_add(ctypes.byref(cp), _p(rop), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))
        return rop

    def ct_smudge(self, p, ct, mag: bytes, sign: int):
        ct = np.ascontiguousarray(ct).copy()
        cp = self.cp(p)
        neg = self.lib.mfo_ct_smudge(ctypes.byref(cp), _p(ct), ctypes.c_char_p(bytes(mag)), ctypes.c_size_t(len(mag)),
                                     ctypes.c_uint8(sign))
        return ct, bool(neg)

    def eval_poly(self, p, seed, off, c8: bytes, coeff, rop=None):
        d = len(coeff)
        rop = np.zeros((p.n + 1, p.L), dtype=np.uint64) if rop is None else np.ascontiguousarray(rop).copy()
        r = self.rng(seed, off)
        cp = self.cp(p)
        co = np.ascontiguousarray(coeff, dtype=np.uint64)
        self.lib.mfo_eval_poly(ctypes.byref(cp), _p(rop), ctypes.byref(r), ctypes.c_char_p(bytes(c8)), _p(co), ctypes.c_size_t(d))
        return rop

    # ---- poly / ssp ----
    def poly_eval(self, poly, x) -> int:
        poly = np.ascontiguousarray(poly, dtype=np.uint64)
        return int(self.lib.mfo_poly_eval(_p(poly), ctypes.c_size_t(len(poly)), ctypes.c_uint64(x)))

    def poly_h(self, v, t):
        v = np.ascontiguousarray(v, dtype=np.uint64)
        t = np.ascontiguousarray(t, dtype=np.uint64)
        q = np.zeros(len(v), dtype=np.uint64)
        self.lib.mfo_poly_h(_p(q), _p(v), _p(t), ctypes.c_size_t(len(v)))
        return q

    def poly_divides(self, v, t) -> bool:
        v = np.ascontiguousarray(v, dtype=np.uint64)
        t = np.ascontiguousarray(t, dtype=np.uint64)
        return bool(self.lib.mfo_poly_divides(_p(v), _p(t), ctypes.c_size_t(len(v))))

    def ssp_from_tape(self, p, tape: np.ndarray, witness_bits: bytes):
        ssp = np.zeros((p.m + 3) * p.d, dtype=np.uint64)
        cp = self.cp(p)
        tape = np.ascontiguousarray(tape)
        self.lib.mfo_ssp_from_tape(ctypes.byref(cp), _p(ssp), _p(tape), ctypes.c_char_p(bytes(witness_bits)))
        return ssp

    # ---- snark ----
    def setup(self, p, seed, ssp, alpha, beta, s, sk, etape):
        cs = np.zeros(p.d * p.ctb, dtype=np.uint8)
        cas = np.zeros(p.d * p.ctb, dtype=np.uint8)
        cv = np.zeros(p.m * p.ctb, dtype=np.uint8)
        ctt = np.zeros(p.ctb, dtype=np.uint8)
        cp = self.cp(p)
        self.lib.mfo_setup(ctypes.byref(cp), _p(cs), _p(cas), _p(cv), _p(ctt), ctypes.c_char_p(bytes(seed)), _p(ssp),
                           ctypes.c_uint64(alpha), ctypes.c_uint64(beta), ctypes.c_uint64(s), _p(sk),
                           _p(np.ascontiguousarray(etape, dtype=np.uint64)))
        return dict(seed=bytes(seed), s=cs, as_=cas, v=cv, t=ctt)

    def prover(self, p, crs, ssp, witness_bits: bytes, delta, smudge_tape: bytes, maglen=80, want_pre=True):
        proof = np.zeros((5, p.n + 1, p.L), dtype=np.uint64)
        pre = np.zeros_like(proof) if want_pre else None
        w = np.zeros(p.d, dtype=np.uint64)
        h = np.zeros(p.d, dtype=np.uint64)
        seedbuf = ctypes.create_string_buffer(bytes(crs["seed"]), 40)
        c = CCrs(ctypes.cast(seedbuf, ctypes.c_void_p), _p(crs["s"]), _p(crs["as_"]), _p(crs["v"]), _p(crs["t"]))
        cp = self.cp(p)
        self.lib.mfo_prover(ctypes.byref(cp), _p(proof), _p(pre), ctypes.byref(c), _p(ssp), ctypes.c_char_p(bytes(witness_bits)),
                            ctypes.c_uint64(delta), ctypes.c_char_p(bytes(smudge_tape)), ctypes.c_size_t(maglen), _p(w), _p(h))
        return dict(proof=proof, pre=pre, w=w, h=h)

    def verifier(self, p, ssp, alpha, beta, s, sk, proof) -> bool:
        cp = self.cp(p)
        return bool(self.lib.mfo_verifier(ctypes.byref(cp), _p(ssp), ctypes.c_uint64(alpha), ctypes.c_uint64(beta),
                                          ctypes.c_uint64(s), _p(sk), _p(np.ascontiguousarray(proof))))

    # ---- cpu baseline ----
    def bench_eval_rows(self, p, seed, rows):
        cp = self.cp(p)
        return int(self.lib.mfo_bench_eval_rows(ctypes.byref(cp), ctypes.c_char_p(bytes(seed)), ctypes.c_size_t(rows)))

    def bench_decrypt(self, p, seed, count):
        cp = self.cp(p)
        return int(self.lib.mfo_bench_decrypt(ctypes.byref(cp), ctypes.c_char_p(bytes(seed)), ctypes.c_size_t(count)))

    def bench_encrypt(self, p, seed, count):
        cp = self.cp(p)
        return int(self.lib.mfo_bench_encrypt(ctypes.byref(cp), ctypes.c_char_p(bytes(seed)), ctypes.c_size_t(count)))


def limbs_to_int(v) -> int:
    return int.from_bytes(np.ascontiguousarray(v, dtype=np.uint64).tobytes(), "little")


def int_to_limbs(x: int, L: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(8 * L, "little"), dtype=np.uint64).copy()


def rand_values(rng: np.random.Generator, count: int, L: int, bits: int) -> np.ndarray:
    """count random values of `bits` bits as (count, L) uint64"""
    nb = (bits + 7) // 8
    raw = rng.integers(0, 256, size=(count, nb), dtype=np.uint8)
    if bits % 8:
        raw[:, -1] &= (1 << (bits % 8)) - 1
    out = np.zeros((count, L * 8), dtype=np.uint8)
    out[:, :nb] = raw
    return out.view(np.uint64).reshape(count, L)
