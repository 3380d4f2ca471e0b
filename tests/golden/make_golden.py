#!/usr/bin/env python3
"""Regenerates the committed golden vectors.  Run in the build container (needs /root/reference):

    make -C oracle all ref && python tests/golden/make_golden.py

Sources of truth:
  * reference_stream.json -- produced by the REAL reference: src/aes.c + src/entropy.c compiled in place into
    oracle/_ref/libmfref.so (oracle/Makefile target `ref`) and driven through oracle/ref_wrap.c.
    Covers aesctr_prg / rng_seek / rng_gen (stateless and stateful, incl. the ctr/rem state) and mpz2_urandomb.
  * gmp_lwe.json -- LWE arithmetic vectors produced by oracle/gmp_check.c, i.e. by the same libgmp 6.2.1 calls the
    reference's lwe.c makes (lwe.c itself cannot be compiled here: FLINT is absent).  NOT reference output.
Only data is stored (inputs and expected outputs); no reference source text.
"""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ref_stream():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmfref.so"))
    out = {"generator": "oracle/_ref/libmfref.so (reference src/aes.c + src/entropy.c, unmodified, compiled in place)", "stateless": [],
           "stateful": [], "urandomb": []}
    seeds = [bytes(range(40)), bytes(40), bytes([0xFF] * 40), bytes((i * 73 + 19) & 0xFF for i in range(40))]
    cases = [(0, 48), (92, 8), (5, 1), (15, 2), (16, 64), (135240, 100), (135232, 40), (8863223880, 92), (11817406440 - 7, 23),
             ((1 << 40) + 3, 50), (4431544320, 32)]
    for s in seeds:
        for off, n in cases:
            buf = ctypes.create_string_buffer(n)
            lib.ref_keystream(ctypes.c_char_p(s), ctypes.c_uint64(off), buf, ctypes.c_size_t(n))
            out["stateless"].append({"seed": s.hex(), "off": off, "n": n, "out": buf.raw.hex()})
    seqs = [[92] * 5, [1, 5, 32, 40, 65, 64, 92, 94], [16, 16, 3, 13, 92, 7], [8, 69, 1, 80, 1], [15, 1, 16, 17]]
    for s in seeds[:2]:
        for off in (0, 512, 135240 + 7):
            for sizes in seqs:
                arr = (ctypes.c_uint32 * len(sizes))(*sizes)
                buf = ctypes.create_string_buffer(sum(sizes))
                lib.ref_gen_sequence(ctypes.c_char_p(s), ctypes.c_uint64(off), arr, ctypes.c_size_t(len(sizes)), buf)
                ctr, rem = ctypes.c_uint64(), ctypes.c_uint64()
                lib.ref_state_after(ctypes.c_char_p(s), ctypes.c_uint64(off), arr, ctypes.c_size_t(len(sizes)), ctypes.byref(ctr), ctypes.byref(rem))
                out["stateful"].append({"seed": s.hex(), "off": off, "sizes": sizes, "out": buf.raw.hex(), "ctr": ctr.value, "rem": rem.value})
    # widths exercised by src/test_entropy.c:24-78 plus the parameter sets' widths
    for nbits in [64, 1, 5, 32, 40, 520, 512, 700] + list(range(736, 752)) + [1472]:
        limbs = (nbits + 63) // 64
        cnt = 3
        arr = np.zeros(cnt * limbs, dtype=np.uint64)
        lib.ref_urandomb(ctypes.c_char_p(seeds[0]), ctypes.c_uint64(92 * 3), ctypes.c_size_t(nbits), ctypes.c_size_t(cnt),
                         arr.ctypes.data_as(ctypes.c_void_p))
        out["urandomb"].append({"seed": seeds[0].hex(), "off": 92 * 3, "nbits": nbits, "count": cnt, "limbs": [int(x) for x in arr]})
    return out


def gmp_lwe():
    import oracle_lib as ol

    g = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmf_gmpcheck.so"))
    g.gx_decrypt.restype = ctypes.c_uint64
    L = 12
    rng = np.random.default_rng(2026)
    out = {"generator": "oracle/gmp_check.c on libgmp 6.2.1 (same mpz calls as reference src/lwe.c; not reference output)", "modq": [],
           "addmul_ui": [], "mul_ui": [], "add": [], "encrypt_b": [], "decrypt": [], "smudge": []}

    def P(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    specials = [(1 << 720) + 5, (1 << 704), (1 << 704) - 1, (1 << 736) - 1, 0, 1, (1 << 767) + (1 << 703)]
    for x in specials + [int.from_bytes(rng.bytes(96), "little") for _ in range(4)]:
        a = ol.int_to_limbs(x, L)
        o = np.zeros(L, dtype=np.uint64)
        g.gx_modq(P(o), P(a), ctypes.c_size_t(L))
        out["modq"].append({"in": hex(x), "out": hex(ol.limbs_to_int(o))})
    for _ in range(6):
        r = ol.rand_values(rng, 1, L, 704)[0]
        a = ol.rand_values(rng, 1, L, 736)[0]
        b = int(rng.integers(0, ol.P, dtype=np.uint64))
        r2 = r.copy()
        g.gx_addmul_ui(P(r2), P(a), ctypes.c_uint64(b))
        m = np.zeros(L, dtype=np.uint64)
        g.gx_mul_ui(P(m), P(a), ctypes.c_uint64(b))
        s = np.zeros(L, dtype=np.uint64)
        g.gx_add(P(s), P(r), P(a))
        out["addmul_ui"].append({"rop": hex(ol.limbs_to_int(r)), "a": hex(ol.limbs_to_int(a)), "b": b, "out": hex(ol.limbs_to_int(r2))})
        out["mul_ui"].append({"a": hex(ol.limbs_to_int(a)), "b": b, "out": hex(ol.limbs_to_int(m))})
        out["add"].append({"a": hex(ol.limbs_to_int(r)), "b": hex(ol.limbs_to_int(a)), "out": hex(ol.limbs_to_int(s))})
    n = 24  # a short vector keeps the fixture small; the arithmetic per term is the same
    for _ in range(3):
        a = ol.rand_values(rng, n, L, 736)
        sk = ol.rand_values(rng, n, L, 736)
        e = ol.rand_values(rng, 1, L, 559)[0]
        m = int(rng.integers(0, ol.P, dtype=np.uint64))
        b = np.zeros(L, dtype=np.uint64)
        g.gx_encrypt_b(P(b), P(a), P(sk), ctypes.c_size_t(n), ctypes.c_uint64(m), P(e))
        dec = g.gx_decrypt(P(a), P(b), P(sk), ctypes.c_size_t(n))
        out["encrypt_b"].append({"n": n, "a": [hex(ol.limbs_to_int(x)) for x in a], "sk": [hex(ol.limbs_to_int(x)) for x in sk],
                                 "e": hex(ol.limbs_to_int(e)), "m": m, "b": hex(ol.limbs_to_int(b))})
        out["decrypt"].append({"case": len(out["encrypt_b"]) - 1, "m": int(dec)})
    for sign in (0, 1):
        b = ol.rand_values(rng, 1, L, 704)[0]
        mag = rng.bytes(80)
        b2 = b.copy()
        neg = g.gx_smudge(P(b2), ctypes.c_char_p(mag), ctypes.c_size_t(80), ctypes.c_uint8(sign))
        out["smudge"].append({"b": hex(ol.limbs_to_int(b)), "mag": mag.hex(), "sign": sign, "out": hex(ol.limbs_to_int(b2)), "negative": int(neg)})
    return out


if __name__ == "__main__":
    json.dump(ref_stream(), open(os.path.join(HERE, "reference_stream.json"), "w"), indent=0)
    json.dump(gmp_lwe(), open(os.path.join(HERE, "gmp_lwe.json"), "w"), indent=0)
    print("wrote", os.listdir(HERE))
