"""CPU: the oracle's AES-256-CTR stream and sampler against (i) vectors generated from the REAL reference build
(tests/golden/reference_stream.json), (ii) the live reference build oracle/_ref when it is present, (iii) FIPS-197,
(iv) the properties src/test_aes.c and src/test_entropy.c assert."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_stream.json")))
REF_SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libmfref.so")


def test_fips197_aes256_kat(oracle):
    # FIPS-197 Appendix C.3
    key = bytes(range(32))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    rk = (ctypes.c_uint32 * 60)()
    oracle.lib.mfo_aes256_expand_key(rk, ctypes.c_char_p(key))
    out = ctypes.create_string_buffer(16)
    oracle.lib.mfo_aes256_encrypt_block(rk, ctypes.c_char_p(pt), out)
    assert out.raw.hex() == "8ea2b7ca516745bfeafc49904b496089"


def test_survey_kats(oracle):
    seed = bytes(range(40))
    assert oracle.keystream(seed, 0, 16).hex() == "8477f45516027713a26a881ae67882bf"
    assert oracle.keystream(seed, 92, 8).hex() == "ed29f6cae21f9e67"


def test_stateless_golden(oracle):
    for c in GOLD["stateless"]:
        assert oracle.keystream(bytes.fromhex(c["seed"]), c["off"], c["n"]).hex() == c["out"], c


def test_stateful_golden_including_ctr_and_rem(oracle):
    # callers interleave rng_seek / ct_import / rng_gen, so ctr and rem after partial reads are part of the contract
    for c in GOLD["stateful"]:
        r = oracle.rng(bytes.fromhex(c["seed"]), c["off"])
        got = b"".join(oracle.rng_gen(r, n) for n in c["sizes"])
        assert got.hex() == c["out"]
        assert (r.ctr, r.rem) == (c["ctr"], c["rem"])


def test_urandomb_golden(oracle):
    for c in GOLD["urandomb"]:
        r = oracle.rng(bytes.fromhex(c["seed"]), c["off"])
        limbs = (c["nbits"] + 63) // 64
        got = np.concatenate([oracle.urandomb(r, c["nbits"]) for _ in range(c["count"])])
        assert [int(x) for x in got] == c["limbs"], c["nbits"]
        assert len(got) == limbs * c["count"]


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference; `make -C oracle ref`)")
def test_against_live_reference_build(oracle):
    lib = ctypes.CDLL(REF_SO)
    rng = np.random.default_rng(0)
    for _ in range(300):
        seed = rng.bytes(40)
        off = int(rng.integers(0, 1 << 44))
        n = int(rng.integers(1, 5000))
        buf = ctypes.create_string_buffer(n)
        lib.ref_keystream(ctypes.c_char_p(seed), ctypes.c_uint64(off), buf, ctypes.c_size_t(n))
        assert buf.raw == oracle.keystream(seed, off, n)
    # bulk vs 92-byte chunks over 1000 elements (src/test_entropy.c:111-137, reduced)
    seed = rng.bytes(40)
    n = 92 * 1000
    buf = ctypes.create_string_buffer(n)
    sizes = (ctypes.c_uint32 * 1000)(*([92] * 1000))
    lib.ref_gen_sequence(ctypes.c_char_p(seed), ctypes.c_uint64(0), sizes, ctypes.c_size_t(1000), buf)
    assert buf.raw == oracle.keystream(seed, 0, n)


def test_properties_of_reference_tests(oracle):
    seed = os.urandom(40)
    # src/test_aes.c: non-trivial, consecutive blocks differ
    a, b = oracle.keystream(seed, 0, 16), oracle.keystream(seed, 16, 16)
    assert a != b and any(a)
    # src/test_entropy.c:24-78: same seed => same draws, for all the widths it lists
    r1, r2 = oracle.rng(seed), oracle.rng(seed)
    for nbits in [64, 1, 5, 32, 40, 520, 512] + list(range(736, 752)):
        assert np.array_equal(oracle.urandomb(r1, nbits), oracle.urandomb(r2, nbits))
    # :138-156 seek(512) == reading past 512 bytes
    r = oracle.rng(seed)
    oracle.rng_gen(r, 512)
    r_seek = oracle.rng(seed, 512)
    assert oracle.rng_gen(r, 8) == oracle.rng_gen(r_seek, 8)
    # sampling a ciphertext row == 1470 consecutive 92-byte elements of the stream, top limb holds 32 bits
    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    rows = oracle.sample_rows(p, seed, p.ctr_bv, 1)[0]
    raw = oracle.keystream(seed, p.ctr_bv, p.ctr_ct)
    for j in (0, 1, 733, 1469):
        assert ol.limbs_to_int(rows[j]) == int.from_bytes(raw[92 * j: 92 * j + 92], "little")
