#!/usr/bin/env python3
"""Golden hashes of ONE complete prover() run at the NDEBUG default size (D = 2^15, M = 21845), produced by the CPU oracle.

    python tests/golden/make_default_size_golden.py        (about 2-3 minutes, ~12 GB of RAM, one core)

The instance is deterministic and reproducible anywhere: the SSP is the generator-defined one of csrc/ssp_prg.hpp (restated here
in numpy), the CRS bytes / witness / delta / smudging draws come from numpy's default_rng with fixed seeds.  The CRS is random
bytes (a prover does not care), so the proof does not verify -- what is pinned is every bit of the five ciphertexts, of the
witness polynomial w and of h = (v^2-1)/t.  tests/test_gpu_fullsize.py::test_default_size_proof_matches_oracle_hashes rebuilds
the same instance on the GPU and compares hashes.  Source of truth: oracle/mf_oracle.c (NOT the reference itself).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

P = 0xFFFFFFFB
PRG_SEED = 0x00C0FFEE12345678
M32 = np.uint64(0xFFFFFFFF)


def rowkey(seed, slot):
    r = (int(seed) + slot * 0x9E3779B1) & 0xFFFFFFFF
    r ^= r >> 15
    r = (r * 0x2C1B3C6D) & 0xFFFFFFFF
    r ^= (seed >> 32) & 0xFFFFFFFF
    return r | 1


def coeff_row(seed, slot, d):
    k = np.arange(d, dtype=np.uint64)
    x = ((k + np.uint64(0x632BE5AB)) * np.uint64(rowkey(seed, slot))) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return np.where(x >= np.uint64(P), x - np.uint64(P), x)


def instance(p):
    rng = np.random.default_rng(20261003)
    bits = rng.bytes((p.m + 7) // 8)
    c8 = rng.integers(0, 256, size=(2 * p.d + p.m) * p.ctb, dtype=np.uint8)
    delta = int(rng.integers(0, P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes()
    signs = bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist())
    seed = bytes((29 * i + 3) & 0xFF for i in range(40))
    return dict(bits=bits, c8=c8, delta=delta, mags=mags, signs=signs, seed=seed)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


if __name__ == "__main__":
    import c_lwe_snarks_amd as mf
    import oracle_lib as ol

    p = mf.DEFAULT
    I = instance(p)
    t0 = time.time()
    ssp = np.zeros((p.m + 3, p.d), dtype=np.uint64)
    t = coeff_row(PRG_SEED, 1, p.d).copy()
    for slot in range(1, p.m + 1):
        ssp[slot] = coeff_row(PRG_SEED, slot, p.d)
        i = slot - 1
        if i >= 1 and (I["bits"][(i - 1) >> 3] >> ((i - 1) & 7)) & 1:
            t = (t + ssp[slot]) % np.uint64(P)
    t[0] = (t[0] + np.uint64(P - 1)) % np.uint64(P)
    ssp[0] = t
    print(f"ssp built in {time.time() - t0:.0f} s", flush=True)
    o = ol.Oracle()
    c8 = I["c8"]
    crs = dict(seed=I["seed"], s=c8[: p.d * p.ctb].copy(), as_=c8[p.d * p.ctb: 2 * p.d * p.ctb].copy(),
               t=c8[2 * p.d * p.ctb: (2 * p.d + 1) * p.ctb].copy(),
               v=np.concatenate([c8[(2 * p.d + 1) * p.ctb:], np.zeros(p.ctb, dtype=np.uint8)]))
    tape = b"".join(I["mags"][80 * k: 80 * k + 80] + I["signs"][k: k + 1] for k in range(5))
    t0 = time.time()
    out = o.prover(p, crs, ssp.reshape(-1), I["bits"], I["delta"], tape, 80)
    print(f"oracle prover in {time.time() - t0:.0f} s", flush=True)
    names = ["h", "hat_h", "hat_v", "v_w", "b_w"]
    gold = {"generator": "oracle/mf_oracle.c mfo_prover at D=32768, M=21845 (CPU restatement, not the reference)", "prg_seed": PRG_SEED,
            "t_sha256": sha(t.astype(np.uint32)), "w_sha256": sha(out["w"].astype(np.uint32)), "h_sha256": sha(out["h"].astype(np.uint32)),
            "proof_sha256": {n: sha(out["proof"][k]) for k, n in enumerate(names)},
            "pre_smudge_sha256": {n: sha(out["pre"][k]) for k, n in enumerate(names)},
            "proof_all_sha256": sha(out["proof"])}
    json.dump(gold, open(os.path.join(HERE, "default_size_proof.json"), "w"), indent=1)
    print(json.dumps(gold, indent=1))
