"""A/B of the polynomial step's exact-division path (mfh_set_poly_exact) inside mfh_prove_batch at the default instance: $AB_NB statements (default 1020) carrying the
satisfying witness (what the reference's benchmark_snark proves) or, with $AB_INVALID=k, every k-th one a random witness.  Alternates on / off, prints ms per step,
the statements that fell back, and whether the proofs are bit-identical.  dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf
MODE = int(os.environ.get("EXACT_MODE", "2"))  # 2: every batch tries the exact path; 1: the default (64 batches of Euclidean division alone after a failed check)
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
rng = np.random.default_rng(5)
nb = int(os.environ.get("AB_NB", "1020"))
every = int(os.environ.get("AB_INVALID", "0"))
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
signs = [bytes(5)] * nb
bits = [rng.bytes(len(inst["bits"])) if every and i % every == every - 1 else inst["bits"] for i in range(nb)]
print("exact path offered:", ctx.poly_exact_fallbacks() >= 0, flush=True)
outs = {}
for rnd in range(3):
    for on in (True, False):
        ctx.set_poly_exact(MODE if on else 0)
        out = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
        torch.cuda.synchronize()
        ctx.poly_exact_fallbacks()
        t0 = time.perf_counter()
        for _ in range(5):
            out = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        fb = ctx.poly_exact_fallbacks()
        outs[on] = out.clone()
        print(f"exact={int(on)}  {ms:8.3f} ms per step  {nb / ms * 1e3:9.1f} proofs/s   statements that failed the check in 5 steps: {fb}", flush=True)
print("proofs identical:", bool(torch.equal(outs[True], outs[False])))
