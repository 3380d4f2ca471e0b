"""Throughput of mfh_verify (the device verifier: 5 decryptions + the four equations per proof) on a batch of proofs. dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
rng = np.random.default_rng(5)
nb = 248
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
proofs = ctx.prove_batch(d_crs, inst["d_ssp"], [inst["bits"]] * nb, deltas, mags, [bytes(5)] * nb)
ok = ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], proofs, nb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ok = ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], proofs, nb)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"verify {nb} proofs: {dt*1e3:.2f} ms = {nb/dt:.0f} proofs/s; accepted {int(ctx.to_host(ok).sum())}/{nb}")
