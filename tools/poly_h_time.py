"""The polynomial step alone: mfh_poly_h_multi for $NB (default 255) statements at the default size, exact-division path on / off, ms per call; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times without anything running beside them.  $PH_INVALID=k: every k-th statement does not divide.  dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf
MODE = int(os.environ.get("EXACT_MODE", "2"))  # 2: every batch tries the exact path; 1: the default (64 batches of Euclidean division alone after a failed check)
p = mf.DEFAULT
ctx = mf.Context(p, 0)
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
nb = int(os.environ.get("NB", "255"))
every = int(os.environ.get("PH_INVALID", "0"))
rng = np.random.default_rng(3)
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
bits = [rng.bytes(len(inst["bits"])) if every and i % every == every - 1 else inst["bits"] for i in range(nb)]
w = ctx.witness_poly_many(inst["d_ssp"], bits, deltas).view(torch.int32).view(nb, p.d).to(torch.int64) & 0xFFFFFFFF
v0 = inst["d_ssp"].view(torch.int32).view(p.m + 3, p.d)[1].to(torch.int64) & 0xFFFFFFFF
v = ((w + v0) % mf.P).to(torch.int32).contiguous().view(torch.uint8).view(-1)
res = {}
for on in (True, False, True, False):
    ctx.set_poly_exact(MODE if on else 0)
    h = ctx.poly_h_many(v, nb)
    torch.cuda.synchronize()
    ctx.poly_exact_fallbacks()
    t0 = time.perf_counter()
    for _ in range(20):
        h = ctx.poly_h_many(v, nb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    res[on] = h.clone()
    print(f"exact={int(on)}  {ms:7.3f} ms per call of {nb} statements   failed the check in 20 calls: {ctx.poly_exact_fallbacks()}", flush=True)
print("identical:", bool(torch.equal(res[True], res[False])))
