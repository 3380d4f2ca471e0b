"""Single proof (mfh_prove) at the NDEBUG default size with the tile kernel (k_eval, path 0) and the wave-autonomous eval kernel (k_eval_w, path 1):
ms per proof, AES rate of the S-region launch, proofs identical.  dev tool (same-box A/B)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import c_lwe_snarks_amd as mf  # noqa: E402

p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
rng = np.random.default_rng(99)
delta = int(rng.integers(0, mf.P, dtype=np.uint64))
mags = rng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes()
signs = bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist())
proofs = {}
for rep in range(2):
    for path in (0, 1):
        ctx.set_eval_path(path)
        for _ in range(3):
            pr = ctx.prove(d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs)
        ctx.set_timing(True)
        ctx.timing_drain("eval")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            pr = ctx.prove(d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ctx.set_timing(False)
        n2, ms2, rows2 = ctx.timing_drain("eval2")
        n1, ms1, rows1 = ctx.timing_drain("eval1")
        gblk = rows2 * (p.ctr_ct / 16.0) / (ms2 * 1e-3) / 1e9 if n2 else 0
        proofs[path] = pr.clone()
        print(f"path {path} ({'k_eval_w' if path == 1 else 'k_eval'}): {ms:6.3f} ms/proof; eval2 {ms2 / max(n2, 1):6.3f} ms/launch = {gblk:5.1f} Gblock/s; "
              f"eval1 {ms1 / max(n1, 1):6.3f} ms/launch ({rows1 / max(n1, 1):.0f} rows)", flush=True)
print("proofs identical:", bool(torch.equal(proofs[0], proofs[1])))
ok = bench.verify_on_gpu(mf, ctx, inst, proofs[0])
print("verifier accepts:", ok)
sys.exit(0 if ok and torch.equal(proofs[0], proofs[1]) else 1)
