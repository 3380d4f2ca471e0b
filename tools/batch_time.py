"""Time mfh_prove_batch at the default instance. dev tool.
usage: python tools/batch_time.py [nproofs ...] [--resident] [--launch=NGL,MERGE[:NGL,MERGE...]]   (groups per streaming launch, S+AS in one launch)"""
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
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--resident" in sys.argv:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    image = ctx.crs_expand_mm(d_crs)
    torch.cuda.synchronize()
    print(f"matrix-core CRS image: {image.numel()/1e9:.2f} GB expanded in {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    ctx.set_resident_mm(image)
launches = [(4, 1)]
for a in sys.argv[1:]:
    if a.startswith("--launch="):
        launches = [tuple(int(x) for x in t.split(",")) for t in a[9:].split(":")]
for (ngl, merge), nb in [(l, int(a)) for l in launches for a in (args or ["12", "24"])]:
    ctx.set_batch_launch(ngl, bool(merge))
    deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
    signs = [bytes(5)] * nb
    out = ctx.prove_batch(d_crs, inst["d_ssp"], [inst["bits"]] * nb, deltas, mags, signs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 3
    for _ in range(reps): ctx.prove_batch(d_crs, inst["d_ssp"], [inst["bits"]] * nb, deltas, mags, signs, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    ok = ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], out, nb))
    one = ctx.prove(d_crs, inst["d_ssp"], inst["bits"], deltas[nb - 1], mags[nb - 1], signs[nb - 1])
    same = torch.equal(out.view(nb, -1)[nb - 1], one)
    print(f"launch ngl={ngl} merge={merge}  batch of {nb:3d}: {dt*1e3:8.2f} ms = {dt*1e3/nb:6.3f} ms/proof = {nb/dt:7.1f} proofs/s; accepted {int(ok.sum())}/{nb}; last == single-proof path: {same}", flush=True)
