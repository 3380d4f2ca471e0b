"""four mfh_prove_batch steps of $BATCH_PROF_NB (default 1020) statements at the default instance, for rocprofv3 --kernel-trace (tools/step_breakdown.py). dev tool.
usage: python tools/batch_prof.py [merge 0|1]      $MFUOCO_MM_PACK=0: int32 partial products (the round-5 epilogue); $MFUOCO_POLY_EXACT=0: no exact-division path"""
import os, sys
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
nb = int(os.environ.get("BATCH_PROF_NB", "1020"))
if os.environ.get("MFUOCO_MM_PACK") == "0":
    ctx.set_mm_pack(False)
if os.environ.get("MFUOCO_POLY_EXACT") == "0":
    ctx.set_poly_exact(0)  # the polynomial step by Euclidean division only (rounds 1 - 5)
ctx.set_batch_launch(8, bool(int(sys.argv[1])) if len(sys.argv) > 1 else True)
if os.environ.get("MFUOCO_MM_WAVE1") == "1":
    ctx.set_mm_stream(1, 2, 0, 0)
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
signs = [bytes(5)] * nb
out = None
for _ in range(4):
    out = ctx.prove_batch(d_crs, inst["d_ssp"], [inst["bits"]] * nb, deltas, mags, signs, out=out)
torch.cuda.synchronize()
