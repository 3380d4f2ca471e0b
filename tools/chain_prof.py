"""The statement chain (witness GEMM + polynomial step) of one super-group of 248 statements at the default instance, alone on the GPU:
wall time per call, and -- under rocprofv3 --kernel-trace --stats -- its kernels.  dev tool.
usage: python tools/chain_prof.py [nstmt=248] [reps=5] [config4|config5]   (generator-defined SSP at 2^20 constraints)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf
big = [a for a in sys.argv[1:] if a.startswith("config")]
p = mf.DEFAULT if not big else mf.Params(logq=736 if big[0] == "config4" else 1472, d=1 << 20, m=699050)
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
if big:
    inst = bench.build_prg_instance(mf, ctx, torch, p, 20260101)
    ctx.ssp_prepare(None)
else:
    inst = bench.build_instance(mf, ctx, torch, p, 20260101)
    ctx.ssp_prepare(inst["d_ssp"])
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 248
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(5)
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
out = ctx.batch_chain(inst["d_ssp"], [inst["bits"]] * nb, deltas)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): ctx.batch_chain(inst["d_ssp"], [inst["bits"]] * nb, deltas, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"chain of {nb} statements: {dt*1e3:.3f} ms per call ({dt*1e6/nb:.2f} us per statement)", flush=True)
