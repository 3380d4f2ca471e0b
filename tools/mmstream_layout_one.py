"""One mfh_prove_batch call (default instance) under ONE k_mmstream launch layout -- the program tools/mmstream_layout_pmc.py runs under rocprofv3 --pmc.  dev tool.
usage: python3 tools/mmstream_layout_one.py map,persistent,sync,spin[,ngl] [nb]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf
v = [int(x) for x in sys.argv[1].split(",")]
v += [4] * (5 - len(v))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 510
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
rng = np.random.default_rng(5)
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
ctx.set_batch_launch(v[4], True)
ctx.set_mm_stream(v[0], bool(v[1]), v[2], v[3])
out = ctx.prove_batch(d_crs, inst["d_ssp"], [inst["bits"]] * nb, deltas, mags, [bytes(5)] * nb)
torch.cuda.synchronize()
print("digest", int(out.view(torch.int64).sum().item()))
