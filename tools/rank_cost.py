"""Per-rank cost of the row-sharded prover without the collective: one process runs rank 0's share of an N-rank job
(prove_sharded with world = N outside a process group, so the two lane all-reduces are no-ops).  The result is a
partial proof -- only the time is meaningful: it bounds the strong-scaling efficiency bench.py --gpus N can reach.
usage: python tools/rank_cost.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import c_lwe_snarks_amd as mf  # noqa: E402
from c_lwe_snarks_amd import dist as mfdist  # noqa: E402

worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
p = mf.DEFAULT
ctx = mf.Context(p, 0)
if os.environ.get("MFUOCO_OVERLAP"):  # 0 off, 1 auto, 2 b_w first, 3 chain first
    ctx.lib.mfh_set_overlap(ctx._h, int(os.environ["MFUOCO_OVERLAP"]))
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
rng = np.random.default_rng(99)
delta = int(rng.integers(0, mf.P, dtype=np.uint64))
mags = rng.integers(0, 256, size=400, dtype=np.uint8).tobytes()
signs = bytes(5)
for resident in (False, True):
    for w in worlds:
        bufs = {}
        if resident:
            image = ctx.crs_expand_share(d_crs, 0, w)
            ctx.set_resident_share(image, 0, w)
        for _ in range(3):
            mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs, 0, w, bufs=bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs, 0, w, bufs=bufs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{'resident' if resident else 'regenerate'} world={w}: rank-0 share {ms:7.3f} ms/proof -> <= {1e3 / ms:7.1f} proofs/s "
              f"(ideal {w}x of world=1 would be {ms * w:6.2f} ms-equivalents)", flush=True)
