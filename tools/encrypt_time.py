"""mfh_encrypt_rows: VALU kernel (path 1) against the matrix-core kernel (path 2) at benchmark_lwe parameters. dev tool.
usage: python tools/encrypt_time.py [nrows ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import c_lwe_snarks_amd as mf


def rand_values(rng, count, L, bits):
    """count random values of `bits` bits as (count, L) uint64 limbs"""
    nb = (bits + 7) // 8
    raw = rng.integers(0, 256, size=(count, nb), dtype=np.uint8)
    if bits % 8:
        raw[:, -1] &= (1 << (bits % 8)) - 1
    out = np.zeros((count, L * 8), dtype=np.uint8)
    out[:, :nb] = raw
    return out.view(np.uint64).reshape(count, L)

logq = int(os.environ.get("LOGQ", "736"))
p = mf.Params(logq=logq)
ctx = mf.Context(p, 0)
ctx.set_seed(bytes(range(40)))
rng = np.random.default_rng(1)
d_sk = ctx.to_device(rand_values(rng, p.n, p.L, p.logq))
for B in [int(a) for a in sys.argv[1:]] or [65536, 87381, 8192, 1000]:
    d_msg = ctx.to_device(rng.integers(0, mf.P, size=B, dtype=np.uint64).astype(np.uint32))
    d_err = ctx.to_device(rand_values(rng, B, p.L, 559))
    outs = {}
    for path in (1, 2):
        ctx.set_encrypt_path(path)
        out = ctx.encrypt_rows(0, B, d_sk, d_msg, d_err)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): ctx.encrypt_rows(0, B, d_sk, d_msg, d_err, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        outs[path] = out.clone()
        print(f"logq {logq} B={B:6d} path {path}: {dt*1e3:8.3f} ms  {B/dt/1e6:6.2f} M enc/s  {B*p.ctr_ct/16/dt/1e9:6.1f} Gblock/s", flush=True)
    print("   identical:", bool(torch.equal(outs[1], outs[2])), flush=True)
