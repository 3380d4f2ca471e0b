"""mfh_encrypt_rows on the matrix cores, three batches of 65536 rows, for rocprofv3. dev tool."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
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

p = mf.Params()
ctx = mf.Context(p, 0); ctx.set_seed(bytes(range(40)))
rng = np.random.default_rng(1)
d_sk = ctx.to_device(rand_values(rng, p.n, p.L, p.logq))
B = 65536
d_msg = ctx.to_device(rng.integers(0, mf.P, size=B, dtype=np.uint64).astype(np.uint32))
d_err = ctx.to_device(rand_values(rng, B, p.L, 559))
ctx.set_encrypt_path(2)
for _ in range(3): out = ctx.encrypt_rows(0, B, d_sk, d_msg, d_err)
torch.cuda.synchronize()
