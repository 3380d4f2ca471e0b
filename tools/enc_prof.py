import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
import oracle_lib as ol
import c_lwe_snarks_amd as mf
p = mf.Params()
ctx = mf.Context(p, 0); ctx.set_seed(bytes(range(40)))
rng = np.random.default_rng(1)
d_sk = ctx.to_device(ol.rand_values(rng, p.n, p.L, p.logq))
B = 65536
d_msg = ctx.to_device(rng.integers(0, mf.P, size=B, dtype=np.uint64).astype(np.uint32))
d_err = ctx.to_device(ol.rand_values(rng, B, p.L, 559))
ctx.set_encrypt_path(2)
for _ in range(3): out = ctx.encrypt_rows(0, B, d_sk, d_msg, d_err)
torch.cuda.synchronize()
