import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c_lwe_snarks_amd as mf, torch
p = mf.DEFAULT
c = mf.Context(p, 0); c.set_seed(bytes(range(40))); c.set_timing(True)
rng = np.random.default_rng(0)
for nrows in (4096, 32768):
    c8 = c.to_device(rng.integers(0,256,size=nrows*p.ctb,dtype=np.uint8))
    co0 = c.to_device(rng.integers(1,mf.P,size=nrows,dtype=np.uint64).astype(np.uint32))
    co1 = c.to_device(rng.integers(1,mf.P,size=nrows,dtype=np.uint64).astype(np.uint32))
    for nacc in (1,2):
        for it in range(3):
            c.eval_rows(0, nrows, c8, co0, co1 if nacc==2 else None); c.sync()
            ms = c.last_kernel_ms("eval")
        gb = nrows*1471*92/1e9
        print(f"eval nrows={nrows} nacc={nacc}: {ms:.3f} ms  {gb/ms*1e3:.1f} GB/s eff  {nrows*8452.5/ms/1e6:.2f} Gblk/s", flush=True)
nrows=8192
sk = c.to_device(rng.integers(0,2**63,size=p.n*p.L,dtype=np.uint64))
msg = c.to_device(rng.integers(0,mf.P,size=nrows,dtype=np.uint64).astype(np.uint32))
err = c.to_device(rng.integers(0,2**40,size=nrows*p.L,dtype=np.uint64))
for it in range(3):
    c.encrypt_rows(0,nrows,sk,msg,err); c.sync(); ms=c.last_kernel_ms("encrypt")
print(f"encrypt nrows={nrows}: {ms:.3f} ms -> {nrows/ms*1e3:.0f} enc/s", flush=True)
n = 1<<30
buf = c.empty(n)
for it in range(3):
    c.keystream(0, n, buf); c.sync(); ms=c.last_kernel_ms("keystream")
print(f"keystream 1GiB: {ms:.3f} ms -> {n/ms/1e6:.1f} GB/s", flush=True)
