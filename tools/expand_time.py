"""mfh_crs_expand_mm at the default instance: barrier-free writer (path 0) against the LDS-tile writer (path 1). dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import c_lwe_snarks_amd as mf
p = mf.Params(logq=int(os.environ.get("LOGQ", "736")))
ctx = mf.Context(p, 0)
ctx.set_seed(bytes(range(40)))
rng = np.random.default_rng(1)
d_crs = ctx.to_device(rng.integers(0, 256, size=(2 * p.d + p.m) * p.ctb, dtype=np.uint8))
img = ctx.crs_expand_mm(d_crs)
rows = 2 * p.d + p.m
for path in (1, 0, 1, 0):
    ctx.set_expand_path(path)
    ctx.crs_expand_mm(d_crs, out=img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ctx.crs_expand_mm(d_crs, out=img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"logq {p.logq} path {path}: {dt*1e3:7.3f} ms  {rows * p.ctr_ct / 16 / dt / 1e9:6.1f} Gblock/s (minimal blocks)  write {img.numel() / dt / 1e9:6.0f} GB/s", flush=True)
