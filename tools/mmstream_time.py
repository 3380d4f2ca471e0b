"""Time k_mmstream (mfh_eval_rows_multi from a registered fragment image) alone on one stream: S region of the default instance, 62 vectors. dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import c_lwe_snarks_amd as mf
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes(range(40)))
g = torch.Generator(device="cuda").manual_seed(1)
crs = torch.randint(0, 256, ((2 * p.d + p.m) * p.ctb,), dtype=torch.uint8, device="cuda", generator=g)
img = ctx.crs_expand_mm(crs)
ctx.set_resident_mm(img)
nvec = 62
co = torch.randint(0, 2**32 - 6, (nvec, p.d), dtype=torch.int64, device="cuda", generator=g).to(torch.int32).view(torch.uint8)
out = ctx.eval_rows_multi(p.ctr_s, p.d, crs, co, nvec)
ctx.set_timing(True); ctx.timing_drain("evalmm_resident")
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N): ctx.eval_rows_multi(p.ctr_s, p.d, crs, co, nvec, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
cnt, ms, rows = ctx.timing_drain("evalmm_resident"); ctx.set_timing(False)
k = ms / max(cnt, 1)
print(f"k_mmstream alone: {k:7.3f} ms/launch ({dt*1e3:7.3f} ms/call) -> {736*11*16*p.d/ (k*1e-3)/1e12:5.2f} TB/s of A fragments", flush=True)
