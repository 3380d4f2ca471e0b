"""Time mfh_eval_rows_multi against mfh_eval_rows at the default instance's S region (32768 rows). dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import c_lwe_snarks_amd as mf
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes(range(40)))
nrows = int(sys.argv[1]) if len(sys.argv) > 1 else p.d
g = torch.Generator(device="cuda").manual_seed(1)
c8 = torch.randint(0, 256, (nrows * p.ctb,), dtype=torch.uint8, device="cuda", generator=g)
for nvec in (2, 30, 62):
    co = torch.randint(0, 2**32 - 6, (nvec, nrows), dtype=torch.int64, device="cuda", generator=g).to(torch.int32).view(torch.uint8)
    out = ctx.eval_rows_multi(p.ctr_s, nrows, c8, co, nvec)
    ctx.set_timing(True); ctx.timing_drain("evalmm")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ctx.eval_rows_multi(p.ctr_s, nrows, c8, co, nvec, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    cnt, ms, rows = ctx.timing_drain("evalmm"); ctx.set_timing(False)
    print(f"multi nvec={nvec:2d}: {dt*1e3:7.3f} ms/call (kernel {ms/max(cnt,1):7.3f} ms) -> {dt*1e3/nvec:6.3f} ms per vector", flush=True)
co2 = torch.randint(0, 2**32 - 6, (2, nrows), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
a, b = ctx.eval_rows(p.ctr_s, nrows, c8, co2[0].view(torch.uint8), co2[1].view(torch.uint8))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ctx.eval_rows(p.ctr_s, nrows, c8, co2[0].view(torch.uint8), co2[1].view(torch.uint8), a, b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"k_eval 2 vectors: {dt*1e3:7.3f} ms/call -> {dt*1e3/2:6.3f} ms per vector")
