"""regev_decrypt (src/lwe.c:105-111) as a batch: mfh_decrypt over B FULL ciphertexts resident in HBM ((n+1) x 96 B each: 141 KB), built from
real encryptions (a-vectors sampled from the stream, b from mfh_encrypt_rows) so that the messages can be checked.  dev tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import c_lwe_snarks_amd as mf  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
g = torch.Generator(device=ctx.device)
g.manual_seed(7)
sk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (p.n, p.L), dtype=torch.int64, device=ctx.device, generator=g)
sk[:, p.L - 1] &= (1 << (p.logq - 64 * (p.L - 1))) - 1
err = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, p.L), dtype=torch.int64, device=ctx.device, generator=g)
err[:, 8] &= (1 << (559 - 512)) - 1
err[:, 9:] = 0
msg = torch.randint(0, mf.P, (B,), dtype=torch.int64, device=ctx.device, generator=g).to(torch.int32)
sk8, err8 = sk.view(torch.uint8).reshape(-1), err.view(torch.uint8).reshape(-1)
c8 = ctx.encrypt_rows(0, B, sk8, msg.view(torch.uint8), err8)  # B x 92 bytes
cts = torch.zeros((B, p.n + 1, p.L), dtype=torch.int64, device=ctx.device)
step = 4096
for r0 in range(0, B, step):
    r1 = min(B, r0 + step)
    a = ctx.sample_rows(r0 * p.ctr_ct, r1 - r0).view(torch.int64).view(r1 - r0, p.n, p.L)
    cts[r0:r1, : p.n] = a
bpad = torch.zeros((B, p.L * 8), dtype=torch.uint8, device=ctx.device)
bpad[:, : p.ctb] = c8.view(B, p.ctb)
cts[:, p.n] = bpad.view(torch.int64)
del bpad
flat = cts.view(torch.uint8).reshape(-1)
torch.cuda.synchronize()
gb = B * (p.n + 1) * p.L * 8 / 1e9
allok = True
for path, name in ((1, "k_decrypt (VALU)"), (2, "k_decrypt_mm (matrix cores)")):
    ctx.set_decrypt_path(path)
    out = ctx.decrypt(sk8, flat, B)
    torch.cuda.synchronize()
    ok = bool(torch.equal(out.view(torch.int32), msg))
    allok = allok and ok
    ctx.set_timing(True)
    ctx.timing_drain("decrypt")
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ctx.decrypt(sk8, flat, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ctx.set_timing(False)
    kn, kms, _ = ctx.timing_drain("decrypt")
    print(f"mfh_decrypt {name}: {B} full ciphertexts ({gb:.2f} GB): {dt * 1e3:.3f} ms per call = {B / dt / 1e6:.2f} M dec/s = {gb / dt / 1e3:.2f} TB/s of ciphertext "
          f"reads; kernel {kms / max(kn, 1):.3f} ms; messages correct: {ok}", flush=True)
ctx.set_decrypt_path(0)
out = ctx.decrypt_rows(0, B, sk8, c8)
torch.cuda.synchronize()
ok = bool(torch.equal(out.view(torch.int32), msg))
allok = allok and ok
t0 = time.perf_counter()
for _ in range(3):
    ctx.decrypt_rows(0, B, sk8, c8)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"mfh_decrypt_rows (seed-compressed, a regenerated): {dt * 1e3:.3f} ms per call = {B / dt / 1e6:.2f} M dec/s = {B * (p.ctr_ct / 16) / dt / 1e9:.1f} Gblock/s of AES; "
      f"messages correct: {ok}", flush=True)
sys.exit(0 if allok else 1)
