"""can the CRS expansion (k_expand_mm: AES, LDS / VALU bound) run UNDER the streaming GEMM (k_mmstream: matrix cores / HBM)?
Two contexts on one GPU: A proves 992 statements from a registered image (no expansion inside the call: only chains, b_w and the S / AS rounds), B expands the
CRS into a second image on its own stream -- unrestricted, or on a stream masked to N CUs (hipExtStreamCreateWithCUMask).  Alone, then both at once.
If the GPU work were complementary the wall time of both would approach max(A, B); if every kernel fills the CUs it gets, it approaches A + B.  dev tool."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import c_lwe_snarks_amd as mf  # noqa: E402

p = mf.DEFAULT
seed = bytes((37 * i + 11) & 0xFF for i in range(40))
A = mf.Context(p, 0)
A.set_seed(seed)
inst = bench.build_instance(mf, A, torch, p, 20260101)
A.ssp_prepare(inst["d_ssp"])
d_crs = A.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
image = A.crs_expand_mm(d_crs)
A.set_resident_mm(image)
nb = 992
rng = np.random.default_rng(1)
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
signs = [bytes(5)] * nb
bits = [inst["bits"]] * nb
out = A.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
torch.cuda.synchronize()

hip = ctypes.CDLL("libamdhip64.so")
sB = torch.cuda.Stream()
with torch.cuda.stream(sB):
    B = mf.Context(p, 0)
B.set_seed(seed)
image2 = torch.empty_like(image)


def masked_stream(ncu):
    """a stream restricted to the first ncu / 8 CUs of every XCD (mask bit i = CU i in the driver's interleaved numbering: spread evenly)"""
    words = (ctypes.c_uint32 * 8)()
    total = 256
    step = total / ncu
    for k in range(ncu):
        i = int(k * step)
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s


def time_both(run_a, run_b, n=3):
    best = None
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if run_a:
            A.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs, out=out)
        if run_b:
            B.crs_expand_mm(d_crs, out=image2)
        A.sync()
        B.sync()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


ta = time_both(True, False)
print(f"A alone (prove_batch, 992 statements, image registered): {ta:7.2f} ms", flush=True)
for ncu in (256, 128, 64, 32):
    if ncu == 256:
        B._chk(B.lib.mfh_set_stream(B._h, ctypes.c_void_p(sB.cuda_stream)))
    else:
        ms = masked_stream(ncu)
        B._chk(B.lib.mfh_set_stream(B._h, ms))
    tb = time_both(False, True)
    tab = time_both(True, True)
    print(f"B = crs_expand_mm on {ncu:3d} CUs: alone {tb:6.2f} ms; A and B at once {tab:7.2f} ms (A + B = {ta + tb:7.2f}, max = {max(ta, tb):7.2f}): "
          f"{(ta + tb - tab):+.2f} ms against running them one after the other", flush=True)
same = bool(torch.equal(image, image2))
print("second image identical:", same)
sys.exit(0 if same else 1)
