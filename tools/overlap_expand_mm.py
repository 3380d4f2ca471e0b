"""Does the CRS expansion (AES on the vector ALUs and LDS) overlap with the streaming GEMM (matrix cores, power-limited) when both are on the GPU at once?  Two
contexts on two streams: one proves batches from a resident image (k_mmstream_p + chains, no expansion), the other expands the same CRS into a second image over and
over (k_expand_mm).  Times each alone and both together.  If together = the sum, pipelining a call's expansion under its own streaming launches (by row chunks) cannot
pay; if together = the longer of the two, it would be worth up to the expansion's 10 ms per call.  dev tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf
p = mf.DEFAULT
SEED = bytes((37 * i + 11) & 0xFF for i in range(40))
c1 = mf.Context(p, 0)
c1.set_seed(SEED)
inst = bench.build_instance(mf, c1, torch, p, 20260101)
c1.ssp_prepare(inst["d_ssp"])
d_crs = c1.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
image = c1.crs_expand_mm(d_crs)
c1.set_resident_mm(image)
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    c2 = mf.Context(p, 0)
    c2.set_seed(SEED)
    img2 = c2.empty(image.numel())
rng = np.random.default_rng(5)
nb = 1020
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
signs = [bytes(5)] * nb
bits = [inst["bits"]] * nb
out = c1.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
with torch.cuda.stream(s2):
    c2.crs_expand_mm(d_crs, out=img2)
torch.cuda.synchronize()
assert torch.equal(img2, image)
NB, NE = 4, 24  # 4 calls of 1020 statements (~ 250 ms) and 24 expansions (~ 245 ms)


def mm():
    for _ in range(NB):
        c1.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs, out=out)


def ex():
    with torch.cuda.stream(s2):
        for _ in range(NE):
            c2.crs_expand_mm(d_crs, out=img2)


def timed(*fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for rnd in range(3):
    a, b = timed(mm), timed(ex)
    both = timed(ex, mm)
    print(f"{NB} resident calls alone {a:7.1f} ms   {NE} expansions alone {b:7.1f} ms   sum {a + b:7.1f}   both at once {both:7.1f} ms   hidden {a + b - both:6.1f} ms "
          f"({(a + b - both) / min(a, b) * 100:.0f} % of the shorter one)", flush=True)
