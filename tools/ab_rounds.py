"""Same-box A/B of two builds of libmfhip.so on the headline call (mfh_prove_batch, 992 statements, default instance) and on the LWE batch
(mfh_encrypt_rows, 65 536 rows), through the C ABI only -- the entry points both builds export.  dev tool.
usage: python tools/ab_rounds.py <libA.so> <libB.so> ..."""
import ctypes, os, sys, time
import numpy as np, torch

P = 0xFFFFFFFB


class CParams(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint32), ("logq", ctypes.c_uint32), ("d", ctypes.c_uint32), ("m", ctypes.c_uint32)]


def run(path):
    lib = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    h = vp()
    n, logq, d, m = 1470, 736, 1 << 15, 21845
    L, ctb = 12, 92
    assert lib.mfh_ctx_create(ctypes.byref(h), 0, ctypes.byref(CParams(n, logq, d, m))) == 0
    dev = torch.device("cuda", 0)
    ptr = lambda t: vp(t.data_ptr())
    lib.mfh_set_stream(h, vp(torch.cuda.current_stream(dev).cuda_stream))
    lib.mfh_set_seed(h, bytes((37 * i + 11) & 0xFF for i in range(40)))
    g = torch.Generator(device=dev); g.manual_seed(1)
    ssp = torch.randint(0, P, ((m + 3), d), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    sk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (n, L), dtype=torch.int64, device=dev, generator=g); sk[:, L - 1] &= (1 << 32) - 1
    rows = 2 * d + m
    err = torch.randint(-(2 ** 63), 2 ** 63 - 1, (rows, L), dtype=torch.int64, device=dev, generator=g); err[:, 8] &= (1 << 47) - 1; err[:, 9:] = 0
    crs = torch.empty(rows * ctb, dtype=torch.uint8, device=dev)
    assert lib.mfh_ssp_prepare(h, ptr(ssp)) == 0
    u32, sz = ctypes.c_uint32, ctypes.c_size_t
    lib.mfh_setup.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp]
    assert lib.mfh_setup(h, ptr(ssp), 5, 7, 11, ptr(sk), ptr(err), ptr(crs)) == 0
    nb = 992
    rng = np.random.default_rng(5)
    stride = (m + 6) // 8
    bits = rng.integers(0, 256, size=nb * stride, dtype=np.uint8).tobytes()
    dl = (ctypes.c_uint32 * nb)(*[int(x) for x in rng.integers(0, P, size=nb, dtype=np.uint64)])
    mags = rng.integers(0, 256, size=nb * 400, dtype=np.uint8).tobytes()
    signs = bytes(nb * 5)
    out = torch.empty(nb * 5 * (n + 1) * L * 8, dtype=torch.uint8, device=dev)
    lib.mfh_prove_batch.argtypes = [vp, vp, vp, u32, ctypes.c_char_p, sz, vp, ctypes.c_char_p, sz, ctypes.c_char_p, vp]
    call = lambda: lib.mfh_prove_batch(h, ptr(crs), ptr(ssp), nb, bits, stride, ctypes.cast(dl, vp), mags, 80, signs, ptr(out))
    assert call() == 0
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    digest = int(out.view(torch.int64).sum().item())
    # LWE batch
    B = 65536
    msg = torch.randint(0, P, (B,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    c8 = torch.empty(B * ctb, dtype=torch.uint8, device=dev)
    lib.mfh_encrypt_rows.argtypes = [vp, ctypes.c_uint64, sz, vp, vp, vp, vp]
    enc = lambda: lib.mfh_encrypt_rows(h, 0, B, ptr(sk), ptr(msg), ptr(err), ptr(c8))
    assert enc() == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): enc()
    torch.cuda.synchronize(); de = (time.perf_counter() - t0) / 3
    print(f"{os.path.basename(path):28s} prove_batch(992): {dt*1e3:7.2f} ms = {nb/dt:8.0f} proofs/s   encrypt(65536): {de*1e3:6.2f} ms = {B/de/1e6:5.2f} M enc/s   digest {digest & 0xffffffff:08x} / {int(c8.view(torch.int64).sum().item()) & 0xffffffff:08x}", flush=True)
    lib.mfh_ctx_destroy(h)


for pth in sys.argv[1:]:
    run(os.path.abspath(pth))
