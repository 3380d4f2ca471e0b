"""device-to-host copy rate into pinned memory with 1 .. 4 streams (slabs of 45 MB, as the shim drains proofs): what the tail of a typed batch call pays.  dev tool."""
import torch, time
n = 45 * 1024 * 1024
src = torch.empty(4 * n, dtype=torch.uint8, device="cuda")
dst = torch.empty(4 * n, dtype=torch.uint8).pin_memory()
s = [torch.cuda.Stream() for _ in range(4)]
def run(k, parts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        for i in range(4):          # 4 slabs of 45 MB
            for j in range(parts):  # each slab in `parts` pieces on `k` streams
                lo = i * n + j * (n // parts); hi = lo + n // parts
                with torch.cuda.stream(s[j % k]):
                    dst[lo:hi].copy_(src[lo:hi], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"streams {k} parts {parts}: {dt*1e3:.2f} ms for 180 MB = {4*n/dt/1e9:.1f} GB/s", flush=True)
for k, parts in [(1, 1), (2, 2), (4, 4), (1, 1), (2, 2), (3, 3)]:
    run(k, parts)
