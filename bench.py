#!/usr/bin/env python3
"""bench.py -- proofs/s of the reference's benchmark_snark prover (src/benchmark_snark.c:70-74) on MI355X.

Default (--mode batch): one "step" = ONE mfh_prove_batch call over --batch (1020 = 4 super-groups of 255) statements per GPU on the NDEBUG default SSP
instance (D = 2^15 constraints, M = 21845 wires, N = 1470, log q = 736; src/lwe.h:14-31), starting from the COMPRESSED CRS exactly as
the reference's prover() does (src/snark.c:117-190): the call expands all (2D+M) x 135 240 B of AES-256-CTR keystream on the CU into
a transient image inside the timed region, then streams it -- two passes over a region's image per 255 proofs -- with the multiply-accumulate on the matrix cores;
witness polynomials, h = (v^2-1)/t, the five evaluations and the smudging of every statement are inside the timed region.  Inputs
(CRS bytes, SSP, witnesses) are resident in HBM when the clock starts.  Nothing is cached between steps except per-circuit constants
(AES tables, the power-series inverse of rev(t), the SSP's matrix-core image -- all functions of the SSP only, like the SSP itself).
Every proof is bit-identical to the single-proof prover()'s; the batch figure is a prover-SERVICE throughput and is always printed
beside `single_proof` (one prover() call per step: what the reference's benchmark_snark times).

N > 1 (one process per GPU, torch.distributed, backend nccl = RCCL over xGMI), default workload:
  * headline `value`: every rank proves its own --batch statements per step, no data-path collective ("scaling": "weak");
  * `single_proof`: ONE proof computed cooperatively, CRS rows sharded over the ranks, two lane all-reduces ("strong");
  * `row_sharded_batch`: --batch statements per step for the WHOLE job with the CRS rows sharded over the ranks: chain per statement
    slab, all-to-all of the coefficient row slices, one reduce-scatter of uint64 lanes ("strong"; the configuration BASELINE configs
    3/4 name).  With --workload config4/config5 and --mode batch this leg IS the headline when N > 1 (the 363 GB image only fits
    sharded: 45 GB per GPU on 8).

Prints ONE JSON line on rank 0 (driver contract), with "roofline" and "cpu_baseline" objects.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

ROW_BYTES = 1471 * 92  # algorithmic bytes of one expanded ciphertext row (keystream + b); x2 at logq 1472
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 = 2x BF16 per clock, dense BF16 ~2.5 PF -> ~5 POPS (no sparsity)

# The ruler of the AES-bound kernels (round 6; it replaces the LDS-pipe figure of 91 Gblock/s, which round 5's own measurement disproved).  The T-table loop of
# csrc/aes_dev.hpp is bound by the address formation of its lookups (three v_perm_b32 per column), not by the LDS pipe: a timing-only build of the same loop whose lookup
# addresses are all ONE full-rate v_bitop3 -- every ds_read, every combine, the same dependences -- is the fastest this table layout can run on the chip, and it was
# measured (tools/aes3_ubench.hip -DCHEAP_ADDR, profiles/r05_aes_address_bound.txt): 108.0 Gblock/s at 4 waves per SIMD, 113.5 at 8.  (The microbenchmark does all 224
# lookups of a block, the kernels 201 with the counter-mode shortcut; the ceiling is NOT scaled up for that -- the shortcut's span constants cost lookups of their own.)
AES_CEILING_GBLOCKS = {4: 108.0, 8: 113.5}
AES_KERNEL_WAVES_PER_SIMD = {"k_eval": 4, "k_evalmm16": 4, "k_expand_mm": 8, "k_encrypt_mm": 8}


def aes_ceiling(kernel, gblk):
    """`gblk` Gblock/s of AES-256 achieved by `kernel`, against the cheap-address ceiling at that kernel's occupancy"""
    w = AES_KERNEL_WAVES_PER_SIMD[kernel]
    peak = AES_CEILING_GBLOCKS[w]
    return {"kernel": kernel, "achieved_gblocks_per_s": gblk, "peak_gblocks_per_s": peak, "frac": (gblk / peak) if gblk else None, "waves_per_simd": w,
            "what": "the table AES loop with every lookup address formed by one full-rate instruction (timing-only build), measured on the MI355X",
            "source": "profiles/r05_aes_address_bound.txt (tools/aes3_ubench.hip, cheap-addr rows)"}

CONC_NOTE = ("avg_launch_ms is the start-to-end time of one launch (HIP events on the launch stream; what rocprofv3 --stats reports), "
             "busy_ms_per_launch the union of all launches' spans / launches: the two agree when launches of this kernel do not overlap; "
             "achieved = algorithmic bytes per launch / avg_launch_ms")


def drop_in_leg():
    """host/bench_snark (C, reference function names and types over libmfuoco_gpu.so) as a child process: seconds per call INCLUDING PCIe and the mpz_t conversions.
    Returns a dict for the JSON line, or {"error": ...}: this leg never takes the line with it."""
    import re
    import subprocess

    exe = os.path.join(ROOT, "c-lwe-snarks_amd", "host", "bench_snark")
    if not os.path.exists(exe):
        return {"error": "c-lwe-snarks_amd/host/bench_snark has not been built (make -C c-lwe-snarks_amd shim)"}
    nb, nenc = 1020, 65536
    try:
        r = subprocess.run([exe, "1", "300", str(nb), "0", str(nenc)], capture_output=True, text=True, timeout=120)
    except Exception as e:  # (timeout, exec failure)
        return {"error": f"{type(e).__name__}: {e}"}
    if r.returncode != 0:
        return {"error": f"bench_snark exited with {r.returncode}: {r.stderr[-300:]}"}
    vals = {}
    for ln in r.stdout.splitlines():
        m = re.match(r"([a-z_]+)\t([0-9.]+)", ln)
        if m:
            vals.setdefault(m.group(1), []).append(float(m.group(2)))
    try:
        warm = min(vals["prover_batch"][1:])
        encb = min(vals["encryption_batch"][1:])
        return {
            "what": "reference function names and types (proof_t, crs_t, mpz_t, ...) through libmfuoco_gpu.so, PCIe and mpz_t conversion INCLUDED; the C driver host/bench_snark in a child process",
            "all_proofs_verified_and_all_decryptions_correct": "all proofs verified, all decryptions correct" in r.stderr,
            "setup_s": vals["setup"][0],
            "prover_first_call_s": vals["prover"][0],
            "verifier_s": vals["verifier"][0],
            "prover_batch": {"statements_per_call": nb, "cold_s": vals["prover_batch"][0], "warm_s": warm, "warm_proofs_per_s": nb / warm,
                             "note": "warm: the expanded CRS image kept by the shim is streamed; the call drains super-group k (PCIe + mpz_t) under the kernels of k + 1"},
            "verifier_batch_s": vals["verifier_batch"][0],
            "regev_encrypt_single_s": vals["encryption"][0],
            "regev_decrypt_single_s": vals["decryption"][0],
            "encrypt_batch": {"messages_per_call": nenc, "first_call_s": vals["encryption_batch"][0], "warm_s": encb, "enc_per_s": nenc / encb,
                              "note": "mfuoco_encrypt_batch: OS entropy, PCIe and the export included"},
            "decrypt_rows_batch_dec_per_s": nenc / vals["decryption_rows_batch"][0],
        }
    except Exception as e:  # (a line missing from the driver's output)
        return {"error": f"could not parse bench_snark's output: {type(e).__name__}: {e}", "stdout_tail": r.stdout[-400:]}


def build_instance(mf, ctx, torch, p, seed_int):
    """Synthetic, VALID default-size instance, deterministic in seed_int (identical on every rank)."""
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed_int)
    P = mf.P
    # SSP: (m+3) x d uint32 coefficients in [0, p)
    d_ssp = ctx.empty((p.m + 3) * p.d * 4)
    ssp32 = d_ssp.view(torch.int32).view(p.m + 3, p.d)
    step = 2048
    for r0 in range(0, p.m + 3, step):
        r1 = min(p.m + 3, r0 + step)
        x = torch.randint(0, P, (r1 - r0, p.d), dtype=torch.int64, device=dev, generator=g)
        ssp32[r0:r1] = x.to(torch.int32)  # keeps the low 32 bits
    ssp32[p.m + 1:] = 0  # slots m+1, m+2 are unused padding in the reference layout (src/ssp.h:6)
    rng = np.random.default_rng(seed_int)
    bits = bytearray(rng.integers(0, 256, size=(p.m + 7) // 8, dtype=np.uint8).tobytes())
    # t = v_0 + sum_{bit} v_i - 1  (random_ssp, src/ssp.c:59-71) so that t | v^2 - 1
    ssp32[0] = 0
    acc = ctx.witness_poly(d_ssp, bytes(bits), 0).view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    v0 = ssp32[1].to(torch.int64) & 0xFFFFFFFF
    t = (acc + v0) % P
    t[0] = (t[0] + (P - 1)) % P  # minus the constant polynomial 1 (nmod_poly_sub(t, t, one), src/ssp.c:71)
    ssp32[0] = t.to(torch.int32)
    alpha, beta, s = (int(x) for x in rng.integers(1, P, size=3, dtype=np.uint64))
    # secret key: n uniform 736-bit values; errors: 559-bit (src/lwe.c:30-34,60-63)
    sk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (p.n, p.L), dtype=torch.int64, device=dev, generator=g)
    top_bits = p.logq - 64 * (p.L - 1)
    if top_bits < 64:
        sk[:, p.L - 1] &= (1 << top_bits) - 1
    rows = 2 * p.d + p.m
    err = torch.randint(-(2 ** 63), 2 ** 63 - 1, (rows, p.L), dtype=torch.int64, device=dev, generator=g)
    err[:, 8] &= (1 << (559 - 512)) - 1
    err[:, 9:] = 0
    return dict(d_ssp=d_ssp, bits=bytes(bits), alpha=alpha, beta=beta, s=s, sk=sk.view(torch.uint8).reshape(-1),
                err=err.view(torch.uint8).reshape(-1), t_host=t.cpu().numpy(), v0_host=v0.cpu().numpy())


def build_prg_instance(mf, ctx, torch, p, seed_int):
    """BASELINE configs 4/5: generator-defined SSP (csrc/ssp_prg.hpp), only t is stored; d_ssp = None selects it."""
    rng = np.random.default_rng(seed_int)
    bits = rng.bytes((p.m + 7) // 8)
    prg_seed = 0x5EED5EED00000000 | seed_int
    d_t = ctx.ssp_prg_make_t(prg_seed, bits)
    ctx.ssp_set_prg(prg_seed, d_t)
    alpha, beta, s = (int(x) for x in rng.integers(1, mf.P, size=3, dtype=np.uint64))
    g = torch.Generator(device=ctx.device)
    g.manual_seed(seed_int)
    sk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (p.n, p.L), dtype=torch.int64, device=ctx.device, generator=g)
    top_bits = p.logq - 64 * (p.L - 1)
    if top_bits < 64:
        sk[:, p.L - 1] &= (1 << top_bits) - 1
    rows = 2 * p.d + p.m
    err = torch.randint(-(2 ** 63), 2 ** 63 - 1, (rows, p.L), dtype=torch.int64, device=ctx.device, generator=g)
    err[:, 8] &= (1 << (559 - 512)) - 1
    err[:, 9:] = 0
    return dict(d_ssp=None, bits=bits, alpha=alpha, beta=beta, s=s, sk=sk.view(torch.uint8).reshape(-1), err=err.view(torch.uint8).reshape(-1))


def horner(coeffs, x, P):
    r = 0
    for c in reversed(coeffs.tolist()):
        r = (r * x + c) % P
    return r


def verify_on_gpu(mf, ctx, inst, proof):
    """verifier() (src/snark.c:192-250) with the five decryptions on the GPU; returns True iff it accepts."""
    P = mf.P
    dec = [int(x) for x in ctx.to_host(ctx.decrypt(inst["sk"], proof, 5), np.uint32)]
    h_s, hath_s, hatv_s, w_s, b_s = dec
    t_s = horner(inst["t_host"], inst["s"], P)
    v_s = (horner(inst["v0_host"], inst["s"], P) + w_s) % P
    ok = (h_s * inst["alpha"] % P == hath_s and v_s * inst["alpha"] % P == hatv_s and (v_s * v_s - 1 - h_s * t_s) % P == 0
          and w_s * inst["beta"] % P == b_s)
    return ok


def usable_cores(cap=32):
    """cores this process may really use: the affinity mask, cut down to the cgroup CPU quota (a one-GPU box shares a big host)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except Exception:
            continue
    return max(1, min(n, cap))


def _cpu_rows_worker(args):
    """one process of the N-process CPU aggregate: `rows` reference-faithful row touches of the oracle (rows are independent)"""
    d, m, logq, seed, rows = args
    import oracle_lib as ol

    import c_lwe_snarks_amd as mf

    o = ol.Oracle()
    p = mf.Params(d=d, m=m, logq=logq)
    o.bench_eval_rows(p, seed, 64)
    t0 = time.perf_counter()
    o.bench_eval_rows(p, seed, rows)
    return time.perf_counter() - t0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and wait for them.  This parent has
    not touched the GPU (no torch import, no HIP call) and never does: every child is a new interpreter that reads RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* like a rank started by `python -m torch.distributed.run`.  Rank 0 prints the JSON line on the inherited
    stdout.  Exit code: 0 only if every rank exited 0; when one fails the others are ended (by PID) and its code is returned."""
    import socket
    import subprocess

    with socket.socket() as sk_:
        sk_.bind(("127.0.0.1", 0))
        port = sk_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MFUOCO_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    deadline = time.time() + float(os.environ.get("MFUOCO_BENCH_DEADLINE_S", "3000"))  # a hung rank must not hold the GPUs for ever
    while live:
        if time.time() > deadline:
            print(f"[bench] {len(live)} rank process(es) still running at the deadline: ending them", file=sys.stderr, flush=True)
            for other in live:
                other.terminate()
            deadline = float("inf")
            rc = rc or 124
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank process {procs.index(pr)} exited with {code}: ending the other ranks", file=sys.stderr, flush=True)
                for other in live:
                    other.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the PCIe / mpz_t-inclusive leg (the reference-typed C shim driven by host/bench_snark as a child process)")
    ap.add_argument("--no-resident", action="store_true", help="skip the resident-CRS regime")
    ap.add_argument("--sharding", choices=["rows", "proofs"], default="rows",
                    help="N > 1: 'rows' = every proof is computed cooperatively, CRS rows sharded over the ranks + all-reduce (strong scaling, "
                         "default); 'proofs' = every rank proves its own statements, no collective (weak scaling)")
    ap.add_argument("--mode", choices=["batch", "single"], default=None,
                    help="batch (default at the default workload): a step = --batch statements per GPU proved by mfh_prove_batch (CRS regions "
                         "expanded once per group of 31, MAC on the matrix cores), ranks take disjoint statements, no collective; "
                         "single: a step = one prover() call, CRS rows sharded over the ranks + lane all-reduces")
    ap.add_argument("--batch", type=int, default=1020, help="statements per GPU per step in batch mode (4 super-groups of 255)")
    ap.add_argument("--invalid-every", type=int, default=0,
                    help="batch mode: every K-th statement of the headline batch carries a random witness instead of the satisfying one (0 = none, the reference's "
                         "benchmark_snark case; rounds 1 - 5 ran 2).  The mixed batch is always run beside the headline as its own leg")
    ap.add_argument("--no-overlap", action="store_true", help="run the prover on one stream (A/B check of the side-stream overlap)")
    ap.add_argument("--groups-per-launch", type=int, default=8, help="batch mode: groups of 63 / 64 coefficient vectors per streaming launch and region (8 = a super-group's S and AS regions in one launch; 4 = rounds 1-3)")
    ap.add_argument("--no-merge", action="store_true", help="batch mode: S and AS groups of a round as two launches on two streams (A/B check)")
    ap.add_argument("--mm-width", type=int, default=32, help="batch mode: workgroups per XCD of the persistent S / AS launch (32 = every CU; A/B knob of the round-5 power experiment)")
    ap.add_argument("--early-chain", action="store_true", help="batch mode: chain of super-group k + 1 and epilogues of k queued beside the streaming launches (with --mm-width < 32)")
    ap.add_argument("--sharded-batch", type=int, default=None,
                    help="N > 1: statements per step of the row-sharded batch leg for the whole job (default: --batch at the default workload, 255 for config4/5)")
    ap.add_argument("--resident-gb", type=float, default=200.0, help="HBM budget for the resident CRS image per GPU")
    ap.add_argument("--workload", choices=["default", "config4", "config5"], default="default",
                    help="default = benchmark_snark NDEBUG instance (the driver's workload); config4/config5 = BASELINE's 2^20-constraint "
                         "instance (M = 699050, generator-defined SSP) at logq 736 / 1472")
    ap.add_argument("--cpu-rows", type=int, default=20000, help="rows of the CPU baseline sample (~0.65 ms each on one core)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)  # no launcher: be one (before anything touches the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # the line would claim a rank count the job does not have
        if rank == 0:
            print(f"[bench] error: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        return 2

    # stdout carries the ONE JSON line and nothing else: whatever the libraries print there (gloo's connection messages, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    import c_lwe_snarks_amd as mf
    from c_lwe_snarks_amd import dist as mfdist

    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MFUOCO_DIST_BACKEND=gloo + MFUOCO_SHARE_GPU=1 rehearse the multi-rank path on a one-GPU box (all ranks on cuda:0)
        backend = os.environ.get("MFUOCO_DIST_BACKEND", "nccl")
        if os.environ.get("MFUOCO_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist = None
        torch.cuda.set_device(0)
    # control-plane group on the host (gloo): ranks agree there on what to do after a failed collective -- a rank that is alone with its exception must not
    # change mode while the others sit inside an RCCL call; it finds nobody in the agreement, times out and EXITS non-zero, which ends the job
    ctl_group = None
    if dist is not None:
        import datetime

        ctl_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("MFUOCO_BENCH_AGREE_S", "90"))))

    def ctl_reduce(value, op):
        """MAX / MIN of a scalar over the ranks on the host group: timing and acceptance flags never depend on the data-path backend"""
        if dist is None:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=op, group=ctl_group)
        return float(t.item())

    if dist is not None and dist.get_world_size() != args.gpus:
        print(f"[bench] error: {dist.get_world_size()} ranks joined the process group, --gpus {args.gpus}", file=sys.stderr, flush=True)
        return 2

    big = args.workload != "default"
    p = mf.DEFAULT if not big else mf.Params(logq=736 if args.workload == "config4" else 1472, d=1 << 20, m=699050)
    mode = args.mode or ("batch" if not big else "single")
    run_single = True  # one prover() per step is always reported: on N > 1 ranks row-sharded with its two lane all-reduces
    single_steps = args.steps if mode == "single" else min(args.steps, 10 if not big else 3)
    backend = os.environ.get("MFUOCO_DIST_BACKEND", "nccl") if world > 1 else None
    ctx = mf.Context(p, local_rank)
    if args.no_overlap:
        ctx.set_overlap(False)
    merge_regions = not args.no_merge
    ctx.set_batch_launch(args.groups_per_launch, merge_regions)
    if args.mm_width != 32 or args.early_chain:
        ctx.set_mm_width(args.mm_width, args.early_chain)
    seed = bytes((37 * i + 11) & 0xFF for i in range(40))
    ctx.set_seed(seed)
    if not big:
        inst = build_instance(mf, ctx, torch, p, 20260101)
    else:
        inst = build_prg_instance(mf, ctx, torch, p, 20260101)
        # the expanded CRS of these configs is 362 / 724 GB: a rank's share fits from 2-4 GPUs on (45 GB/GPU on 8);
        # on one GPU a prefix of --resident-gb stays resident and the rest of the rows is regenerated
        share_bytes = int(ctx.lib.mfh_resident_share_rows(ctx._h, rank, world)) * ctx.resident_row_bytes()
        if world > 1 and share_bytes > args.resident_gb * 1e9:
            args.no_resident = True
    ctx.ssp_prepare(inst["d_ssp"])  # per-circuit constant: rev(t)^-1 (depends on the SSP only)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=ctl_group)
        torch.cuda.synchronize()

    # ---- setup(): 2D+M encryptions -> compressed CRS (also the "LWE encryptions/s" figure at scale)
    rows_crs = 2 * p.d + p.m
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    barrier()
    t0 = time.perf_counter()
    ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"], out=d_crs)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0

    rng = np.random.default_rng(99)
    delta = int(rng.integers(0, mf.P, dtype=np.uint64))
    mags = rng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes()
    signs = bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist())
    bufs = {}

    by_rows = args.sharding == "rows"
    # batch mode on N > 1 ranks (what the driver's scaling runs are): the headline needs no collective, so every leg that does -- the row-sharded single
    # proof and the row-sharded batch prover -- runs LAST, under a watchdog (collective_legs below): a backend that hangs costs those legs, not the line
    defer_rows = by_rows and world > 1 and mode == "batch" and not big
    if defer_rows:
        by_rows = False
    eff_rank, eff_world = (rank, world) if by_rows else (0, 1)

    def step():
        return mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs, eff_rank, eff_world, bufs=bufs)

    proof = None
    elapsed = float("nan")
    n2 = n1 = rows2 = rows1 = 0
    ms2 = ms1 = 0.0
    accepted = True
    single_err = None
    if run_single and world > 1 and by_rows:
        # the first collectives of the job.  A backend that refuses them (an exception, the same on every rank) must not take the headline with it:
        # the leg is then reported with its error and one prover() per rank instead
        try:
            proof = step()
            torch.cuda.synchronize()
        except Exception as e:
            single_err = f"{type(e).__name__}: {e}"
        # the outcome is agreed on the host group (MIN of an ok flag): only when EVERY rank arrives -- all of them out of the collective, by success or
        # by exception -- does the job go on, all in the same mode; a rank that waits here alone (the others hang in the collective) times out and exits
        try:
            flag = torch.tensor([0 if single_err else 1], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl_group)
        except Exception as e:
            print(f"[bench] rank {rank}: no agreement after the first row-sharded step ({single_err or 'this rank succeeded'}; {type(e).__name__}): exiting",
                  file=sys.stderr, flush=True)
            os._exit(3)
        if int(flag.item()) == 0:
            single_err = single_err or "another rank's first row-sharded step failed"
            by_rows = False
            eff_rank, eff_world = 0, 1
            bufs.clear()
    if run_single:
        for _ in range(args.warmup):
            proof = step()
        ctx.set_timing(True)
        ctx.timing_drain("eval")
        barrier()
        t0 = time.perf_counter()
        for _ in range(single_steps):
            proof = step()
        barrier()
        elapsed = time.perf_counter() - t0
        ctx.set_timing(False)
        n2, ms2, rows2 = ctx.timing_drain("eval2")
        n1, ms1, rows1 = ctx.timing_drain("eval1")
        if dist is not None:
            elapsed = ctl_reduce(elapsed, dist.ReduceOp.MAX)

        if rank != 0:
            accepted = True
        elif big:
            accepted = bool(int(ctx.to_host(ctx.verify(None, inst["alpha"], inst["beta"], inst["s"], inst["sk"], proof, 1))[0]))
        else:
            accepted = verify_on_gpu(mf, ctx, inst, proof) and bool(
                int(ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], proof, 1))[0]))

    # ---- second regime: the expanded CRS resident in HBM (11.3 GB), streamed at HBM speed
    resident = None
    if not args.no_resident and run_single:
        share_rows = int(ctx.lib.mfh_resident_share_rows(ctx._h, eff_rank, eff_world))
        budget_rows = int(args.resident_gb * 1e9) // ctx.resident_row_bytes()
        partial_res = eff_world == 1 and share_rows > budget_rows  # single GPU and the image does not fit: keep a prefix resident
        if partial_res:
            share_rows = budget_rows
        image = ctx.empty(share_rows * ctx.resident_row_bytes())  # this rank's shares only (= the whole CRS when world == 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if partial_res:
            ctx.crs_expand(0, share_rows, d_crs, out=image)
        else:
            ctx.crs_expand_share(d_crs, eff_rank, eff_world, out=image)
        torch.cuda.synchronize()
        expand_s = time.perf_counter() - t1
        if partial_res:
            ctx.set_resident_prefix(image, share_rows)
        else:
            ctx.set_resident_share(image, eff_rank, eff_world)
        for _ in range(args.warmup):
            proof_r = step()
        ctx.set_timing(True)
        ctx.timing_drain("mac2")
        ctx.timing_drain("mac1")
        barrier()
        t1 = time.perf_counter()
        for _ in range(single_steps):
            proof_r = step()
        barrier()
        el_r = time.perf_counter() - t1
        ctx.set_timing(False)
        m2n, m2ms, m2rows = ctx.timing_drain("mac2")
        m1n, m1ms, m1rows = ctx.timing_drain("mac1")
        if dist is not None:
            el_r = ctl_reduce(el_r, dist.ReduceOp.MAX)
        same = bool(torch.equal(proof_r, proof))
        ctx.set_resident_share(None, 0, 1)
        rb = ctx.resident_row_bytes()
        traffic_r = None
        tf2 = os.path.join(ROOT, "profiles", "traffic_mac2.json")
        if os.path.exists(tf2):
            try:
                traffic_r = json.load(open(tf2)).get("hbm_bytes_per_launch")
            except Exception:
                traffic_r = None
        avg = m2ms / max(m2n, 1)
        lr = m2rows / max(m2n, 1)
        resident = {"value": single_steps / el_r * (1 if by_rows else world), "unit": "proofs/s", "ms_per_step": el_r / single_steps * 1e3, "proof_identical_to_regenerated": same,
                    "crs_expand_s": expand_s, "image_bytes_per_rank": share_rows * rb, "resident_rows": share_rows, "partially_resident": partial_res,
                    "roofline": {"bound": "hbm", "kernel": f"k_mac_resident<{p.logq},2> (streaming 2x MAC over the expanded S / AS rows)",
                                 "achieved": lr * (p.n + 1) * p.ctb / (avg * 1e-3) / 1e9 if m2n else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (lr * (p.n + 1) * p.ctb / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if m2n else None,
                                 "traffic": traffic_r, "bytes_read_per_row": rb, "read_gbs": lr * rb / (avg * 1e-3) / 1e9 if m2n else None,
                                 "launches": m2n, "avg_launch_ms": avg, "rows_per_launch": lr},
                    "mac1": {"launches": m1n, "avg_launch_ms": m1ms / max(m1n, 1), "rows_per_launch": m1rows / max(m1n, 1)}}
        del image

    row_bytes_b = (p.n + 1) * p.ctb
    tile_bytes_per_row = (736 * 11 if p.logq == 736 else 1471 * 12) * 16  # row tiles x 16 byte positions (88 of each value's 92 bytes at 736)

    def traffic_of(name):
        tf_ = os.path.join(ROOT, "profiles", name)
        try:
            return json.load(open(tf_)).get("hbm_bytes_per_launch") if os.path.exists(tf_) else None
        except Exception:
            return None

    def clock_note():
        """effective clock under k_mmstream from the committed PMC summary (GRBM_GUI_ACTIVE / 8 XCDs / the kernel's duration in the same profile), not a constant"""
        try:
            tag = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_by_kernel.json"))[-1]
            pm = json.load(open(os.path.join(ROOT, "profiles", tag)))["kernels"]
            kk = [k for k in pm if "k_mmstream_p(" in k or "k_mmstream(" in k]
            stats = os.path.join(ROOT, "profiles", tag.replace("_pmc_by_kernel.json", "_kernel_stats.csv"))
            avg_ns = None
            for line in open(stats):
                if ("k_mmstream_p(" in line or "k_mmstream(" in line) and not line.startswith("#"):
                    avg_ns = float(line.rsplit('"', 1)[1].split(",")[3])
                    break
            ghz = pm[kk[0]]["GRBM_GUI_ACTIVE"]["mean_of_large_launches"] / 8 / avg_ns
            return (f"the chip holds {ghz:.2f} GHz under this kernel (GRBM_GUI_ACTIVE / 8 / average launch duration, profiles/{tag} and its kernel_stats: 2.4 nominal); "
                    "a bare register-only loop of this MFMA sustains ~4050 TOPS (tools/mfma_i8_rate.hip)")
        except Exception:
            return "effective clock not available (no PMC summary under profiles/); a bare register-only loop of this MFMA sustains ~4050 TOPS (tools/mfma_i8_rate.hip)"

    def mmstream_roofline(ktd, step_ms, steps_=None):
        """roofline of k_mmstream from the launches that serve several groups (the S / AS rounds: the dominant shape); the single-group
        launches of the same kernel (b_w over the BT+BV rows: HBM-bound) are listed beside it and counted in whole_step"""
        kt, other = ktd["mmstream_rounds"], ktd["evalmm_resident"]
        bwk = ktd.get("mmstream_bw", (0, 0.0, 0, 0.0, 0))
        if not kt[0]:
            kt, other = other, (0, 0.0, 0, 0.0, 0)
        n_, ms_, rows_, busy_, work_ = kt
        if not n_:
            return None
        avg, eff, rows, work = ms_ / n_, busy_ / n_, rows_ / n_, work_ / n_
        groups = work / rows                                     # groups of 63 / 64 coefficient vectors served by one launch (S and AS groups together)
        regions = 2.0 if (merge_regions and groups > 1) else 1.0  # image regions one launch passes over
        mtile_rows = 129536 if p.logq == 736 else 1471 * 192     # M: row tiles x 16 byte positions (8096 x 16 at logq 736)
        ops = 2.0 * mtile_rows * 256 * work                      # int8 multiply-adds x 2 per launch (M x N = 256 x K = rows x groups)
        tops = ops / (avg * 1e-3) / 1e12                         # per average launch duration: what rocprofv3 --stats reproduces
        tops_busy = ops / (eff * 1e-3) / 1e12                    # per union of launch spans (= tops when launches do not overlap)
        ops_other = 2.0 * mtile_rows * 256 * (other[4] + bwk[4])    # (all single-group and b_w launches together)
        step_tops = (ops * n_ + ops_other) / (steps_ or args.steps) / (step_ms * 1e-3) / 1e12  # the kernel's operations of a step over the WHOLE step time
        gbs = regions * rows * tile_bytes_per_row / (avg * 1e-3) / 1e9  # one pass over each region's image
        return {"bound": "mfma", "kernel": "k_mmstream_p (persistent grid, one workgroup per CU: A fragments of the expanded CRS streamed once per launch and region for the "
                                           f"{groups / regions:.0f} groups of 63 / 64 coefficient vectors of a super-group (255 proofs) -- HBM for the first workgroup of a "
                                           "row-tile set, that XCD's L2 for the others, 8 tile groups x 4 groups of ONE region on an XCD at a time --, digit fragments "
                                           "through LDS, i8 MFMA 16x16x64; the S and the AS region of a super-group share one launch)",
                "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s (int8; multiply and add counted separately)", "frac": tops / MFMA_I8_PEAK_TOPS,
                "traffic": traffic_of("traffic_mmstream.json"),
                "traffic_source": "static file profiles/traffic_mmstream.json (a separate rocprofv3 --pmc pass of this command, not this run)",
                "launches": n_, "avg_launch_ms": avg, "busy_ms_per_launch": eff, "concurrency": avg / eff, "rows_per_launch": rows,
                "groups_per_launch": groups, "regions_per_launch": regions, "int8_ops_per_launch": ops,
                "single_group_launches": ({"launches": other[0], "avg_launch_ms": other[1] / other[0], "rows_per_launch": other[2] / other[0],
                                           "what": "b_w: one byte column per proof over the BT+BV rows, one pass over that region's image per 255 proofs (HBM-bound)",
                                           "image_gbs": other[2] / other[0] * tile_bytes_per_row / (other[1] / other[0] * 1e-3) / 1e9} if other[0] else None),
                "bw_launches": ({"kernel": "k_mmstream_pb: b_w of up to 8 super-groups (one byte column per proof, 255 per group) in one pass over the BT+BV image",
                                 "launches": bwk[0], "avg_launch_ms": bwk[1] / bwk[0], "rows_per_launch": bwk[2] / bwk[0], "groups_per_launch": bwk[4] / bwk[2],
                                 "int8_tops": 2.0 * mtile_rows * 256 * bwk[4] / bwk[0] / (bwk[1] / bwk[0] * 1e-3) / 1e12} if bwk[0] else None),
                "achieved_by_busy_time": tops_busy, "frac_by_busy_time": tops_busy / MFMA_I8_PEAK_TOPS,
                "whole_step": {"achieved": step_tops, "frac": step_tops / MFMA_I8_PEAK_TOPS,
                               "note": "this kernel's int8 operations of a step / ms_per_step: the matrix-core fraction of the whole job"},
                "note": CONC_NOTE.replace("algorithmic bytes", "int8 operations"),
                "hbm": {"bytes_read_per_row": tile_bytes_per_row, "achieved_gbs": gbs, "peak_gbs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                        "fragment_gbs_consumed_incl_l2": gbs * groups / regions},
                "clock_note": clock_note()}

    def evalmm16_roofline(kt):
        n_, ms_, rows_, busy_, _ = kt
        if not n_:
            return None
        avg, eff, rows = ms_ / n_, busy_ / n_, rows_ / n_
        gbs = rows * row_bytes_b / (eff * 1e-3) / 1e9
        gblk_ = rows * (p.ctr_ct / 16.0) / (eff * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "k_evalmm16 (AES-256-CTR expansion of the rows, once per group of 31 proofs, + i8 MFMA multiply-accumulate of the "
                                          "group's 62 coefficient vectors; the BT+BV region runs once per 255 proofs, one byte column per proof)",
                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic_of("traffic_evalmm.json"),
                "launches": n_, "avg_launch_ms": avg, "busy_ms_per_launch": eff, "concurrency": avg / eff, "rows_per_launch": rows, "bytes_per_row": row_bytes_b,
                "note": CONC_NOTE + "; algorithmic bytes = expanded row bytes, regenerated with AES on the CU (LDS T-tables): LDS-lookup / VALU bound, "
                                    "~0 HBM bytes; the MFMA work (2 x 129448 x 256 x rows int8 ops) is a few % of the kernel",
                "aes_gblocks_per_s": gblk_, "aes_ceiling": aes_ceiling("k_evalmm16", gblk_),
                "mfma_int8_tops": 2.0 * 129448 * 256 * rows / (eff * 1e-3) / 1e12}

    def expand_info(kt, steps_=None):
        n_, ms_, rows_, busy_, _ = kt
        if not n_:
            return None
        avg, rows = ms_ / n_, rows_ / n_
        gblk_ = rows * (p.ctr_ct / 16.0) / (avg * 1e-3) / 1e9
        return {"kernel": "k_expand_mm (AES-256-CTR expansion of a CRS region or row slab, written in MFMA A-fragment order: lane = row, transposition on the matrix cores)", "launches": n_,
                "avg_launch_ms": avg, "rows_per_launch": rows, "ms_per_step": ms_ / (steps_ or args.steps), "aes_gblocks_per_s": gblk_,
                "aes_ceiling": aes_ceiling("k_expand_mm", gblk_),
                "note": ("inside mfh_prove_batch the first super-group's chain (witness GEMM + polynomial step, 1.8 ms alone) runs beside these launches and shares their CUs; "
                         "alone the three launches take 10.1 ms = 73 Gblock/s (profiles/r06_step_breakdown.txt)"),
                "write_gbs": rows * tile_bytes_per_row / (avg * 1e-3) / 1e9}

    # ---- batch mode: --batch statements per GPU per step through mfh_prove_batch (disjoint statements per rank, no collective)
    batched = None
    if mode == "batch" and not (big and world > 1):  # (the 2^20-constraint image only fits sharded: with N > 1 the row-sharded leg below is the headline)
        nb = args.batch
        brng = np.random.default_rng(1000 + rank)  # every rank proves its own statements: same witness, its own deltas and smudging
        b_delta = [int(x) for x in brng.integers(0, mf.P, size=nb, dtype=np.uint64)]
        b_mags = [brng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes() for _ in range(nb)]
        b_signs = [bytes(brng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nb)]
        # statements: the satisfying witness under fresh randomness (delta, smudging terms) each -- what the reference's benchmark_snark proves (random_ssp builds
        # the SSP around ONE witness, src/ssp.c:37-77: a second satisfying bit string does not exist).  Rounds 1 - 5 gave every other statement a random witness (a
        # complete proof is still computed; the verifier must reject it): that batch is the `mixed_witnesses` leg below, and --invalid-every 2 makes it the headline again.
        # Before round 6 no kernel's work depended on the witness; now the polynomial step of a statement that divides exactly is cheaper (csrc/poly.hip).
        b_valid = [not (args.invalid_every and i % args.invalid_every == args.invalid_every - 1) for i in range(nb)]
        b_random = [brng.bytes(len(inst["bits"])) for _ in range(nb)]
        b_bits = [inst["bits"] if b_valid[i] else b_random[i] for i in range(nb)]
        d_ssp_b = inst["d_ssp"]

        def run_batch(out=None, bits=None, steps=None):
            """warm-up + args.steps timed calls of prove_batch; returns (proofs, seconds [max over ranks], per-kind kernel timings)"""
            bits = b_bits if bits is None else bits
            steps = steps or args.steps
            out = ctx.prove_batch(d_crs, d_ssp_b, bits, b_delta, b_mags, b_signs, out=out)
            for _ in range(max(args.warmup - 1, 0)):
                ctx.prove_batch(d_crs, d_ssp_b, bits, b_delta, b_mags, b_signs, out=out)
            ctx.set_timing(True)
            for k in ("evalmm", "mmstream_rounds", "mmstream_bw", "evalmm_resident", "expandmm"):
                ctx.timing_drain(k)
            barrier()
            t_ = time.perf_counter()
            for _ in range(steps):
                ctx.prove_batch(d_crs, d_ssp_b, bits, b_delta, b_mags, b_signs, out=out)
            barrier()
            el = time.perf_counter() - t_
            ctx.set_timing(False)
            kt = {}
            for k in ("evalmm", "mmstream_rounds", "mmstream_bw", "evalmm_resident", "expandmm"):  # (rounds first: "evalmm_resident" then holds the single-group launches, b_w's)
                n_, ms_, rows_ = ctx.timing_drain(k)
                kt[k] = (n_, ms_, rows_, ctx.timing_busy_ms(), ctx.timing_work_rows())  # launches on two streams overlap: busy = union of their spans
            if dist is not None:
                el = ctl_reduce(el, dist.ReduceOp.MAX)
            return out, el, kt

        # headline: the CRS expanded once per call (= per step) into a transient image, streamed for every super-group of 255 proofs
        ctx.poly_exact_fallbacks()
        out_b, el_b, kt_b = run_batch()
        recomputed_b = ctx.poly_exact_fallbacks()  # statements whose polynomial step fell back to Euclidean division, warm-up and timed calls (-1: no exact path for this t)
        ok_b = ctx.to_host(ctx.verify(d_ssp_b, inst["alpha"], inst["beta"], inst["s"], inst["sk"], out_b, nb))
        torch.cuda.synchronize()
        tv = time.perf_counter()
        for _ in range(5):
            ctx.verify(d_ssp_b, inst["alpha"], inst["beta"], inst["s"], inst["sk"], out_b, nb)
        torch.cuda.synchronize()
        verify_per_s = 5 * nb / (time.perf_counter() - tv)
        same_b = True
        for i in (0, nb - 1):  # the first and the last statement against the single-proof path, bit for bit
            one = ctx.prove(d_crs, d_ssp_b, b_bits[i], b_delta[i], b_mags[i], b_signs[i])
            same_b = same_b and bool(torch.equal(out_b.view(nb, -1)[i], one))
        all_ok = [bool(int(x)) for x in ok_b] == b_valid and same_b
        # the batch of rounds 1 - 5: every other statement carries a random witness -- its proof is computed all the same (src/snark.c:117-190 does not look at
        # divisibility), the verifier must reject it, and it must equal the single-proof prover's bit for bit
        mixed = None
        if not args.invalid_every and nb >= 2:
            m_valid = [i % 2 == 0 for i in range(nb)]
            m_bits = [inst["bits"] if m_valid[i] else b_random[i] for i in range(nb)]
            msteps = max(2, args.steps // 4)
            out_m, el_m, _ = run_batch(bits=m_bits, steps=msteps)
            recomputed_m = ctx.poly_exact_fallbacks()
            ctx.set_poly_exact(1)  # (forget what this batch has taught the polynomial step: the legs below prove satisfying witnesses again)
            ok_m = ctx.to_host(ctx.verify(d_ssp_b, inst["alpha"], inst["beta"], inst["s"], inst["sk"], out_m, nb))
            im = nb - 1 if not m_valid[nb - 1] else nb - 2
            one = ctx.prove(d_crs, d_ssp_b, m_bits[im], b_delta[im], b_mags[im], b_signs[im])
            same_m = bool(torch.equal(out_m.view(nb, -1)[im], one)) and bool(torch.equal(out_m.view(nb, -1)[0], out_b.view(nb, -1)[0]))
            mixed_ok = [bool(int(x)) for x in ok_m] == m_valid and same_m
            all_ok = all_ok and mixed_ok
            mixed = {"value": world * nb * msteps / el_m, "unit": "proofs/s", "ms_per_step": el_m / msteps * 1e3, "steps": msteps,
                     "valid_accepted_invalid_rejected_and_identical_to_single_proof_path": mixed_ok,
                     "statements_recomputed_by_euclidean_division": recomputed_m,
                     "note": "the headline batch of rounds 1 - 5: every other statement carries a random witness.  Its polynomial step h = (v^2 - 1) / t does not divide exactly: "
                             "the exact-division path's check fails for it on the device and the Euclidean path recomputes that statement behind it -- in the first call(s): "
                             "once the host has seen a failed check, the next 64 batches of 255 take the Euclidean path alone (mfh_set_poly_exact, mode 1)"}
            del out_m
        if dist is not None:
            all_ok = bool(int(ctl_reduce(1 if all_ok else 0, dist.ReduceOp.MIN)))
        image_bytes = int(ctx.lib.mfh_crs_mm_image_bytes(ctx._h))
        # the memory-light variant: no transient image, every group of 31 proofs regenerates the keystream on the CU (k_evalmm16)
        regen = None
        if not args.no_resident:
            ctx.set_batch_image(False)
            out_g, el_g, kt_g = run_batch()
            regen = {"value": world * nb * args.steps / el_g, "unit": "proofs/s", "ms_per_step": el_g / args.steps * 1e3,
                     "proofs_identical_to_headline": bool(torch.equal(out_g, out_b)), "roofline": evalmm16_roofline(kt_g["evalmm"])}
            del out_g
            ctx.set_batch_image(True)
        # resident regime: the image expanded ONCE, outside the timed region, and kept across calls
        resident_b = None
        if not args.no_resident and image_bytes <= args.resident_gb * 1e9:
            ctx.set_batch_image(False)  # frees the transient image
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            image_mm = ctx.crs_expand_mm(d_crs)
            torch.cuda.synchronize()
            expand_mm_s = time.perf_counter() - t2
            ctx.set_resident_mm(image_mm)
            out_r, el_rb, kt_r = run_batch()
            same_rb = bool(torch.equal(out_r, out_b))
            ctx.set_resident_mm(None)
            ctx.set_batch_image(True)
            img_bytes = image_mm.numel()
            del image_mm, out_r
            resident_b = {"value": world * nb * args.steps / el_rb, "unit": "proofs/s", "ms_per_step": el_rb / args.steps * 1e3,
                          "proofs_identical_to_headline": same_rb, "crs_expand_s": expand_mm_s, "image_bytes_per_rank": img_bytes,
                          "roofline": mmstream_roofline(kt_r, el_rb / args.steps * 1e3)}
        # the same call with twice the statements (8 super-groups): the per-call CRS expansion is shared by twice the proofs.  Reported beside the headline, never as it:
        # the headline stays the 1020-statement call of rounds 3 - 5.
        larger = None
        if not big and not args.no_resident and world == 1:
            nl = 2 * nb
            lb, ld, lm, lsg = b_bits * 2, b_delta * 2, b_mags * 2, b_signs * 2
            out_l = ctx.prove_batch(d_crs, d_ssp_b, lb, ld, lm, lsg)
            torch.cuda.synchronize()
            lsteps = max(2, args.steps // 4)
            tl = time.perf_counter()
            for _ in range(lsteps):
                ctx.prove_batch(d_crs, d_ssp_b, lb, ld, lm, lsg, out=out_l)
            torch.cuda.synchronize()
            el_l = time.perf_counter() - tl
            larger = {"statements_per_call": nl, "value": nl * lsteps / el_l, "unit": "proofs/s", "ms_per_call": el_l / lsteps * 1e3, "calls": lsteps,
                      "second_half_identical_to_the_headline_call": bool(torch.equal(out_l.view(2, -1)[1], out_b.view(-1))),
                      "note": "one mfh_prove_batch call of twice the headline's statements: same kernels, the 10 - 12 ms CRS expansion paid once per call"}
            del out_l, lb, ld, lm, lsg
        used_image = kt_b["evalmm_resident"][0] + kt_b["mmstream_rounds"][0] > 0
        batched = {"value": world * nb * args.steps / el_b, "unit": "proofs/s", "ms_per_step": el_b / args.steps * 1e3, "statements_per_gpu_per_step": nb,
                   "valid_accepted_invalid_rejected_and_identical_to_single_proof_path": all_ok, "resident_crs": resident_b,
                   "witnesses": ("every statement carries the satisfying witness" if not args.invalid_every else f"every {args.invalid_every}. statement carries a random witness"),
                   "polynomial_step": {"exact_division_path_offered": recomputed_b >= 0, "statements_recomputed_by_euclidean_division": max(recomputed_b, 0),
                                       "what": f"h = (v^2 - 1 mod x^N - 1) t^-1 in F_p[x] / (x^N - 1), N = 2^{(p.d - 1).bit_length()}: two cyclic products of half the length, every result checked on "
                                               "the device at four points, a statement that fails recomputed by Euclidean division behind the check (csrc/poly.hip)"},
                   "mixed_witnesses": mixed,
                   "regenerate_per_group": regen, "call_of_twice_the_statements": larger, "device_verifier_proofs_per_s": verify_per_s,
                   "transient_image_bytes_per_rank": image_bytes if used_image else 0, "crs_expansion": expand_info(kt_b["expandmm"]),
                   "roofline": mmstream_roofline(kt_b, el_b / args.steps * 1e3) if used_image else evalmm16_roofline(kt_b["evalmm"])}

    # ---- LWE batch (BASELINE config 1/2: one batch of 65 536 encryptions, rows at stream offset k*135240)
    enc_per_s = None
    if rank == 0:
        B = 65536 if not big else 16384
        msg = ctx.to_device(np.random.default_rng(5).integers(0, mf.P, size=B, dtype=np.uint64).astype(np.uint32))
        errB = inst["err"][: B * p.L * 8]
        outB = ctx.empty(B * p.ctb)
        ctx.encrypt_rows(0, B, inst["sk"], msg, errB, out=outB)
        torch.cuda.synchronize()
        ctx.set_timing(True)
        ctx.timing_drain("encrypt")
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.encrypt_rows(0, B, inst["sk"], msg, errB, out=outB)
        torch.cuda.synchronize()
        enc_per_s = 3 * B / (time.perf_counter() - t1)
        ctx.set_timing(False)
        en_, ems_, _ = ctx.timing_drain("encrypt")
        enc_launch_ms = ems_ / max(en_, 1)
        enc_kernel = ("k_encrypt_mm (AES-256-CTR row expansion, one block per lane, fed straight to i8 MFMA 16x16x64 as the A operand of the "
                      "(rows x keystream bytes) x Toeplitz(sk) product <sk, a>; e p + m in the finishing kernel)")

    # ---- regev_decrypt as a workload (src/lwe.c:105-111, timed by src/benchmark_lwe.c:35-38): the batch just encrypted, (i) as FULL ciphertexts resident in
    # HBM ((n+1) values of L limbs = 141 KB each: HBM-bound, <a, sk> as a Toeplitz int8 GEMM, k_decrypt_mm), (ii) in the seed-compressed form ct_export /
    # the CRS hold (92-byte b; the a part regenerated from the stream: AES-bound like the encryption)
    dec = None
    if enc_per_s is not None:
        Bd = B if not big else 8192
        cts = torch.zeros((Bd, p.n + 1, p.L), dtype=torch.int64, device=ctx.device)
        for r0 in range(0, Bd, 4096):
            r1 = min(Bd, r0 + 4096)
            cts[r0:r1, : p.n] = ctx.sample_rows(r0 * p.ctr_ct, r1 - r0).view(torch.int64).view(r1 - r0, p.n, p.L)
        bpad = torch.zeros((Bd, p.L * 8), dtype=torch.uint8, device=ctx.device)
        bpad[:, : p.ctb] = outB.view(B, p.ctb)[:Bd]
        cts[:, p.n] = bpad.view(torch.int64)
        del bpad
        flat = cts.view(torch.uint8).reshape(-1)
        want = msg.view(torch.int32)[:Bd]
        res = {}
        for name, path in (("matrix_cores", 2), ("valu", 1)):
            ctx.set_decrypt_path(path)
            o = ctx.decrypt(inst["sk"], flat, Bd)
            torch.cuda.synchronize()
            ok_d = bool(torch.equal(o.view(torch.int32), want))
            ctx.set_timing(True)
            ctx.timing_drain("decrypt")
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.decrypt(inst["sk"], flat, Bd, out=o)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t1) / 3
            ctx.set_timing(False)
            dn_, dms_, _ = ctx.timing_drain("decrypt")
            res[name] = (Bd / dts, dts * 1e3, dms_ / max(dn_, 1), ok_d)
        ctx.set_decrypt_path(0)
        o2 = ctx.decrypt_rows(0, Bd, inst["sk"], outB)
        torch.cuda.synchronize()
        ok_r = bool(torch.equal(o2.view(torch.int32), want))
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.decrypt_rows(0, Bd, inst["sk"], outB, out=o2)
        torch.cuda.synchronize()
        rows_per_s = 3 * Bd / (time.perf_counter() - t1)
        del cts, flat
        mm = res["matrix_cores"]
        ct_bytes = (p.n + 1) * p.L * 8
        kern_gbs = Bd * (p.n + 1) * p.ctb / (mm[2] * 1e-3) / 1e9 if mm[2] else None
        dec = {"metric": "lwe_dec_per_s", "value": mm[0], "unit": "dec/s", "batch": Bd, "ms_per_batch": mm[1], "messages_recovered": mm[3] and res["valu"][3] and ok_r,
               "workload": "benchmark_lwe parameters (N=%d, logq=%d): one batch of %d regev_decrypt over FULL ciphertexts resident in HBM (%d B each)" % (p.n, p.logq, Bd, ct_bytes),
               "roofline": {"bound": "hbm", "kernel": "k_decrypt_mm (ciphertext bytes streamed from HBM, one 16-byte load per lane = the A operand of i8 MFMA 16x16x64 after a "
                                                     "ds_bpermute; Toeplitz(sk) fragments through LDS; b - <a, sk> mod p in the finishing kernel)",
                            "achieved": kern_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (kern_gbs / HBM_PEAK_GBS) if kern_gbs else None, "traffic": traffic_of("traffic_decryptmm.json"),
                            "bytes_per_unit": (p.n + 1) * p.ctb, "avg_launch_ms": mm[2], "bytes_read_per_unit": ct_bytes,
                            "read_gbs": Bd * ct_bytes / (mm[2] * 1e-3) / 1e9 if mm[2] else None,
                            "note": "algorithmic bytes = the expanded ciphertext row of SURVEY 8(d) (1471 x 92 B); the kernel reads the values as they lie in memory "
                                    "(1471 x 96 B); avg_launch_ms = HIP events around k_decrypt_mm; ms_per_batch also holds the per-key operand preparation (0.13 ms) and the finishing kernel"},
               "valu_path": {"value": res["valu"][0], "unit": "dec/s", "ms_per_batch": res["valu"][1], "kernel": "k_decrypt (one workgroup per ciphertext, 253 v_mad_u64_u32 per coordinate)"},
               "seed_compressed": {"value": rows_per_s, "unit": "dec/s", "kernel": "k_encrypt_mm + k_decrypt_finish_mm (a regenerated from the public stream: AES-256-CTR on the CU, no HBM reads)",
                                   "aes_gblocks_per_s": rows_per_s * (p.ctr_ct / 16.0) / 1e9,
                                   "aes_ceiling": aes_ceiling("k_encrypt_mm", rows_per_s * (p.ctr_ct / 16.0) / 1e9)}}

    lwe = None
    if enc_per_s is not None:
        enc_gbs = enc_per_s * (p.n + 1) * p.ctb / 1e9
        enc_gblk = enc_per_s * (p.ctr_ct / 16.0) / 1e9
        lwe = {"metric": "lwe_enc_per_s", "value": enc_per_s, "unit": "enc/s", "batch": B, "ms_per_batch": B / enc_per_s * 1e3,
               "workload": "benchmark_lwe parameters (N=1470, logq=%d): one batch of %d regev_encrypt2 + ct_export, row k at stream offset k*CTR_CT" % (p.logq, B),
               "roofline": {"bound": "hbm", "kernel": enc_kernel, "achieved": enc_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": enc_gbs / HBM_PEAK_GBS, "traffic": None,
                            "bytes_per_unit": (p.n + 1) * p.ctb, "avg_launch_ms": enc_launch_ms,
                            "note": "algorithmic bytes = the expanded row (SURVEY 8(d)); the kernel regenerates them with AES on the CU and writes 92 B per "
                                    "encryption: LDS-lookup bound, ~0 HBM bytes",
                            "aes_gblocks_per_s": enc_gblk,
                            "aes_ceiling": aes_ceiling("k_encrypt_mm", enc_gblk)}}

    # ---- the drop-in path, PCIe and mpz_t included (never `value`): the reference's function names and types through libmfuoco_gpu.so, measured by the C driver
    # host/bench_snark in a child process (its own HIP context on the same GPU, this process idle meanwhile): what a maintainer of src/benchmark_snark.c /
    # src/benchmark_lwe.c sees after changing the link line
    drop_in = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if rank == 0 and world == 1 and not args.no_drop_in and not big and not under_profiler:  # (no child process under a profiler's preloaded library)
        drop_in = drop_in_leg()

    # ---- CPU baseline: the oracle's reference-faithful row touch (ct_import + ct_addmul_ui), one thread
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not big:
        import oracle_lib as ol

        o = ol.Oracle()
        o.bench_eval_rows(p, seed, 200)  # warm tables
        c0 = time.perf_counter()
        o.bench_eval_rows(p, seed, args.cpu_rows)
        cpu_s = time.perf_counter() - c0
        rows_per_s = args.cpu_rows / cpu_s
        ref_rows = 4 * p.d + p.m  # row touches of the reference prover: (M-1)+1 + 4D (src/snark.c:143-174)
        cpu = {"value": rows_per_s / ref_rows, "unit": "proofs/s", "cores": 1, "kind": "port",
               "sample": f"{args.cpu_rows} prover row-touches (ct_import + ct_addmul_ui, one AES block per call) of the oracle in {cpu_s:.1f} s "
                         f"= {rows_per_s:.0f} rows/s, scaled to the reference prover's {ref_rows} row-touches; h=(v^2-1)/t excluded",
               "rows_per_s": rows_per_s}
        # the reference is single-threaded by construction; its rows are independent, so an N-process run is the fair many-core figure
        # (SURVEY 8(d)): every usable core runs its own bounded sample at the same time
        ncores = usable_cores()
        if ncores > 1:
            import multiprocessing as mpx

            rows_each = max(2000, args.cpu_rows // 4)
            with mpx.get_context("spawn").Pool(ncores) as pool:
                a0 = time.perf_counter()
                times = pool.map(_cpu_rows_worker, [(p.d, p.m, p.logq, seed, rows_each)] * ncores)
                wall = time.perf_counter() - a0
            agg = sum(rows_each / t for t in times)
            cpu["all_cores"] = {"value": agg / ref_rows, "unit": "proofs/s", "cores": ncores, "kind": "port", "rows_per_s": agg,
                                "sample": f"{ncores} processes x {rows_each} row-touches each, concurrently ({wall:.1f} s wall incl. process start); "
                                          f"sum of the per-process rates", "slowest_process_s": max(times)}
        # LWE encryption on one core (src/benchmark_lwe.c:28-33): regev_encrypt2 of the oracle, own AES + truncated 704-bit dot product
        o.bench_encrypt(p, seed, 50)
        e0 = time.perf_counter()
        n_enc = 6000
        o.bench_encrypt(p, seed, n_enc)
        es = time.perf_counter() - e0
        cpu["lwe_encrypt"] = {"value": n_enc / es, "unit": "enc/s", "cores": 1, "kind": "port",
                              "sample": f"{n_enc} regev_encrypt2 calls of the oracle (row expansion + <sk,a> + e p + m) in {es:.1f} s"}
        # LWE decryption on one core (src/benchmark_lwe.c:35-38): regev_decrypt of the oracle (truncated 704-bit dot product + mod p)
        o.bench_decrypt(p, seed, 50)
        e0 = time.perf_counter()
        n_dec = 20000
        o.bench_decrypt(p, seed, n_dec)
        ds_ = time.perf_counter() - e0
        cpu["lwe_decrypt"] = {"value": n_dec / ds_, "unit": "dec/s", "cores": 1, "kind": "port", "sample": f"{n_dec} regev_decrypt calls of the oracle in {ds_:.1f} s"}
        # calibration against the REAL reference where its build travelled (oracle/_ref = reference src/aes.c + src/entropy.c):
        # its keystream generator is ~97 % of a reference prover row, so this bounds the reference's rows/s.
        ref_so = os.path.join(ROOT, "oracle", "_ref", "libmfref.so")
        if os.path.exists(ref_so):
            import ctypes

            rl = ctypes.CDLL(ref_so)
            rl.ref_bench_keystream.restype = ctypes.c_uint64
            nbytes = 135240 * 4000
            r0 = time.perf_counter()
            rl.ref_bench_keystream(ctypes.c_char_p(seed), ctypes.c_size_t(nbytes), ctypes.c_size_t(92))
            rs = time.perf_counter() - r0
            cpu["reference_keystream_MBps"] = nbytes / rs / 1e6
            cpu["reference_rows_per_s_upper_bound"] = 4000 / rs
            cpu["note"] = ("the oracle's table AES is faster than the reference's OpenSSL AES_encrypt path; reference_* fields time the real "
                           "reference aes.c/entropy.c (92-byte reads, as ct_import does) on this host")

    def emit(sharded_b, single_rows):
        """rank 0: assemble and print THE json line (called once: at the end, or by the watchdog of the collective legs); returns the overall acceptance"""
        launch_rows = rows2 / max(n2, 1)
        avg_ms = ms2 / max(n2, 1)
        row_bytes = (p.n + 1) * p.ctb
        achieved = launch_rows * row_bytes / (avg_ms * 1e-3) / 1e9 if n2 else None
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_eval2.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        gblk = (launch_rows * (p.ctr_ct / 16.0) / (avg_ms * 1e-3) / 1e9) if n2 else None
        single = None
        if run_single:
            single = {
                "value": single_steps / elapsed * (1 if by_rows else world), "unit": "proofs/s", "steps": single_steps,
                "ms_per_step": elapsed / single_steps * 1e3, "proof_accepted": bool(accepted),
                "sharding": ((f"CRS rows over {world} rank(s), 2 lane all-reduces per proof" if by_rows else f"{world} independent provers, no collective")
                             if world > 1 else "single GPU"),
                "scaling": "strong" if (by_rows and world > 1) else ("weak" if world > 1 else None),
                "collectives_per_proof": ([{"op": "all_reduce(sum, int64 lanes)", "what": "this rank's share of sum_bits v_i: the witness polynomial", "bytes": p.d * 8},
                                           {"op": "all_reduce(sum, int64 lanes)", "what": "the five partial ciphertexts, 56 bits per uint64 lane",
                                            "bytes": 5 * (p.n + 1) * p.lanes * 8}] if (by_rows and world > 1) else []),
                "ranks": world, "backend": (backend + (" (RCCL)" if backend == "nccl" else " (host-staged rehearsal)")) if world > 1 else None,
                "row_sharding_error": single_err,
                "roofline": {"bound": "hbm", "kernel": f"k_eval<{p.logq},2> (fused AES-256-CTR expansion + 2x MAC, S and AS regions)",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                             "traffic": traffic, "launches": n2, "avg_launch_ms": avg_ms, "rows_per_launch": launch_rows, "bytes_per_row": row_bytes,
                             "note": "algorithmic bytes = expanded row bytes; the kernel regenerates them with AES on the CU (LDS T-tables), "
                                     "it is LDS-lookup/VALU bound and moves ~0 HBM bytes: see DESIGN.md",
                             "aes_gblocks_per_s": gblk,
                             # the kernel's real limiter: the AES table loop (address formation of its lookups), priced against the measured cheap-address ceiling
                             "aes_ceiling": aes_ceiling("k_eval", gblk)},
                "eval1": {"launches": n1, "avg_launch_ms": ms1 / max(n1, 1), "rows_per_launch": rows1 / max(n1, 1)},
                "resident_crs": resident,
            }
        base_workload = ("benchmark_snark default SSP instance (NDEBUG): D=32768, M=21845, N=1470, logq=736; full prover() from the compressed "
                         "CRS, keystream regenerated in the timed region") if not big else (
                             f"BASELINE {args.workload}: D=2^20, M=699050, N=1470, logq={p.logq}, generator-defined SSP; full prover() from the "
                             "compressed CRS, keystream regenerated in the timed region")
        if mode == "batch" and batched is None:  # 2^20-constraint workloads on N > 1 ranks: the row-sharded batch prover is the job
            head = {"value": sharded_b["value"], "ms_per_step": sharded_b["ms_per_step"], "scaling": "strong", "roofline": sharded_b["roofline"],
                    "proof_accepted": sharded_b["own_proofs_accepted_rejected_as_expected_and_identical_to_prover"],
                    "config": {"workload": base_workload + f"; a step = {sharded_b['statements_per_step_whole_job']} statements for the whole job through the row-sharded batch "
                                           "prover: every rank expands and streams only its row shares of the matrix-core image, chain per statement slab, "
                                           "all-to-all of the coefficient row slices, one reduce-scatter of uint64 lanes per step",
                               "rows_per_proof": rows_crs, "sharding": f"CRS rows over {world} ranks"}}
            ok_all = head["proof_accepted"] and bool(accepted)
        elif mode == "batch":
            head = {"value": batched["value"], "ms_per_step": batched["ms_per_step"], "scaling": "weak", "roofline": batched["roofline"],
                    "proof_accepted": batched["valid_accepted_invalid_rejected_and_identical_to_single_proof_path"],
                    "config": {"workload": base_workload + f"; a step = {args.batch} statements per GPU (same circuit and CRS, " + ("the satisfying witness of the reference's random_ssp" if not args.invalid_every else f"every {args.invalid_every}. one with a random witness") + ", own delta and smudging terms each) through "
                                           "mfh_prove_batch: every call expands the compressed CRS once (AES on the CU) into a transient image in HBM and streams it for every "
                                           "super-group of 255 proofs (one persistent launch of 8 + 8 groups of 63 / 64 coefficient vectors over the S / AS images per super-group, one launch over BT+BV for b_w of all super-groups), "
                                           "the groups' multiply-accumulate on the matrix cores; every proof is bit-identical to "
                                           "the single-proof prover()'s",
                               "rows_per_proof": rows_crs, "statements_per_gpu_per_step": args.batch,
                               "sharding": f"{world} ranks, disjoint statements, no collective" if world > 1 else "single GPU"}}
            ok_all = head["proof_accepted"] and bool(accepted) and (sharded_b is None or bool(sharded_b.get("error")) or sharded_b["own_proofs_accepted_rejected_as_expected_and_identical_to_prover"])
        else:
            head = {"value": single["value"], "ms_per_step": single["ms_per_step"], "scaling": "strong" if by_rows else "weak",
                    "roofline": single["roofline"], "proof_accepted": bool(accepted),
                    "config": {"workload": base_workload, "rows_per_proof": rows_crs, "sharding": single["sharding"]}}
            ok_all = bool(accepted)
        out = {
            "metric": "snark_proofs_per_sec",
            "value": head["value"],
            "unit": "proofs/s",
            "n_gpus": world,
            "ranks": world,
            "launcher": os.environ.get("MFUOCO_BENCH_LAUNCHER", "external (torch.distributed.run)" if world > 1 else None),
            "backend": (backend + (" (RCCL over xGMI, device tensors)" if backend == "nccl" else " (host-staged rehearsal)")) if world > 1 else None,
            "collectives": {
                "in_value": ([c["op"] for c in sharded_b["collectives_per_step"]] if (mode == "batch" and batched is None and sharded_b and not sharded_b.get("error"))
                             else ([c["op"] for c in single["collectives_per_proof"]] if mode == "single" and single else [])),
                "timing_only": (["barrier before and after the timed steps", "all_reduce(MAX) of the elapsed time", "all_reduce(MIN) of the acceptance flag"] if world > 1 else []),
                "data_path_calls_on_rank0_all_legs": mfdist.collectives_snapshot(),
                "note": ("value = replicas: ranks prove disjoint statements, nothing is exchanged inside the timed steps" if (mode == "batch" and batched is not None and world > 1)
                         else None)},
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": head["scaling"],
            "vs_baseline": None,
            "dtype": "u32 limbs (704-bit integers mod 2^704) + AES-256 bytes" + ("; int8 digits on the matrix cores, int32 accumulation (exact)" if mode == "batch" else ""),
            "data": "synthetic (valid random SSP, random witness, seeded secrets)",
            "mode": mode,
            "config": head["config"],
            "proof_accepted": head["proof_accepted"],
            # SURVEY 8(d)'s per-proof accounting (every CRS row touched once per proof + the selected SSP rows): what the whole job
            # "moves" by that count.  The batch kernels serve 31 - 32 proofs from one read of a row, so this exceeds the HBM peak; the
            # roofline objects below count the bytes a launch actually streams
            "algorithmic_bytes_per_proof": rows_crs * (p.n + 1) * p.ctb + (p.m // 2) * p.d * 4,
            "effective_gbs_by_per_proof_accounting": head["value"] * (rows_crs * (p.n + 1) * p.ctb + (p.m // 2) * p.d * 4) / 1e9,
            "lwe_enc_per_s": enc_per_s,
            "setup_s": setup_s,
            "setup_enc_per_s": rows_crs / setup_s,
            "roofline": head["roofline"],
            "crs_expansion": batched["crs_expansion"] if batched else None,
            "transient_image_bytes_per_rank": batched["transient_image_bytes_per_rank"] if batched else None,
            "regenerate_per_group_batch": batched["regenerate_per_group"] if batched else None,
            "resident_crs_batch": batched["resident_crs"] if batched else None,
            "call_of_twice_the_statements": batched["call_of_twice_the_statements"] if batched else None,
            "witnesses": batched["witnesses"] if batched else None,
            "polynomial_step": batched["polynomial_step"] if batched else None,
            "mixed_witnesses": batched["mixed_witnesses"] if batched else None,
            "device_verifier_proofs_per_s": batched["device_verifier_proofs_per_s"] if batched else None,
            "row_sharded_batch": sharded_b,
            "single_proof_row_sharded": single_rows,
            "lwe": lwe,
            "lwe_dec_per_s": dec["value"] if dec else None,
            "lwe_decrypt": dec,
            "single_proof": single if mode == "batch" else None,
            "eval1": single["eval1"] if mode == "single" else None,
            "resident_crs": resident if mode == "single" else None,
            "cpu_baseline": cpu,
            "reference_typed_api": drop_in,
            "aes_note": ("every AES-bound kernel (k_eval, k_expand_mm, k_encrypt_mm, k_evalmm16) carries an aes_ceiling object: its AES-256 block rate against the fastest the "
                         "T-table loop can run on this chip at the kernel's occupancy -- the timing-only build whose lookup addresses cost one full-rate instruction each: 108 Gblock/s "
                         "at 4 waves per SIMD, 113.5 at 8 (profiles/r05_aes_address_bound.txt).  What separates the kernels from it is the three v_perm_b32 per column that put a "
                         "state byte at bits 8..15 of a lookup address; the LDS pipe is not the limiter (rounds 1-5 printed a 91 Gblock/s LDS ruler here: withdrawn)"),
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return ok_all
    # ---- the legs that need the data-path backend (RCCL), last and under a watchdog ---------------------------------------------------------------------
    sharded_b, single_rows = None, None
    emitted = threading.Event()

    def on_timeout(limit):
        # a collective has not completed: the main thread sits inside it and cannot be recovered.  Rank 0 prints the line with what has been measured (the
        # headline needs no collective) and the legs marked, then every rank leaves; exit code 0 -- the measurement is valid, the marked legs are not
        if emitted.is_set():
            return
        emitted.set()
        msg = f"timed out after {limit:.0f} s: a collective of this leg did not complete on the {backend} backend"
        print(f"[bench] rank {rank}: {msg}", file=sys.stderr, flush=True)
        if rank == 0:
            sb = sharded_b if isinstance(sharded_b, dict) else {"error": msg, "value": None, "unit": "proofs/s", "scaling": "strong", "ranks": world, "roofline": None,
                                                                  "own_proofs_accepted_rejected_as_expected_and_identical_to_prover": False, "ms_per_step": None}
            sr = single_rows if isinstance(single_rows, dict) else {"error": msg, "value": None}
            try:
                emit(sb, sr)
            except BaseException as e:  # (never silent: without the line the run is lost)
                print(f"[bench] rank 0: could not print the line from the watchdog: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                os._exit(5)
            os._exit(0)
        os._exit(0)

    watchdog = None
    if world > 1 and (mode == "batch" or defer_rows):
        limit = float(os.environ.get("MFUOCO_BENCH_COLLECTIVE_S", "300"))
        watchdog = threading.Timer(limit, on_timeout, args=(limit,))
        watchdog.daemon = True
        watchdog.start()
    if world > 1 and os.environ.get("MFUOCO_BENCH_TEST_HANG") == str(rank):  # (rehearsal of the watchdog: this rank never joins the collectives)
        time.sleep(10 ** 6)
    if defer_rows:
        # one proof computed by all ranks together: CRS rows sharded, two lane all-reduces per proof (dist.prove_sharded)
        try:
            rb = {}
            pr = None
            for _ in range(max(args.warmup, 1)):
                pr = mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs, rank, world, bufs=rb)
            barrier()
            t0 = time.perf_counter()
            for _ in range(single_steps):
                pr = mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], inst["bits"], delta, mags, signs, rank, world, bufs=rb)
            barrier()
            el_sr = ctl_reduce(time.perf_counter() - t0, dist.ReduceOp.MAX)
            same_sr = bool(int(ctl_reduce(1 if torch.equal(pr, proof) else 0, dist.ReduceOp.MIN)))
            single_rows = {"value": single_steps / el_sr, "unit": "proofs/s", "steps": single_steps, "ms_per_step": el_sr / single_steps * 1e3, "scaling": "strong",
                           "sharding": f"CRS rows over {world} rank(s), 2 lane all-reduces per proof", "proof_identical_to_the_single_gpu_proof_on_every_rank": same_sr,
                           "collectives_per_proof": [{"op": "all_reduce(sum, int64 lanes)", "what": "this rank's share of sum_bits v_i: the witness polynomial", "bytes": p.d * 8},
                                                     {"op": "all_reduce(sum, int64 lanes)", "what": "the five partial ciphertexts, 56 bits per uint64 lane",
                                                      "bytes": 5 * (p.n + 1) * p.lanes * 8}],
                           "backend": backend + (" (RCCL)" if backend == "nccl" else " (host-staged rehearsal)")}
            del rb, pr
        except Exception as e:
            single_rows = {"error": f"{type(e).__name__}: {e}", "value": None}
            # the other ranks may sit in the collective this rank left: agree on the host group or leave (see the first-collectives block above)
            try:
                ctl_reduce(0, dist.ReduceOp.MIN)
            except Exception:
                os._exit(3)
    # ---- N > 1: the row-sharded BATCH prover (BASELINE configs 3/4): one statement list for the whole job, CRS rows sharded over the ranks,
    # all-to-all of the coefficient row slices + ONE reduce-scatter of uint64 lanes per step (dist.prove_batch_sharded)
    sharded_b = None
    if mode == "batch" and world > 1:
        nbt = args.sharded_batch or (args.batch if not big else 255)
        srng = np.random.default_rng(4000)  # the same statements on every rank
        s_delta = [int(x) for x in srng.integers(0, mf.P, size=nbt, dtype=np.uint64)]
        s_mags = [srng.integers(0, 256, size=5 * 80, dtype=np.uint8).tobytes() for _ in range(nbt)]
        s_signs = [bytes(srng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nbt)]
        s_valid = [i % 2 == 0 for i in range(nbt)]
        s_bits = [inst["bits"] if s_valid[i] else srng.bytes(len(inst["bits"])) for i in range(nbt)]
        sbufs = {}
        sh_steps = min(args.steps, 10 if not big else 3)

        def sharded_step():
            return mfdist.prove_batch_sharded(ctx, d_crs, inst["d_ssp"], s_bits, s_delta, s_mags, s_signs, rank, world, bufs=sbufs)

        try:
            for _ in range(max(args.warmup, 1) if not big else 1):
                s_first, s_count, s_pr = sharded_step()
            sharded_err = None
        except Exception as e:  # (a collective the backend refuses fails on every rank alike: report it, keep the other legs)
            sharded_err = f"{type(e).__name__}: {e}"
    if mode == "batch" and world > 1 and sharded_err is not None:
        sharded_b = {"error": sharded_err, "value": None, "unit": "proofs/s", "scaling": "strong", "ranks": world, "roofline": None,
                     "own_proofs_accepted_rejected_as_expected_and_identical_to_prover": False, "ms_per_step": None,
                     "statements_per_step_whole_job": nbt}
    elif mode == "batch" and world > 1:
        ctx.set_timing(True)
        for k in ("evalmm", "mmstream_rounds", "mmstream_bw", "evalmm_resident", "expandmm"):
            ctx.timing_drain(k)
        barrier()
        ts = time.perf_counter()
        for _ in range(sh_steps):
            s_first, s_count, s_pr = sharded_step()
        barrier()
        el_s = time.perf_counter() - ts
        ctx.set_timing(False)
        kt_s = {}
        for k in ("evalmm", "mmstream_rounds", "mmstream_bw", "evalmm_resident", "expandmm"):
            n_, ms_, rows_ = ctx.timing_drain(k)
            kt_s[k] = (n_, ms_, rows_, ctx.timing_busy_ms(), ctx.timing_work_rows())
        el_s = ctl_reduce(el_s, dist.ReduceOp.MAX)
        ok_s = True
        if s_count:  # the rank's own proofs: accepted / rejected as the witnesses demand, and the first one bit for bit against prover()
            okv = ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], s_pr, s_count))
            ok_s = [bool(int(x)) for x in okv] == s_valid[s_first:s_first + s_count]
            one = ctx.prove(d_crs, inst["d_ssp"], s_bits[s_first], s_delta[s_first], s_mags[s_first], s_signs[s_first])
            ok_s = ok_s and bool(torch.equal(s_pr.view(s_count, -1)[0], one))
        ok_s = bool(int(ctl_reduce(1 if ok_s else 0, dist.ReduceOp.MIN)))
        per_s = -(-nbt // world)
        lps = 5 * (p.n + 1) * p.lanes
        share_img = int(ctx.lib.mfh_crs_mm_share_bytes(ctx._h, rank, world))
        sharded_b = {"value": nbt * sh_steps / el_s, "unit": "proofs/s", "scaling": "strong", "steps": sh_steps, "ms_per_step": el_s / sh_steps * 1e3,
                     "statements_per_step_whole_job": nbt, "ranks": world, "backend": backend + (" (RCCL)" if backend == "nccl" else " (host-staged rehearsal)"),
                     "own_proofs_accepted_rejected_as_expected_and_identical_to_prover": ok_s,
                     "collectives_per_step": ([{"op": "all_to_all_single", "what": "generator-defined SSP: the rank's coefficient range of w of every statement to the statement's owner "
                                                                                    "(the witness pass sharded by coefficient range: 1/N of the generation per rank)",
                                                "bytes_sent_per_rank": nbt * 4 * (p.d // world) * (world - 1) // world}] if inst["d_ssp"] is None and all(
                                                    (p.d * r // world) % 128 == 0 for r in range(world + 1)) else []) + [
                         {"op": "all_to_all_single", "what": "rows [d r/N, d (r+1)/N) of w | h | v of every statement to rank r",
                          "bytes_sent_per_rank": per_s * 3 * 4 * (p.d - p.d // world)},
                         {"op": "reduce_scatter_tensor(sum, int64 lanes)", "what": "the 5 partial ciphertexts of every statement as uint64 lanes of 56 bits (13 per 704-bit value, 27 at logq 1472)",
                          "input_bytes_per_rank": per_s * world * lps * 8, "output_bytes_per_rank": per_s * lps * 8}],
                     "reduce_scatter_bytes_per_rank_per_step": {"in": per_s * world * lps * 8, "out": per_s * lps * 8},
                     "note": ("the reduce-scatter carries 0.76 MB of uint64 lanes per statement (1.29 MB in rounds 1 - 3: one lane per 32-bit word) into every rank whatever the instance size, while the row work per "
                              "rank shrinks with D / N: at the default instance (D = 2^15) this leg is collective-bound and REPLICAS (the headline `value`) are "
                              "the better use of N GPUs; row sharding is for CRS images that do not fit one GPU (config 4/5: 45 / 90 GB per GPU on 8)"),
                     "image_share_bytes_per_rank": share_img,
                     "image": "transient: every call expands the rank's row shares (AES on the CU) inside the timed region" if kt_s["expandmm"][0] else
                              ("regenerated per group" if kt_s["evalmm"][0] else "resident"),
                     "roofline": mmstream_roofline(kt_s, el_s / sh_steps * 1e3, sh_steps) if (kt_s["evalmm_resident"][0] + kt_s["mmstream_rounds"][0]) else None,
                     "crs_expansion": expand_info(kt_s["expandmm"], sh_steps)}
        del sbufs, s_pr

    if watchdog is not None:
        watchdog.cancel()
    if emitted.is_set():  # (the watchdog fired while the last collective was returning: it prints and exits)
        time.sleep(3600)

    if rank == 0:
        emitted.set()
        accepted = emit(sharded_b, single_rows)
    if dist is not None:
        try:
            dist.barrier(group=ctl_group)
            dist.destroy_process_group()
        except Exception as e:  # (a rank that left through its watchdog is not coming: the line is out, nothing to wait for)
            print(f"[bench] rank {rank}: closing barrier failed ({type(e).__name__})", file=sys.stderr, flush=True)
    return 0 if accepted else 1


if __name__ == "__main__":
    sys.exit(main())
