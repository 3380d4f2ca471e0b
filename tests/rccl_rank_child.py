"""Child process of tests/test_gpu_rccl.py: ONE rank of a torch.distributed group on backend "nccl" (= RCCL on ROCm), on cuda:0.

Runs the two N-rank host sequences of c_lwe_snarks_amd.dist -- prove_sharded (two int64 lane all-reduces) and prove_batch_sharded
(all_to_all_single with split lists, reduce_scatter_tensor on int64 lanes; also with the witness pass sharded by coefficient range:
a second all-to-all) -- with `force_collectives`, so that a one-rank communicator still pushes every device tensor through librccl,
and compares the proofs bit for bit with mfh_prove / mfh_prove_batch.  Also grows the statement count on reused buffers (ADVICE r2:
`bpartial` must be re-allocated).  Writes "ok" or the failure to argv[1]; started fresh (nothing has touched the GPU before it).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path, port):
    import torch
    import torch.distributed as dist

    import c_lwe_snarks_amd as mf
    from c_lwe_snarks_amd import dist as mfdist
    from test_gpu_batch_sharded import SEED, _world

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    fails = []

    p = mf.Params(d=1152, m=1000)
    ctx = mf.Context(p, 0)
    ctx.set_seed(SEED)
    nb = 37
    inst, d_crs, bits, deltas, mags, signs = _world(mf, ctx, p, 77, nb)
    want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
    bufs = {}
    # a smaller call first, then the full one on the same buffers
    f0, c0, pr0 = mfdist.prove_batch_sharded(ctx, d_crs, inst["d_ssp"], bits[:5], deltas[:5], mags[:5], signs[:5], 0, 1, bufs=bufs, force_collectives=True)
    if not (f0 == 0 and c0 == 5 and torch.equal(pr0, want.view(nb, -1)[:5].reshape(-1))):
        fails.append("prove_batch_sharded(5 statements) != prove_batch")
    first, count, proofs = mfdist.prove_batch_sharded(ctx, d_crs, inst["d_ssp"], bits, deltas, mags, signs, 0, 1, bufs=bufs, force_collectives=True)
    if not (first == 0 and count == nb and torch.equal(proofs, want)):
        fails.append("prove_batch_sharded != prove_batch (reused, grown buffers)")
    # the pipelined form (round 6): stages of 7 statements -- 6 of them, the collectives issued async_op on RCCL's own stream and waited for by the compute stream only
    # where their result is used --, first with the call's own transient image share (none is registered: expanded once for the six stages), then with a registered one
    for registered in (False, True):
        if registered:
            ctx.set_resident_mm_share(ctx.crs_expand_mm_share(d_crs, 0, 1), 0, 1)
        before = mfdist.collectives_snapshot().get("reduce_scatter_tensor", {}).get("calls", 0)
        fs, cs_, prs = mfdist.prove_batch_sharded(ctx, d_crs, inst["d_ssp"], bits, deltas, mags, signs, 0, 1, bufs=bufs, force_collectives=True, stage=7)
        torch.cuda.synchronize()
        if not (fs == 0 and cs_ == nb and torch.equal(prs, want)):
            fails.append(f"prove_batch_sharded(stage=7, image share {'registered' if registered else 'transient'}) != prove_batch")
        if mfdist.collectives_snapshot()["reduce_scatter_tensor"]["calls"] - before != 6:
            fails.append("stage=7 over 37 statements should issue 6 reduce-scatters")
        if not registered and getattr(ctx, "_resident_mm", None) is not None:
            fails.append("the call's transient image share stayed registered")
    ctx.set_resident_mm_share(None, 0, 1)
    one = mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], bits[0], deltas[0], mags[0], signs[0], 0, 1, force_collectives=True)
    if not torch.equal(one, ctx.prove(d_crs, inst["d_ssp"], bits[0], deltas[0], mags[0], signs[0])):
        fails.append("prove_sharded != prove")
    ctx.close()

    p2 = mf.Params(d=1024, m=700)
    ctx2 = mf.Context(p2, 0)
    ctx2.set_seed(SEED)
    nb2 = 35
    inst2, d_crs2, bits2, deltas2, mags2, signs2 = _world(mf, ctx2, p2, 78, nb2)
    _, c2, pr2 = mfdist.prove_batch_sharded(ctx2, d_crs2, inst2["d_ssp"], bits2, deltas2, mags2, signs2, 0, 1, witness_by_cols=True, force_collectives=True)
    if not (c2 == nb2 and torch.equal(pr2, ctx2.prove_batch(d_crs2, inst2["d_ssp"], bits2, deltas2, mags2, signs2))):
        fails.append("prove_batch_sharded(witness_by_cols) != prove_batch")
    ctx2.close()

    ran = mfdist.collectives_snapshot()
    need = {"all_reduce": 2, "all_to_all_single": 4, "reduce_scatter_tensor": 3}
    for op, n in need.items():
        if ran.get(op, {}).get("calls", 0) < n:
            fails.append(f"{op}: {ran.get(op)} calls went through the backend, expected >= {n}")
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    maps = open("/proc/self/maps").read()
    if "librccl" not in maps and "libnccl" not in maps and "libtorch_hip" not in maps:
        fails.append("no RCCL-carrying library mapped")
    with open(out_path, "w") as f:
        f.write("ok " + repr(ran) if not fails else "FAIL: " + "; ".join(fails))
    return 0 if not fails else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], int(sys.argv[2])))
