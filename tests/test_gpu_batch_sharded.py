"""The row-sharded batch prover (include/mfhip.h: mfh_batch_chain -> exchange -> mfh_prove_batch_partial -> lane reduction ->
mfh_prove_batch_finish; SURVEY 8(e), BASELINE configs 3/4) on ONE GPU:

  * a single-process emulation of `world` ranks -- every rank's shares computed one after the other, the exchange and the lane sum
    done with torch -- must give mfh_prove_batch's proofs bit for bit (which test_gpu_batch_sizes.py pins to the oracle), with the
    rank's share of the matrix-core image registered, with the transient image, and regenerating the keystream;
  * the real host sequence (c_lwe_snarks_amd.dist.prove_batch_sharded) in two PROCESSES sharing the GPU over a gloo process group.
"""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = bytes((17 * i + 1) & 0xFF for i in range(40))


def _world(mf, ctx, p, seed_int, nb):
    """a valid instance built on the GPU (bench.build_instance: random_ssp's construction, src/ssp.c:59-71) + nb statements"""
    import torch

    sys.path.insert(0, ROOT)
    import bench

    inst = bench.build_instance(mf, ctx, torch, p, seed_int)
    ctx.ssp_prepare(inst["d_ssp"])
    d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
    rng = np.random.default_rng(seed_int + 1)
    nbytes = (p.m + 7) // 8
    bits = [inst["bits"] if b % 2 == 0 else rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for b in range(nb)]
    deltas = [int(x) for x in rng.integers(0, ol.P, size=nb, dtype=np.uint64)]
    mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
    signs = [bytes(rng.integers(0, 2, size=5, dtype=np.uint8).tolist()) for _ in range(nb)]
    return inst, d_crs, bits, deltas, mags, signs


@pytest.mark.parametrize("d,m,nb,world,chunk_rows,mode", [
    (256, 64, 40, 3, 0, "share_image"),       # the reference's debug size: shares of 85 / 85 / 86 S rows and 21 / 21 / 22 BT+BV rows
    (256, 64, 40, 3, 0, "regenerate"),
    (256, 64, 70, 2, 0, "transient"),         # the call expands the rank's shares itself
    (1152, 1000, 70, 3, 256, "share_image"),  # shares of 384 S rows in 2 row chunks, 333 / 333 / 334 BT+BV rows; 3 groups per rank launch
    (1152, 1000, 35, 5, 0, "transient"),      # 230 / 231-row shares: no share is a multiple of 64
    (256, 10, 33, 12, 0, "share_image"),      # more ranks than BT+BV rows: two ranks own none of them
])
def test_sharded_batch_emulation_equals_prove_batch(gpu_ctx_factory, d, m, nb, world, chunk_rows, mode):
    import torch

    import c_lwe_snarks_amd as mf
    from c_lwe_snarks_amd import dist as mfdist

    p = mf.Params(d=d, m=m)
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst, d_crs, bits, deltas, mags, signs = _world(mf, ctx, p, 1000 * d + m, nb)
    want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).clone()
    ctx.set_mm_chunk_rows(chunk_rows)
    ctx.set_batch_image(mode != "regenerate")
    per, owned = mfdist.statement_shares(nb, world)
    shares = mfdist.row_shares(p.d, world)
    # step 1 on every "rank": the chain of its own statements
    whv = [ctx.batch_chain(inst["d_ssp"], bits[a:b], deltas[a:b]).clone() for a, b in owned]
    lps = 5 * (p.n + 1) * p.lanes
    total = torch.zeros(per * world * lps, dtype=torch.int64, device=ctx.device)
    try:
        for r in range(world):
            lo, hi = shares[r]
            cs = hi - lo
            # step 2, the all-to-all as rank r sees its result: [statement][w | h | v][rows lo..hi)
            recv = torch.cat([w[:, :, lo:hi].permute(1, 0, 2).reshape(-1) for w in whv])
            assert recv.numel() == nb * 3 * cs
            image = None
            if mode == "share_image":
                image = ctx.crs_expand_mm_share(d_crs, r, world)
                assert image.numel() == int(ctx.lib.mfh_crs_mm_share_bytes(ctx._h, r, world))
                ctx.set_resident_mm_share(image, r, world)
            ctx.set_timing(True)
            try:
                partial = ctx.prove_batch_partial(d_crs, r, world, bits, recv, recv[cs:], recv[2 * cs:], 3 * cs)  # step 3
            finally:
                ctx.set_timing(False)
                if image is not None:
                    ctx.set_resident_mm_share(None, 0, 1)
            # the path that ran: streamed from the (registered or transient) image of the rank's shares, or AES per group
            streamed, regen = ctx.timing_drain("evalmm_resident")[0], ctx.timing_drain("evalmm")[0]
            assert (streamed > 0 and regen == 0) if mode != "regenerate" else (regen > 0 and streamed == 0), (streamed, regen)
            assert (ctx.timing_drain("expandmm")[0] > 0) == (mode == "transient")
            lanes = torch.zeros_like(total)
            ctx.ct_to_lanes(partial, nb * 5, out=lanes)  # step 4: what the reduce-scatter sums
            total += lanes
            del image
    finally:
        ctx.set_mm_chunk_rows(0)
        ctx.set_batch_image(True)
    got = []
    for r, (a, b) in enumerate(owned):  # step 5 on the owner of each slab
        if b == a:
            continue
        proofs = ctx.ct_from_lanes(total[r * per * lps:(r * per + (b - a)) * lps], (b - a) * 5)
        ctx.prove_batch_finish(d_crs, deltas[a:b], mags[a:b], signs[a:b], proofs)
        got.append(proofs)
    assert torch.equal(torch.cat(got), want)
    ok = ctx.to_host(ctx.verify(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], torch.cat(got), nb))
    assert [bool(x) for x in ok] == [b % 2 == 0 for b in range(nb)]
    ctx.close()


def test_sharded_batch_argument_checks(gpu_ctx_factory):
    import c_lwe_snarks_amd as mf

    p = mf.DEBUG
    ctx = gpu_ctx_factory(p)
    ctx.set_seed(SEED)
    inst, d_crs, bits, deltas, mags, signs = _world(mf, ctx, p, 5, 4)
    whv = ctx.batch_chain(inst["d_ssp"], bits, deltas)
    with pytest.raises(mf.MfhError):  # rank >= world
        ctx.prove_batch_partial(d_crs, 2, 2, bits, whv[0], whv[1], whv[2], p.d)
    with pytest.raises(mf.MfhError):  # stride shorter than the share
        ctx.prove_batch_partial(d_crs, 0, 2, bits, whv[0], whv[1], whv[2], 100)
    image = ctx.crs_expand_mm_share(d_crs, 1, 2)
    ctx.set_resident_mm_share(image, 1, 2)
    try:
        with pytest.raises(mf.MfhError):  # the image holds another rank's shares
            ctx.prove_batch_partial(d_crs, 0, 2, bits, whv[0], whv[1], whv[2], p.d)
        with pytest.raises(mf.MfhError):  # ... and the unsharded call must not silently regenerate beside it
            ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
    finally:
        ctx.set_resident_mm_share(None, 0, 1)
    assert ctx.prove_batch(d_crs, inst["d_ssp"], [], [], [], []).numel() == 0  # no statements: nothing to do, no error
    ctx.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import c_lwe_snarks_amd as mf
    from c_lwe_snarks_amd import dist as mfdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = mf.Params(d=1152, m=1000)
    ctx = mf.Context(p, 0)  # both ranks on the one GPU of the box
    ctx.set_seed(SEED)
    nb = 37
    inst, d_crs, bits, deltas, mags, signs = _world(mf, ctx, p, 77, nb)  # deterministic: identical on both ranks
    first, count, proofs = mfdist.prove_batch_sharded(ctx, d_crs, inst["d_ssp"], bits, deltas, mags, signs, rank, world)
    want = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs).view(nb, -1)[first:first + count].reshape(-1)
    ok = bool(torch.equal(proofs, want)) and count > 0
    # the single-proof row-sharded path with its two lane all-reduces, same process group
    one = mfdist.prove_sharded(ctx, d_crs, inst["d_ssp"], bits[0], deltas[0], mags[0], signs[0], rank, world)
    ok = ok and bool(torch.equal(one, ctx.prove(d_crs, inst["d_ssp"], bits[0], deltas[0], mags[0], signs[0])))
    ctx.close()
    # the chain cut in two (coefficient ranges of w of all statements per rank + a first all-to-all to the statement owners): d = 1024, so
    # that the ranges start at multiples of 128
    p2 = mf.Params(d=1024, m=700)
    ctx2 = mf.Context(p2, 0)
    ctx2.set_seed(SEED)
    nb2 = 35
    inst2, d_crs2, bits2, deltas2, mags2, signs2 = _world(mf, ctx2, p2, 78, nb2)
    first2, count2, proofs2 = mfdist.prove_batch_sharded(ctx2, d_crs2, inst2["d_ssp"], bits2, deltas2, mags2, signs2, rank, world, witness_by_cols=True)
    want2 = ctx2.prove_batch(d_crs2, inst2["d_ssp"], bits2, deltas2, mags2, signs2).view(nb2, -1)[first2:first2 + count2].reshape(-1)
    ok = ok and bool(torch.equal(proofs2, want2)) and count2 > 0
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "MISMATCH")
    ctx2.close()
    dist.destroy_process_group()


def test_prove_batch_sharded_two_processes_one_gpu(tmp_path):
    """dist.prove_batch_sharded end to end: 2 ranks (processes) share cuda:0, collectives over gloo (host staged); every rank's own
    proofs equal mfh_prove_batch's -- with the chain per statement owner, and with the witness pass sharded by coefficient range"""
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
