"""N > 1 path on CPU: world_size-2 gloo run of the row-sharded evaluation + lazy-carry lane all-reduce.

The per-rank compute stand-in is the oracle (allowed in tests); what is under test is the distributed logic that
bench.py / c_lwe_snarks_amd.dist use on GPUs: contiguous row shares addressed by stream offset, one SUM all-reduce of
int64 lanes, carry propagation + modq afterwards, bit-identical to the unsharded result.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    import c_lwe_snarks_amd as mf
    import oracle_lib as ol
    from c_lwe_snarks_amd import dist as mfdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = mf.DEBUG
    o = ol.Oracle()
    seed = bytes(range(40))
    nrows = 11  # odd on purpose: uneven shares
    rng = np.random.default_rng(123)
    c8 = rng.integers(0, 256, size=nrows * p.ctb, dtype=np.uint8)
    co = rng.integers(0, ol.P, size=nrows, dtype=np.uint64)
    lo = nrows * rank // world
    hi = nrows * (rank + 1) // world
    off = p.ctr_as + lo * p.ctr_ct
    part = o.eval_poly(p, seed, off, c8[lo * p.ctb: hi * p.ctb].tobytes(), co[lo:hi])
    lanes = torch.from_numpy(mfdist.lanes_from_limbs_cpu(part, p.K))
    mfdist.allreduce_lanes(lanes)
    got = mfdist.limbs_from_lanes_cpu(lanes.numpy(), p.L, p.K)
    full = o.eval_poly(p, seed, p.ctr_as, c8.tobytes(), co)
    ok = np.array_equal(got.reshape(full.shape), full)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "MISMATCH")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_eval_allreduce_gloo(tmp_path, world):
    """the lazy-carry all-reduce of 56-bit lanes against the oracle's eval_poly over all rows; world = 8: the machine's rank count (11 rows: shares of 1 or 2)"""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


@pytest.mark.parametrize("L,K", [(12, 11), (23, 23)])
def test_lane_roundtrip_and_carries(L, K):
    """56-bit lanes (13 per 704-bit value, 27 per 1472-bit value): lane-wise uint64 sums of up to 256 maximal values do not overflow, and the carry
    propagation + modq of the restatement gives the big-integer sum mod 2^(64 K)"""
    from c_lwe_snarks_amd import dist as mfdist

    bits = 64 * K
    assert mfdist.lanes_per_value(K) == {11: 13, 23: 27}[K]
    rng = np.random.default_rng(0)
    vals = [int.from_bytes(rng.bytes(bits // 8), "little") for _ in range(7)] + [(1 << bits) - 1] * 249
    limbs = np.array([list(int(v).to_bytes(8 * L, "little")) for v in vals], dtype=np.uint8).view(np.uint64)
    lanes = mfdist.lanes_from_limbs_cpu(limbs, K)
    assert lanes.shape == (256, mfdist.lanes_per_value(K)) and int(lanes.max()) < 1 << 56
    back = mfdist.limbs_from_lanes_cpu(lanes, L, K)
    assert np.array_equal(back, limbs)
    total = lanes.astype(np.uint64).sum(axis=0, keepdims=True, dtype=np.uint64)  # what ncclSum on ncclUint64 computes
    assert sum(int(x) for x in lanes[:, 0]) == int(total[0, 0])                  # no wrap-around at 256 summands
    got = mfdist.limbs_from_lanes_cpu(total, L, K)
    assert int.from_bytes(got.tobytes(), "little") == sum(vals) % (1 << bits)


class _FakeCtx:
    """Records the order of C-ABI calls prove_sharded makes (no GPU needed): the host sequencing is what is under test."""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        def f(*a, **k):
            self.calls.append(name)
            return name
        return f


@pytest.mark.parametrize("world,expect", [
    (1, ["prove"]),
    (4, ["witness_lanes", "ALLREDUCE", "prove_partial_w", "ct_to_lanes", "ALLREDUCE", "ct_from_lanes", "prove_finish"]),
])
def test_prove_sharded_call_sequence(monkeypatch, world, expect):
    """world == 1 issues no collective (independent provers inside a process group); world > 1 issues exactly the two
    lane all-reduces of SURVEY 8(e), witness lanes first."""
    from c_lwe_snarks_amd import dist as mfdist

    ctx = _FakeCtx()
    monkeypatch.setattr(mfdist, "allreduce_lanes", lambda lanes, group=None, force=False: ctx.calls.append("ALLREDUCE"))
    mfdist.prove_sharded(ctx, None, None, None, None, None, None, 0, world)
    assert ctx.calls == expect


# ---- the row-sharded BATCH prover's host sequencing (dist.prove_batch_sharded) under a real gloo process group -----------------------
class _LinearCtx:
    """CPU stand-in for the C ABI with LINEAR arithmetic whose result exposes any mix-up of statements, rows or ownership:
    chain: w | h | v [k][b][i] = 1000 k + 10 id_b + i  (id_b = the statement's delta);  partial: component k of statement b =
    sum over the rank's rows i of coef_k[i] * (i + 1) (absolute row index), so the summed proof is a known closed form;
    finish adds 7 id_b.  What is under test is prove_batch_sharded itself: split sizes and layout of the all-to-all, the padded
    reduce-scatter, which statements a rank finishes."""

    class _P:
        d, m, n, K, L, lanes = 13, 9, 1, 1, 1, 2  # (one 64-bit limb per value: two 56-bit lanes)

    def __init__(self, resident=True):
        import torch

        self.torch = torch
        self.params = self._P()
        self.params.ct_limbs = (self.params.n + 1) * self.params.L
        self.finished = []
        self.device = torch.device("cpu")
        self.partial_calls = 0
        self.chain_passes = []  # own statements per chain pass: one pass of up to 255 covers every stage (not one pass per stage)
        self._resident_mm = "image share" if resident else None  # (without one a call of several stages expands the rank's shares itself, once)
        self.expansions = 0
        self.params.ctb = 8

    def crs_expand_mm_share(self, d_crs, rank, world, out=None):
        self.expansions += 1
        return "transient share"

    def set_resident_mm_share(self, image, rank, world):
        self._resident_mm = image

    def empty(self, nbytes):
        return self.torch.empty(int(nbytes), dtype=self.torch.uint8)

    def batch_chain(self, d_ssp, bits_list, deltas, out=None):
        t, p = self.torch, self.params
        self.chain_passes.append(len(deltas))
        res = t.zeros((3, len(deltas), p.d), dtype=t.int32)
        for k in range(3):
            for b, idb in enumerate(deltas):
                res[k, b] = 1000 * k + 10 * idb + t.arange(p.d, dtype=t.int32)
        if out is None:
            return res
        out.copy_(res)
        return out

    witness_cols_align = 1  # (the C ABI wants coefficient ranges at multiples of 128; the stand-in takes any)

    def batch_witness_cols(self, d_ssp, bits_list, deltas, col0, ncols):
        t = self.torch
        out = t.zeros((len(deltas), ncols), dtype=t.int32)
        for b, idb in enumerate(deltas):
            out[b] = 10 * idb + t.arange(col0, col0 + ncols, dtype=t.int32)
        return out

    def batch_chain_from_w(self, d_ssp, whv):
        whv[1] = whv[0] + 1000
        whv[2] = whv[0] + 2000
        return whv

    def prove_batch_partial(self, d_crs, rank, world, bits_list, d_w, d_h, d_v, stride, out=None):
        t, p = self.torch, self.params
        lo, hi = p.d * rank // world, p.d * (rank + 1) // world
        wts = t.arange(lo + 1, hi + 1, dtype=t.int64)
        nb = len(bits_list)
        self.partial_calls += 1
        part = t.zeros((nb, 5, p.n + 1, p.L), dtype=t.int64)
        for b in range(nb):
            for k, src in enumerate((d_w, d_h, d_v)):
                part[b, k, 0, 0] = int((src[b * stride: b * stride + (hi - lo)].to(t.int64) * wts).sum())
            part[b, 4, 1, 0] = sum((bits_list[b][i // 8] >> (i % 8)) & 1 for i in range(p.m * rank // world, p.m * (rank + 1) // world))
        return part

    def ct_to_lanes(self, partial, count, out=None):
        from c_lwe_snarks_amd import dist as mfdist

        p = self.params
        lanes = self.torch.from_numpy(mfdist.lanes_from_limbs_cpu(partial.numpy().astype(np.uint64).reshape(-1, p.L), p.K)).reshape(-1)
        out[: lanes.numel()] = lanes
        return out

    def ct_from_lanes(self, lanes, count, out=None):
        from c_lwe_snarks_amd import dist as mfdist

        p = self.params
        n = count * (p.n + 1)
        res = self.torch.from_numpy(mfdist.limbs_from_lanes_cpu(lanes[: n * p.lanes].numpy().reshape(n, p.lanes), p.L, p.K).astype(np.int64)).reshape(-1)
        if out is None:
            return res
        out.view(self.torch.int64)[: res.numel()] = res
        return out

    def prove_batch_finish(self, d_crs, deltas, mags, signs, proofs, maglen=80):
        p = self.params
        v = proofs.view(self.torch.int64).view(len(deltas), 5, p.n + 1, p.L) if len(deltas) else proofs
        for b, idb in enumerate(deltas):
            v[b, 4, 0, 0] += 7 * idb
        self.finished += list(deltas)
        return proofs


def _batch_worker(rank, world, port, out_dir, nb, by_cols=False, stage=None):
    import torch
    import torch.distributed as dist

    from c_lwe_snarks_amd import dist as mfdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _LinearCtx(resident=stage != "no image")
    p = ctx.params
    ids = [3 + 2 * b for b in range(nb)]
    bits = [bytes([(37 * b + 1) & 0xFF, (11 * b) & 0xFF]) for b in range(nb)]
    mfdist.collectives_snapshot(reset=True)
    first, count, proofs = mfdist.prove_batch_sharded(ctx, None, None, bits, ids, [b""] * nb, [b""] * nb, rank, world, witness_by_cols=by_cols,
                                                      stage=2 if stage == "no image" else stage)
    per = -(-nb // world)
    ok = first == min(nb, rank * per) and count == min(nb, first + per) - first and ctx.finished == ids[first:first + count]
    # stages: as planned (one when no image share is registered), one row-work call each; the bytes handed to the backend do not depend on the cut
    sper, nst = mfdist.stage_plan(nb, world, 2 if stage == "no image" else stage)
    ok = ok and ctx.partial_calls == nst
    ok = ok and (by_cols or ctx.chain_passes == ([count] if count else []))  # (fewer than 255 own statements here: ONE chain pass whatever the stages)
    # no image share registered: a call of several stages expands the rank's shares once, streams them for every stage and leaves nothing registered
    ok = ok and ctx.expansions == (1 if stage == "no image" and nst > 1 else 0) and (stage != "no image" or ctx._resident_mm is None)
    snap = mfdist.collectives_snapshot()
    lps = 5 * (p.n + 1) * p.lanes
    ok = ok and snap["reduce_scatter_tensor"] == {"calls": nst, "bytes": per * world * lps * 8}
    a2a_bytes = count * 3 * p.d * 4 + (nb * (p.d * (rank + 1) // world - p.d * rank // world) * 4 if by_cols else 0)
    ok = ok and snap["all_to_all_single"] == {"calls": nst * (2 if by_cols else 1), "bytes": a2a_bytes}
    got = proofs.view(torch.int64).view(count, 5, p.n + 1, p.L) if count else None
    wsum = sum(i * (i + 1) for i in range(p.d))   # sum_i i (i + 1)
    w1 = sum(i + 1 for i in range(p.d))
    for b in range(count):
        idb = ids[first + b]
        for k in range(3):
            ok = ok and int(got[b, k, 0, 0]) == (1000 * k + 10 * idb) * w1 + wsum
        ok = ok and int(got[b, 3, 0, 0]) == 0
        ok = ok and int(got[b, 4, 0, 0]) == 7 * idb
        ok = ok and int(got[b, 4, 1, 0]) == sum((bits[first + b][i // 8] >> (i % 8)) & 1 for i in range(p.m))
    with open(os.path.join(out_dir, f"brank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "MISMATCH")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nb,by_cols,stage", [(2, 5, False, None), (2, 4, False, 1), (3, 2, False, None), (2, 5, True, 2), (3, 2, True, None), (3, 7, True, 1),
                                                    (8, 20, False, None), (8, 5, True, None), (8, 5, False, None), (8, 20, True, 2), (8, 20, False, 1), (2, 9, False, 2),
                                                    (2, 9, True, "no image"), (8, 20, False, 0)])
def test_prove_batch_sharded_sequencing_gloo(tmp_path, world, nb, by_cols, stage):
    """uneven statement slabs (5 over 2), even ones, and a rank that owns no statement (2 over 3; d = 13 and m = 9 never divide);
    by_cols: the chain cut in two -- coefficient ranges of w of all statements per rank, a first all-to-all to the statement owners.
    world = 8 is the machine's real rank count (one process per GPU of an 8 x MI355X node): 20 statements = slabs of 3, 3, 3, 3, 3, 3, 2, 0 (the last rank owns
    none), 5 statements = three ranks without any; row shares of 13 and 9 rows over 8 ranks are 1 or 2 rows each.
    stage: statements per rank and pipeline stage (None = the plan of a real call -- one stage at these sizes --, 1 / 2 = several stages, the last one ragged and
    some ranks empty in it: 20 statements over 8 ranks in stages of 2 are 16 + 4 with ranks 0 .. 5 holding one more and rank 6 done; 0 = the one-shot sequence;
    "no image" = no image share registered with the context: the call (stages of 2) expands the rank's shares itself, once, and unregisters them at the end)"""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_batch_worker, args=(world, port, str(tmp_path), nb, by_cols, stage), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"brank{r}.txt").read() == "ok"
