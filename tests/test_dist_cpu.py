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


def test_sharded_eval_allreduce_gloo(tmp_path):
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_lane_roundtrip_and_carries():
    from c_lwe_snarks_amd import dist as mfdist

    L, K = 12, 11
    rng = np.random.default_rng(0)
    vals = [int.from_bytes(rng.bytes(88), "little") for _ in range(7)] + [(1 << 704) - 1] * 3
    limbs = np.array([list(int(v).to_bytes(96, "little")) for v in vals], dtype=np.uint8).view(np.uint64)
    lanes = mfdist.lanes_from_limbs_cpu(limbs, K)
    total = lanes.sum(axis=0, keepdims=True)
    got = mfdist.limbs_from_lanes_cpu(total, L, K)
    assert int.from_bytes(got.tobytes(), "little") == sum(vals) % (1 << 704)


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
    monkeypatch.setattr(mfdist, "allreduce_lanes", lambda lanes, group=None: ctx.calls.append("ALLREDUCE"))
    mfdist.prove_sharded(ctx, None, None, None, None, None, None, 0, world)
    assert ctx.calls == expect
