"""Row-sharded prover across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

SURVEY 8(e): every proof element is sum_i coeff_i * row_i over CRS rows and the AES-CTR stream is seekable, so
each rank takes a contiguous share of the rows of every region and produces five partial ciphertexts.  The
exchange steps are one all-reduce per proof of 5 x 1471 x 22 uint64 lanes (1.3 MB) and, so that the SSP pass shards
as well, one of D uint64 lanes for the witness polynomial (256 KB): each 32-bit limb travels in its own
64-bit lane so RCCL's integer sum cannot overflow (2^32 ranks of headroom), carries are propagated once afterwards,
and because sums mod 2^704 are order-independent the result is bit-identical to the single-GPU proof.
"""
from __future__ import annotations


def allreduce_lanes(lanes, group=None):
    """Sum the int64 lane tensor over all ranks in place (no-op outside a process group)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(lanes, op=dist.ReduceOp.SUM, group=group)
    return lanes


def prove_sharded(ctx, d_crs, d_ssp, witness_bits, delta, smudge_mag, smudge_sign, rank, world, maglen=80, bufs=None, group=None):
    """prover() (reference src/snark.c:117-190) with the CRS rows sharded over `world` ranks.  Every rank returns the
    complete proof.  `bufs` may hold reusable device buffers {"partial", "lanes", "proof"}.  world == 1 proves alone: no
    collective is issued even inside a process group (independent provers per rank)."""
    bufs = {} if bufs is None else bufs
    if world == 1:  # nothing to exchange: the partial proof is the proof (mfh_prove = mfh_prove_partial + mfh_prove_finish)
        proof = ctx.prove(d_crs, d_ssp, witness_bits, delta, smudge_mag, smudge_sign, maglen, out=bufs.get("proof"))
        bufs["proof"] = proof
        return proof
    # first exchange: the SSP pass is sharded too (each rank sums its share of the selected v_i), d uint64 lanes
    wl = ctx.witness_lanes(d_ssp, witness_bits, rank, world, out=bufs.get("wlanes"))
    allreduce_lanes(wl, group)
    partial = ctx.prove_partial_w(d_crs, d_ssp, witness_bits, delta, rank, world, wl, out=bufs.get("partial"))
    # second exchange: the five partial ciphertexts, one 32-bit word per uint64 lane
    lanes = ctx.ct_to_lanes(partial, 5, out=bufs.get("lanes"))
    allreduce_lanes(lanes, group)
    proof = ctx.ct_from_lanes(lanes, 5, out=bufs.get("proof"))
    ctx.prove_finish(proof, smudge_mag, smudge_sign, maglen)
    bufs.update(wlanes=wl, partial=partial, lanes=lanes, proof=proof)
    return proof


def lanes_from_limbs_cpu(cts_u64, K):
    """CPU/torch restatement of mfh_ct_to_lanes for the gloo tests: (..., L) uint64 limbs -> (..., 2K) int64 lanes."""
    import numpy as np

    w = np.ascontiguousarray(cts_u64).view(np.uint32).reshape(*cts_u64.shape[:-1], -1)[..., : 2 * K]
    return w.astype(np.int64)


def limbs_from_lanes_cpu(lanes, L, K):
    """CPU restatement of mfh_ct_from_lanes: propagate carries over the 2K lanes, drop what exceeds 2^(64K) (modq)."""
    import numpy as np

    lanes = np.asarray(lanes).astype(np.uint64)
    out = np.zeros(lanes.shape[:-1] + (2 * L,), dtype=np.uint32)
    carry = np.zeros(lanes.shape[:-1], dtype=np.uint64)
    for w in range(2 * K):
        x = lanes[..., w]
        lo = (x & np.uint64(0xFFFFFFFF)) + (carry & np.uint64(0xFFFFFFFF))
        out[..., w] = (lo & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        carry = (x >> np.uint64(32)) + (carry >> np.uint64(32)) + (lo >> np.uint64(32))
    return out.view(np.uint64)
