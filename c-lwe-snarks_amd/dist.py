"""Row-sharded prover across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

SURVEY 8(e): every proof element is sum_i coeff_i * row_i over CRS rows and the AES-CTR stream is seekable, so
each rank takes a contiguous share of the rows of every region and produces five partial ciphertexts.  The
exchange steps are one all-reduce per proof of 5 x 1471 x 13 uint64 lanes (0.76 MB) and, so that the SSP pass shards
as well, one of D uint64 lanes for the witness polynomial (256 KB): each 32-bit limb travels in its own
64-bit lane so RCCL's integer sum cannot overflow (2^32 ranks of headroom), carries are propagated once afterwards,
and because sums mod 2^704 are order-independent the result is bit-identical to the single-GPU proof.
"""
from __future__ import annotations

# what actually went through the backend in this process: {op: [calls, bytes handed to the backend]} -- bench.py prints it at the top
# level of its JSON line so that a run shows which collectives its number contains
COLLECTIVES = {}


def _count(op, nbytes):
    c = COLLECTIVES.setdefault(op, [0, 0])
    c[0] += 1
    c[1] += int(nbytes)


def collectives_snapshot(reset=False):
    snap = {k: {"calls": v[0], "bytes": v[1]} for k, v in COLLECTIVES.items()}
    if reset:
        COLLECTIVES.clear()
    return snap


def allreduce_lanes(lanes, group=None, force=False):
    """Sum the int64 lane tensor over all ranks in place (no-op outside a process group; inside a one-rank group only with `force`)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(lanes, op=dist.ReduceOp.SUM, group=group)  # (gloo takes device tensors here and stages them itself)
        _count("all_reduce", lanes.numel() * lanes.element_size())
    return lanes


def prove_sharded(ctx, d_crs, d_ssp, witness_bits, delta, smudge_mag, smudge_sign, rank, world, maglen=80, bufs=None, group=None,
                  force_collectives=False):
    """prover() (reference src/snark.c:117-190) with the CRS rows sharded over `world` ranks.  Every rank returns the
    complete proof.  `bufs` may hold reusable device buffers {"partial", "lanes", "proof"}.  world == 1 proves alone: no
    collective is issued even inside a process group (independent provers per rank) -- unless `force_collectives`, which
    runs the whole N-rank sequence (lane conversion, both all-reduces) through the backend of a one-rank group: how the RCCL
    branch is exercised on a one-GPU box."""
    bufs = {} if bufs is None else bufs
    if world == 1 and not force_collectives:  # nothing to exchange: the partial proof is the proof (mfh_prove = mfh_prove_partial + mfh_prove_finish)
        proof = ctx.prove(d_crs, d_ssp, witness_bits, delta, smudge_mag, smudge_sign, maglen, out=bufs.get("proof"))
        bufs["proof"] = proof
        return proof
    # first exchange: the SSP pass is sharded too (each rank sums its share of the selected v_i), d uint64 lanes
    wl = ctx.witness_lanes(d_ssp, witness_bits, rank, world, out=bufs.get("wlanes"))
    allreduce_lanes(wl, group, force_collectives)
    partial = ctx.prove_partial_w(d_crs, d_ssp, witness_bits, delta, rank, world, wl, out=bufs.get("partial"))
    # second exchange: the five partial ciphertexts, one 32-bit word per uint64 lane
    lanes = ctx.ct_to_lanes(partial, 5, out=bufs.get("lanes"))
    allreduce_lanes(lanes, group, force_collectives)
    proof = ctx.ct_from_lanes(lanes, 5, out=bufs.get("proof"))
    ctx.prove_finish(proof, smudge_mag, smudge_sign, maglen)
    bufs.update(wlanes=wl, partial=partial, lanes=lanes, proof=proof)
    return proof


def _on_host_backend(t, group=None):
    """gloo (the CPU rehearsal backend: tests, and two ranks sharing one GPU) is given host tensors; nccl (= RCCL) device tensors"""
    import torch.distributed as dist

    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_to_all_rows(recv, send, out_splits, in_splits, group=None):
    import torch.distributed as dist

    if _on_host_backend(send, group):
        r = recv.cpu()
        dist.all_to_all_single(r, send.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    _count("all_to_all_single", send.numel() * send.element_size())
    return recv


def reduce_scatter_lanes(own, lanes, group=None):
    """own = this rank's equal slab of the element-wise sum of `lanes` over the ranks (int64 stands in for uint64: two's-complement sums
    are the same bits, and 2^32 ranks of 32-bit words fit)"""
    import torch.distributed as dist

    if _on_host_backend(lanes, group):
        o = own.cpu()
        dist.reduce_scatter_tensor(o, lanes.cpu(), op=dist.ReduceOp.SUM, group=group)
        own.copy_(o)
    else:
        dist.reduce_scatter_tensor(own, lanes, op=dist.ReduceOp.SUM, group=group)
    _count("reduce_scatter_tensor", lanes.numel() * lanes.element_size())
    return own


def row_shares(total, world):
    """contiguous shares [total r / world, total (r+1) / world) -- the split every mfh_*_partial / *_share entry point uses"""
    return [(total * r // world, total * (r + 1) // world) for r in range(world)]


def statement_shares(nb, world):
    """statements are owned in equal slabs of ceil(nb / world) (the last ranks may own fewer or none): the slab is the unit of the
    reduce-scatter, which needs equal pieces"""
    per = -(-nb // world) if nb else 0
    return per, [(min(nb, r * per), min(nb, (r + 1) * per)) for r in range(world)]


def prove_batch_sharded(ctx, d_crs, d_ssp, witness_bits_list, deltas, smudge_mags, smudge_signs, rank, world, maglen=80, group=None, bufs=None,
                        witness_by_cols=None, force_collectives=False):
    """prover() (reference src/snark.c:117-190) for len(witness_bits_list) statements with the CRS ROWS sharded over `world` ranks
    (BASELINE configs 3/4: "ciphertexts sharded across 8 x MI355X + RCCL reduce"; include/mfhip.h, row-sharded batch prover).

    Every rank is given the same statement list.  Data path per call:
      chain (own statements)  ->  all-to-all of the w | h | v row slices (3 x 4 B x d x nb / world sent per rank)
      ->  row shares of all five ciphertexts of all statements on the matrix cores
      ->  ONE reduce-scatter (sum) of uint64 lanes, nb x 5 x (n+1) x ceil(64 K / 56) lanes of 8 B  ->  carries, modq, delta ct_t, smudging (own statements).
    witness_by_cols (default: on for a generator-defined SSP, d_ssp = None, where the witness pass is the chain's cost): the chain is cut
    in two -- every rank computes its COEFFICIENT RANGE [d r / world, d (r+1) / world) of w of ALL statements (1 / world of the
    generation of the selected rows, no reduction), one more all-to-all (4 B x d x nb / world sent per rank) hands every statement's
    slices to its owner, who finishes the chain (v = w + v_0, h = (v^2 - 1) / t).  Needs the ranges to start at multiples of 128.
    Returns (first, count, proofs): the rank's own statements [first, first + count) and their finished proofs (count x 5 ciphertexts,
    bit-identical to prove_batch's).  world == 1 proves alone: no collective -- unless `force_collectives`, which runs the whole
    sequence (chain, all-to-all with split lists, row shares, lane conversion, reduce-scatter, finish) through a one-rank group."""
    import torch

    nb = len(witness_bits_list)
    if world == 1 and not force_collectives:
        return 0, nb, ctx.prove_batch(d_crs, d_ssp, witness_bits_list, deltas, smudge_mags, smudge_signs, maglen)
    p = ctx.params
    bufs = {} if bufs is None else bufs
    per, owned = statement_shares(nb, world)
    first, last = owned[rank]
    count = last - first
    shares = row_shares(p.d, world)
    lo, hi = shares[rank]
    cs = hi - lo
    align = getattr(ctx, "witness_cols_align", 128)
    use_cols = (d_ssp is None) if witness_by_cols is None else bool(witness_by_cols)
    use_cols = use_cols and all(a % align == 0 and b % align == 0 for a, b in shares)
    # 1. the chain of the rank's own statements: w | h | v, [3][count][d]
    if use_cols:
        # 1a. this rank's coefficient range of w of ALL statements; 1b. all-to-all by statement owner; 1c. the owner finishes the chain
        wsl = ctx.batch_witness_cols(d_ssp, witness_bits_list, deltas, lo, cs)  # [nb][cs]
        wrecv = torch.empty(count * p.d, dtype=wsl.dtype, device=wsl.device)
        all_to_all_rows(wrecv, wsl.reshape(-1), [count * (b - a) for a, b in shares], [(b - a) * cs for a, b in owned], group)
        whv = torch.empty((3, count, p.d), dtype=wsl.dtype, device=wsl.device)
        off = 0
        for a, b in shares:  # the block from rank q: [count][its range]
            whv[0][:, a:b] = wrecv[off:off + count * (b - a)].view(count, b - a)
            off += count * (b - a)
        del wsl, wrecv
        ctx.batch_chain_from_w(d_ssp, whv)
    else:
        whv = ctx.batch_chain(d_ssp, witness_bits_list[first:last], deltas[first:last])
    # 2. all-to-all: rank r gets rows [d r / world, d (r+1) / world) of w | h | v of every statement, laid out [statement][w | h | v][rows]
    send = torch.cat([whv[:, :, a:b].permute(1, 0, 2).reshape(-1) for a, b in shares]) if count else whv.reshape(-1)
    in_splits = [count * 3 * (b - a) for a, b in shares]
    out_splits = [(b - a) * 3 * cs for a, b in owned]
    recv = torch.empty(nb * 3 * cs, dtype=send.dtype, device=send.device)
    all_to_all_rows(recv, send, out_splits, in_splits, group)
    # 3. the rank's row shares of every statement's five ciphertexts
    partial = bufs.get("bpartial")
    if partial is not None and partial.numel() * partial.element_size() < nb * 5 * p.ct_limbs * 8:
        partial = None  # a buffer kept from a smaller call: mfh_prove_batch_partial writes nb x 5 ciphertexts
    partial = ctx.prove_batch_partial(d_crs, rank, world, witness_bits_list, recv, recv[cs:], recv[2 * cs:], 3 * cs, out=partial)
    # 4. the partial ciphertexts as uint64 lanes of 56 bits (13 per 704-bit value), statements padded to world equal slabs; reduce-scatter: the rank receives its slab summed
    lps = 5 * (p.n + 1) * p.lanes  # lanes per statement
    lanes = bufs.get("blanes")
    if lanes is None or lanes.numel() != per * world * lps:
        lanes = torch.zeros(per * world * lps, dtype=torch.int64, device=send.device)
    ctx.ct_to_lanes(partial, nb * 5, out=lanes)
    own = bufs.get("bown")
    if own is None or own.numel() != per * lps:
        own = torch.empty(per * lps, dtype=torch.int64, device=send.device)
    reduce_scatter_lanes(own, lanes, group)
    bufs.update(bpartial=partial, blanes=lanes, bown=own)
    # 5. carries + modq, then delta ct_t and the smudging of the rank's own statements
    proofs = ctx.ct_from_lanes(own, count * 5) if count else ctx.empty(0)
    ctx.prove_batch_finish(d_crs, deltas[first:last], smudge_mags[first:last], smudge_signs[first:last], proofs, maglen)
    return first, count, proofs


def lanes_per_value(K):
    return (64 * K + 55) // 56


def lanes_from_limbs_cpu(cts_u64, K):
    """CPU restatement of mfh_ct_to_lanes for the gloo tests: (..., L) uint64 limbs -> (..., ceil(64 K / 56)) int64 lanes, lane j = bits [56 j, 56 j + 56)."""
    import numpy as np

    x = np.ascontiguousarray(cts_u64).astype(np.uint64)
    nl = lanes_per_value(K)
    out = np.zeros(x.shape[:-1] + (nl,), dtype=np.uint64)
    for j in range(nl):
        l, sh = (56 * j) >> 6, (56 * j) & 63
        r = x[..., l] >> np.uint64(sh)
        if sh > 8 and l + 1 < K:
            r = r | (x[..., l + 1] << np.uint64(64 - sh))
        out[..., j] = r & np.uint64((1 << 56) - 1)
    return out.astype(np.int64)


def limbs_from_lanes_cpu(lanes, L, K):
    """CPU restatement of mfh_ct_from_lanes: sum_j lane_j 2^(56 j) mod 2^(64 K) (carries propagated, modq) as (..., L) uint64 limbs."""
    import numpy as np

    lanes = np.asarray(lanes).astype(np.uint64)
    nl = lanes_per_value(K)
    assert lanes.shape[-1] == nl
    flat = lanes.reshape(-1, nl)
    out = np.zeros((flat.shape[0], L), dtype=np.uint64)
    mask = (1 << (64 * K)) - 1
    for i in range(flat.shape[0]):
        v = 0
        for j in range(nl):
            v += int(flat[i, j]) << (56 * j)
        v &= mask
        for l in range(K):
            out[i, l] = (v >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
    return out.reshape(lanes.shape[:-1] + (L,))
