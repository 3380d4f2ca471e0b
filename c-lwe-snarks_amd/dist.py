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


class _Done:
    """a finished collective (host-staged backends complete inside the call)"""

    def wait(self):
        return True


def all_to_all_rows(recv, send, out_splits, in_splits, group=None, async_op=False):
    """async_op (device tensors on RCCL only): the collective is queued on the backend's own stream behind what the current stream holds and a handle is returned;
    handle.wait() makes the CURRENT STREAM wait for it (no host wait) -- kernels queued in between run beside the transfer"""
    import torch.distributed as dist

    work = _Done()
    if _on_host_backend(send, group):
        r = recv.cpu()
        dist.all_to_all_single(r, send.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        recv.copy_(r)
    else:
        w = dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group, async_op=async_op)
        work = w if async_op and w is not None else work
    _count("all_to_all_single", send.numel() * send.element_size())
    return work if async_op else recv


def reduce_scatter_lanes(own, lanes, group=None, async_op=False):
    """own = this rank's equal slab of the element-wise sum of `lanes` over the ranks (int64 stands in for uint64: two's-complement sums
    are the same bits, and 2^32 ranks of 32-bit words fit).  async_op: as in all_to_all_rows."""
    import torch.distributed as dist

    work = _Done()
    if _on_host_backend(lanes, group):
        o = own.cpu()
        dist.reduce_scatter_tensor(o, lanes.cpu(), op=dist.ReduceOp.SUM, group=group)
        own.copy_(o)
    else:
        w = dist.reduce_scatter_tensor(own, lanes, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        work = w if async_op and w is not None else work
    _count("reduce_scatter_tensor", lanes.numel() * lanes.element_size())
    return work if async_op else own


def row_shares(total, world):
    """contiguous shares [total r / world, total (r+1) / world) -- the split every mfh_*_partial / *_share entry point uses"""
    return [(total * r // world, total * (r + 1) // world) for r in range(world)]


def statement_shares(nb, world):
    """statements are owned in equal slabs of ceil(nb / world) (the last ranks may own fewer or none): the slab is the unit of the
    reduce-scatter, which needs equal pieces"""
    per = -(-nb // world) if nb else 0
    return per, [(min(nb, r * per), min(nb, (r + 1) * per)) for r in range(world)]


def stage_plan(nb, world, stage=None):
    """How a row-sharded batch call is cut into pipeline stages (host/mfuoco_dist.c uses the same arithmetic): stage k holds, from every rank, the statements
    [k sper, (k + 1) sper) of that rank's slab.  sper: the fewest equally long stages of at most 255 statements each (one super-group of the row work = one pass over
    the rank's image share per stage; 1020 statements on 8 ranks: 5 stages of 26 per rank; 255 on 8: one stage); `stage` forces it (0 = one stage).
    Returns (sper, number of stages)."""
    per, _ = statement_shares(nb, world)
    if per == 0:
        return 0, 0
    if stage is None:
        nst = -(-nb // 255)
        while True:
            sper = -(-per // nst)
            if sum(min(sper, max(0, min(nb, (q + 1) * per) - min(nb, q * per))) for q in range(world)) <= 255 or sper == 1:
                break
            nst += 1
    else:
        sper = int(stage)
    if sper <= 0 or sper > per:
        sper = per
    return sper, -(-per // sper)


def prove_batch_sharded(ctx, d_crs, d_ssp, witness_bits_list, deltas, smudge_mags, smudge_signs, rank, world, maglen=80, group=None, bufs=None,
                        witness_by_cols=None, force_collectives=False, stage=None):
    """prover() (reference src/snark.c:117-190) for len(witness_bits_list) statements with the CRS ROWS sharded over `world` ranks
    (BASELINE configs 3/4: "ciphertexts sharded across 8 x MI355X + RCCL reduce"; include/mfhip.h, row-sharded batch prover).

    Every rank is given the same statement list.  Data path:
      chain (own statements)  ->  all-to-all of the w | h | v row slices (3 x 4 B x d x nb / world sent per rank)
      ->  row shares of all five ciphertexts of all statements on the matrix cores
      ->  reduce-scatter (sum) of uint64 lanes, nb x 5 x (n+1) x ceil(64 K / 56) lanes of 8 B  ->  carries, modq, delta ct_t, smudging (own statements),
    PIPELINED in stages (stage_plan: world x sper <= 255 statements, sper from every rank's slab): with device tensors on RCCL the collectives are issued
    async_op -- they run on the backend's stream -- in the order  C0 A0 | C1 A1 P0 L0 R0 | C2 A2 P1 L1 R1 F0 | ...  (C chain, A all-to-all, P row shares, L lanes,
    R reduce-scatter, F finish), so stage k + 1's all-to-all runs under the row work of stage k and stage k's reduce-scatter under the chain and row work behind it.
    Bytes handed to the backend are those of the one-shot sequence (stage=0).  When no image share is registered with the context a call of several stages expands
    the rank's shares itself, once (the row work would otherwise expand its transient image once per stage).
    witness_by_cols (default: on for a generator-defined SSP, d_ssp = None, where the witness pass is the chain's cost): the chain is cut
    in two -- every rank computes its COEFFICIENT RANGE [d r / world, d (r+1) / world) of w of the stage's statements (1 / world of the
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
    sper, nst = stage_plan(nb, world, stage)
    # No image share registered with the context: every mfh_prove_batch_partial call would expand its own transient image -- once per STAGE.  A call of several stages
    # therefore expands the rank's shares itself, once, streams them for all its stages and drops the registration at the end ("from the compressed CRS", like
    # mfh_prove_batch's transient image; the buffer is kept in `bufs`).
    own_image = None
    if nst > 1 and getattr(ctx, "_resident_mm", None) is None:
        if (p.n * p.ctb) % 8 == 0:
            own_image = ctx.crs_expand_mm_share(d_crs, rank, world, out=bufs.get("bimage"))
            ctx.set_resident_mm_share(own_image, rank, world)
            bufs["bimage"] = own_image
        else:
            sper, nst = stage_plan(nb, world, 0)  # (rows the matrix-core image cannot hold: the groups regenerate the keystream; one stage)
    lps = 5 * (p.n + 1) * p.lanes  # lanes per statement
    dev = ctx.device
    proofs = ctx.empty(count * 5 * p.ct_limbs * 8) if count else ctx.empty(0)
    if nb == 0:
        return first, count, proofs

    def cnt(q, k):  # statements of rank q's slab in stage k
        nq = owned[q][1] - owned[q][0]
        return max(0, min(sper, nq - k * sper))

    def stage_ids(k):  # the stage's statements in stage order: by rank, then by place in the rank's slab
        return [owned[q][0] + k * sper + i for q in range(world) for i in range(cnt(q, k))]

    st = [None] * nst  # per stage: what the later steps need
    chain_whv, chain_hi = None, 0  # w | h | v of all own statements, filled up to statement chain_hi by chain passes of 255
    for it in range(nst + 2):
        if it < nst:  # ---- C(it): the chain of the OWN statements of the stage, operands laid out for the all-to-all; A(it)
            k = it
            on = cnt(rank, k)
            ids = stage_ids(k)
            own_ids = [first + k * sper + i for i in range(on)]
            cnts = [cnt(q, k) for q in range(world)]
            if use_cols:
                # this rank's coefficient range of w of the stage's statements; all-to-all by statement owner; the owner finishes the chain
                wsl = ctx.batch_witness_cols(d_ssp, [witness_bits_list[i] for i in ids], [deltas[i] for i in ids], lo, cs)  # [stage statements][cs]
                wrecv = torch.empty(on * p.d, dtype=wsl.dtype, device=dev)
                all_to_all_rows(wrecv, wsl.reshape(-1), [on * (b - a) for a, b in shares], [c * cs for c in cnts], group)
                whv = torch.empty((3, on, p.d), dtype=wsl.dtype, device=dev)
                off = 0
                for a, b in shares:  # the block from rank q: [on][its range]
                    whv[0][:, a:b] = wrecv[off:off + on * (b - a)].view(on, b - a)
                    off += on * (b - a)
                del wsl, wrecv
                ctx.batch_chain_from_w(d_ssp, whv)
            else:
                # the chain runs in passes of up to 255 OWN statements (one read of the SSP, one set of NTT launches per pass: a pass per stage would read the SSP once
                # per 26 statements on 8 ranks), queued when the first stage that needs them comes up; a stage takes its slice of the pass
                o0 = k * sper
                if chain_whv is None:
                    chain_whv = torch.empty((3, count, p.d), dtype=torch.int32, device=dev)
                while chain_hi < min(count, o0 + on):  # (o0 may lie beyond a short slab: nothing to chain for this stage then)
                    c0, c1 = chain_hi, min(count, chain_hi + 255)
                    ctx.batch_chain(d_ssp, witness_bits_list[first + c0:first + c1], deltas[first + c0:first + c1], out=chain_whv[:, c0:c1, :])
                    chain_hi = c1
                whv = chain_whv[:, o0:o0 + on, :]
            # rank r gets rows [d r / world, d (r+1) / world) of w | h | v of the own statements, laid out [statement][w | h | v][rows]
            send = torch.cat([whv[:, :, a:b].permute(1, 0, 2).reshape(-1) for a, b in shares]) if on else whv.reshape(-1)
            recv = torch.empty(len(ids) * 3 * cs, dtype=send.dtype, device=dev)
            work = all_to_all_rows(recv, send, [c * 3 * cs for c in cnts], [on * 3 * (b - a) for a, b in shares], group, async_op=True)
            st[k] = {"ids": ids, "on": on, "own_ids": own_ids, "recv": recv, "send": send, "a2a": work, "sl": cnts[0]}
            del whv
        if 1 <= it <= nst:  # ---- P(it - 1): the rank's row shares of the stage's statements; L: as uint64 lanes, padded to world equal slabs; R: the reduce-scatter
            k = it - 1
            s_ = st[k]
            s_["a2a"].wait()
            nk, sl = len(s_["ids"]), s_["sl"]
            key = "bpartial%d" % (k & 1)
            partial = bufs.get(key)
            if partial is not None and partial.numel() * partial.element_size() < nk * 5 * p.ct_limbs * 8:
                partial = None  # a buffer kept from a smaller call: mfh_prove_batch_partial writes nk x 5 ciphertexts
            recv = s_["recv"]
            partial = ctx.prove_batch_partial(d_crs, rank, world, [witness_bits_list[i] for i in s_["ids"]], recv, recv[cs:], recv[2 * cs:], 3 * cs, out=partial)
            bufs[key] = partial
            lanes = torch.zeros(sl * world * lps, dtype=torch.int64, device=dev) if sl * world > nk else torch.empty(sl * world * lps, dtype=torch.int64, device=dev)
            ctx.ct_to_lanes(partial, nk * 5, out=lanes)
            own = torch.empty(sl * lps, dtype=torch.int64, device=dev)
            s_["rs"] = reduce_scatter_lanes(own, lanes, group, async_op=True)
            s_["own"], s_["lanes"] = own, lanes
            s_["recv"] = s_["send"] = None
        if it >= 2:  # ---- F(it - 2): carries + modq, then delta ct_t and the smudging of the rank's own statements
            k = it - 2
            s_ = st[k]
            s_["rs"].wait()
            on = s_["on"]
            if on:
                o0 = (k * sper) * 5 * p.ct_limbs * 8
                out = proofs[o0:o0 + on * 5 * p.ct_limbs * 8]
                ctx.ct_from_lanes(s_["own"], on * 5, out=out)
                oi = s_["own_ids"]
                ctx.prove_batch_finish(d_crs, [deltas[i] for i in oi], [smudge_mags[i] for i in oi], [smudge_signs[i] for i in oi], out, maglen)
            st[k] = None
    if own_image is not None:
        ctx.set_resident_mm_share(None, rank, world)
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
