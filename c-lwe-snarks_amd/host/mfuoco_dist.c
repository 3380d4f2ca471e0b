/*
 * mfuoco_dist.c -- libmfuoco_gpu_dist.so: the multi-GPU entry points of the host shim (host/include/mfuoco/mfuoco_dist.h).
 * One process per GPU; the collectives are RCCL calls made directly from C (ncclGroupStart + ncclSend / ncclRecv for the all-to-all,
 * ncclReduceScatter and ncclAllReduce on ncclUint64 lanes, ncclBroadcast for 413 bytes of entropy); all GPU work is libmfhip's
 * (include/mfhip.h: mfh_batch_chain, mfh_prove_batch_partial, mfh_ct_to_lanes / from_lanes, mfh_prove_batch_finish, mfh_witness_lanes,
 * mfh_prove_partial_w, mfh_prove_finish).  The reference has no counterpart: its prover() (src/snark.c:117-190) is one thread on one
 * device; the loops split here are src/snark.c:147-155 and :157-174.  c-lwe-snarks_amd/dist.py is the same sequence over
 * torch.distributed and is what bench.py drives; the split arithmetic below is the same (row_shares, statement_shares).
 *
 * Streams: kernels and device copies run on the shim's stream (the NULL stream); the collectives of the batch prover run on the communicator's OWN non-blocking
 * stream, ordered with the kernels by events -- stage k's all-to-all under the row work of stage k - 1, its reduce-scatter under the row work of stage k + 1
 * (mfuoco_prover_batch_sharded below).  The single-proof sequence (two small all-reduces) stays on the NULL stream.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <gmp.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "mfhip.h"
#include "mfuoco/mfuoco_dist.h"
#include "mfuoco_rendezvous.h"

/* shared with host/mfuoco_gpu.c (libmfuoco_gpu) */
mfh_ctx *mfuoco_gpu_ctx(void);
const uint8_t *mfuoco_gpu_stage_crs(crs_t crs);
const uint32_t *mfuoco_gpu_stage_ssp(ssp_t ssp);
size_t mfuoco_gpu_bits_stride(void);
void mfuoco_gpu_witness_bits(uint8_t *bits, mpz_t witness);
void mfuoco_gpu_prover_entropy(uint32_t *delta, uint8_t *mag, uint8_t *sign);
void mfuoco_gpu_prover_entropy_batch(uint32_t *delta, uint8_t *mag, uint8_t *sign, size_t count);
void mfuoco_gpu_proofs_to_host(proof_t *pis, const uint64_t *d_proofs, size_t count);
void mfuoco_gpu_proofs_to_host_after(proof_t *pis, const uint64_t *d_proofs, size_t count, void *hip_event);
int mfuoco_gpu_image_resident_share(const uint8_t *d_crs, uint32_t rank, uint32_t world, int expand);

#define L_LIMBS 12
#define K_LIMBS 11
#define CTL ((size_t)(GAMMA_N + 1) * L_LIMBS)
#define LANES_PER_CT ((size_t)(GAMMA_N + 1) * ((64 * K_LIMBS + 55) / 56)) /* uint64 lanes of 56 bits: 13 per 704-bit value (mfh_lanes_per_value) */
#define MAGLEN (GAMMA_LOG_SMUDGING / 8)
#define MAXW 64
#define NBUF 16

enum { ST_A2A = 0, ST_RS = 1, ST_AR = 2, ST_BC = 3 };

/* The communicator: rank / world / device, the transport that carries the four collectives (RCCL here; mfuoco_comm_create_transport for a caller's own),
 * call statistics and device scratch.  The product library contains the RCCL transport only. */
struct mfuoco_comm {
  int rank, world, device;
  const mfuoco_transport *t; /* NULL: no backend (comm == NULL callers, world 1) */
  void *impl;
  uint64_t calls[4], bytes[4];
  void *buf[NBUF]; /* device scratch, grown on demand */
  size_t cap[NBUF];
  /* the batch prover's pipeline: the collectives' stream and the events that order it with the shim's stream (made on first use) */
  hipStream_t cstream;
  hipEvent_t ev_sent[2], ev_recv[2], ev_lanes[2], ev_own[2]; /* by stage parity: operands packed / all-to-all done / lanes written / reduce-scatter done */
  hipEvent_t *ev_done;                                        /* per stage: the stage's proofs are final */
  size_t n_done;
};

static void dist_die(const char *what, const char *detail)
{
  fprintf(stderr, "libmfuoco_gpu_dist: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
  abort();
}
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) dist_die(#call, hipGetErrorString(e_)); } while (0)
#define NK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) dist_die(#call, ncclGetErrorString(r_)); } while (0)
#define CK(call) do { if ((call) != MFH_OK) dist_die(#call, mfh_last_error(mfuoco_gpu_ctx())); } while (0)

static void *scratch(mfuoco_comm *c, int slot, size_t bytes)
{
  if (bytes > c->cap[slot]) {
    if (c->buf[slot]) HK(hipFree(c->buf[slot]));
    HK(hipMalloc(&c->buf[slot], bytes ? bytes : 8));
    c->cap[slot] = bytes;
  }
  return c->buf[slot];
}
static void *xmalloc(size_t bytes)
{
  void *p = malloc(bytes ? bytes : 1);
  if (!p) dist_die("out of host memory", NULL);
  return p;
}

static void comm_release(mfuoco_comm *c)
{
  for (int i = 0; i < NBUF; i++) {
    if (c->buf[i]) hipFree(c->buf[i]);
    c->buf[i] = NULL;
    c->cap[i] = 0;
  }
  if (c->cstream) {
    for (int i = 0; i < 2; i++) { hipEventDestroy(c->ev_sent[i]); hipEventDestroy(c->ev_recv[i]); hipEventDestroy(c->ev_lanes[i]); hipEventDestroy(c->ev_own[i]); }
    hipStreamDestroy(c->cstream);
    c->cstream = NULL;
  }
  for (size_t i = 0; i < c->n_done; i++) hipEventDestroy(c->ev_done[i]);
  free(c->ev_done);
  c->ev_done = NULL;
  c->n_done = 0;
}
static void comm_pipeline(mfuoco_comm *c, size_t stages)
{
  if (!c->cstream) {
    HK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
      HK(hipEventCreateWithFlags(&c->ev_sent[i], hipEventDisableTiming));
      HK(hipEventCreateWithFlags(&c->ev_recv[i], hipEventDisableTiming));
      HK(hipEventCreateWithFlags(&c->ev_lanes[i], hipEventDisableTiming));
      HK(hipEventCreateWithFlags(&c->ev_own[i], hipEventDisableTiming));
    }
  }
  if (stages > c->n_done) {
    c->ev_done = realloc(c->ev_done, stages * sizeof *c->ev_done);
    if (!c->ev_done) dist_die("out of host memory", NULL);
    for (size_t i = c->n_done; i < stages; i++) HK(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
    c->n_done = stages;
  }
}

/* ---- split arithmetic (dist.py: row_shares, statement_shares) ------------------------------------------------------------ */
static size_t row_lo(size_t total, int r, int world) { return total * (size_t)r / (size_t)world; }
static size_t stmt_per(size_t nb, int world) { return nb ? (nb + world - 1) / world : 0; }
static size_t stmt_lo(size_t nb, int r, int world)
{
  size_t lo = (size_t)r * stmt_per(nb, world);
  return lo < nb ? lo : nb;
}

/* ---- the RCCL transport ---------------------------------------------------------------------------------------------------- */
static void rccl_alltoallv_u32(void *impl, int rank, int world, const uint32_t *d_send, const size_t *scnt, const size_t *sdsp, uint32_t *d_recv,
                               const size_t *rcnt, const size_t *rdsp, void *stream)
{
  (void)rank;
  ncclComm_t nc = impl;
  NK(ncclGroupStart());
  for (int q = 0; q < world; q++) {
    if (scnt[q]) NK(ncclSend(d_send + sdsp[q], scnt[q], ncclUint32, q, nc, (hipStream_t)stream));
    if (rcnt[q]) NK(ncclRecv(d_recv + rdsp[q], rcnt[q], ncclUint32, q, nc, (hipStream_t)stream));
  }
  NK(ncclGroupEnd());
}
static void rccl_reduce_scatter_u64(void *impl, int rank, int world, const uint64_t *d_send, uint64_t *d_recv, size_t n, void *stream)
{
  (void)rank; (void)world;
  NK(ncclReduceScatter(d_send, d_recv, n, ncclUint64, ncclSum, (ncclComm_t)impl, (hipStream_t)stream));
}
static void rccl_allreduce_u64(void *impl, int rank, int world, uint64_t *d_buf, size_t n, void *stream)
{
  (void)rank; (void)world;
  NK(ncclAllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, (ncclComm_t)impl, (hipStream_t)stream));
}
static void rccl_bcast_bytes(void *impl, int rank, int world, uint8_t *d_buf, size_t n, int root, void *stream)
{
  (void)rank; (void)world;
  NK(ncclBroadcast(d_buf, d_buf, n, ncclUint8, root, (ncclComm_t)impl, (hipStream_t)stream));
}
static void rccl_destroy(void *impl) { ncclCommDestroy((ncclComm_t)impl); }
static const mfuoco_transport rccl_transport = { "rccl", rccl_alltoallv_u32, rccl_reduce_scatter_u64, rccl_allreduce_u64, rccl_bcast_bytes, rccl_destroy };

/* ---- communicator ---------------------------------------------------------------------------------------------------------- */
/* The shim's GPU and the communicator's must be the same one: every mfh_* call switches the current device to the shim's, so scratch and collectives
 * on another device would end in a wrong-device collective or a hang.  Checked here instead of trusted. */
static int bind_device(int device)
{
  if (mfuoco_gpu_set_device(device)) return -1;
  if (hipSetDevice(device) != hipSuccess) {
    fprintf(stderr, "libmfuoco_gpu_dist: no HIP device %d\n", device);
    return -1;
  }
  return 0;
}
int mfuoco_comm_create_transport(mfuoco_comm **out, int rank, int world, int device, const mfuoco_transport *t, void *impl)
{
  if (!out || world < 1 || world > MAXW || rank < 0 || rank >= world) {
    fprintf(stderr, "libmfuoco_gpu_dist: bad rank %d / world %d (at most %d ranks)\n", rank, world, MAXW);
    return -1;
  }
  if (t && (!t->alltoallv_u32 || !t->reduce_scatter_u64 || !t->allreduce_u64 || !t->bcast_bytes)) {
    fprintf(stderr, "libmfuoco_gpu_dist: incomplete transport\n");
    return -1;
  }
  if (bind_device(device)) return -1;
  mfuoco_comm *c = calloc(1, sizeof *c);
  if (!c) {
    fprintf(stderr, "libmfuoco_gpu_dist: out of host memory\n");
    return -1;
  }
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->t = t;
  c->impl = impl;
  *out = c;
  return 0;
}

void mfuoco_comm_unique_id(uint8_t id[MFUOCO_UNIQUE_ID_BYTES])
{
  ncclUniqueId u;
  NK(ncclGetUniqueId(&u));
  memcpy(id, u.internal, MFUOCO_UNIQUE_ID_BYTES);
}

int mfuoco_comm_create_from_id(mfuoco_comm **out, int rank, int world, int device, const uint8_t id[MFUOCO_UNIQUE_ID_BYTES])
{
  if (world < 1 || world > MAXW || rank < 0 || rank >= world) {
    fprintf(stderr, "libmfuoco_gpu_dist: bad rank %d / world %d (at most %d ranks)\n", rank, world, MAXW);
    return -1;
  }
  if (bind_device(device)) return -1;
  ncclUniqueId u;
  memcpy(u.internal, id, MFUOCO_UNIQUE_ID_BYTES);
  ncclComm_t nc;
  ncclResult_t r = ncclCommInitRank(&nc, world, u, rank);
  if (r != ncclSuccess) {
    fprintf(stderr, "libmfuoco_gpu_dist: ncclCommInitRank(rank %d of %d): %s\n", rank, world, ncclGetErrorString(r));
    return -1;
  }
  if (mfuoco_comm_create_transport(out, rank, world, device, &rccl_transport, nc)) {
    ncclCommDestroy(nc);
    return -1;
  }
  return 0;
}

int mfuoco_comm_create(mfuoco_comm **out, int rank, int world, int device, const char *id_file)
{
  if (!out || !id_file || world < 1 || world > MAXW || rank < 0 || rank >= world) {
    fprintf(stderr, "libmfuoco_gpu_dist: bad arguments (rank %d, world %d, at most %d ranks)\n", rank, world, MAXW);
    return -1;
  }
  if (bind_device(device)) return -1;
  uint8_t id[MFUOCO_UNIQUE_ID_BYTES];
  memset(id, 0, sizeof id);
  if (rank == 0) mfuoco_comm_unique_id(id);
  if (world > 1) {
    const char *te = getenv("MFUOCO_RENDEZVOUS_S");
    if (mfuoco_rendezvous_files(rank, world, id_file, id, te && atof(te) > 0 ? atof(te) : 120.0)) return -1;
  }
  int rc = mfuoco_comm_create_from_id(out, rank, world, device, id);
  /* every rank is through ncclCommInitRank now (it returns when all have joined): the files have done their work */
  if (world > 1) mfuoco_rendezvous_cleanup(rank, id_file);
  return rc;
}

void mfuoco_comm_destroy(mfuoco_comm *c)
{
  if (!c) return;
  hipDeviceSynchronize();
  comm_release(c);
  if (c->t && c->t->destroy) c->t->destroy(c->impl);
  free(c);
}

int mfuoco_comm_rank(const mfuoco_comm *c) { return c ? c->rank : 0; }
int mfuoco_comm_world(const mfuoco_comm *c) { return c ? c->world : 1; }
const char *mfuoco_comm_backend(const mfuoco_comm *c) { return c && c->t ? c->t->name : "none"; }
void mfuoco_comm_stats(const mfuoco_comm *c, uint64_t calls[4], uint64_t bytes[4])
{
  for (int i = 0; i < 4; i++) {
    calls[i] = c ? c->calls[i] : 0;
    bytes[i] = c ? c->bytes[i] : 0;
  }
}

/* ---- collectives (device buffers; counts in elements) ---------------------------------------------------------------------- */
/* all-to-all of uint32 words: scnt[q] words from d_send + sdsp[q] go to rank q; rcnt[q] words from rank q land at d_recv + rdsp[q] */
static void comm_alltoallv_u32(mfuoco_comm *c, const uint32_t *d_send, const size_t *scnt, const size_t *sdsp, uint32_t *d_recv, const size_t *rcnt,
                               const size_t *rdsp, hipStream_t stream)
{
  size_t total = 0;
  for (int q = 0; q < c->world; q++) total += scnt[q];
  c->calls[ST_A2A]++;
  c->bytes[ST_A2A] += total * 4;
  c->t->alltoallv_u32(c->impl, c->rank, c->world, d_send, scnt, sdsp, d_recv, rcnt, rdsp, stream);
}
/* d_recv[0 .. n) = sum over ranks of their d_send[rank * n .. (rank + 1) * n)   (uint64 lanes, wrap-around sum) */
static void comm_reduce_scatter_u64(mfuoco_comm *c, const uint64_t *d_send, uint64_t *d_recv, size_t n, hipStream_t stream)
{
  c->calls[ST_RS]++;
  c->bytes[ST_RS] += n * c->world * 8;
  c->t->reduce_scatter_u64(c->impl, c->rank, c->world, d_send, d_recv, n, stream);
}
static void comm_allreduce_u64(mfuoco_comm *c, uint64_t *d_buf, size_t n)
{
  c->calls[ST_AR]++;
  c->bytes[ST_AR] += n * 8;
  c->t->allreduce_u64(c->impl, c->rank, c->world, d_buf, n, NULL);
}
static void comm_bcast_bytes(mfuoco_comm *c, uint8_t *d_buf, size_t n, int root)
{
  c->calls[ST_BC]++;
  c->bytes[ST_BC] += n;
  c->t->bcast_bytes(c->impl, c->rank, c->world, d_buf, n, root, NULL);
}

/* ---- the row-sharded batch prover ------------------------------------------------------------------------------------------- */
/* A call is cut into STAGES.  Stage k holds, from every rank, the statements k * sper .. (k + 1) * sper - 1 of that rank's slab (sper = 255 / world, so a stage is at most one
 * super-group of mfh_prove_batch_partial: one pass over the rank's share of the image per stage), in rank order -- ownership and results are those of the one-shot sequence
 * (sums mod 2^(64K) do not depend on the order, and a proof does not depend on which statements share its launches).  Per stage, the five steps of the header:
 *     C  chain of the OWN statements (passes of 255, queued when a stage first needs them), the stage's operands packed            shim stream
 *     A  all-to-all of the w | h | v row slices                               communicator stream, after C
 *     P  the rank's row shares of the stage's statements; L  lanes            shim stream, after A
 *     R  reduce-scatter of the lanes                                          communicator stream, after L
 *     F  carries, modq, delta ct_t, smudging of the own statements            shim stream, after R
 *     D  the drain: device to host on the shim's copy stream, mpz_t's on host threads   after F, an iteration later (the call blocks in it, with work queued behind)
 * queued as   C0 | C1 P0 L0 | C2 P1 L1 F0 | C3 P2 L2 F1 [D0] | ...   on the shim's stream and   A0 | A1 R0 | A2 R1 | ...   on the communicator's, so that A(k + 1) runs under
 * P(k), R(k) under C(k + 2) and P(k + 1), and the host turns stage k into mpz_t's under the kernels of the stages behind it.  (One rank: F(k) directly behind R(k).)  Everything is double-buffered by stage parity;
 * the order above is what makes that safe (a buffer's next writer is queued behind an event its last reader precedes).  The cut depends on count, world and the
 * environment only, never on what a rank happens to have resident: every rank issues the same collectives. */
void mfuoco_prover_batch_sharded(proof_t *pis, crs_t crs, ssp_t ssp, mpz_t *witnesses, size_t count, mfuoco_comm *comm, size_t *own_first,
                                 size_t *own_count)
{
  mfuoco_comm local = { .rank = 0, .world = 1 };
  mfuoco_comm *c = comm ? comm : &local;
  const int rank = c->rank, world = c->world;
  const size_t nb = count, d = GAMMA_D;
  const size_t per = stmt_per(nb, world), first = stmt_lo(nb, rank, world), last = stmt_lo(nb, rank + 1, world), nown = last - first;
  if (own_first) *own_first = first;
  if (own_count) *own_count = nown;
  if (!nb) return;
  mfh_ctx *ctx = mfuoco_gpu_ctx();
  const uint8_t *d_crs = mfuoco_gpu_stage_crs(crs);
  const uint32_t *d_ssp = mfuoco_gpu_stage_ssp(ssp);
  const size_t stride = mfuoco_gpu_bits_stride();
  const size_t lo = row_lo(d, rank, world), cs = row_lo(d, rank + 1, world) - lo;
  (void)lo;
  if (world > 1 && !c->t) dist_die("mfuoco_prover_batch_sharded", "more than one rank needs a communicator with a transport");

  /* the rank's share of the image, kept across calls while the CRS is the same; stages only when the row work streams it */
  const int streamed = mfuoco_gpu_image_resident_share(d_crs, (uint32_t)rank, (uint32_t)world, nb > 31);
  const char *se = getenv("MFUOCO_DIST_STAGE"); /* statements per rank and stage (tests, tuning); 0 = one stage */
  /* the fewest equally long stages of at most 255 statements (one super-group of the row work, one pass over the image share) each: 1020 statements on 8 ranks are 5 stages
   * of 26 per rank, 255 on 8 ranks one stage (dist.py: stage_plan) */
  size_t sper = per;
  for (size_t ns = (nb + 254) / 255;; ns++) {
    sper = (per + ns - 1) / ns;
    size_t in_stage = 0;
    for (int q = 0; q < world; q++) {
      const size_t nq = stmt_lo(nb, q + 1, world) - stmt_lo(nb, q, world);
      in_stage += nq < sper ? nq : sper;
    }
    if (in_stage <= 255 || sper == 1) break;
  }
  if (se && *se) sper = (size_t)atol(se);
  if (!sper || sper > per) sper = per;
  const size_t nst = (per + sper - 1) / sper; /* (a function of count, world and the environment only: every rank cuts the call the same way) */
  comm_pipeline(c, nst);
  /* No image share kept by the shim ($MFUOCO_GPU_RESIDENT_CRS=0, or no room beside the scratch): mfh_prove_batch_partial would expand its own transient image -- once per
   * STAGE.  A call of several stages therefore expands the rank's shares itself, once, streams them for all its stages and drops the registration at the end.  (When even
   * that does not fit, the row work expands or regenerates per stage: slower, same proofs, same collectives.) */
  bool own_image = false;
  if (!streamed && nst > 1 && nb > 31) {
    const size_t ib = mfh_crs_mm_share_bytes(ctx, (uint32_t)rank, (uint32_t)world);
    size_t fr = 0, tot = 0;
    if (ib <= c->cap[13] || (hipMemGetInfo(&fr, &tot) == hipSuccess && ib + ((size_t)8 << 30) <= fr)) {
      uint8_t *img = scratch(c, 13, ib);
      CK(mfh_crs_expand_mm_share(ctx, d_crs, (uint32_t)rank, (uint32_t)world, img));
      CK(mfh_crs_set_resident_mm_share(ctx, img, (uint32_t)rank, (uint32_t)world));
      own_image = true;
    }
  }
  hipStream_t const cst = c->t ? c->cstream : NULL; /* (no backend: the stand-in copies run in line) */

  /* host inputs: the bit strings of ALL statements in stage order (they select the rank's BT+BV rows), entropy of the OWN statements in the order of nown prover() calls */
  uint8_t *bits = xmalloc(nb * stride), *mag = xmalloc((nown ? nown : 1) * 5 * MAGLEN), *sign = xmalloc((nown ? nown : 1) * 5);
  uint32_t *delta = xmalloc((nown ? nown : 1) * 4);
  memset(bits, 0, nb * stride);
  size_t *soff = xmalloc((nst + 1) * sizeof *soff); /* first statement of stage k in stage order */
  {
    size_t pos = 0;
    for (size_t k = 0; k < nst; k++) {
      soff[k] = pos;
      for (int q = 0; q < world; q++) {
        const size_t fq = stmt_lo(nb, q, world), nq = stmt_lo(nb, q + 1, world) - fq;
        for (size_t i = k * sper; i < (k + 1) * sper && i < nq; i++) mfuoco_gpu_witness_bits(bits + pos++ * stride, witnesses[fq + i]);
      }
    }
    soff[nst] = pos;
  }
  if (nown) mfuoco_gpu_prover_entropy_batch(delta, mag, sign, nown);
  /* ... and the own statements' bit strings once more in slab order: the chain runs in passes of up to 255 own statements (one read of the SSP each), whatever the stages */
  uint8_t *obits = xmalloc((nown ? nown : 1) * stride);
  memset(obits, 0, (nown ? nown : 1) * stride);
  for (size_t i = 0; i < nown; i++) mfuoco_gpu_witness_bits(obits + i * stride, witnesses[first + i]);
  size_t chained = 0; /* own statements whose chain has been queued */

  /* device buffers, all of them before anything is queued (growing one frees it, and hipFree waits for the device) */
  const size_t lps = 5 * LANES_PER_CT, smax = sper * (size_t)world;
  uint32_t *send[2], *recv[2];
  uint64_t *partial[2], *lanes[2], *own[2];
  uint32_t *whv = scratch(c, 0, 3 * (nown ? nown : 1) * d * 4); /* w | h | v of ALL own statements: [kind][own statement][d] */
  for (int b = 0; b < 2; b++) {
    send[b] = scratch(c, 2 + b, 3 * sper * d * 4);
    recv[b] = scratch(c, 4 + b, smax * 3 * cs * 4);
    partial[b] = scratch(c, 6 + b, smax * 5 * CTL * 8);
    lanes[b] = scratch(c, 8 + b, smax * lps * 8);
    own[b] = scratch(c, 10 + b, sper * lps * 8);
  }
  uint64_t *proofs = scratch(c, 12, (nown ? nown : 1) * 5 * CTL * 8);

  /* cnt(q, k): how many statements of rank q's slab stage k holds (own indices k * sper ...); rank 0's is the largest: the slab of the stage's reduce-scatter */
#define STAGE_CNT(q, k) ({ const size_t nq_ = stmt_lo(nb, (q) + 1, world) - stmt_lo(nb, (q), world), lo_ = (k) * sper; nq_ > lo_ ? (nq_ - lo_ < sper ? nq_ - lo_ : sper) : (size_t)0; })
  size_t scnt[MAXW], sdsp[MAXW], rcnt[MAXW], rdsp[MAXW];
  /* F(k) is queued f_lag iterations after C(k): with several ranks behind the row work of stage k + 1, so that the reduce-scatter of stage k has that long to complete
   * before the shim's stream waits for it; with ONE rank the "reduce-scatter" is a device copy and nothing is gained by waiting -- F(k) follows it at once, and the
   * stage's proofs leave for the host a whole stage earlier (one rank drains every proof of the call: 706 KB each) */
  const size_t f_lag = world > 1 ? 2 : 1;
  size_t next_drain = 0;
  for (size_t it = 0; it < nst + f_lag; it++) {
    /* ---- C(it): chain of the own statements of stage `it` (src/snark.c:141-169), operands packed; A(it): the all-to-all */
    if (it < nst) {
      const size_t k = it, on = STAGE_CNT(rank, k), olo = k * sper;
      const int b = (int)(k & 1);
      /* the chain passes that cover this stage's own statements: 255 statements per pass (one read of the SSP, one set of NTT launches: a pass per STAGE would read
       * the SSP once per 26 statements on 8 ranks), queued when the first stage that needs them comes up */
      while (chained < nown && chained < olo + on) { /* (olo may lie beyond a short slab: nothing to chain for this stage then) */
        const size_t np = nown - chained < 255 ? nown - chained : 255;
        CK(mfh_batch_chain(ctx, d_ssp, (uint32_t)np, obits + chained * stride, stride, delta + chained, whv + chained * d, whv + (nown + chained) * d,
                           whv + (2 * nown + chained) * d));
        chained += np;
      }
      /* rank q gets rows [d q / world, d (q+1) / world) of w | h | v of the own statements, laid out [statement][w | h | v][rows]; this rank receives its rows of the
       * stage's statements of every rank, in stage order */
      size_t off = 0, at = 0;
      for (int q = 0; q < world; q++) {
        const size_t a = row_lo(d, q, world), w = row_lo(d, q + 1, world) - a, nq = STAGE_CNT(q, k);
        sdsp[q] = off;
        scnt[q] = on * 3 * w;
        if (scnt[q])
          for (int x = 0; x < 3; x++) /* a strided column slice per polynomial kind: one 2-D device copy */
            HK(hipMemcpy2DAsync(send[b] + off + (size_t)x * w, 3 * w * 4, whv + ((size_t)x * nown + olo) * d + a, d * 4, w * 4, on, hipMemcpyDeviceToDevice, NULL));
        off += scnt[q];
        rdsp[q] = at * 3 * cs;
        rcnt[q] = nq * 3 * cs;
        at += nq;
      }
      if (c->t) {
        HK(hipEventRecord(c->ev_sent[b], NULL));
        HK(hipStreamWaitEvent(cst, c->ev_sent[b], 0));
        comm_alltoallv_u32(c, send[b], scnt, sdsp, recv[b], rcnt, rdsp, cst);
        HK(hipEventRecord(c->ev_recv[b], cst));
      } else HK(hipMemcpyAsync(recv[b], send[b], scnt[0] * 4, hipMemcpyDeviceToDevice, NULL));
    }
    /* ---- P(it - 1), L(it - 1): the rank's row shares of the five ciphertexts of the stage's statements (matrix cores; no delta ct_t term, un-smudged), as uint64 lanes
     * of 56 bits padded to `world` equal slabs; R(it - 1): the reduce-scatter -- the own slab, summed */
    if (it >= 1 && it - 1 < nst) {
      const size_t k = it - 1, nk = soff[k + 1] - soff[k], sl = STAGE_CNT(0, k);
      const int b = (int)(k & 1);
      if (c->t) HK(hipStreamWaitEvent(NULL, c->ev_recv[b], 0));
      CK(mfh_prove_batch_partial(ctx, d_crs, (uint32_t)rank, (uint32_t)world, (uint32_t)nk, bits + soff[k] * stride, stride, recv[b], recv[b] + cs, recv[b] + 2 * cs, 3 * cs,
                                 partial[b]));
      if (sl * (size_t)world > nk) HK(hipMemsetAsync(lanes[b] + nk * lps, 0, (sl * (size_t)world - nk) * lps * 8, NULL));
      CK(mfh_ct_to_lanes(ctx, partial[b], nk * 5, lanes[b]));
      if (c->t) {
        HK(hipEventRecord(c->ev_lanes[b], NULL));
        HK(hipStreamWaitEvent(cst, c->ev_lanes[b], 0));
        comm_reduce_scatter_u64(c, lanes[b], own[b], sl * lps, cst);
        HK(hipEventRecord(c->ev_own[b], cst));
      } else HK(hipMemcpyAsync(own[b], lanes[b], sl * lps * 8, hipMemcpyDeviceToDevice, NULL));
    }
    /* ---- F(it - f_lag): carries + modq, then b_w += delta ct_t and the smudging of the own statements (src/snark.c:143-145,185-189) */
    if (it >= f_lag) {
      const size_t k = it - f_lag, on = STAGE_CNT(rank, k), olo = k * sper;
      const int b = (int)(k & 1);
      if (c->t) HK(hipStreamWaitEvent(NULL, c->ev_own[b], 0));
      if (on) {
        CK(mfh_ct_from_lanes(ctx, own[b], on * 5, proofs + olo * 5 * CTL));
        CK(mfh_prove_batch_finish(ctx, d_crs, (uint32_t)on, delta + olo, mag + olo * 5 * MAGLEN, MAGLEN, sign + olo * 5, proofs + olo * 5 * CTL));
      }
      HK(hipEventRecord(c->ev_done[k], NULL));
    }
    /* ---- D(it - f_lag - 1), the drain: the stage crosses PCIe (the shim's own copy stream, behind its ev_done only) and becomes mpz_t's on the host threads while
     * the GPU runs what this iteration has just queued.  One stage further back than F: the call blocks here for the copy, and a drain of the stage whose F was queued
     * a moment ago would wait with nothing behind it for the GPU to go on with. */
    if (it >= f_lag + 1) {
      const size_t k = next_drain++, on = STAGE_CNT(rank, k), olo = k * sper;
      if (on) mfuoco_gpu_proofs_to_host_after(pis + first + olo, proofs + olo * 5 * CTL, on, c->ev_done[k]);
    }
  }
  while (next_drain < nst) { /* ... and the last stage, which nothing is left to hide behind */
    const size_t k = next_drain++, on = STAGE_CNT(rank, k), olo = k * sper;
    if (on) mfuoco_gpu_proofs_to_host_after(pis + first + olo, proofs + olo * 5 * CTL, on, c->ev_done[k]);
  }
#undef STAGE_CNT
  HK(hipStreamSynchronize(NULL));
  if (c->t) HK(hipStreamSynchronize(cst)); /* (a rank without own statements has waited for nothing so far) */
  if (own_image) CK(mfh_crs_set_resident_mm_share(ctx, NULL, (uint32_t)rank, (uint32_t)world));
  if (c == &local) comm_release(&local);
  CK(mfh_scrub_staging(ctx)); /* (as in prover(): every copy of the call has run; witness bits, deltas and smudging terms leave the context's pinned staging) */
  explicit_bzero(mag, (nown ? nown : 1) * 5 * MAGLEN); /* smudging terms and deltas are the proofs' zero-knowledge: not left on the heap */
  explicit_bzero(sign, (nown ? nown : 1) * 5);
  explicit_bzero(delta, (nown ? nown : 1) * 4);
  free(bits); free(obits); free(mag); free(sign); free(delta); free(soff);
}

/* ---- one proof, rows sharded (dist.py: prove_sharded) ------------------------------------------------------------------------ */
void mfuoco_prover_sharded(proof_t pi, crs_t crs, ssp_t ssp, mpz_t witness, mfuoco_comm *comm)
{
  mfuoco_comm local = { .rank = 0, .world = 1 };
  mfuoco_comm *c = comm ? comm : &local;
  mfh_ctx *ctx = mfuoco_gpu_ctx();
  const uint8_t *d_crs = mfuoco_gpu_stage_crs(crs);
  const uint32_t *d_ssp = mfuoco_gpu_stage_ssp(ssp);
  uint8_t bits[(GAMMA_M + 7) / 8 + 8] = { 0 };
  mfuoco_gpu_witness_bits(bits, witness);
  /* every rank needs the same delta (w = delta t + ...) and the same smudging: rank 0 draws, the others receive */
  struct { uint32_t delta; uint8_t mag[5 * MAGLEN], sign[5]; } ent;
  memset(&ent, 0, sizeof ent);
  if (c->rank == 0) mfuoco_gpu_prover_entropy(&ent.delta, ent.mag, ent.sign);
  if (c->t) {
    uint8_t *d_ent = scratch(c, 0, sizeof ent);
    HK(hipMemcpy(d_ent, &ent, sizeof ent, hipMemcpyHostToDevice));
    comm_bcast_bytes(c, d_ent, sizeof ent, 0);
    HK(hipMemcpy(&ent, d_ent, sizeof ent, hipMemcpyDeviceToHost));
  }
  uint64_t *wl = scratch(c, 1, (size_t)GAMMA_D * 8), *partial = scratch(c, 2, 5 * CTL * 8), *lanes = scratch(c, 3, 5 * LANES_PER_CT * 8),
           *proof = scratch(c, 4, 5 * CTL * 8);
  CK(mfh_witness_lanes(ctx, d_ssp, bits, (uint32_t)c->rank, (uint32_t)c->world, wl));
  if (c->t) comm_allreduce_u64(c, wl, GAMMA_D);
  CK(mfh_prove_partial_w(ctx, d_crs, d_ssp, bits, ent.delta, (uint32_t)c->rank, (uint32_t)c->world, wl, partial));
  CK(mfh_ct_to_lanes(ctx, partial, 5, lanes));
  if (c->t) comm_allreduce_u64(c, lanes, 5 * LANES_PER_CT);
  CK(mfh_ct_from_lanes(ctx, lanes, 5, proof));
  CK(mfh_prove_finish(ctx, proof, ent.mag, MAGLEN, ent.sign));
  explicit_bzero(&ent, sizeof ent);
  proof_t *one = (proof_t *)pi;
  mfuoco_gpu_proofs_to_host(one, proof, 1);
  CK(mfh_scrub_staging(ctx)); /* (the proof is out: every copy of the call has run) */
  if (c == &local) comm_release(&local);
}
