/*
 * mfuoco_dist.c -- libmfuoco_gpu_dist.so: the multi-GPU entry points of the host shim (host/include/mfuoco/mfuoco_dist.h).
 * One process per GPU; the collectives are RCCL calls made directly from C (ncclGroupStart + ncclSend / ncclRecv for the all-to-all,
 * ncclReduceScatter and ncclAllReduce on ncclUint64 lanes, ncclBroadcast for 413 bytes of entropy); all GPU work is libmfhip's
 * (include/mfhip.h: mfh_batch_chain, mfh_prove_batch_partial, mfh_ct_to_lanes / from_lanes, mfh_prove_batch_finish, mfh_witness_lanes,
 * mfh_prove_partial_w, mfh_prove_finish).  The reference has no counterpart: its prover() (src/snark.c:117-190) is one thread on one
 * device; the loops split here are src/snark.c:147-155 and :157-174.  c-lwe-snarks_amd/dist.py is the same sequence over
 * torch.distributed and is what bench.py drives; the split arithmetic below is the same (row_shares, statement_shares).
 *
 * Everything runs on the NULL stream (the shim's stream), so kernels, copies and collectives are ordered without events.
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
void mfuoco_gpu_image_resident_share(const uint8_t *d_crs, uint32_t rank, uint32_t world, int expand);

#define L_LIMBS 12
#define K_LIMBS 11
#define CTL ((size_t)(GAMMA_N + 1) * L_LIMBS)
#define LANES_PER_CT ((size_t)(GAMMA_N + 1) * ((64 * K_LIMBS + 55) / 56)) /* uint64 lanes of 56 bits: 13 per 704-bit value (mfh_lanes_per_value) */
#define MAGLEN (GAMMA_LOG_SMUDGING / 8)
#define MAXW 64

enum { ST_A2A = 0, ST_RS = 1, ST_AR = 2, ST_BC = 3 };

/* The communicator: rank / world / device, the transport that carries the four collectives (RCCL here; mfuoco_comm_create_transport for a caller's own),
 * call statistics and device scratch.  The product library contains the RCCL transport only. */
struct mfuoco_comm {
  int rank, world, device;
  const mfuoco_transport *t; /* NULL: no backend (comm == NULL callers, world 1) */
  void *impl;
  uint64_t calls[4], bytes[4];
  void *buf[8]; /* device scratch, grown on demand */
  size_t cap[8];
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
                               const size_t *rcnt, const size_t *rdsp)
{
  (void)rank;
  ncclComm_t nc = impl;
  NK(ncclGroupStart());
  for (int q = 0; q < world; q++) {
    if (scnt[q]) NK(ncclSend(d_send + sdsp[q], scnt[q], ncclUint32, q, nc, NULL));
    if (rcnt[q]) NK(ncclRecv(d_recv + rdsp[q], rcnt[q], ncclUint32, q, nc, NULL));
  }
  NK(ncclGroupEnd());
}
static void rccl_reduce_scatter_u64(void *impl, int rank, int world, const uint64_t *d_send, uint64_t *d_recv, size_t n)
{
  (void)rank; (void)world;
  NK(ncclReduceScatter(d_send, d_recv, n, ncclUint64, ncclSum, (ncclComm_t)impl, NULL));
}
static void rccl_allreduce_u64(void *impl, int rank, int world, uint64_t *d_buf, size_t n)
{
  (void)rank; (void)world;
  NK(ncclAllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, (ncclComm_t)impl, NULL));
}
static void rccl_bcast_bytes(void *impl, int rank, int world, uint8_t *d_buf, size_t n, int root)
{
  (void)rank; (void)world;
  NK(ncclBroadcast(d_buf, d_buf, n, ncclUint8, root, (ncclComm_t)impl, NULL));
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
  for (int i = 0; i < 8; i++)
    if (c->buf[i]) hipFree(c->buf[i]);
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
                               const size_t *rdsp)
{
  size_t total = 0;
  for (int q = 0; q < c->world; q++) total += scnt[q];
  c->calls[ST_A2A]++;
  c->bytes[ST_A2A] += total * 4;
  c->t->alltoallv_u32(c->impl, c->rank, c->world, d_send, scnt, sdsp, d_recv, rcnt, rdsp);
}
/* d_recv[0 .. n) = sum over ranks of their d_send[rank * n .. (rank + 1) * n)   (uint64 lanes, wrap-around sum) */
static void comm_reduce_scatter_u64(mfuoco_comm *c, const uint64_t *d_send, uint64_t *d_recv, size_t n)
{
  c->calls[ST_RS]++;
  c->bytes[ST_RS] += n * c->world * 8;
  c->t->reduce_scatter_u64(c->impl, c->rank, c->world, d_send, d_recv, n);
}
static void comm_allreduce_u64(mfuoco_comm *c, uint64_t *d_buf, size_t n)
{
  c->calls[ST_AR]++;
  c->bytes[ST_AR] += n * 8;
  c->t->allreduce_u64(c->impl, c->rank, c->world, d_buf, n);
}
static void comm_bcast_bytes(mfuoco_comm *c, uint8_t *d_buf, size_t n, int root)
{
  c->calls[ST_BC]++;
  c->bytes[ST_BC] += n;
  c->t->bcast_bytes(c->impl, c->rank, c->world, d_buf, n, root);
}

/* ---- the row-sharded batch prover ------------------------------------------------------------------------------------------- */
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

  /* host inputs: the bit strings of ALL statements (they select the rank's BT+BV rows), entropy of the OWN statements */
  if (world > 1 && !c->t) dist_die("mfuoco_prover_batch_sharded", "more than one rank needs a communicator with a transport");
  uint8_t *bits = xmalloc(nb * stride), *mag = xmalloc((nown ? nown : 1) * 5 * MAGLEN), *sign = xmalloc((nown ? nown : 1) * 5);
  uint32_t *delta = xmalloc((nown ? nown : 1) * 4);
  memset(bits, 0, nb * stride);
  for (size_t k = 0; k < nb; k++) mfuoco_gpu_witness_bits(bits + k * stride, witnesses[k]);
  if (nown) mfuoco_gpu_prover_entropy_batch(delta, mag, sign, nown); /* (one draw, cut up in the order of nown prover() calls) */

  /* 1. the chain of the own statements (src/snark.c:141-169): w | h | v, nown x d coefficients each */
  uint32_t *whv = scratch(c, 0, 3 * nown * d * 4);
  uint32_t *W = whv, *H = whv + nown * d, *V = whv + 2 * nown * d;
  if (nown) CK(mfh_batch_chain(ctx, d_ssp, (uint32_t)nown, bits + first * stride, stride, delta, W, H, V));

  /* 2. all-to-all: rank q gets rows [d q / world, d (q+1) / world) of w | h | v of the own statements, laid out [statement][w | h | v][rows];
   *    this rank receives its rows of ALL statements (the statements of rank q are contiguous: [first_q, last_q)) */
  uint32_t *send = scratch(c, 1, 3 * nown * d * 4), *recv = scratch(c, 2, nb * 3 * cs * 4);
  size_t scnt[MAXW], sdsp[MAXW], rcnt[MAXW], rdsp[MAXW], off = 0;
  for (int q = 0; q < world; q++) {
    const size_t a = row_lo(d, q, world), w = row_lo(d, q + 1, world) - a;
    sdsp[q] = off;
    scnt[q] = nown * 3 * w;
    if (scnt[q])
      for (int x = 0; x < 3; x++) /* a strided column slice per polynomial kind: one 2-D device copy */
        HK(hipMemcpy2DAsync(send + off + (size_t)x * w, 3 * w * 4, whv + (size_t)x * nown * d + a, d * 4, w * 4, nown, hipMemcpyDeviceToDevice, NULL));
    off += scnt[q];
    const size_t fq = stmt_lo(nb, q, world), nq = stmt_lo(nb, q + 1, world) - fq;
    rdsp[q] = fq * 3 * cs;
    rcnt[q] = nq * 3 * cs;
  }
  if (c->t) comm_alltoallv_u32(c, send, scnt, sdsp, recv, rcnt, rdsp);
  else HK(hipMemcpyAsync(recv, send, scnt[0] * 4, hipMemcpyDeviceToDevice, NULL));

  /* 3. the rank's row shares of the five ciphertexts of every statement (matrix cores; no delta ct_t term, un-smudged) */
  uint64_t *partial = scratch(c, 3, nb * 5 * CTL * 8);
  mfuoco_gpu_image_resident_share(d_crs, (uint32_t)rank, (uint32_t)world, nb > 31); /* the rank's share of the image, kept across calls while the CRS is the same */
  CK(mfh_prove_batch_partial(ctx, d_crs, (uint32_t)rank, (uint32_t)world, (uint32_t)nb, bits, stride, recv, recv + cs, recv + 2 * cs, 3 * cs, partial));

  /* 4. the partial ciphertexts as uint64 lanes of 56 bits, statements padded to `world` equal slabs; ONE reduce-scatter: the own slab, summed */
  const size_t lps = 5 * LANES_PER_CT;
  uint64_t *lanes = scratch(c, 4, per * world * lps * 8), *own = scratch(c, 5, per * lps * 8);
  if (per * world > nb) HK(hipMemsetAsync(lanes + nb * lps, 0, (per * world - nb) * lps * 8, NULL));
  CK(mfh_ct_to_lanes(ctx, partial, nb * 5, lanes));
  if (c->t) comm_reduce_scatter_u64(c, lanes, own, per * lps);
  else HK(hipMemcpyAsync(own, lanes, per * lps * 8, hipMemcpyDeviceToDevice, NULL));

  /* 5. carries + modq, then b_w += delta ct_t and the smudging of the own statements (src/snark.c:143-145,185-189) */
  if (nown) {
    uint64_t *proofs = scratch(c, 6, nown * 5 * CTL * 8);
    CK(mfh_ct_from_lanes(ctx, own, nown * 5, proofs));
    CK(mfh_prove_batch_finish(ctx, d_crs, (uint32_t)nown, delta, mag, MAGLEN, sign, proofs));
    mfuoco_gpu_proofs_to_host(pis + first, proofs, nown);
  }
  HK(hipDeviceSynchronize());
  if (c == &local)
    for (int i = 0; i < 8; i++)
      if (local.buf[i]) hipFree(local.buf[i]);
  explicit_bzero(mag, (nown ? nown : 1) * 5 * MAGLEN); /* smudging terms and deltas are the proofs' zero-knowledge: not left on the heap */
  explicit_bzero(sign, (nown ? nown : 1) * 5);
  explicit_bzero(delta, (nown ? nown : 1) * 4);
  free(bits); free(mag); free(sign); free(delta);
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
  if (c == &local)
    for (int i = 0; i < 8; i++)
      if (local.buf[i]) hipFree(local.buf[i]);
}
