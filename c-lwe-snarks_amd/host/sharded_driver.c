/*
 * sharded_driver.c -- one rank (= one process = one GPU) of the multi-GPU prover through the C entry points of
 * host/include/mfuoco/mfuoco_dist.h.  Built twice: host/test_sharded (debug parameters D = 256, M = 64: checks) and
 * host/bench_snark_sharded (NDEBUG parameters: the measurement src/benchmark_snark.c:70-74 makes, for N GPUs, `label\tseconds` lines).
 *
 *   RANK, WORLD_SIZE, LOCAL_RANK       as a launcher (mpirun, torchrun, a shell loop) sets them; default 0 / 1 / 0
 *   MFUOCO_COMM_ID_FILE                file through which rank 0 publishes the ncclUniqueId: REQUIRED with more than one rank on RCCL -- a path in the job's
 *                                      own directory (mode 0700), never a shared /tmp (mfuoco_dist.h: mfuoco_comm_create)
 *   MFUOCO_REHEARSAL_SHM=name          rehearsal backend instead of RCCL (host shared memory; with MFUOCO_SHARE_GPU=1 all ranks use GPU 0)
 *   usage: test_sharded [statements]        bench_snark_sharded [statements] [calls]
 *
 * Every rank must build the SAME instance, so OS entropy is replaced by a deterministic, position-addressable tape: this file defines
 * getrandom(), which the shim libraries then resolve to (the executable precedes libc in symbol lookup).  A statement consumes exactly
 * 8 + 5 x 81 = 413 tape bytes (src/snark.c:140,185-189), so the owner of statement k seeks to base + 413 k and draws what a
 * single-process mfuoco_prover_batch() draws for it: the sharded proofs must then equal the single-GPU ones bit for bit.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/types.h>

#include "mfuoco/mfuoco_dist.h"
#include "mfuoco_dist_rehearsal.h" /* test scaffolding compiled into this driver, not into libmfuoco_gpu_dist.so */

/* ---- the tape: byte i = byte (i % 8) of mix(seed + i / 8) -------------------------------------------------------------------- */
static uint64_t tape_pos;
static const uint64_t tape_seed = 0x6d667575636f2121ULL;
static uint64_t mix(uint64_t x)
{
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}
static void tape_seek(uint64_t pos) { tape_pos = pos; }
ssize_t getrandom(void *buf, size_t n, unsigned flags)
{
  (void)flags;
  uint8_t *out = buf;
  size_t i = 0;
  while (i < n) {
    uint64_t w = mix(tape_seed + tape_pos / 8);
    size_t o = tape_pos % 8, take = 8 - o < n - i ? 8 - o : n - i;
    memcpy(out + i, (uint8_t *)&w + o, take);
    i += take;
    tape_pos += take;
  }
  return (ssize_t)n;
}

static double now(void)
{
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return tv.tv_sec + tv.tv_usec * 1e-6;
}
static int env_int(const char *name, int dflt)
{
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

static int ct_equal(ct_t a, ct_t b)
{
  for (size_t j = 0; j <= GAMMA_N; j++)
    if (mpz_cmp(a[j], b[j])) return 0;
  return 1;
}
static int proof_equal(proof_t a, proof_t b)
{
  return ct_equal(a->h, b->h) && ct_equal(a->hat_h, b->hat_h) && ct_equal(a->hat_v, b->hat_v) && ct_equal(a->v_w, b->v_w) && ct_equal(a->b_w, b->b_w);
}

#define STMT_TAPE 413u
#define BASE_BATCH (1ULL << 40)
#define BASE_SINGLE (1ULL << 41)

int main(int argc, char **argv)
{
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
  const int device = env_int("MFUOCO_SHARE_GPU", 0) ? 0 : env_int("LOCAL_RANK", 0);
  size_t count = argc > 1 ? (size_t)atol(argv[1]) : 40;
  int calls = argc > 2 ? atoi(argv[2]) : 3;
  mfuoco_comm *comm = NULL;
  const char *shm = getenv("MFUOCO_REHEARSAL_SHM");
  const char *idfile = getenv("MFUOCO_COMM_ID_FILE");
  if (!(shm && *shm) && world > 1 && !(idfile && *idfile)) {
    fprintf(stderr, "rank %d: set MFUOCO_COMM_ID_FILE to a path in a directory of this job's own (the ranks meet there)\n", rank);
    return 2;
  }
  int rc = shm && *shm ? mfuoco_comm_create_rehearsal(&comm, rank, world, device, shm) : mfuoco_comm_create(&comm, rank, world, device, idfile ? idfile : "");
  if (rc) return 2; /* (non-zero exit: a launcher then ends the other ranks; no retry in-process) */

  /* the same instance on every rank */
  tape_seek(0);
  ssp_t ssp = calloc(1, SSP_SIZE);
  mpz_t witness;
  mpz_init(witness);
  random_ssp(witness, ssp);
  crs_t crs;
  crs_init(crs);
  vrs_t vrs;
  setup(crs, vrs, ssp);

  mpz_t *wit = malloc(count * sizeof *wit);
  proof_t *pis = malloc(count * sizeof *pis), *ref = malloc(count * sizeof *ref);
  tape_seek(1ULL << 39);
  for (size_t k = 0; k < count; k++) {
    mpz_init(wit[k]);
    if (k % 2 == 0) mpz_set(wit[k], witness); /* even statements carry the satisfying witness, odd ones random bits */
    else mpz2_urandomb2(wit[k], GAMMA_M - 1);
    proof_init(pis[k]);
    proof_init(ref[k]);
  }
  const size_t per = count ? (count + world - 1) / world : 0;
  const size_t first_expected = (size_t)rank * per < count ? (size_t)rank * per : count;
  size_t first = 0, nown = 0;
  int ok = 1;

#ifndef NDEBUG
  (void)calls;
  /* single-GPU reference run of all statements, then the sharded run with the owner's part of the same tape */
  tape_seek(BASE_BATCH);
  mfuoco_prover_batch(ref, crs, ssp, wit, count);
  tape_seek(BASE_BATCH + STMT_TAPE * first_expected);
  mfuoco_prover_batch_sharded(pis, crs, ssp, wit, count, comm, &first, &nown);
  ok = ok && first == first_expected;
  { /* the bytes handed to the backend are those of the one-shot sequence however the call was cut into stages ($MFUOCO_DIST_STAGE): every own statement's w | h | v once
     * through the all-to-all, ceil(count / world) x world statements of 5 x (N + 1) x 13 uint64 lanes into the reduce-scatter */
    uint64_t c0[4], b0[4];
    mfuoco_comm_stats(comm, c0, b0);
    const uint64_t want_a2a = (uint64_t)nown * 3 * GAMMA_D * 4, want_rs = (uint64_t)per * world * 5 * (GAMMA_N + 1) * 13 * 8;
    if (b0[0] != want_a2a || b0[1] != want_rs) {
      fprintf(stderr, "rank %d: all-to-all %llu bytes (expected %llu), reduce-scatter %llu (expected %llu)\n", rank, (unsigned long long)b0[0],
              (unsigned long long)want_a2a, (unsigned long long)b0[1], (unsigned long long)want_rs);
      ok = 0;
    }
  }
  for (size_t k = first; k < first + nown; k++) {
    int same = proof_equal(pis[k], ref[k]);
    int acc = verifier(ssp, vrs, pis[k]);
    if (!same || acc != (k % 2 == 0)) {
      fprintf(stderr, "rank %d: statement %zu: %s, verifier says %d\n", rank, k, same ? "identical" : "DIFFERS from mfuoco_prover_batch", acc);
      ok = 0;
    }
  }
  /* a second call on the same communicator with fewer statements than ranks (scratch reuse; the last rank owns none) */
  size_t few = world > 1 ? (size_t)world - 1 : 1;
  if (few > count) few = count;
  size_t f2 = (size_t)rank < few ? (size_t)rank : few, n2 = 0, g2 = 0; /* slabs of ceil(few / world) = 1 statement */
  tape_seek(BASE_BATCH + STMT_TAPE * f2);
  mfuoco_prover_batch_sharded(pis, crs, ssp, wit, few, comm, &g2, &n2);
  ok = ok && g2 == f2 && n2 == ((size_t)rank < few ? 1u : 0u);
  for (size_t k = g2; k < g2 + n2; k++) ok = ok && proof_equal(pis[k], ref[k]);
  /* one proof computed by all ranks together against prover() */
  proof_t one, one_ref;
  proof_init(one);
  proof_init(one_ref);
  tape_seek(BASE_SINGLE);
  prover(one_ref, crs, ssp, witness);
  tape_seek(BASE_SINGLE);
  mfuoco_prover_sharded(one, crs, ssp, witness, comm);
  if (!proof_equal(one, one_ref) || !verifier(ssp, vrs, one)) {
    fprintf(stderr, "rank %d: mfuoco_prover_sharded differs from prover() or is rejected\n", rank);
    ok = 0;
  }
  uint64_t ncalls[4], nbytes[4];
  mfuoco_comm_stats(comm, ncalls, nbytes);
  printf("rank %d/%d backend=%s statements=%zu own=[%zu,%zu) all_to_all=%llu reduce_scatter=%llu all_reduce=%llu broadcast=%llu: %s\n", rank, world,
         mfuoco_comm_backend(comm), count, first, first + nown, (unsigned long long)ncalls[0], (unsigned long long)ncalls[1], (unsigned long long)ncalls[2],
         (unsigned long long)ncalls[3], ok ? "sharded ok" : "FAILURE");
  ok = ok && ncalls[0] >= 2 && ncalls[1] >= 2 && ncalls[2] >= 2;
#else
  /* the measurement: wall clock around the call, as src/benchmark_snark.c:70-74 times prover() */
  tape_seek(BASE_BATCH + STMT_TAPE * first_expected);
  mfuoco_prover_batch_sharded(pis, crs, ssp, wit, count, comm, &first, &nown); /* warm-up: SSP upload, images, scratch */
  for (int i = 0; i < calls; i++) {
    tape_seek(BASE_BATCH + STMT_TAPE * first_expected);
    double t0 = now();
    mfuoco_prover_batch_sharded(pis, crs, ssp, wit, count, comm, &first, &nown);
    double dt = now() - t0;
    if (rank == 0) printf("prover_batch_sharded\t%lf\t(%zu statements over %d rank(s), backend %s: %.1f proofs/s incl. PCIe and mpz_t conversion)\n", dt, count, world,
                          mfuoco_comm_backend(comm), count / dt);
  }
  for (size_t k = first; k < first + nown && k < first + 4; k++) ok = ok && verifier(ssp, vrs, pis[k]) == (k % 2 == 0);
  proof_t one;
  proof_init(one);
  for (int i = 0; i < calls; i++) {
    double t0 = now();
    mfuoco_prover_sharded(one, crs, ssp, witness, comm);
    if (rank == 0) printf("prover_sharded\t%lf\n", now() - t0);
  }
  ok = ok && verifier(ssp, vrs, one);
  fprintf(stderr, "rank %d: %s\n", rank, ok ? "own proofs verified" : "FAILURE");
#endif
  mfuoco_comm_destroy(comm);
  return ok ? 0 : 1;
}
