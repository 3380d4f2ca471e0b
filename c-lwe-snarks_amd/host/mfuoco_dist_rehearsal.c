/*
 * mfuoco_dist_rehearsal.c -- TEST SCAFFOLDING (see mfuoco_dist_rehearsal.h): the four collectives of mfuoco_transport staged through host shared memory.
 * Every operation: wait for the stream it is ordered on, copy the rank's contribution into its mailbox, barrier, read the others' mailboxes (copies on that stream,
 * waited for), barrier.  The host blocks in every call, so a pipelined caller is correct on it but overlaps nothing.  Never linked into a product library.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "mfuoco_dist_rehearsal.h"

#define MAXW 64

struct shm_hdr {
  _Atomic uint32_t ready, arrived, generation;
  uint32_t world;
  uint64_t slot_bytes;
  uint64_t disp[MAXW][MAXW], cnt[MAXW][MAXW]; /* all-to-all: bytes rank r sends to q, and where they start in r's mailbox */
};
struct rehearsal {
  int rank, world;
  struct shm_hdr *shm;
  size_t total;
};

static void die(const char *what, const char *detail)
{
  fprintf(stderr, "mfuoco rehearsal transport: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
  abort();
}
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die(#call, hipGetErrorString(e_)); } while (0)

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static void barrier(struct rehearsal *r)
{
  struct shm_hdr *h = r->shm;
  uint32_t gen = atomic_load(&h->generation);
  if (atomic_fetch_add(&h->arrived, 1) + 1 == (uint32_t)r->world) {
    atomic_store(&h->arrived, 0);
    atomic_fetch_add(&h->generation, 1);
    return;
  }
  double t0 = now_s();
  while (atomic_load(&h->generation) == gen) {
    usleep(20);
    if (now_s() - t0 > 300.0) die("barrier", "a rank did not arrive within 300 s");
  }
}
static uint8_t *mailbox(struct rehearsal *r, int q) { return (uint8_t *)r->shm + sizeof(struct shm_hdr) + (size_t)q * r->shm->slot_bytes; }
static void put(struct rehearsal *r, const void *d_src, size_t bytes, hipStream_t st)
{
  if (bytes > r->shm->slot_bytes) die("mailbox too small", "raise MFUOCO_REHEARSAL_SLOT_MB");
  if (bytes) HK(hipMemcpyAsync(mailbox(r, r->rank), d_src, bytes, hipMemcpyDeviceToHost, st));
  HK(hipStreamSynchronize(st)); /* (what the caller queued on the stream before the collective has run, and the mailbox is written) */
}
static void get(void *d_dst, const void *src, size_t bytes, hipStream_t st)
{
  HK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, st));
  HK(hipStreamSynchronize(st)); /* (pageable source: it may be reused or freed right after) */
}

static void reh_alltoallv_u32(void *impl, int rank, int world, const uint32_t *d_send, const size_t *scnt, const size_t *sdsp, uint32_t *d_recv,
                              const size_t *rcnt, const size_t *rdsp, void *stream)
{
  struct rehearsal *r = impl;
  hipStream_t st = stream;
  size_t end = 0;
  for (int q = 0; q < world; q++) {
    r->shm->disp[rank][q] = sdsp[q] * 4;
    r->shm->cnt[rank][q] = scnt[q] * 4;
    if (sdsp[q] + scnt[q] > end) end = sdsp[q] + scnt[q];
  }
  put(r, d_send, end * 4, st);
  barrier(r);
  for (int q = 0; q < world; q++) {
    if (r->shm->cnt[q][rank] != rcnt[q] * 4) die("all-to-all", "send and receive counts disagree");
    if (rcnt[q]) get(d_recv + rdsp[q], mailbox(r, q) + r->shm->disp[q][rank], rcnt[q] * 4, st);
  }
  barrier(r);
}
static void reh_reduce_scatter_u64(void *impl, int rank, int world, const uint64_t *d_send, uint64_t *d_recv, size_t n, void *stream)
{
  struct rehearsal *r = impl;
  hipStream_t st = stream;
  put(r, d_send, n * world * 8, st);
  barrier(r);
  uint64_t *acc = calloc(n ? n : 1, 8);
  if (!acc) die("out of host memory", NULL);
  for (int q = 0; q < world; q++) {
    const uint64_t *src = (const uint64_t *)mailbox(r, q) + (size_t)rank * n;
    for (size_t i = 0; i < n; i++) acc[i] += src[i];
  }
  if (n) get(d_recv, acc, n * 8, st);
  free(acc);
  barrier(r);
}
static void reh_allreduce_u64(void *impl, int rank, int world, uint64_t *d_buf, size_t n, void *stream)
{
  (void)rank;
  struct rehearsal *r = impl;
  hipStream_t st = stream;
  put(r, d_buf, n * 8, st);
  barrier(r);
  uint64_t *acc = calloc(n ? n : 1, 8);
  if (!acc) die("out of host memory", NULL);
  for (int q = 0; q < world; q++) {
    const uint64_t *src = (const uint64_t *)mailbox(r, q);
    for (size_t i = 0; i < n; i++) acc[i] += src[i];
  }
  if (n) get(d_buf, acc, n * 8, st);
  free(acc);
  barrier(r);
}
static void reh_bcast_bytes(void *impl, int rank, int world, uint8_t *d_buf, size_t n, int root, void *stream)
{
  (void)world;
  struct rehearsal *r = impl;
  hipStream_t st = stream;
  if (rank == root) put(r, d_buf, n, st);
  barrier(r);
  if (rank != root) get(d_buf, mailbox(r, root), n, st);
  barrier(r);
}
static void reh_destroy(void *impl)
{
  struct rehearsal *r = impl;
  barrier(r);
  munmap(r->shm, r->total);
  free(r);
}
static const mfuoco_transport rehearsal_transport = { "rehearsal (host shared memory)", reh_alltoallv_u32, reh_reduce_scatter_u64, reh_allreduce_u64,
                                                      reh_bcast_bytes, reh_destroy };

int mfuoco_comm_create_rehearsal(mfuoco_comm **out, int rank, int world, int device, const char *shm_name)
{
  if (world < 1 || world > MAXW || rank < 0 || rank >= world) return -1;
  struct rehearsal *r = calloc(1, sizeof *r);
  if (!r) return -1;
  r->rank = rank;
  r->world = world;
  char name[96];
  snprintf(name, sizeof name, "/%s", shm_name[0] == '/' ? shm_name + 1 : shm_name);
  const char *mb = getenv("MFUOCO_REHEARSAL_SLOT_MB");
  size_t slot = (size_t)(mb ? atol(mb) : 256) << 20;
  r->total = sizeof(struct shm_hdr) + (size_t)world * slot;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)r->total)) {
      fprintf(stderr, "mfuoco rehearsal transport: shm_open(%s): %s\n", name, strerror(errno));
      free(r);
      return -1;
    }
  } else {
    struct stat st;
    for (int tries = 0; tries < 2400; tries++) {
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0 && !fstat(fd, &st) && (size_t)st.st_size == r->total) break;
      if (fd >= 0) close(fd);
      fd = -1;
      usleep(50000);
    }
    if (fd < 0) {
      fprintf(stderr, "mfuoco rehearsal transport: rank %d: segment %s did not appear\n", rank, name);
      free(r);
      return -1;
    }
  }
  r->shm = mmap(NULL, r->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (r->shm == MAP_FAILED) {
    fprintf(stderr, "mfuoco rehearsal transport: mmap(%s): %s\n", name, strerror(errno));
    free(r);
    return -1;
  }
  if (rank == 0) {
    r->shm->world = (uint32_t)world;
    r->shm->slot_bytes = slot;
    atomic_store(&r->shm->ready, 1);
  } else {
    double t0 = now_s();
    while (!atomic_load(&r->shm->ready)) {
      usleep(100);
      if (now_s() - t0 > 120.0) die("segment", "rank 0 never initialised it");
    }
    if (r->shm->world != (uint32_t)world) die("segment", "world size differs from rank 0's");
  }
  barrier(r);
  if (rank == 0) shm_unlink(name); /* every rank has mapped it: the name can go now, so that a rank that dies later leaks nothing in /dev/shm */
  if (mfuoco_comm_create_transport(out, rank, world, device, &rehearsal_transport, r)) {
    munmap(r->shm, r->total);
    free(r);
    return -1;
  }
  return 0;
}
