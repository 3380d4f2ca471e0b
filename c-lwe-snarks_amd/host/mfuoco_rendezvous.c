#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "mfuoco_rendezvous.h"

/*
 * mfuoco_rendezvous.c -- rendezvous through files (mfuoco_comm_create)
 * A rank must never enter ncclCommInitRank with an id the others do not have: it would wait there forever, holding its GPU.  A file left behind by a killed job
 * (or planted by another user of a shared /tmp) is exactly that, so the id file alone is not trusted.  Three files, all created with O_CREAT | O_EXCL |
 * O_NOFOLLOW and mode 0600 under a private temporary name and renamed into place:
 *   <id_file>          rank 0: { magic, world, session nonce N (fresh from the OS), the ncclUniqueId }   -- rank 0 unlinks any older one first
 *   <id_file>.ack.<k>  rank k: { magic, the nonce it read, its own fresh nonce R_k }
 *   <id_file>.go       rank 0, once every ack carries N: { magic, N, R_1 .. R_(world-1) }
 * Rank k proceeds only when the go file names ITS R_k (a stale go file cannot); an ack with an old nonce is deleted by rank 0 and rank k, seeing it gone,
 * reads the id file again.  Everybody gives up after MFUOCO_RENDEZVOUS_S seconds (default 120) with -1: the caller exits non-zero, it does not retry. */
#define RDV_MAGIC 0x6d66756f636f4944ULL /* "mfuocoID" */
struct rdv_id { uint64_t magic, nonce; uint32_t world, pad; uint8_t id[MFUOCO_RDV_ID_BYTES]; };
struct rdv_ack { uint64_t magic, nonce, mine; };
struct rdv_go { uint64_t magic, nonce; uint64_t r[MFUOCO_RDV_MAXW]; };

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static uint64_t fresh_nonce(void)
{
  uint64_t x = 0;
  errno = 0;
  while (getrandom(&x, sizeof x, 0) != (ssize_t)sizeof x) {
    if (errno != EINTR && errno != EAGAIN) (perror("getrandom"), abort());
    errno = 0;
  }
  /* (a harness may interpose getrandom with a deterministic tape: process id and clock keep sessions apart even then) */
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  x ^= ((uint64_t)getpid() << 40) ^ ((uint64_t)ts.tv_sec << 20) ^ (uint64_t)ts.tv_nsec;
  return x ? x : 1;
}
static int write_private(const char *path, const void *buf, size_t n)
{
  char tmp[4096];
  if (snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid()) >= (int)sizeof tmp) return -1;
  unlink(tmp);
  int fd = open(tmp, O_CREAT | O_EXCL | O_NOFOLLOW | O_WRONLY | O_CLOEXEC, 0600);
  if (fd < 0) return -1;
  const uint8_t *p = buf;
  size_t left = n;
  while (left) {
    ssize_t w = write(fd, p, left);
    if (w < 0 && errno == EINTR) continue;
    if (w <= 0) { close(fd); unlink(tmp); return -1; }
    p += w;
    left -= (size_t)w;
  }
  if (close(fd) || rename(tmp, path)) { unlink(tmp); return -1; }
  return 0;
}
/* n bytes of a regular file owned by this user, or -1 */
static int read_private(const char *path, void *buf, size_t n)
{
  int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return -1;
  struct stat st;
  if (fstat(fd, &st) || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (size_t)st.st_size != n) { close(fd); return -1; }
  uint8_t *p = buf;
  size_t left = n;
  while (left) {
    ssize_t r = read(fd, p, left);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) { close(fd); return -1; }
    p += r;
    left -= (size_t)r;
  }
  close(fd);
  return 0;
}

int mfuoco_rendezvous_files(int rank, int world, const char *id_file, uint8_t id[MFUOCO_RDV_ID_BYTES], double limit)
{
  if (!id_file || !*id_file || world < 2 || world > MFUOCO_RDV_MAXW || rank < 0 || rank >= world) {
    fprintf(stderr, "libmfuoco_gpu_dist (rendezvous): bad arguments (rank %d of %d, file %s)\n", rank, world, id_file ? id_file : "(null)");
    return -1;
  }
  const double t0 = now_s();
  char ack[4096], go[4096];
  if (snprintf(go, sizeof go, "%s.go", id_file) >= (int)sizeof go) return -1;
  struct rdv_id rid;
  memset(&rid, 0, sizeof rid);
  if (rank == 0) {
    unlink(id_file);
    unlink(go);
    for (int k = 1; k < world; k++) {
      snprintf(ack, sizeof ack, "%s.ack.%d", id_file, k);
      unlink(ack);
    }
    rid.magic = RDV_MAGIC;
    rid.nonce = fresh_nonce();
    rid.world = (uint32_t)world;
    memcpy(rid.id, id, sizeof rid.id);
    if (write_private(id_file, &rid, sizeof rid)) {
      fprintf(stderr, "libmfuoco_gpu_dist (rendezvous): cannot publish the communicator id in %s: %s\n", id_file, strerror(errno));
      return -1;
    }
    struct rdv_go g;
    memset(&g, 0, sizeof g);
    g.magic = RDV_MAGIC;
    g.nonce = rid.nonce;
    for (int missing = world - 1; missing;) {
      missing = 0;
      for (int k = 1; k < world; k++) {
        if (g.r[k]) continue;
        struct rdv_ack a;
        snprintf(ack, sizeof ack, "%s.ack.%d", id_file, k);
        if (!read_private(ack, &a, sizeof a) && a.magic == RDV_MAGIC) {
          if (a.nonce == rid.nonce && a.mine) g.r[k] = a.mine;
          else unlink(ack); /* answered an older id file: rank k reads again */
        }
        if (!g.r[k]) missing++;
      }
      if (missing) {
        if (now_s() - t0 > limit) {
          fprintf(stderr, "libmfuoco_gpu_dist (rendezvous): rank 0: %d of %d ranks did not answer %s within %.0f s\n", missing, world - 1, id_file, limit);
          unlink(id_file);
          return -1;
        }
        usleep(20000);
      }
    }
    if (write_private(go, &g, sizeof g)) {
      fprintf(stderr, "libmfuoco_gpu_dist (rendezvous): cannot write %s: %s\n", go, strerror(errno));
      unlink(id_file);
      return -1;
    }
  } else {
    const uint64_t mine = fresh_nonce();
    uint64_t acked = 0;
    snprintf(ack, sizeof ack, "%s.ack.%d", id_file, rank);
    for (;;) {
      struct rdv_id cand;
      if (!read_private(id_file, &cand, sizeof cand) && cand.magic == RDV_MAGIC && cand.world == (uint32_t)world && cand.nonce) {
        if (acked != cand.nonce || access(ack, F_OK)) {
          struct rdv_ack a = { RDV_MAGIC, cand.nonce, mine };
          if (!write_private(ack, &a, sizeof a)) acked = cand.nonce;
        }
        struct rdv_go g;
        if (acked == cand.nonce && !read_private(go, &g, sizeof g) && g.magic == RDV_MAGIC && g.nonce == cand.nonce && g.r[rank] == mine) {
          rid = cand;
          break;
        }
      }
      if (now_s() - t0 > limit) {
        fprintf(stderr, "libmfuoco_gpu_dist (rendezvous): rank %d: no rendezvous through %s within %.0f s\n", rank, id_file, limit);
        unlink(ack);
        return -1;
      }
      usleep(20000);
    }
  }
  if (rank) memcpy(id, rid.id, MFUOCO_RDV_ID_BYTES);
  return 0;
}

void mfuoco_rendezvous_cleanup(int rank, const char *id_file)
{
  char path[4096];
  if (rank == 0) {
    unlink(id_file);
    if (snprintf(path, sizeof path, "%s.go", id_file) < (int)sizeof path) unlink(path);
  } else if (snprintf(path, sizeof path, "%s.ack.%d", id_file, rank) < (int)sizeof path) {
    unlink(path);
  }
}
