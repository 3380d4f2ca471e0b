/*
 * mfuoco_files.c -- flat on-disk images of the CRS, the SSP, ciphertext-row files and proofs.
 *
 * The reference sizes these images but only ever maps them in commented-out scaffolding:
 *   CRS_SIZE = CT_BYTES * (2D + M + 1 + 2)            src/snark.h:6;  "crs.mfuoco", src/benchmark_snark.c:23,47-53,68-69
 *   SSP_SIZE = D * 8 * (M + 3)                        src/ssp.h:6;    "ssp.mfuoco", src/benchmark_snark.c:24,34-42
 *   "coeffs" = d rows of CT_BYTES, ct_export of b     src/benchmark_eval.c:20,44-66 (this one is live code)
 * The images here are exactly those sizes.  The CRS image keeps the rows in keystream order, which is also the order
 * struct crs's four arrays are consumed in and the order of the device CRS (mfh_setup / mfh_prove, include/mfhip.h):
 *
 *   row 0 .. D-1        s[i]           (Enc(s^i))
 *   row D .. 2D-1       as[i]          (Enc(alpha s^i))
 *   row 2D              t              (Enc(beta t(s)))
 *   row 2D+1 .. 2D+M    v[0..M)        (Enc(beta v_i(s)); the reference allocates M rows and uses M-1)
 *   row 2D+M+1, 2D+M+2  trailer: the 40-byte public seed (rseed_t), then zeros
 *
 * so a mapped image's first (2D+M) rows can be copied to the GPU in one piece.  Nothing here touches the GPU, with one optional hook: when this file is part of
 * libmfuoco_gpu (where mfuoco_gpu_prefetch_crs exists) a CRS mapped read-only is handed to it, so that its expansion is under way before the first prover().
 * All integers inside rows are the little-endian ct_export bytes (src/lwe.c:36-43); SSP slots are little-endian u64.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mangiafuoco_api.h"

/* (weak: absent when this file is compiled on its own, e.g. by the CPU tests) */
extern void mfuoco_gpu_prefetch_crs(crs_t crs) __attribute__((weak));

#define ROW_T (2 * (size_t)GAMMA_D)
#define ROW_V (ROW_T + 1)
#define ROW_TRAILER (ROW_V + (size_t)GAMMA_M)

static int write_all(int fd, const void *buf, size_t n)
{
  const uint8_t *p = buf;
  while (n) {
    ssize_t w = write(fd, p, n > (1u << 30) ? (1u << 30) : n);
    if (w < 0) {
      if (errno == EINTR) continue;
      return -1;
    }
    p += w;
    n -= (size_t)w;
  }
  return 0;
}

static int save_image(const char *path, const void *const *parts, const size_t *sizes, int nparts)
{
  int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, S_IRUSR | S_IWUSR);
  if (fd < 0) return -1;
  for (int i = 0; i < nparts; i++)
    if (write_all(fd, parts[i], sizes[i]) != 0) {
      int e = errno;
      close(fd);
      errno = e;
      return -1;
    }
  return close(fd);
}

/* map `path`; when expect != 0 the file must have exactly that size (EINVAL otherwise) */
static void *map_image(const char *path, size_t expect, size_t *size_out, int writable)
{
  int fd = open(path, writable ? O_RDWR : O_RDONLY);
  if (fd < 0) return NULL;
  struct stat st;
  if (fstat(fd, &st) != 0 || (expect && (size_t)st.st_size != expect) || st.st_size == 0) {
    close(fd);
    errno = EINVAL;
    return NULL;
  }
  void *m = mmap(NULL, (size_t)st.st_size, writable ? PROT_READ | PROT_WRITE : PROT_READ, writable ? MAP_SHARED : MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return NULL;
  madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
  if (size_out) *size_out = (size_t)st.st_size;
  return m;
}

/* ---- CRS ---------------------------------------------------------------------------------------------------- */
int mfuoco_crs_save(const char *path, const struct crs *crs)
{
  uint8_t trailer[2 * CT_BYTES] = { 0 };
  memcpy(trailer, crs->seed, sizeof(rseed_t));
  const void *parts[5] = { crs->s, crs->as, crs->t, crs->v, trailer };
  const size_t sizes[5] = { CT_BYTES * GAMMA_D, CT_BYTES * GAMMA_D, CT_BYTES, CT_BYTES * GAMMA_M, sizeof trailer };
  return save_image(path, parts, sizes, 5);
}

int mfuoco_crs_map(struct crs *crs, const char *path, int writable)
{
  uint8_t *m = map_image(path, CRS_SIZE, NULL, writable);
  if (!m) return -1;
  memcpy(crs->seed, m + ROW_TRAILER * CT_BYTES, sizeof(rseed_t));
  crs->s = (uint8_t(*)[CT_BYTES])m;
  crs->as = (uint8_t(*)[CT_BYTES])(m + (size_t)GAMMA_D * CT_BYTES);
  crs->t = m + ROW_T * CT_BYTES;
  crs->v = (uint8_t(*)[CT_BYTES])(m + ROW_V * CT_BYTES);
  if (!writable && mfuoco_gpu_prefetch_crs) mfuoco_gpu_prefetch_crs(crs); /* (a writable mapping is about to be filled by setup(), which writes the image itself) */
  return 0;
}

void mfuoco_crs_unmap(struct crs *crs)
{
  if (crs->s) munmap(crs->s, CRS_SIZE);
  crs->s = crs->as = crs->v = NULL;
  crs->t = NULL;
}

/* a writable mapping's seed lives in the struct; store it back into the trailer before unmapping a CRS that setup() filled */
void mfuoco_crs_sync_seed(struct crs *crs) { memcpy((uint8_t *)crs->s + ROW_TRAILER * CT_BYTES, crs->seed, sizeof(rseed_t)); }

int mfuoco_crs_create(struct crs *crs, const char *path)
{
  int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, S_IRUSR | S_IWUSR);
  if (fd < 0) return -1;
  if (ftruncate(fd, CRS_SIZE) != 0) {
    close(fd);
    return -1;
  }
  close(fd);
  rseed_t seed;
  memcpy(seed, crs->seed, sizeof seed);
  if (mfuoco_crs_map(crs, path, 1) != 0) return -1;
  memcpy(crs->seed, seed, sizeof seed); /* the caller's seed (crs_init draws it), not the zeros of the new file */
  mfuoco_crs_sync_seed(crs);
  return 0;
}

/* ---- SSP ---------------------------------------------------------------------------------------------------- */
int mfuoco_ssp_save(const char *path, const uint8_t *ssp)
{
  const void *parts[1] = { ssp };
  const size_t sizes[1] = { SSP_SIZE };
  return save_image(path, parts, sizes, 1);
}
uint8_t *mfuoco_ssp_map(const char *path, int writable) { return map_image(path, SSP_SIZE, NULL, writable); }
void mfuoco_ssp_unmap(uint8_t *ssp)
{
  if (ssp) munmap(ssp, SSP_SIZE);
}

/* ---- ciphertext-row files ("coeffs") ------------------------------------------------------------------------ */
int mfuoco_rows_save(const char *path, uint8_t (*c8)[CT_BYTES], size_t rows)
{
  const void *parts[1] = { c8 };
  const size_t sizes[1] = { rows * CT_BYTES };
  return save_image(path, parts, sizes, 1);
}
uint8_t (*mfuoco_rows_map(const char *path, size_t *rows))[CT_BYTES]
{
  size_t size = 0;
  uint8_t *m = map_image(path, 0, &size, 0);
  if (!m) return NULL;
  if (size % CT_BYTES) {
    munmap(m, size);
    errno = EINVAL;
    return NULL;
  }
  *rows = size / CT_BYTES;
  return (uint8_t(*)[CT_BYTES])m;
}
void mfuoco_rows_unmap(uint8_t (*c8)[CT_BYTES], size_t rows)
{
  if (c8) munmap(c8, rows * CT_BYTES);
}

/* ---- proofs: 5 ciphertexts in struct order, each N+1 values of CT_BYTES little-endian bytes (a_0..a_{N-1}, b) ---- */
#define PROOF_SIZE (5 * (size_t)(GAMMA_N + 1) * CT_BYTES)
static void value_export(uint8_t *out, const mpz_t z)
{
  memset(out, 0, CT_BYTES);
  if (mpz_sgn(z) < 0 || mpz_sizeinbase(z, 2) > 8 * CT_BYTES) {
    fprintf(stderr, "mfuoco_proof_save: value outside [0, 2^%lu)\n", 8 * CT_BYTES);
    abort();
  }
  mpz_export(out, NULL, -1, 1, -1, 0, z);
}
int mfuoco_proof_save(const char *path, proof_t pi)
{
  uint8_t *buf = malloc(PROOF_SIZE);
  if (!buf) return -1;
  mpz_t *cts[5] = { pi->h, pi->hat_h, pi->hat_v, pi->v_w, pi->b_w };
  for (int k = 0; k < 5; k++)
    for (size_t j = 0; j <= GAMMA_N; j++) value_export(buf + (k * (size_t)(GAMMA_N + 1) + j) * CT_BYTES, cts[k][j]);
  const void *parts[1] = { buf };
  const size_t sizes[1] = { PROOF_SIZE };
  int rc = save_image(path, parts, sizes, 1);
  free(buf);
  return rc;
}
int mfuoco_proof_load(proof_t pi, const char *path)
{
  uint8_t *m = map_image(path, PROOF_SIZE, NULL, 0);
  if (!m) return -1;
  mpz_t *cts[5] = { pi->h, pi->hat_h, pi->hat_v, pi->v_w, pi->b_w };
  for (int k = 0; k < 5; k++)
    for (size_t j = 0; j <= GAMMA_N; j++) mpz_import(cts[k][j], CT_BYTES, -1, 1, -1, 0, m + (k * (size_t)(GAMMA_N + 1) + j) * CT_BYTES);
  munmap(m, PROOF_SIZE);
  return 0;
}
