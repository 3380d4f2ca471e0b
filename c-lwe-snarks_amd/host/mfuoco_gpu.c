/*
 * mfuoco_gpu.c -- libmfuoco_gpu.so: the reference library's C interface (names, argument meaning, ownership and error
 * behaviour of src/aes.h, src/entropy.h, src/lwe.h, src/ssp.h, src/snark.h) implemented on top of the MI355X C ABI
 * (include/mfhip.h).  Host code here only (a) converts mpz_t <-> dense little-endian limbs, (b) moves bytes across
 * PCIe, (c) draws OS entropy where the reference calls getrandom(2).  All arithmetic on the path -- AES-CTR expansion,
 * big-integer multiply-accumulate, dot products, the polynomial step -- runs on the GPU.  No CPU fallback: if the GPU
 * context cannot be created the first call aborts with a message.
 *
 * Error behaviour mirrors the reference: functions return void, preconditions the reference asserts (message < p,
 * scalar < p, non-negative values) abort() here in every build, allocation failures perror() and continue where the
 * reference does.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/random.h>
#include <unistd.h>

#include <gmp.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "mfhip.h"
#include "mfuoco/mangiafuoco_api.h"

#define L_LIMBS 12
#define K_LIMBS 11
#define CTL ((size_t)(GAMMA_N + 1) * L_LIMBS)
#define PIN_BYTES ((size_t)4 << 20)
#define ENC_CHUNK ((size_t)16384)
#define ENC_UP ((size_t)L_LIMBS * 8 + 4) /* per row: the error's limbs, then the message */
/* (CTR_CT, CTR_S, CTR_AS, CTR_BT come from the header, as in src/snark.h:8-12: CT_BYTES is 92UL, so they are 64-bit values) */

static struct {
  mfh_ctx *ctx;
  int device;
  uint8_t seed[40];
  bool have_seed;
  /* persistent device scratch */
  uint64_t *d_ct[3];  /* three ciphertexts */
  uint64_t *d_sk;     /* n values */
  uint8_t *d_c8;      /* up to max(D, M) * CT_BYTES, or the whole CRS */
  uint32_t *d_co;     /* coefficients */
  uint64_t *d_proof;  /* 5 ciphertexts */
  uint32_t *d_ssp;    /* (M+3) * D uint32 */
  const void *ssp_host; /* host pointer the resident SSP was uploaded from */
  uint8_t *d_crs;     /* (2D+M) * CT_BYTES */
  uint64_t *d_err;
  uint64_t *d_out;    /* mfuoco_prover_batch: out_cap proofs */
  size_t out_cap;
  uint64_t *d_up;     /* mfuoco_verifier_batch / mfuoco_decrypt_batch: up_cap ciphertexts + up_cap result words */
  size_t up_cap;
  /* the secret key last uploaded: host copy of its limbs (exact comparison, no digest) so that a caller encrypting / decrypting in a loop under one key uploads it once */
  uint64_t *h_sk, *pin_sk;
  bool sk_valid;
  uint8_t *pin;       /* PIN_BYTES of pinned host memory: staging of the single-ciphertext calls */
  hipEvent_t ev_small; /* ... and an event for their split downloads */
  hipEvent_t ev_img;   /* setup(): the a parts of the row image have been written (queued on s2 beside the SSP upload) */
  hipStream_t s2;      /* ... and a second stream: the public half of a regev_encrypt2 beside its secret half */
  /* mfuoco_encrypt_batch: two chunks of ENC_CHUNK rows in flight (error limbs and messages up, exported b's down) */
  uint8_t *enc_pin[2], *d_enc[2];
  hipEvent_t enc_ev[2];
  /* the expanded CRS kept across prover calls (mfuoco_gpu_set_resident_crs): the matrix-core image of (seed, compressed CRS) */
  int resident_crs;     /* -1 = not decided yet ($MFUOCO_GPU_RESIDENT_CRS, default on), 0 / 1 */
  uint8_t *d_img;
  size_t img_bytes;
  bool img_valid, img_registered;
  uint32_t img_rank, img_world;
  uint8_t img_seed[40];
  uint64_t img_digest[2];
  /* ... and the single-proof form (mfh_crs_expand: limb planes, streamed by k_mac_resident): written by setup() as a by-product of its encryptions and by
   * mfuoco_gpu_prefetch_crs() (what mfuoco_crs_map calls), so that the FIRST prover() under such a CRS streams it; for a CRS that reached the shim by neither way, kept
   * from the second prover() under it on */
  void *d_rows;
  size_t rows_bytes;
  bool rows_valid;
  uint8_t rows_seed[40], seen_seed[40];
  uint64_t rows_digest[2], seen_digest[2];
  bool seen;
  uint64_t staged_digest[2]; /* of the compressed CRS as last staged (mfuoco_gpu_stage_crs) */
  bool staged_digest_valid;
  int last_prover_path;      /* what the last prover() ran on: 0 regenerated keystream, 1 the resident rows (mfuoco_gpu_last_prover_path) */
} G = { .device = -1, .resident_crs = -1 };

static void die(const char *what)
{
  fprintf(stderr, "libmfuoco_gpu: %s%s%s\n", what, G.ctx ? ": " : "", G.ctx ? mfh_last_error(G.ctx) : "");
  abort();
}
/* errno is the CALLER's: the reference's drivers test it after their own system calls (src/benchmark_eval.c:22-27 exits when errno > 0 after open / write / mmap), and
 * the HIP runtime leaves ENOENT and friends behind on success (a missing amdgpu.ids on first use is enough).  Every exported function hands errno back as it found it. */
static inline void errno_restore_(int *saved) { errno = *saved; }
/* ... and the process starts with errno == 0, as it does with the reference's objects: the initialisers of the ROCm libraries this one pulls in probe files that need not
 * exist (they run before this constructor: dependencies are initialised first) */
__attribute__((constructor)) static void shim_clean_errno(void) { errno = 0; }
#define KEEP_ERRNO int errno_keep_ __attribute__((cleanup(errno_restore_), unused)) = errno
#define CK(call) do { if ((call) != MFH_OK) die(#call); } while (0)
#define HK(call) do { if ((call) != hipSuccess) die(#call); } while (0)

/* $MFUOCO_TRACE=1: wall-clock phases of the batch entry points on stderr (where a call's time goes: staging, queueing, draining) */
#include <time.h>
static double tnow(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static bool tracing(void)
{
  static int on = -1;
  if (on < 0) { const char *e = getenv("MFUOCO_TRACE"); on = e && *e && atoi(e) != 0; }
  return on == 1;
}
static void *xmalloc(size_t bytes)
{
  void *p = malloc(bytes ? bytes : 1);
  if (!p) die("out of host memory");
  return p;
}
static void *xcalloc(size_t n, size_t size)
{
  void *p = calloc(n ? n : 1, size ? size : 1);
  if (!p) die("out of host memory");
  return p;
}
/* mpz_t <-> limb conversions of a batch on a few host threads (distinct mpz_t: GMP is re-entrant).  Plain pthreads, created and joined per call: no runtime that
 * keeps spinning worker threads around a process whose other threads feed a GPU (an OpenMP team did exactly that on the 16-core share of a many-core box and made every
 * call 10 x slower).  $MFUOCO_HOST_THREADS (default 8, at most 64; 1 = the calling thread only). */
struct par_job { void (*fn)(size_t lo, size_t hi, void *arg); void *arg; size_t lo, hi; };
static void *par_worker(void *p)
{
  struct par_job *j = p;
  j->fn(j->lo, j->hi, j->arg);
  return NULL;
}
static void parallel_for_grain(size_t n, size_t serial_below, void (*fn)(size_t lo, size_t hi, void *arg), void *arg)
{
  static int nthreads;
  if (!nthreads) {
    const char *e = getenv("MFUOCO_HOST_THREADS");
    long want = e && *e ? atol(e) : 8, have = sysconf(_SC_NPROCESSORS_ONLN);
    nthreads = (int)(want < 1 ? 1 : want > 64 ? 64 : want);
    if (have > 0 && nthreads > have) nthreads = (int)have;
  }
  int nt = nthreads;
  if (n < serial_below || nt < 2) {
    fn(0, n, arg);
    return;
  }
  pthread_t th[64];
  struct par_job job[64];
  int started = 0;
  for (int t = 0; t < nt; t++) {
    job[t] = (struct par_job){ fn, arg, n * (size_t)t / nt, n * (size_t)(t + 1) / nt };
    if (t == nt - 1 || pthread_create(&th[t], NULL, par_worker, &job[t])) { /* the last slice (and any slice whose thread cannot be had) runs here */
      job[t].hi = n;
      fn(job[t].lo, n, arg);
      break;
    }
    started++;
  }
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}
/* (a single proof is 7 355 values: not worth a thread) */
static void parallel_for(size_t n, void (*fn)(size_t lo, size_t hi, void *arg), void *arg) { parallel_for_grain(n, 65536, fn, arg); }

/* OS entropy where the reference calls getrandom(2) (src/entropy.h, src/snark.c:40,62-65,140,185-189): EINTR and short reads are retried, the pool not
 * being initialised yet is waited for, and any other failure ends the process -- a buffer left undrawn would silently cost zero-knowledge */
static void shim_random(void *buf, size_t bytes)
{
  uint8_t *p = buf;
  while (bytes) {
    ssize_t got = getrandom(p, bytes, 0);
    if (got < 0) {
      if (errno == EINTR || errno == EAGAIN) continue;
      perror("getrandom");
      die("no OS entropy");
    }
    p += got;
    bytes -= (size_t)got;
  }
}

int mfuoco_gpu_set_device(int device)
{
  KEEP_ERRNO;
  if (G.ctx && G.device != device) {
    fprintf(stderr, "libmfuoco_gpu: the shim already runs on GPU %d; mfuoco_gpu_set_device(%d) must precede the first call\n", G.device, device);
    return -1;
  }
  G.device = device;
  return 0;
}
int mfuoco_gpu_device(void) { return G.ctx ? G.device : -1; }
static void sk_forget(void);
static void drop_image(void)
{
  if (G.ctx && G.img_registered) {
    mfh_crs_set_resident_mm(G.ctx, NULL);
    G.img_registered = false;
  }
  G.img_valid = false;
  G.rows_valid = false;
  G.seen = false;
}
void mfuoco_gpu_invalidate(void)
{
  KEEP_ERRNO;
  G.ssp_host = NULL;
  drop_image();
  /* ... and the batch calls' device scratch (720 MB per 1020 proofs), which otherwise stays for the next call */
  if (G.d_out) { (void)hipFree(G.d_out); G.d_out = NULL; G.out_cap = 0; }
  if (G.d_up) { (void)hipFree(G.d_up); G.d_up = NULL; G.up_cap = 0; }
  sk_forget(); /* ... and forget the cached key */
}
void mfuoco_gpu_set_resident_crs(int on)
{
  KEEP_ERRNO;
  G.resident_crs = on ? 1 : 0;
  if (!on) {
    drop_image();
    if (G.d_img) hipFree(G.d_img);
    if (G.d_rows) hipFree(G.d_rows);
    G.d_img = NULL;
    G.d_rows = NULL;
    G.img_bytes = G.rows_bytes = 0;
  }
}

static bool shim_warm_on(void)
{
  const char *e = getenv("MFUOCO_GPU_WARM");
  return !(e && *e && !atoi(e));
}
static mfh_ctx *gpu(void)
{
  if (G.ctx) return G.ctx;
  if (G.device < 0) {
    const char *e = getenv("MFUOCO_GPU");
    G.device = e ? atoi(e) : 0;
  }
  mfh_params P = { GAMMA_N, GAMMA_LOGQ, (uint32_t)GAMMA_D, (uint32_t)GAMMA_M };
  if (mfh_ctx_create(&G.ctx, G.device, &P) != MFH_OK) {
    fprintf(stderr, "libmfuoco_gpu: no usable HIP device %d (there is no CPU fallback)\n", G.device);
    abort();
  }
  CK(mfh_set_stream(G.ctx, NULL)); /* default stream: hipMemcpy below is ordered with the kernels */
  {
    /* mfuoco_prover_batch's polynomial step: the exact-division path (include/mfhip.h: mfh_set_poly_exact) pays for itself when the statements' witnesses satisfy the
     * SSP, which is what a prover is given, and backs off by itself when they do not; $MFUOCO_GPU_POLY_EXACT=0 / 2: never / always (same proofs either way) */
    const char *e = getenv("MFUOCO_GPU_POLY_EXACT");
    if (e && *e && atoi(e) >= 0 && atoi(e) <= 2) CK(mfh_set_poly_exact(G.ctx, atoi(e)));
  }
  size_t rows = 2 * (size_t)GAMMA_D + GAMMA_M;
  for (int i = 0; i < 3; i++) HK(hipMalloc((void **)&G.d_ct[i], CTL * 8 + 256)); /* (+ room for a keystream block behind a ciphertext: regev_encrypt2) */
  HK(hipMalloc((void **)&G.d_sk, (size_t)GAMMA_N * L_LIMBS * 8));
  HK(hipMalloc((void **)&G.d_c8, rows * CT_BYTES));
  HK(hipMalloc((void **)&G.d_co, rows * 4));
  HK(hipMalloc((void **)&G.d_proof, 5 * CTL * 8));
  HK(hipMalloc((void **)&G.d_crs, rows * CT_BYTES));
  HK(hipHostMalloc((void **)&G.pin, PIN_BYTES, hipHostMallocDefault));
  HK(hipEventCreateWithFlags(&G.ev_small, hipEventDisableTiming));
  HK(hipEventCreateWithFlags(&G.ev_img, hipEventDisableTiming));
  HK(hipStreamCreateWithFlags(&G.s2, hipStreamNonBlocking));
  HK(hipHostMalloc((void **)&G.pin_sk, (size_t)GAMMA_N * L_LIMBS * 8, hipHostMallocDefault));
  G.h_sk = xcalloc((size_t)GAMMA_N * L_LIMBS, 8);
  /* eval_poly's share of the context's scratch and code objects, paid here instead of by the first eval_poly a caller times (src/benchmark_eval.c:70-74 times exactly one,
   * the first): one evaluation of D rows under an all-zero seed, result discarded.  $MFUOCO_GPU_WARM=0 skips it (and setup()'s warm-up proof). */
  if (shim_warm_on()) {
    static const uint8_t zero_seed[40];
    CK(mfh_set_seed(G.ctx, zero_seed)); /* (G.have_seed stays false: the first real call sets its own) */
    HK(hipMemset(G.d_c8, 0, (size_t)GAMMA_D * CT_BYTES));
    HK(hipMemset(G.d_co, 1, (size_t)GAMMA_D * 4));
    CK(mfh_eval_rows(G.ctx, 0, GAMMA_D, G.d_c8, G.d_co, NULL, G.d_ct[2], NULL, 0));
    CK(mfh_sync(G.ctx));
    /* ... and the runtime's copy engines: the first host-to-device copy of more than a few KB in a process takes 8 ms whatever memory it comes from (measured inside
     * benchmark_eval's one timed eval_poly: 131 KB of coefficients from pinned memory, 8.2 ms; the next copy 0.03) */
    memset(G.pin, 0, (size_t)1 << 20);
    HK(hipMemcpy(G.d_c8, G.pin, (size_t)1 << 20 < rows * CT_BYTES ? (size_t)1 << 20 : rows * CT_BYTES, hipMemcpyHostToDevice));
    HK(hipMemcpy(G.pin, G.d_c8, (size_t)1 << 20 < rows * CT_BYTES ? (size_t)1 << 20 : rows * CT_BYTES, hipMemcpyDeviceToHost));
  }
  return G.ctx;
}

static void use_seed(const uint8_t seed[40])
{
  mfh_ctx *c = gpu();
  if (!G.have_seed || memcmp(G.seed, seed, 40)) {
    CK(mfh_set_seed(c, seed));
    memcpy(G.seed, seed, 40);
    G.have_seed = true;
  }
}

/* ---- mpz <-> limbs -------------------------------------------------------------------------------------- */
static inline void to_limbs(uint64_t *out, const mpz_t z)
{
  if (mpz_sgn(z) < 0) die("negative value (the reference asserts SIZ >= 0, src/lwe.h:109)");
  const size_t n = mpz_size(z);
  if (n > L_LIMBS) die("value wider than 768 bits");
  memcpy(out, mpz_limbs_read(z), n * 8); /* (whole native limbs, least significant first: what mpz_export(out, NULL, -1, 8, 0, 0, z) writes) */
  memset(out + n, 0, (L_LIMBS - n) * 8);
}
/* (what mpz_import(z, L_LIMBS, -1, 8, 0, 0, in) does for whole native limbs -- normalise, make room, copy -- without its per-call word-order / nails dispatch:
 * a batch of 1020 proofs is 7.5 M values) */
static inline void from_limbs(mpz_t z, const uint64_t *in)
{
  _Static_assert(sizeof(mp_limb_t) == 8 && GMP_NAIL_BITS == 0, "64-bit limbs without nails");
  mp_size_t n = L_LIMBS;
  while (n > 0 && !in[n - 1]) n--;
  if (n) memcpy(mpz_limbs_write(z, n), in, (size_t)n * 8);
  mpz_limbs_finish(z, n);
}

/* pageable host bytes to the device through the shim's own pinned scratch, PIN_BYTES at a time.  (Handed to hipMemcpy as they are, the FIRST pageable copy of a process
 * makes the runtime set up its internal staging: 9 ms inside the one eval_poly src/benchmark_eval.c times, once every other copy of the shim had moved to pinned memory.) */
static void h2d_staged(void *d, const void *src, size_t bytes)
{
  for (size_t o = 0; o < bytes; o += PIN_BYTES) {
    const size_t nb = bytes - o < PIN_BYTES ? bytes - o : PIN_BYTES;
    memcpy(G.pin, (const uint8_t *)src + o, nb);
    HK(hipMemcpy((uint8_t *)d + o, G.pin, nb, hipMemcpyHostToDevice));
  }
}
/* `count` values up / down, synchronously, through the pinned scratch when they fit it (a ciphertext is 141 KB): no staging copy inside the runtime, no malloc */
static void ct_to_dev(uint64_t *d, mpz_t *ct, size_t count)
{
  const size_t bytes = count * L_LIMBS * 8;
  uint64_t *h = bytes <= PIN_BYTES && G.pin ? (uint64_t *)G.pin : xmalloc(bytes);
  for (size_t j = 0; j < count; j++) to_limbs(h + j * L_LIMBS, ct[j]);
  HK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
  if ((uint8_t *)h != G.pin) free(h);
}
static void ct_from_dev(mpz_t *ct, const uint64_t *d, size_t count)
{
  const size_t bytes = count * L_LIMBS * 8;
  uint64_t *h = bytes <= PIN_BYTES && G.pin ? (uint64_t *)G.pin : xmalloc(bytes);
  HK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
  for (size_t j = 0; j < count; j++) from_limbs(ct[j], h + j * L_LIMBS);
  if ((uint8_t *)h != G.pin) free(h);
}

/* the key on the device (G.d_sk), uploaded only when its limbs differ from the ones uploaded last (src/benchmark_lwe.c:28-38 and the encryption loops of
 * src/snark.c:75-110 pass the same sk to every call) */
static const uint64_t *sk_resident(mpz_t *sk)
{
  gpu();
  const size_t bytes = (size_t)GAMMA_N * L_LIMBS * 8;
  for (size_t j = 0; j < GAMMA_N; j++) to_limbs(G.pin_sk + j * L_LIMBS, sk[j]);
  if (G.sk_valid && !memcmp(G.pin_sk, G.h_sk, bytes)) return G.d_sk;
  HK(hipMemcpy(G.d_sk, G.pin_sk, bytes, hipMemcpyHostToDevice));
  memcpy(G.h_sk, G.pin_sk, bytes);
  G.sk_valid = true;
  return G.d_sk;
}

/* ---- L0/L1: stream -------------------------------------------------------------------------------------- */
/* what struct aesctr's opaque `key` points to: the 40-byte seed and a host window of the stream.  The reference's callers draw the stream in small pieces (92 bytes per
 * mpz2_urandomb, 16-byte tails for remb): served one GPU round trip each, src/test_entropy.c took 68 s; the window fetches 64 KiB at a time. */
#define WIN_BYTES ((size_t)1 << 16)
typedef struct { uint8_t seed[40]; uint64_t win_pos; size_t win_len; uint8_t *win; } shim_key;
static uint8_t *d_win; /* device staging of the window and of large reads, grown on demand */
static size_t d_win_bytes;
static void stream_fetch(shim_key *k, uint64_t pos, void *out, size_t bytes)
{
  use_seed(k->seed);
  if (bytes > d_win_bytes) {
    if (d_win) HK(hipFree(d_win));
    d_win = NULL;
    HK(hipMalloc((void **)&d_win, bytes < WIN_BYTES ? WIN_BYTES : bytes));
    d_win_bytes = bytes < WIN_BYTES ? WIN_BYTES : bytes;
  }
  CK(mfh_keystream(G.ctx, pos, d_win, bytes));
  HK(hipMemcpy(out, d_win, bytes, hipMemcpyDeviceToHost));
}
/* stream bytes [pos, pos + bytes) into out */
static void stream_read(shim_key *k, uint64_t pos, void *out, size_t bytes)
{
  if (bytes > WIN_BYTES / 2) { /* bulk reads go straight through */
    stream_fetch(k, pos, out, bytes);
    return;
  }
  if (!k->win || pos < k->win_pos || pos + bytes > k->win_pos + k->win_len) {
    if (!k->win) k->win = xmalloc(WIN_BYTES);
    k->win_pos = pos & ~(uint64_t)15;
    k->win_len = WIN_BYTES;
    stream_fetch(k, k->win_pos, k->win, WIN_BYTES);
  }
  memcpy(out, k->win + (pos - k->win_pos), bytes);
}

void aesctr_init(aesctr_ptr s, const uint8_t *key, const uint64_t nonce)
{
  KEEP_ERRNO;
  s->rem = 0;
  s->ctr = 0;
  shim_key *k = calloc(1, sizeof *k);
  if (!k) { perror("Failed malloc"); return; }
  memcpy(k->seed, &nonce, 8);
  memcpy(k->seed + 8, key, 32);
  s->key = k;
  s->nonce = nonce;
}

void aesctr_clear(aesctr_ptr s)
{
  KEEP_ERRNO;
  if (!s) return;
  if (s->key) {
    shim_key *k = s->key;
    if (k->win) { explicit_bzero(k->win, WIN_BYTES); free(k->win); }
    memset(k, 0, sizeof(shim_key));
    free(k);
  }
  memset(s, 0, sizeof(struct aesctr));
}

/* absolute stream position of the next byte */
static uint64_t stream_pos(const struct aesctr *s) { return s->ctr * 16 - s->rem; }
/* blk: the 16 stream bytes at ((pos + 15) / 16 - 1) * 16 when the caller already has them (NULL: fetched through the window if pos is not block-aligned) */
static void stream_set_pos_blk(struct aesctr *s, uint64_t pos, const uint8_t *blk)
{
  s->ctr = (pos + 15) / 16;
  s->rem = (size_t)(s->ctr * 16 - pos);
  /* remb mirrors the reference: the unread tail of the last generated block (src/aes.c:135-142) */
  if (s->rem) {
    uint8_t fetched[16];
    if (!blk) {
      stream_read((shim_key *)s->key, (s->ctr - 1) * 16, fetched, 16);
      blk = fetched;
    }
    memcpy(s->remb, blk + 16 - s->rem, s->rem);
  }
}
static void stream_set_pos(struct aesctr *s, uint64_t pos) { stream_set_pos_blk(s, pos, NULL); }

void aesctr_prg(aesctr_ptr s, void *out, size_t bytes)
{
  KEEP_ERRNO;
  if (!bytes) return;
  gpu();
  uint64_t pos = stream_pos(s);
  stream_read((shim_key *)s->key, pos, out, bytes);
  stream_set_pos(s, pos + bytes);
}

void rng_init(rng_t rs, uint8_t *rseed)
{
  KEEP_ERRNO;
  uint64_t nonce;
  memcpy(&nonce, rseed, 8);
  aesctr_init((aesctr_ptr)rs, rseed + 8, nonce);
}
void rng_clear(rng_t rs) { aesctr_clear((aesctr_ptr)rs); }
void rng_seek(rng_t rs, size_t count) { stream_set_pos((aesctr_ptr)rs, count); }

void mpz2_urandomb(mpz_ptr rop, rng_t rs, size_t nbits)
{
  KEEP_ERRNO;
  size_t limbs = (nbits + 63) / 64, bytes = nbits / 8;
  uint64_t *buf = calloc(limbs ? limbs : 1, 8);
  aesctr_prg((aesctr_ptr)rs, buf, bytes);
  if (limbs) buf[limbs - 1] &= ~0ULL >> (limbs * 64 - nbits);
  mpz_import(rop, limbs, -1, 8, 0, 0, buf);
  free(buf);
}
void mpz2_urandomb2(mpz_ptr rop, size_t nbits)
{
  KEEP_ERRNO;
  size_t limbs = (nbits + 63) / 64, bytes = nbits / 8;
  uint64_t *buf = calloc(limbs ? limbs : 1, 8);
  shim_random(buf, bytes);
  if (limbs) buf[limbs - 1] &= ~0ULL >> (limbs * 64 - nbits);
  mpz_import(rop, limbs, -1, 8, 0, 0, buf);
  free(buf);
}

/* ---- L2: LWE -------------------------------------------------------------------------------------------- */
static void initv(mpz_t *v, size_t n) { for (size_t i = 0; i < n; i++) mpz_init2(v[i], GAMMA_LOGQ); }
static void clearv(mpz_t *v, size_t n) { for (size_t i = 0; i < n; i++) mpz_clear(v[i]); }
void key_gen(sk_t sk) { initv(sk, GAMMA_N); for (size_t i = 0; i < GAMMA_N; i++) mpz2_urandomb2(sk[i], GAMMA_LOGQ); }
/* (the reference frees the limbs as they are, src/lwe.c:36-43.  The shim holds copies of the key last used -- pageable, pinned and on the device, sk_resident() -- and they
 * go with it: a caller that is done with its key leaves none behind in this library.  The next call under another key uploads that one, as after any key change.) */
static void sk_forget(void)
{
  if (!G.h_sk) return;
  explicit_bzero(G.h_sk, (size_t)GAMMA_N * L_LIMBS * 8);
  explicit_bzero(G.pin_sk, (size_t)GAMMA_N * L_LIMBS * 8);
  if (G.ctx) { (void)hipDeviceSynchronize(); (void)hipMemset(G.d_sk, 0, (size_t)GAMMA_N * L_LIMBS * 8); }
  G.sk_valid = false;
}
void key_clear(sk_t sk)
{
  KEEP_ERRNO;
  sk_forget();
  clearv(sk, GAMMA_N);
}
void ct_init(ct_t ct) { initv(ct, GAMMA_N + 1); }
void ct_clear(ct_t ct) { clearv(ct, GAMMA_N + 1); }
void errdist_uniform(mpz_t e) { mpz2_urandomb2(e, GAMMA_LOG_SIGMA + 3); }
void ct_zero(ct_t rop) { for (size_t i = 0; i <= GAMMA_N; i++) mpz_set_ui(rop[i], 0); }

void ct_export(uint8_t *buf, ct_t ct)
{
  KEEP_ERRNO;
  bzero(buf, CT_BYTES);
  if (mpz_sizeinbase(ct[GAMMA_N], 2) > 8 * CT_BYTES) die("ct_export: b does not fit CT_BYTES");
  mpz_export(buf, NULL, -1, 1, -1, 0, ct[GAMMA_N]);
}

/* a-part of a row at the rng's position, advancing it by CTR_CT (mpz2_urandommv, src/lwe.c:90,101,124) */
static void sample_a(ct_t ct, rng_t rng)
{
  struct aesctr *s = (struct aesctr *)rng;
  gpu();
  use_seed(((shim_key *)s->key)->seed);
  const uint64_t pos = stream_pos(s), end = pos + CTR_CT, endblk = ((end + 15) / 16 - 1) * 16;
  /* the row's a part and the stream block it ends in (the caller's remb when the row ends mid-block) in one round trip, through pinned memory */
  uint64_t *down = (uint64_t *)(G.pin + 4096);
  CK(mfh_sample_rows(G.ctx, pos, 1, G.d_ct[0]));
  CK(mfh_keystream(G.ctx, endblk, (uint8_t *)G.d_ct[0] + (size_t)GAMMA_N * L_LIMBS * 8, 16));
  HK(hipMemcpyAsync(down, G.d_ct[0], (size_t)GAMMA_N * L_LIMBS * 8 + 16, hipMemcpyDeviceToHost, NULL));
  HK(hipStreamSynchronize(NULL));
  for (size_t j = 0; j < GAMMA_N; j++) from_limbs(ct[j], down + j * L_LIMBS);
  stream_set_pos_blk(s, end, (const uint8_t *)(down + (size_t)GAMMA_N * L_LIMBS));
}

void ct_import(ct_t ct, rng_t rng, uint8_t *buf)
{
  KEEP_ERRNO;
  sample_a(ct, rng);
  mpz_import(ct[GAMMA_N], LOGQ_BYTES, -1, 1, -1, 0, buf);
}
void decompress_encryption(ct_t c, rng_t rng, mpz_t b)
{
  KEEP_ERRNO;
  sample_a(c, rng);
  mpz_set(c[GAMMA_N], b);
}

void regev_encrypt2(ct_t c, rng_t rs, sk_t sk, mpz_t m, void (*chi)(mpz_t))
{
  KEEP_ERRNO;
  if (mpz_sgn(m) < 0 || mpz_cmp_ui(m, GAMMA_P) >= 0) die("regev_encrypt2: message must be < p (src/lwe.c:80)");
  struct aesctr *s = (struct aesctr *)rs;
  gpu();
  use_seed(((shim_key *)s->key)->seed);
  const uint64_t pos = stream_pos(s);
  const double T0 = tnow();
  /* The call is latency: seven runtime calls of 5 - 10 us each and two short kernels.  So the SECRET half goes first, on the default stream -- error draw, key check
   * (uploaded only when it changed), ONE 100-byte upload of [error limbs | message], the encryption of the row, 96 bytes of b down -- and the PUBLIC half (the row's a
   * part, which the caller receives too, and the stream block the row ends in: its unread tail is the caller's remb, CTR_CT = 16 * 8452 + 8, every other row ends
   * mid-block) is queued behind it on a second stream and runs beside it on the GPU: one copy down, turned into mpz_t's while the encryption finishes. */
  uint64_t *up = (uint64_t *)G.pin, *down = (uint64_t *)(G.pin + 4096), *down_b = (uint64_t *)(G.pin + 2048);
  const uint64_t end = pos + CTR_CT, endblk = ((end + 15) / 16 - 1) * 16;
  mpz_t e;
  mpz_init(e);
  (*chi)(e);
  uint8_t sign;
  shim_random(&sign, 1); /* the reference burns one byte here (src/lwe.c:87) */
  const uint64_t *d_sk = sk_resident(sk);
  to_limbs(up, e);
  mpz_clear(e);
  const uint32_t mh = (uint32_t)mpz_get_ui(m);
  memcpy(up + L_LIMBS, &mh, 4);
  const double T1 = tnow();
  uint8_t *d_b = (uint8_t *)G.d_ct[1] + 128; /* (behind the 100 bytes of [error | message]) */
  HK(hipMemcpyAsync(G.d_ct[1], up, ENC_UP, hipMemcpyHostToDevice, NULL));
  CK(mfh_encrypt_rows(G.ctx, pos, 1, d_sk, (const uint32_t *)(G.d_ct[1] + L_LIMBS), G.d_ct[1], d_b));
  HK(hipMemcpyAsync(down_b, d_b, L_LIMBS * 8, hipMemcpyDeviceToHost, NULL));
  HK(hipEventRecord(G.ev_small, NULL));           /* (what the call waits for: b is down) */
  HK(hipMemsetAsync(G.d_ct[1], 0, ENC_UP, NULL)); /* the error leaves neither the device scratch (wiped behind the call's back, before anything else runs on this stream) ... */
  const double T2 = tnow();
  CK(mfh_set_stream(G.ctx, G.s2));
  CK(mfh_sample_rows(G.ctx, pos, 1, G.d_ct[0]));
  CK(mfh_keystream(G.ctx, endblk, (uint8_t *)G.d_ct[0] + (size_t)GAMMA_N * L_LIMBS * 8, 16));
  CK(mfh_set_stream(G.ctx, NULL));
  HK(hipMemcpyAsync(down, G.d_ct[0], (size_t)GAMMA_N * L_LIMBS * 8 + 16, hipMemcpyDeviceToHost, G.s2));
  const double T3 = tnow();
  HK(hipStreamSynchronize(G.s2));
  const double T4 = tnow();
  for (size_t j = 0; j < GAMMA_N; j++) from_limbs(c[j], down + j * L_LIMBS);
  const double T5 = tnow();
  HK(hipEventSynchronize(G.ev_small));
  const double T6 = tnow();
  explicit_bzero(up, ENC_UP); /* ... nor the pinned one behind (the upload it fed is over: b came after it) */
  mpz_import(c[GAMMA_N], LOGQ_BYTES, -1, 1, -1, 0, down_b);
  stream_set_pos_blk(s, end, (const uint8_t *)(down + (size_t)GAMMA_N * L_LIMBS));
  if (tracing()) {
    static double acc[7];
    static unsigned calls;
    const double T7 = tnow(), t[8] = { T0, T1, T2, T3, T4, T5, T6, T7 };
    for (int q = 0; q < 7; q++) acc[q] += t[q + 1] - t[q];
    if (++calls % 256 == 0) {
      fprintf(stderr, "regev_encrypt2 x256 (us per call): error + key check + pack %.1f, queue encryption %.1f, queue public half %.1f, wait public half %.1f, a -> mpz_t %.1f, wait b %.1f, b + stream state %.1f\n",
              acc[0] * 1e3 / 256, acc[1] * 1e3 / 256, acc[2] * 1e3 / 256, acc[3] * 1e3 / 256, acc[4] * 1e3 / 256, acc[5] * 1e3 / 256, acc[6] * 1e3 / 256);
      memset(acc, 0, sizeof acc);
    }
  }
}

/* regev_encrypt2 + ct_export (src/lwe.c:78-97,115-119) for `count` messages under one key: row k is what regev_encrypt2 produces with the stream at rs + k CTR_CT
 * and the k-th error draw; c8[k] receives its exported b (the a part is public: ct_import regenerates it from the stream position), and rs ends count rows further.
 * chi == NULL or errdist_uniform: the errors are drawn in bulk (per row the GAMMA_LOG_SIGMA + 3 bits of errdist_uniform, then the sign byte the reference burns,
 * src/lwe.c:60-63,85-87) on a few host threads; any other chi is called once per row, in order.  Chunks of ENC_CHUNK rows: the host draws chunk k + 1 while the GPU
 * encrypts chunk k.  Not in the reference, whose loops (src/benchmark_lwe.c:28-33, src/snark.c:75-110) encrypt one message at a time. */
struct enc_fill { uint8_t *up; mpz_t *ms; size_t base; };
static void enc_fill_uniform(size_t lo, size_t hi, void *arg)
{
  struct enc_fill *f = arg;
  enum { EB = (GAMMA_LOG_SIGMA + 3) / 8, DRAW = EB + 1, RUN = 512 };
  uint8_t tmp[RUN * DRAW];
  for (size_t r0 = lo; r0 < hi; r0 += RUN) {
    const size_t nr = hi - r0 < RUN ? hi - r0 : RUN;
    shim_random(tmp, nr * DRAW);
    for (size_t r = 0; r < nr; r++) {
      uint8_t *row = f->up + (r0 + r) * (L_LIMBS * 8);
      memcpy(row, tmp + r * DRAW, EB); /* (mpz2_urandomb2(e, 559) draws 559 / 8 = 69 whole bytes: the value has 552 random bits) */
      memset(row + EB, 0, L_LIMBS * 8 - EB);
      mpz_srcptr m = f->ms[f->base + r0 + r];
      if (mpz_sgn(m) < 0 || mpz_cmp_ui(m, GAMMA_P) >= 0) die("mfuoco_encrypt_batch: message must be < p (src/lwe.c:80)");
      const uint32_t mh = (uint32_t)mpz_get_ui(m);
      memcpy(f->up + ENC_CHUNK * L_LIMBS * 8 + (r0 + r) * 4, &mh, 4);
    }
  }
  explicit_bzero(tmp, sizeof tmp);
}
void mfuoco_encrypt_batch2(uint8_t (*c8)[CT_BYTES], rng_t rs, sk_t sk, mpz_t *ms, size_t count, void (*chi)(mpz_t))
{
  KEEP_ERRNO;
  if (!count) return;
  struct aesctr *s = (struct aesctr *)rs;
  const uint64_t *d_sk = sk_resident(sk);
  use_seed(((shim_key *)s->key)->seed);
  if (!G.d_enc[0])
    for (int i = 0; i < 2; i++) {
      HK(hipMalloc((void **)&G.d_enc[i], ENC_CHUNK * (ENC_UP + CT_BYTES)));
      HK(hipHostMalloc((void **)&G.enc_pin[i], ENC_CHUNK * (ENC_UP + CT_BYTES), hipHostMallocDefault));
      HK(hipEventCreateWithFlags(&G.enc_ev[i], hipEventDisableTiming));
    }
  const uint64_t pos = stream_pos(s);
  const bool bulk = !chi || chi == errdist_uniform;
  size_t done[2] = { 0, 0 }, at[2] = { 0, 0 };
  int slot = 0;
  for (size_t k0 = 0; k0 < count || done[0] || done[1]; k0 += ENC_CHUNK, slot ^= 1) {
    if (done[slot]) { /* the chunk that used this slot two rounds ago: its b's are down */
      HK(hipEventSynchronize(G.enc_ev[slot]));
      memcpy(c8[at[slot]], G.enc_pin[slot] + ENC_CHUNK * ENC_UP, done[slot] * CT_BYTES);
      done[slot] = 0;
    }
    if (k0 >= count) continue;
    const size_t nk = count - k0 < ENC_CHUNK ? count - k0 : ENC_CHUNK;
    uint8_t *up = G.enc_pin[slot];
    if (bulk) {
      struct enc_fill f = { up, ms, k0 };
      parallel_for_grain(nk, 2048, enc_fill_uniform, &f);
    } else {
      mpz_t e;
      mpz_init(e);
      for (size_t r = 0; r < nk; r++) {
        if (mpz_sgn(ms[k0 + r]) < 0 || mpz_cmp_ui(ms[k0 + r], GAMMA_P) >= 0) die("mfuoco_encrypt_batch: message must be < p (src/lwe.c:80)");
        (*chi)(e);
        uint8_t sign;
        shim_random(&sign, 1);
        uint64_t eh[L_LIMBS];
        to_limbs(eh, e);
        memcpy(up + r * sizeof eh, eh, sizeof eh);
        const uint32_t mh = (uint32_t)mpz_get_ui(ms[k0 + r]);
        memcpy(up + ENC_CHUNK * sizeof eh + r * 4, &mh, 4);
      }
      mpz_clear(e);
    }
    /* slot layout, host and device alike: [ENC_CHUNK x L limbs of error | ENC_CHUNK x uint32 message | ENC_CHUNK x CT_BYTES of b] */
    uint8_t *d = G.d_enc[slot];
    HK(hipMemcpyAsync(d, up, nk * L_LIMBS * 8, hipMemcpyHostToDevice, NULL));
    HK(hipMemcpyAsync(d + ENC_CHUNK * L_LIMBS * 8, up + ENC_CHUNK * L_LIMBS * 8, nk * 4, hipMemcpyHostToDevice, NULL));
    CK(mfh_encrypt_rows(G.ctx, pos + k0 * CTR_CT, nk, d_sk, (const uint32_t *)(d + ENC_CHUNK * L_LIMBS * 8), (const uint64_t *)d, d + ENC_CHUNK * ENC_UP));
    HK(hipMemcpyAsync(G.enc_pin[slot] + ENC_CHUNK * ENC_UP, d + ENC_CHUNK * ENC_UP, nk * CT_BYTES, hipMemcpyDeviceToHost, NULL));
    HK(hipEventRecord(G.enc_ev[slot], NULL));
    done[slot] = nk;
    at[slot] = k0;
  }
  for (int i = 0; i < 2; i++) { /* the errors are secret (with the public a and b an error gives <sk, a> away): neither the pinned nor the device staging keeps them */
    explicit_bzero(G.enc_pin[i], ENC_CHUNK * ENC_UP);
    HK(hipMemsetAsync(G.d_enc[i], 0, ENC_CHUNK * ENC_UP, NULL));
  }
  stream_set_pos(s, pos + count * CTR_CT);
}
void mfuoco_encrypt_batch(uint8_t (*c8)[CT_BYTES], rng_t rs, sk_t sk, mpz_t *ms, size_t count) { mfuoco_encrypt_batch2(c8, rs, sk, ms, count, NULL); }

/* regev_decrypt (src/lwe.c:105-111) of `count` SEED-COMPRESSED ciphertexts -- the form mfuoco_encrypt_batch / ct_export produce and a CRS holds: row k's a part is
 * regenerated on the device from the stream at rs + k CTR_CT (what ct_import does, src/lwe.c:122-126), c8[k] is its b.  ms[k] initialised by the caller; rs ends count
 * rows further. */
void mfuoco_decrypt_rows_batch(mpz_t *ms, rng_t rs, sk_t sk, uint8_t (*c8)[CT_BYTES], size_t count)
{
  KEEP_ERRNO;
  if (!count) return;
  struct aesctr *s = (struct aesctr *)rs;
  const uint64_t pos = stream_pos(s);
  if (pos & 7) { /* the batched kernel wants every row to start at byte 0 or 8 of an AES block (any position a row-wise caller reaches); anything else: one row at a time */
    ct_t ct;
    ct_init(ct);
    for (size_t k = 0; k < count; k++) {
      ct_import(ct, rs, c8[k]);
      regev_decrypt(ms[k], sk, ct);
    }
    ct_clear(ct);
    return;
  }
  const uint64_t *d_sk = sk_resident(sk);
  use_seed(((shim_key *)s->key)->seed);
  uint8_t *d_c8 = NULL;
  uint32_t *d_m = NULL, *hm = xmalloc(count * 4);
  HK(hipMalloc((void **)&d_c8, count * CT_BYTES));
  HK(hipMalloc((void **)&d_m, count * 4));
  HK(hipMemcpy(d_c8, c8, count * CT_BYTES, hipMemcpyHostToDevice));
  CK(mfh_decrypt_rows(G.ctx, pos, count, d_sk, d_c8, d_m));
  HK(hipMemcpy(hm, d_m, count * 4, hipMemcpyDeviceToHost));
  for (size_t k = 0; k < count; k++) mpz_set_ui(ms[k], hm[k]);
  free(hm);
  HK(hipFree(d_c8));
  HK(hipFree(d_m));
  stream_set_pos(s, pos + count * CTR_CT);
}

void regev_decrypt(mpz_t m, sk_t sk, ct_t ct)
{
  KEEP_ERRNO;
  const uint64_t *d_sk = sk_resident(sk);
  uint64_t *up = (uint64_t *)G.pin;
  for (size_t j = 0; j <= GAMMA_N; j++) to_limbs(up + j * L_LIMBS, ct[j]);
  HK(hipMemcpyAsync(G.d_ct[0], up, CTL * 8, hipMemcpyHostToDevice, NULL));
  CK(mfh_decrypt(G.ctx, d_sk, G.d_ct[0], 1, G.d_co));
  uint32_t *out = (uint32_t *)(G.pin + CTL * 8);
  HK(hipMemcpyAsync(out, G.d_co, 4, hipMemcpyDeviceToHost, NULL));
  HK(hipStreamSynchronize(NULL));
  mpz_set_ui(m, *out);
}

void mpz_add_dotp(mpz_t rop, mpz_t a[], mpz_t b[], size_t len)
{
  KEEP_ERRNO;
  gpu();
  uint64_t r[L_LIMBS], *da, *db, *dr;
  /* rop may be an unreduced accumulator in the reference; only its value mod 2^704 survives the final modq */
  mpz_t t;
  mpz_init(t);
  mpz_fdiv_r_2exp(t, rop, 64 * K_LIMBS);
  to_limbs(r, t);
  mpz_clear(t);
  HK(hipMalloc((void **)&da, len * L_LIMBS * 8 + 8));
  HK(hipMalloc((void **)&db, len * L_LIMBS * 8 + 8));
  HK(hipMalloc((void **)&dr, L_LIMBS * 8));
  ct_to_dev(da, a, len);
  ct_to_dev(db, b, len);
  HK(hipMemcpy(dr, r, sizeof r, hipMemcpyHostToDevice));
  CK(mfh_add_dotp(G.ctx, dr, da, db, len));
  HK(hipMemcpy(r, dr, sizeof r, hipMemcpyDeviceToHost));
  from_limbs(rop, r);
  HK(hipFree(da)); HK(hipFree(db)); HK(hipFree(dr));
}

void ct_smudge(ct_t ct)
{
  KEEP_ERRNO;
  gpu();
  uint8_t mag[GAMMA_LOG_SMUDGING / 8], sign;
  shim_random(mag, sizeof mag);
  shim_random(&sign, 1);
  ct_to_dev(G.d_ct[0], ct, GAMMA_N + 1);
  CK(mfh_ct_smudge(G.ctx, G.d_ct[0], 1, mag, sizeof mag, &sign));
  ct_from_dev(ct + GAMMA_N, G.d_ct[0] + (size_t)GAMMA_N * L_LIMBS, 1);
}

void ct_add(ct_t rop, ct_t a, ct_t b)
{
  KEEP_ERRNO;
  gpu();
  ct_to_dev(G.d_ct[0], a, GAMMA_N + 1);
  ct_to_dev(G.d_ct[1], b, GAMMA_N + 1);
  CK(mfh_ct_add(G.ctx, G.d_ct[2], G.d_ct[0], G.d_ct[1], 1));
  ct_from_dev(rop, G.d_ct[2], GAMMA_N + 1);
}
void ct_mul_ui(ct_t rop, ct_t a, uint64_t b)
{
  KEEP_ERRNO;
  gpu();
  if (b >= GAMMA_P) die("ct_mul_ui: scalar must be < p (src/lwe.c:133)");
  ct_to_dev(G.d_ct[0], a, GAMMA_N + 1);
  CK(mfh_ct_mul_ui(G.ctx, G.d_ct[2], G.d_ct[0], (uint32_t)b, 1));
  ct_from_dev(rop, G.d_ct[2], GAMMA_N + 1);
}
void ct_addmul_ui(ct_t rop, ct_t a, uint64_t b)
{
  KEEP_ERRNO;
  gpu();
  if (b >= GAMMA_P) die("ct_addmul_ui: scalar must be < p (src/lwe.c:143)");
  ct_to_dev(G.d_ct[0], a, GAMMA_N + 1);
  ct_to_dev(G.d_ct[2], rop, GAMMA_N + 1);
  CK(mfh_ct_addmul_ui(G.ctx, G.d_ct[2], G.d_ct[0], (uint32_t)b, 1));
  ct_from_dev(rop, G.d_ct[2], GAMMA_N + 1);
}

void eval_poly(ct_t rop, rng_t rng, uint8_t (*c8)[CT_BYTES], nmod_poly_t p, size_t d)
{
  KEEP_ERRNO;
  struct aesctr *s = (struct aesctr *)rng;
  use_seed(((shim_key *)s->key)->seed);
  if (d > 2 * (size_t)GAMMA_D + GAMMA_M) die("eval_poly: more rows than a CRS holds");
  uint32_t *co = malloc(d * 4 + 4);
  for (size_t i = 0; i < d; i++) {
    mp_limb_t c = nmod_poly_get_coeff_ui(p, i);
    if (c >= GAMMA_P) die("eval_poly: coefficient must be < p (src/lwe.c:143)");
    co[i] = (uint32_t)c;
  }
  uint64_t pos = stream_pos(s);
  const double t_in = tnow();
  h2d_staged(G.d_co, co, d * 4);
  const double t_co = tnow();
  h2d_staged(G.d_c8, c8, d * CT_BYTES); /* (c8 is typically a read-only file mapping: src/benchmark_eval.c:62-66) */
  const double t_c8 = tnow();
  free(co);
  ct_to_dev(G.d_ct[2], rop, GAMMA_N + 1); /* eval_poly accumulates into rop (src/lwe.c:183) */
  const double t_up = tnow();
  if (tracing()) fprintf(stderr, "eval_poly uploads: coefficients %.2f ms, rows %.2f, accumulator %.2f\n", t_co - t_in, t_c8 - t_co, t_up - t_c8);
  CK(mfh_eval_rows(G.ctx, pos, d, G.d_c8, G.d_co, NULL, G.d_ct[2], NULL, 1));
  if (tracing()) CK(mfh_sync(G.ctx));
  const double t_gpu = tnow();
  ct_from_dev(rop, G.d_ct[2], GAMMA_N + 1);
  stream_set_pos(s, pos + d * CTR_CT);
  if (tracing()) fprintf(stderr, "eval_poly(%zu rows): uploads %.2f ms, GPU %.2f, download + mpz_t + stream state %.2f\n", d, t_up - t_in, t_gpu - t_up, tnow() - t_gpu);
}

/* ---- L3: SSP (host harness, src/ssp.c) --------------------------------------------------------------------- */
void nmod_poly_export(void *buf_, nmod_poly_t *pp, size_t degree)
{
  KEEP_ERRNO;
  uint64_t *buf = buf_;
  for (size_t i = 0; i < degree; i++) buf[i] = nmod_poly_get_coeff_ui(*pp, i);
}
void nmod_poly_import(nmod_poly_t *pp, void *buf_, size_t degree)
{
  KEEP_ERRNO;
  uint64_t *buf = buf_;
  for (size_t i = 0; i < degree; i++) nmod_poly_set_coeff_ui(*pp, i, buf[i]);
}
void random_ssp(mpz_t input, uint8_t *circuit)
{
  KEEP_ERRNO;
  const size_t buflen = 8 * GAMMA_D;
  uint8_t *buf = malloc(buflen);
  uint64_t *t = calloc(GAMMA_D, 8), *out;
  mpz2_urandomb2(input, GAMMA_M);
  for (size_t i = 0; i < GAMMA_M; i++) {
    shim_random(buf, buflen);
    out = (uint64_t *)(circuit + 8 * GAMMA_D * (i + 1));
    int take = i == 0 || mpz_tstbit(input, i - 1);
    for (size_t k = 0; k < GAMMA_D; k++) {
      uint64_t x;
      memcpy(&x, buf + 8 * k, 8);
      out[k] = x % GAMMA_P;
      if (take) t[k] = (t[k] + out[k]) % GAMMA_P;
    }
  }
  t[0] = (t[0] + GAMMA_P - 1) % GAMMA_P;
  memcpy(circuit, t, buflen);
  free(buf); free(t);
}

/* ---- L4: SNARK -------------------------------------------------------------------------------------------- */
void proof_init(proof_t pi) { ct_init(pi->h); ct_init(pi->hat_h); ct_init(pi->hat_v); ct_init(pi->v_w); ct_init(pi->b_w); }
void proof_clear(proof_t pi) { ct_clear(pi->h); ct_clear(pi->hat_h); ct_clear(pi->hat_v); ct_clear(pi->v_w); ct_clear(pi->b_w); }
void crs_init(crs_t crs)
{
  KEEP_ERRNO;
  shim_random(crs->seed, sizeof(rseed_t));
  crs->s = malloc(CT_BYTES * GAMMA_D);
  crs->as = malloc(CT_BYTES * GAMMA_D);
  crs->v = malloc(CT_BYTES * GAMMA_M);
  crs->t = malloc(CT_BYTES);
  if (!crs->s || !crs->as || !crs->v || !crs->t) perror("Error allocating memory");
}
void crs_clear(crs_t crs) { free(crs->s); free(crs->as); free(crs->v); free(crs->t); }

static uint64_t rand_modp_(void)
{
  uint64_t r;
  shim_random(&r, 8);
  return r % GAMMA_P;
}

static void ssp_resident(ssp_t ssp)
{
  if (G.ssp_host == ssp && G.d_ssp) return;
  if (!G.d_ssp) HK(hipMalloc((void **)&G.d_ssp, (size_t)(GAMMA_M + 3) * GAMMA_D * 4));
  CK(mfh_ssp_upload(G.ctx, ssp, G.d_ssp, 0, GAMMA_M + 3));
  CK(mfh_ssp_prepare(G.ctx, G.d_ssp));
  G.ssp_host = ssp;
}

/* Everything a first prover() would otherwise pay for besides the proof itself -- the context's scratch (hipMalloc), the code objects of the prover's kernels (loaded
 * on first launch), the NTT tables of the polynomial step -- is paid here, by ONE proof of the all-zero witness with zero randomness over the CRS and SSP setup() has
 * just put on the device, its result discarded: public inputs only.  src/benchmark_snark.c:70-74 times the first prover() after setup(); without this it measured
 * 22 ms for 9.9 ms of GPU work.  $MFUOCO_GPU_WARM=0 skips it. */
static void shim_warm_prover(const void *rows)
{
  if (!shim_warm_on()) return;
  static const uint8_t zero_bits[(GAMMA_M + 7) / 8 + 8], zero_mag[5 * (GAMMA_LOG_SMUDGING / 8)], zero_sign[5];
  /* (over the row image setup() has just written when there is one: the path the first prover() will take) */
  if (rows) CK(mfh_crs_set_resident(G.ctx, rows));
  int rc = mfh_prove(G.ctx, G.d_crs, G.d_ssp, zero_bits, 0, zero_mag, GAMMA_LOG_SMUDGING / 8, zero_sign, G.d_proof);
  if (rows) CK(mfh_crs_set_resident(G.ctx, NULL));
  if (rc != MFH_OK) die("mfh_prove (warm-up)");
  CK(mfh_sync(G.ctx));
}
static void *rows_image_reserve(void);
static void rows_image_register(const uint64_t dg[2]);
static bool resident_on(void);

static void *setup_ssp_thread(void *ssp)
{
  HK(hipSetDevice(G.device)); /* (a new thread starts on device 0) */
  ssp_resident((uint8_t *)ssp);
  return NULL;
}
void setup(crs_t crs, vrs_t vrs, ssp_t ssp)
{
  KEEP_ERRNO;
  gpu();
  use_seed(crs->seed);
  /* the SSP (5.7 GB at the NDEBUG size) starts crossing PCIe at once, on a helper thread (mfh_ssp_upload has its own staging threads; this one only waits for them), while
   * THIS thread draws the secrets from the OS -- the only thread that does, so the draws keep the reference's order.  Nothing below touches the GPU before the join. */
  G.ssp_host = NULL;
  /* "... writing the expanded rows to HBM as a by-product, so the prover starts with a materialised CRS".  The a parts of the 2D + M rows depend on the
   * public seed alone, so they are written out FIRST, in the layout prover() streams, on the shim's second stream: 12 ms of AES on the CU that run while the 5.7 GB of
   * the SSP cross PCIe (in 16 row slices, so that the upload's small reduction kernels find CUs between them).  The b's are filled in below, once the encryptions
   * exist, and the image is registered under (seed, digest of the compressed CRS): the FIRST prover() under this CRS streams it.  Not when the image does not fit beside
   * the calls' scratch, nor with $MFUOCO_GPU_RESIDENT_CRS=0.  (Queued before the upload thread starts: one thread at a time talks to the context.) */
  drop_image(); /* (the image buffer and G.d_crs are about to be rewritten) */
  G.staged_digest_valid = false;
  void *img = resident_on() ? rows_image_reserve() : NULL;
  if (img) {
    const size_t nrows_img = 2 * (size_t)GAMMA_D + GAMMA_M, rb = mfh_resident_row_bytes(G.ctx), slice = (nrows_img + 15) / 16;
    CK(mfh_set_stream(G.ctx, G.s2));
    for (size_t r0 = 0; r0 < nrows_img; r0 += slice)
      CK(mfh_crs_expand(G.ctx, (uint64_t)r0 * CTR_CT, nrows_img - r0 < slice ? nrows_img - r0 : slice, NULL, (uint8_t *)img + r0 * rb));
    HK(hipEventRecord(G.ev_img, G.s2));
    CK(mfh_set_stream(G.ctx, NULL));
  }
  pthread_t ssp_th;
  const bool ssp_bg = pthread_create(&ssp_th, NULL, setup_ssp_thread, ssp) == 0;
  vrs->alpha = rand_modp_();
  vrs->beta = rand_modp_();
  vrs->s = rand_modp_();
  key_gen(vrs->sk);
  const size_t rows = 2 * (size_t)GAMMA_D + GAMMA_M;
  /* errors: 559-bit draws (errdist_uniform) + the ineffective sign byte per encryption (src/lwe.c:85-87) */
  /* (drawn in ONE piece on this thread, in the reference's order -- 69 + 1 bytes per encryption --, then cut up: 175 000 system calls were 0.1 s of this function) */
  const double t_in = tnow();
  enum { EB = (GAMMA_LOG_SIGMA + 3) / 8, DRAW = EB + 1 };
  uint64_t *err = xcalloc(rows * L_LIMBS, 8);
  uint8_t *tape = xmalloc(rows * DRAW);
  shim_random(tape, rows * DRAW);
  for (size_t i = 0; i < rows; i++) memcpy(err + i * L_LIMBS, tape + i * DRAW, EB);
  explicit_bzero(tape, rows * DRAW);
  free(tape);
  const double t_drawn = tnow();
  if (ssp_bg) pthread_join(ssp_th, NULL); else ssp_resident(ssp);
  const double t_ssp = tnow();
  if (!G.d_err) HK(hipMalloc((void **)&G.d_err, rows * L_LIMBS * 8));
  HK(hipMemcpy(G.d_err, err, rows * L_LIMBS * 8, hipMemcpyHostToDevice));
  explicit_bzero(err, rows * L_LIMBS * 8); /* the encryption errors are part of the trapdoor: not left on the heap */
  free(err);
  sk_resident(vrs->sk);
  CK(mfh_setup(G.ctx, G.d_ssp, (uint32_t)vrs->alpha, (uint32_t)vrs->beta, (uint32_t)vrs->s, G.d_sk, G.d_err, G.d_crs));
  if (img) { /* the image's b column, behind the a parts (s2) and the encryptions (this stream) */
    HK(hipStreamWaitEvent(NULL, G.ev_img, 0));
    CK(mfh_crs_image_set_b(G.ctx, 0, rows, G.d_crs, img));
  }
  HK(hipMemsetAsync(G.d_err, 0, rows * L_LIMBS * 8, NULL)); /* (the errors are not kept on the device either) */
  HK(hipMemcpy(crs->s, G.d_crs, CT_BYTES * GAMMA_D, hipMemcpyDeviceToHost));
  HK(hipMemcpy(crs->as, G.d_crs + CT_BYTES * GAMMA_D, CT_BYTES * GAMMA_D, hipMemcpyDeviceToHost));
  HK(hipMemcpy(crs->t, G.d_crs + 2 * CT_BYTES * GAMMA_D, CT_BYTES, hipMemcpyDeviceToHost));
  HK(hipMemcpy(crs->v, G.d_crs + (2 * GAMMA_D + 1) * CT_BYTES, CT_BYTES * (GAMMA_M - 1), hipMemcpyDeviceToHost));
  const double t_crs = tnow();
  if (img) {
    CK(mfh_digest128(G.ctx, G.d_crs, rows * CT_BYTES, G.staged_digest));
    G.staged_digest_valid = true;
    rows_image_register(G.staged_digest);
  }
  const double t_img = tnow();
  shim_warm_prover(img);
  if (tracing())
    fprintf(stderr, "setup(): key + error draws %.2f ms (the SSP upload runs beside them), rest of the SSP upload (%.2f GB) + quotient precomputation %.2f, uploads + encryptions + CRS download %.2f (row image for prover(): %s, its a parts expanded beside the SSP upload), digest %.2f, prover warm-up %.2f\n",
            t_drawn - t_in, SSP_SIZE / 1e9, t_ssp - t_drawn, t_crs - t_ssp, img ? "written" : "not kept", t_img - t_crs, tnow() - t_img);
}

/* ---- pieces shared with the multi-GPU entry points (host/mfuoco_dist.c, libmfuoco_gpu_dist): not part of the reference interface ---- */
mfh_ctx *mfuoco_gpu_ctx(void) { return gpu(); }

/* the CRS in keystream order s | as | t | v on the device (the order setup() encrypts in, src/snark.c:75-110) */
const uint8_t *mfuoco_gpu_stage_crs(crs_t crs)
{
  KEEP_ERRNO;
  gpu();
  use_seed(crs->seed);
  HK(hipMemcpy(G.d_crs, crs->s, CT_BYTES * GAMMA_D, hipMemcpyHostToDevice));
  HK(hipMemcpy(G.d_crs + CT_BYTES * GAMMA_D, crs->as, CT_BYTES * GAMMA_D, hipMemcpyHostToDevice));
  HK(hipMemcpy(G.d_crs + 2 * CT_BYTES * GAMMA_D, crs->t, CT_BYTES, hipMemcpyHostToDevice));
  HK(hipMemcpy(G.d_crs + (2 * GAMMA_D + 1) * CT_BYTES, crs->v, CT_BYTES * (GAMMA_M - 1), hipMemcpyHostToDevice));
  /* an image kept from an earlier call serves this CRS only if seed and bytes are the same: checked on EVERY staging, whatever the caller does next */
  G.staged_digest_valid = false;
  if (G.resident_crs != 0) {
    CK(mfh_digest128(G.ctx, G.d_crs, (2 * (size_t)GAMMA_D + GAMMA_M) * CT_BYTES, G.staged_digest));
    G.staged_digest_valid = true;
    if (G.img_valid && (memcmp(G.img_seed, G.seed, 40) || G.staged_digest[0] != G.img_digest[0] || G.staged_digest[1] != G.img_digest[1])) {
      if (G.img_registered) CK(mfh_crs_set_resident_mm(G.ctx, NULL));
      G.img_registered = G.img_valid = false;
    }
    if (G.rows_valid && (memcmp(G.rows_seed, G.seed, 40) || G.staged_digest[0] != G.rows_digest[0] || G.staged_digest[1] != G.rows_digest[1])) G.rows_valid = false;
  }
  return G.d_crs;
}

const uint32_t *mfuoco_gpu_stage_ssp(ssp_t ssp)
{
  KEEP_ERRNO;
  gpu();
  ssp_resident(ssp);
  return G.d_ssp;
}

size_t mfuoco_gpu_bits_stride(void) { return (GAMMA_M + 7) / 8 + 8; }

/* the witness as the little-endian bit string mfh_prove* take (bit i-1 selects v_i, src/snark.c:150) */
void mfuoco_gpu_witness_bits(uint8_t *bits, mpz_t witness)
{
  KEEP_ERRNO;
  if (mpz_sgn(witness) < 0 || mpz_sizeinbase(witness, 2) > GAMMA_M + 8) die("prover: witness negative or wider than M bits");
  /* (little-endian bytes = the limbs as they lie in memory: what mpz_export(bits, NULL, -1, 1, -1, 0, witness) writes byte by byte -- 2.5 us per statement, 2.5 ms per
   * 1020, before the GPU can start; at most ceil((M + 8) / 64) limbs = 2736 bytes of the stride's (M + 7) / 8 + 8) */
  _Static_assert(((GAMMA_M + 8 + 63) / 64) * 8 <= (GAMMA_M + 7) / 8 + 8, "a witness's limbs fit the bit-string stride");
  memcpy(bits, mpz_limbs_read(witness), mpz_size(witness) * sizeof(mp_limb_t));
}

/* entropy of one prover() call in the reference's order: delta (8 B), then 5 x [80 B magnitude, 1 B sign] (src/snark.c:140,185-189) */
void mfuoco_gpu_prover_entropy(uint32_t *delta, uint8_t *mag, uint8_t *sign)
{
  KEEP_ERRNO;
  const size_t maglen = GAMMA_LOG_SMUDGING / 8;
  *delta = (uint32_t)rand_modp_();
  for (int q = 0; q < 5; q++) {
    shim_random(mag + q * maglen, maglen);
    shim_random(sign + q, 1);
  }
}

/* the same for `count` calls, drawn from the OS in one piece and cut up in the reference's order (per call: 8 bytes of delta, then 5 x [80 + 1]): one system call
 * instead of eleven per proof (5 ms per 1020 proofs) */
void mfuoco_gpu_prover_entropy_batch(uint32_t *delta, uint8_t *mag, uint8_t *sign, size_t count)
{
  KEEP_ERRNO;
  const size_t maglen = GAMMA_LOG_SMUDGING / 8, per = 8 + 5 * (maglen + 1);
  uint8_t *tape = xmalloc(count * per);
  shim_random(tape, count * per);
  for (size_t k = 0; k < count; k++) {
    const uint8_t *t = tape + k * per;
    uint64_t r;
    memcpy(&r, t, 8);
    delta[k] = (uint32_t)(r % GAMMA_P);
    for (int q = 0; q < 5; q++) {
      memcpy(mag + (k * 5 + q) * maglen, t + 8 + q * (maglen + 1), maglen);
      sign[k * 5 + q] = t[8 + q * (maglen + 1) + maglen];
    }
  }
  explicit_bzero(tape, count * per);
  free(tape);
}

struct conv_arg { proof_t *pis; ct_t *cts; uint64_t *h; };
static mpz_t *proof_ct(struct proof *pk, size_t q) { return q == 0 ? pk->h : q == 1 ? pk->hat_h : q == 2 ? pk->hat_v : q == 3 ? pk->v_w : pk->b_w; }
static void proofs_from_limbs(size_t lo, size_t hi, void *arg)
{
  struct conv_arg *a = arg;
  const size_t per = 5 * (size_t)(GAMMA_N + 1);
  for (size_t i = lo; i < hi; i++) from_limbs(proof_ct(a->pis[i / per], (i % per) / (GAMMA_N + 1))[i % (GAMMA_N + 1)], a->h + i * L_LIMBS);
}
static void proofs_to_limbs(size_t lo, size_t hi, void *arg)
{
  struct conv_arg *a = arg;
  const size_t per = 5 * (size_t)(GAMMA_N + 1);
  for (size_t i = lo; i < hi; i++) to_limbs(a->h + i * L_LIMBS, proof_ct(a->pis[i / per], (i % per) / (GAMMA_N + 1))[i % (GAMMA_N + 1)]);
}
static void cts_to_limbs(size_t lo, size_t hi, void *arg)
{
  struct conv_arg *a = arg;
  for (size_t i = lo; i < hi; i++) to_limbs(a->h + i * L_LIMBS, a->cts[i / (GAMMA_N + 1)][i % (GAMMA_N + 1)]);
}

/* Device-to-host drain of proofs: slabs of DRAIN_SLAB proofs copied on a stream of the shim's own (non-blocking: independent of the kernels' stream) into two pinned
 * host buffers, slab j + 1 in flight while the host threads turn slab j into mpz_t's.  Nothing is allocated per call: the pinned pair (2 x 45 MB) and the stream are made once. */
#define DRAIN_SLAB ((size_t)64)
static struct {
  hipStream_t stream;
  hipEvent_t ev[2], ev_src;
  uint64_t *pin[2];
} DR;
static void drain_init(void)
{
  if (DR.stream) return;
  HK(hipStreamCreateWithFlags(&DR.stream, hipStreamNonBlocking));
  HK(hipEventCreateWithFlags(&DR.ev_src, hipEventDisableTiming));
  for (int i = 0; i < 2; i++) {
    HK(hipEventCreateWithFlags(&DR.ev[i], hipEventDisableTiming));
    HK(hipHostMalloc((void **)&DR.pin[i], DRAIN_SLAB * 5 * CTL * 8, hipHostMallocDefault));
  }
}
/* count proofs (5 ciphertexts each, struct proof order) from device limbs into initialised proof_t's.  batch != 0: d_proofs is being written by the mfh_prove_batch call
 * just queued, super-group by super-group; a slab's copy waits (on the device, mfh_prove_batch_stream_wait) for its super-group only, so super-group k crosses PCIe and
 * becomes mpz_t's under the kernels of k + 1.  batch == 0: d_proofs is final once the work queued on the shim's (default) stream so far has run. */
static void proofs_drain(proof_t *pis, const uint64_t *d_proofs, size_t count, int batch, hipEvent_t after)
{
  if (count * 5 * CTL * 8 <= PIN_BYTES) { /* up to five proofs (706 KB each): the small pinned scratch, no pipeline (and none of its 90 MB of page-locking on a first prover() call) */
    if (batch) CK(mfh_sync(G.ctx));
    if (after) { /* d_proofs is final once `after` has happened (the caller queued more work behind it: not waited for) */
      drain_init();
      HK(hipStreamWaitEvent(DR.stream, after, 0));
      HK(hipMemcpyAsync(G.pin, d_proofs, count * 5 * CTL * 8, hipMemcpyDeviceToHost, DR.stream));
      HK(hipStreamSynchronize(DR.stream));
    } else {
      HK(hipMemcpyAsync(G.pin, d_proofs, count * 5 * CTL * 8, hipMemcpyDeviceToHost, NULL));
      HK(hipStreamSynchronize(NULL));
    }
    struct conv_arg a = { pis, NULL, (uint64_t *)G.pin };
    proofs_from_limbs(0, count * 5 * (size_t)(GAMMA_N + 1), &a);
    return;
  }
  drain_init();
  const size_t sg = batch ? mfh_prove_batch_supergroup(G.ctx) : 0;
  if (!sg) batch = 0; /* (a call that hands out no per-super-group completion is drained like any finished buffer) */
  if (!batch && after) HK(hipStreamWaitEvent(DR.stream, after, 0)); /* order the copies after the caller's event ... */
  else if (!batch) {                                                /* ... or after what the default stream holds */
    HK(hipEventRecord(DR.ev_src, NULL));
    HK(hipStreamWaitEvent(DR.stream, DR.ev_src, 0));
  }
  /* slabs never straddle a super-group (a straddling copy would wait for the later one) */
  size_t lo[2] = { 0, 0 }, n[2] = { 0, 0 }, next = 0;
  int slot = 0;
#define DRAIN_ISSUE(sl)                                                                                                        \
  do {                                                                                                                         \
    size_t nk_ = count - next < DRAIN_SLAB ? count - next : DRAIN_SLAB;                                                        \
    if (sg && next / sg != (next + nk_ - 1) / sg) nk_ = (next / sg + 1) * sg - next;                                           \
    if (sg) CK(mfh_prove_batch_stream_wait(G.ctx, (uint32_t)(next + nk_), DR.stream));                                         \
    HK(hipMemcpyAsync(DR.pin[sl], d_proofs + next * 5 * CTL, nk_ * 5 * CTL * 8, hipMemcpyDeviceToHost, DR.stream));            \
    HK(hipEventRecord(DR.ev[sl], DR.stream));                                                                                  \
    lo[sl] = next;                                                                                                             \
    n[sl] = nk_;                                                                                                               \
    next += nk_;                                                                                                               \
  } while (0)
  const double t_dr = tnow();
  double t_wait = 0, t_conv = 0, t_last_arrival = 0;
  DRAIN_ISSUE(0);
  while (n[slot]) {
    if (next < count) DRAIN_ISSUE(slot ^ 1); else n[slot ^ 1] = 0;
    const double t0 = tnow();
    HK(hipEventSynchronize(DR.ev[slot]));
    const double t1 = tnow();
    struct conv_arg a = { pis + lo[slot], NULL, DR.pin[slot] };
    parallel_for(n[slot] * 5 * (size_t)(GAMMA_N + 1), proofs_from_limbs, &a);
    t_wait += t1 - t0;
    t_conv += tnow() - t1;
    if (!n[slot ^ 1]) t_last_arrival = t1;
    slot ^= 1;
  }
  if (tracing() && count > DRAIN_SLAB)
    fprintf(stderr, "  drain of %zu proofs: %.2f ms waiting for slabs, %.2f converting them (in between); the last slab arrived %.2f ms in, the call left %.2f ms later\n", count,
            t_wait, t_conv, t_last_arrival - t_dr, tnow() - t_last_arrival);
#undef DRAIN_ISSUE
}
void mfuoco_gpu_proofs_to_host(proof_t *pis, const uint64_t *d_proofs, size_t count)
{
  KEEP_ERRNO;
  proofs_drain(pis, d_proofs, count, 0, NULL);
}
/* the same once `hip_event` (a hipEvent_t recorded by the caller behind the last writer of d_proofs) has happened: later work of the caller's streams is not waited for */
void mfuoco_gpu_proofs_to_host_after(proof_t *pis, const uint64_t *d_proofs, size_t count, void *hip_event)
{
  KEEP_ERRNO;
  proofs_drain(pis, d_proofs, count, 0, (hipEvent_t)hip_event);
}

/* ---- the expanded CRS kept across prover calls (the materialised-CRS regime behind the reference's types) -------------------------------
 * The reference's prover() regenerates every a-vector from the seed on every call (ct_import, src/lwe.c:122-126); a caller that proves many statements
 * under one CRS (src/benchmark_snark.c:70-74 in a loop) would pay the AES expansion each time.  The shim therefore keeps what it expanded, keyed by the
 * 40-byte seed and a 128-bit digest of the staged compressed CRS taken ON the device (mfh_digest128: 8 MB read at HBM speed, so a caller that rewrites
 * crs->s/as/t/v in place is noticed without an extra API call; mfuoco_gpu_invalidate() drops everything explicitly).  Off: $MFUOCO_GPU_RESIDENT_CRS=0 or
 * mfuoco_gpu_set_resident_crs(0).  An image that does not fit the free device memory is simply not kept. */
static bool resident_on(void)
{
  if (G.resident_crs < 0) {
    const char *e = getenv("MFUOCO_GPU_RESIDENT_CRS");
    G.resident_crs = e && *e ? atoi(e) != 0 : 1;
  }
  return G.resident_crs == 1;
}
static bool fits_device(size_t need, size_t have)
{
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
  return need <= have || need - have + ((size_t)24 << 30) <= fr; /* leave room for the call's own scratch (w | h | v, partial products, the SSP's second image) */
}
/* matrix-core image (mfh_prove_batch / mfh_prove_batch_partial): rank's shares of `world`, the whole regions when world == 1 */
static void image_resident_mm(const uint8_t *d_crs, uint32_t rank, uint32_t world, bool expand)
{
  /* whatever this call does, an image registered for OTHER row shares must not meet it (mfh_prove_batch refuses a share image, mfh_prove_batch_partial one of
   * another rank or world) */
  if (G.img_registered && (G.img_rank != rank || G.img_world != world)) {
    CK(mfh_crs_set_resident_mm(G.ctx, NULL));
    G.img_registered = false;
  }
  if (!expand || !resident_on() || d_crs != G.d_crs) return;
  uint64_t dg[2];
  if (!G.staged_digest_valid) {
    CK(mfh_digest128(G.ctx, d_crs, (2 * (size_t)GAMMA_D + GAMMA_M) * CT_BYTES, G.staged_digest));
    G.staged_digest_valid = true;
  }
  dg[0] = G.staged_digest[0];
  dg[1] = G.staged_digest[1];
  if (G.img_valid && G.img_rank == rank && G.img_world == world && !memcmp(G.img_seed, G.seed, 40) && dg[0] == G.img_digest[0] && dg[1] == G.img_digest[1]) {
    if (!G.img_registered) {
      CK(world == 1 ? mfh_crs_set_resident_mm(G.ctx, G.d_img) : mfh_crs_set_resident_mm_share(G.ctx, G.d_img, rank, world));
      G.img_registered = true;
    }
    return;
  }
  if (G.img_registered) CK(mfh_crs_set_resident_mm(G.ctx, NULL));
  G.img_registered = G.img_valid = false;
  const size_t need = world == 1 ? mfh_crs_mm_image_bytes(G.ctx) : mfh_crs_mm_share_bytes(G.ctx, rank, world);
  if (!fits_device(need, G.img_bytes)) return;
  if (need > G.img_bytes) {
    if (G.d_img) HK(hipFree(G.d_img));
    G.d_img = NULL;
    G.img_bytes = 0;
    if (hipMalloc((void **)&G.d_img, need) != hipSuccess) { (void)hipGetLastError(); G.d_img = NULL; return; }
    G.img_bytes = need;
  }
  CK(world == 1 ? mfh_crs_expand_mm(G.ctx, d_crs, G.d_img) : mfh_crs_expand_mm_share(G.ctx, d_crs, rank, world, G.d_img));
  CK(world == 1 ? mfh_crs_set_resident_mm(G.ctx, G.d_img) : mfh_crs_set_resident_mm_share(G.ctx, G.d_img, rank, world));
  G.img_valid = G.img_registered = true;
  G.img_rank = rank;
  G.img_world = world;
  memcpy(G.img_seed, G.seed, 40);
  G.img_digest[0] = dg[0];
  G.img_digest[1] = dg[1];
}
/* returns 1 when an image of the rank's shares is registered with the context afterwards (the row work streams it), 0 when the row work will regenerate or expand per call */
int mfuoco_gpu_image_resident_share(const uint8_t *d_crs, uint32_t rank, uint32_t world, int expand)
{
  image_resident_mm(d_crs, rank, world, expand != 0);
  return G.img_registered ? 1 : 0;
}
/* single-proof image (mfh_prove): the rows of the CRS staged at d_crs (digest dg) in k_mac_resident's layout, registered under (seed, digest).  The expansion is
 * QUEUED on the shim's stream, not waited for: whatever is queued next finds the rows written.  NULL when the image is not kept (does not fit). */
/* room for the row image (kept across calls; NULL when it does not fit beside the calls' scratch); whatever image was there is no longer valid */
static void *rows_image_reserve(void)
{
  G.rows_valid = false;
  const size_t rows = 2 * (size_t)GAMMA_D + GAMMA_M, need = rows * mfh_resident_row_bytes(G.ctx);
  if (!fits_device(need, G.rows_bytes)) return NULL;
  if (need > G.rows_bytes) {
    if (G.d_rows) HK(hipFree(G.d_rows));
    G.d_rows = NULL;
    G.rows_bytes = 0;
    if (hipMalloc(&G.d_rows, need) != hipSuccess) { (void)hipGetLastError(); G.d_rows = NULL; return NULL; }
    G.rows_bytes = need;
  }
  return G.d_rows;
}
/* G.d_rows holds (or is queued to hold) the rows of the CRS with digest dg under the current seed */
static void rows_image_register(const uint64_t dg[2])
{
  G.rows_valid = true;
  memcpy(G.rows_seed, G.seed, 40);
  G.rows_digest[0] = dg[0];
  G.rows_digest[1] = dg[1];
}
static const void *rows_image_build(const uint8_t *d_crs, const uint64_t dg[2])
{
  void *img = rows_image_reserve();
  if (!img) return NULL;
  CK(mfh_crs_expand(G.ctx, 0, 2 * (size_t)GAMMA_D + GAMMA_M, d_crs, img));
  rows_image_register(dg);
  return img;
}
/* ... looked up by prover(): the image setup() / mfuoco_gpu_prefetch_crs() wrote for this (seed, CRS), else expanded when prover() meets the same (seed, CRS) a
 * second time; returns the image to register, or NULL */
static const void *image_resident_rows(const uint8_t *d_crs)
{
  if (!resident_on() || d_crs != G.d_crs) return NULL;
  uint64_t dg[2];
  if (!G.staged_digest_valid) {
    CK(mfh_digest128(G.ctx, d_crs, (2 * (size_t)GAMMA_D + GAMMA_M) * CT_BYTES, G.staged_digest));
    G.staged_digest_valid = true;
  }
  dg[0] = G.staged_digest[0];
  dg[1] = G.staged_digest[1];
  if (G.rows_valid && !memcmp(G.rows_seed, G.seed, 40) && dg[0] == G.rows_digest[0] && dg[1] == G.rows_digest[1]) return G.d_rows;
  G.rows_valid = false;
  const bool again = G.seen && !memcmp(G.seen_seed, G.seed, 40) && dg[0] == G.seen_digest[0] && dg[1] == G.seen_digest[1];
  G.seen = true;
  memcpy(G.seen_seed, G.seed, 40);
  G.seen_digest[0] = dg[0];
  G.seen_digest[1] = dg[1];
  if (!again) return NULL; /* first proof under a CRS the caller filled in by hand: regenerate, as the reference does */
  return rows_image_build(d_crs, dg);
}

/* A CRS that arrives from disk (mfuoco_crs_map calls this for a read-only mapping; a caller that fills a struct crs by other means may call it too): staged on the
 * device and its row image queued at once, so that the GPU expands it while the caller maps its SSP and builds its witness, and the first prover() under it streams the
 * rows.  Returns without waiting for the expansion.  Nothing happens with $MFUOCO_GPU_RESIDENT_CRS=0 / $MFUOCO_GPU_PREFETCH=0 or when the image does not fit. */
void mfuoco_gpu_prefetch_crs(crs_t crs)
{
  KEEP_ERRNO;
  const char *e = getenv("MFUOCO_GPU_PREFETCH");
  if ((e && *e && !atoi(e)) || !crs->s || !crs->as || !crs->t || !crs->v) return;
  if (!G.ctx) { /* a hint, not a compute call: on a host without a GPU (a tool that only inspects CRS files) mapping still works; prover() is what fails loudly there */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return; }
  }
  gpu();
  if (!resident_on()) return;
  const double t_in = tnow();
  const uint8_t *d_crs = mfuoco_gpu_stage_crs(crs); /* (takes the digest: resident_on() made G.resident_crs 1) */
  const bool had = G.rows_valid;
  if (!had) rows_image_build(d_crs, G.staged_digest);
  if (tracing()) fprintf(stderr, "mfuoco_gpu_prefetch_crs(): staged + %s in %.2f ms of host time\n", had ? "image already there" : G.rows_valid ? "row image queued" : "image not kept", tnow() - t_in);
}
int mfuoco_gpu_last_prover_path(void) { return G.last_prover_path; }

void prover(proof_t pi, crs_t crs, ssp_t ssp, mpz_t witness)
{
  KEEP_ERRNO;
  const double t_in = tnow();
  const uint8_t *d_crs = mfuoco_gpu_stage_crs(crs);
  const uint32_t *d_ssp = mfuoco_gpu_stage_ssp(ssp);
  const double t_staged = tnow();
  uint8_t bits[(GAMMA_M + 7) / 8 + 8] = { 0 };
  mfuoco_gpu_witness_bits(bits, witness);
  uint32_t delta;
  uint8_t mag[5 * (GAMMA_LOG_SMUDGING / 8)], sign[5];
  mfuoco_gpu_prover_entropy(&delta, mag, sign);
  const void *rows = image_resident_rows(d_crs);
  if (rows) CK(mfh_crs_set_resident(G.ctx, rows));
  int rc = mfh_prove(G.ctx, d_crs, d_ssp, bits, delta, mag, GAMMA_LOG_SMUDGING / 8, sign, G.d_proof);
  if (rows) CK(mfh_crs_set_resident(G.ctx, NULL)); /* (mfh_prove_batch must not find the single-proof image) */
  explicit_bzero(mag, sizeof mag);
  explicit_bzero(&delta, sizeof delta);
  if (rc != MFH_OK) die("mfh_prove");
  G.last_prover_path = rows ? 1 : 0;
  const double t_queued = tnow();
  if (tracing()) CK(mfh_sync(G.ctx));
  const double t_done = tnow();
  proof_t *one = (proof_t *)pi; /* proof_t is struct proof[1]: pi is the address of the one element */
  mfuoco_gpu_proofs_to_host(one, G.d_proof, 1);
  CK(mfh_scrub_staging(G.ctx)); /* the proof is out, so every copy of the call has run: witness bits, delta and the smudging terms leave the pinned staging too */
  if (tracing())
    fprintf(stderr, "prover(): stage CRS+SSP %.2f ms, image + queue %.2f, GPU %.2f, copy + mpz_t %.2f (%s)\n", t_staged - t_in, t_queued - t_staged, t_done - t_queued, tnow() - t_done,
            rows ? "resident rows" : "regenerated");
}

/* prover() for `count` statements under one CRS and SSP (not in the reference): the CRS rows are expanded once per group of proofs
 * and the multiply-accumulate runs on the matrix cores (mfh_prove_batch).  Entropy per proof as in prover(): delta, then 5 x
 * [80-byte magnitude, sign byte].  pis[k] must be initialised (proof_init). */
void mfuoco_prover_batch(proof_t *pis, crs_t crs, ssp_t ssp, mpz_t *witnesses, size_t count)
{
  KEEP_ERRNO;
  if (!count) return;
  const double t_in = tnow();
  const uint8_t *d_crs = mfuoco_gpu_stage_crs(crs);
  const uint32_t *d_ssp = mfuoco_gpu_stage_ssp(ssp);
  const double t_staged = tnow();
  const size_t stride = mfuoco_gpu_bits_stride(), maglen = GAMMA_LOG_SMUDGING / 8;
  uint8_t *bits = xcalloc(count, stride), *mag = xmalloc(count * 5 * maglen), *sign = xmalloc(count * 5);
  uint32_t *delta = xmalloc(count * 4);
  if (count > G.out_cap) { /* the call's device output: kept, grown when a call is larger than every one before */
    if (G.d_out) HK(hipFree(G.d_out));
    G.d_out = NULL;
    G.out_cap = 0;
    HK(hipMalloc((void **)&G.d_out, count * 5 * CTL * 8));
    G.out_cap = count;
  }
  uint64_t *d_out = G.d_out;
  const double t_alloc = tnow();
  for (size_t k = 0; k < count; k++) mfuoco_gpu_witness_bits(bits + k * stride, witnesses[k]);
  const double t_bits = tnow();
  mfuoco_gpu_prover_entropy_batch(delta, mag, sign, count);
  const double t_host = tnow();
  if (tracing()) fprintf(stderr, "mfuoco_prover_batch(%zu) host inputs: buffers %.2f ms, witness bits %.2f, entropy %.2f\n", count, t_alloc - t_staged, t_bits - t_alloc, t_host - t_bits);
  image_resident_mm(d_crs, 0, 1, count > 31); /* (smaller calls do not expand an image at all; one kept from an earlier call is used if it still serves this CRS) */
  const double t_image = tnow();
  int rc = mfh_prove_batch(G.ctx, d_crs, d_ssp, (uint32_t)count, bits, stride, delta, mag, maglen, sign, d_out);
  explicit_bzero(mag, count * 5 * maglen); /* the smudging terms and deltas are the proofs' zero-knowledge: not left on the heap */
  explicit_bzero(sign, count * 5);
  explicit_bzero(delta, count * 4);
  free(bits); free(mag); free(sign); free(delta);
  if (rc != MFH_OK) die("mfh_prove_batch");
  /* the call above only QUEUED the work: super-group k is copied and converted while the GPU runs k + 1 */
  const double t_queued = tnow();
  proofs_drain(pis, d_out, count, 1, NULL);
  CK(mfh_scrub_staging(G.ctx)); /* (as in prover(): nothing secret of the call stays in the context's pinned staging) */
  if (tracing())
    fprintf(stderr, "mfuoco_prover_batch(%zu): stage CRS+SSP %.2f ms, witness bits + entropy %.2f, image %.2f, queue %.2f, drain (copy + mpz_t under the GPU work) %.2f\n", count,
            t_staged - t_in, t_host - t_staged, t_image - t_host, t_queued - t_image, tnow() - t_queued);
}

static uint64_t horner_modp(const uint8_t *slot, uint64_t x)
{
  unsigned __int128 r = 0;
  for (size_t i = GAMMA_D; i-- > 0;) {
    uint64_t c;
    memcpy(&c, slot + 8 * i, 8);
    r = (r * x + c % GAMMA_P) % GAMMA_P;
  }
  return (uint64_t)r;
}

bool verifier(ssp_t ssp, vrs_t vrs, proof_t pi)
{
  KEEP_ERRNO;
  gpu();
  uint32_t dec[5];
  sk_resident(vrs->sk);
  { /* the five ciphertexts in struct order through the pinned scratch: one copy */
    _Static_assert(5 * CTL * 8 <= PIN_BYTES, "a proof fits the pinned scratch");
    uint64_t *h = (uint64_t *)G.pin;
    proof_t *one = (proof_t *)pi;
    struct conv_arg a = { one, NULL, h };
    proofs_to_limbs(0, 5 * (size_t)(GAMMA_N + 1), &a);
    HK(hipMemcpy(G.d_proof, h, 5 * CTL * 8, hipMemcpyHostToDevice));
  }
  CK(mfh_decrypt(G.ctx, G.d_sk, G.d_proof, 5, G.d_co)); /* the five regev_decrypt of src/snark.c:204-208 */
  HK(hipMemcpy(dec, G.d_co, sizeof dec, hipMemcpyDeviceToHost));
  const unsigned __int128 P = GAMMA_P;
  uint64_t h_s = dec[0], hath_s = dec[1], hatv_s = dec[2], w_s = dec[3], b_s = dec[4];
  uint64_t t_s = horner_modp(ssp, vrs->s);
  uint64_t v_s = (horner_modp(ssp + 8 * GAMMA_D, vrs->s) + w_s) % GAMMA_P;
  if ((uint64_t)((unsigned __int128)h_s * vrs->alpha % P) != hath_s) return false;               /* eq-pke */
  if ((uint64_t)((unsigned __int128)v_s * vrs->alpha % P) != hatv_s) return false;
  if ((uint64_t)(((unsigned __int128)v_s * v_s + P - 1) % P) != (uint64_t)((unsigned __int128)h_s * t_s % P)) return false; /* eq-div */
  if ((uint64_t)((unsigned __int128)w_s * vrs->beta % P) != b_s) return false;                    /* eq-lin */
  /* the reference's "test-error" bound (src/snark.c:238-241) can never reject: SIZ of a non-positive value is <= 0 */
  return true;
}

/* The way up: `nct` ciphertexts (proof structs are 5 each) converted slab by slab into the drain's two pinned buffers on the host threads and copied on the shim's own
 * stream, slab j + 1 being converted while slab j crosses PCIe; the kernels' (default) stream is made to wait for the last copy.  conv = proofs_to_limbs (src = proof_t's)
 * or cts_to_limbs (src = ct_t's). */
static void cts_upload(uint64_t *d_dst, proof_t *pis, ct_t *cts, size_t nct)
{
  drain_init();
  const size_t slab = DRAIN_SLAB * 5; /* ciphertexts per pinned buffer */
  int slot = 0;
  bool used[2] = { false, false };
  for (size_t c0 = 0; c0 < nct; c0 += slab, slot ^= 1) {
    const size_t nk = nct - c0 < slab ? nct - c0 : slab;
    if (used[slot]) HK(hipEventSynchronize(DR.ev[slot]));
    struct conv_arg a = { pis ? pis + c0 / 5 : NULL, cts ? cts + c0 : NULL, DR.pin[slot] };
    parallel_for(nk * (size_t)(GAMMA_N + 1), pis ? proofs_to_limbs : cts_to_limbs, &a);
    HK(hipMemcpyAsync(d_dst + c0 * CTL, DR.pin[slot], nk * CTL * 8, hipMemcpyHostToDevice, DR.stream));
    HK(hipEventRecord(DR.ev[slot], DR.stream));
    used[slot] = true;
  }
  HK(hipEventRecord(DR.ev_src, DR.stream));
  HK(hipStreamWaitEvent(NULL, DR.ev_src, 0));
}
/* device scratch of the batch verifier / decryption, kept and grown like the prover's output */
static uint64_t *up_reserve(size_t nct)
{
  if (nct > G.up_cap) {
    if (G.d_up) HK(hipFree(G.d_up));
    G.d_up = NULL;
    G.up_cap = 0;
    HK(hipMalloc((void **)&G.d_up, nct * CTL * 8 + nct * 4));
    G.up_cap = nct;
  }
  return G.d_up;
}

/* verifier() (src/snark.c:192-250) for `count` proofs under one SSP and verification key, entirely on the device (mfh_verify: t(s), v_0(s), the
 * 5 x count decryptions on the matrix cores from 820 proofs on, the four equations): ok[k] = 1 iff proof k is accepted.  Not in the reference. */
void mfuoco_verifier_batch(ssp_t ssp, vrs_t vrs, proof_t *pis, size_t count, uint8_t *ok)
{
  KEEP_ERRNO;
  if (!count) return;
  gpu();
  ssp_resident(ssp);
  sk_resident(vrs->sk);
  uint64_t *d_proofs = up_reserve(count * 5);
  uint8_t *d_ok = (uint8_t *)(d_proofs + G.up_cap * CTL);
  cts_upload(d_proofs, pis, NULL, count * 5);
  CK(mfh_verify(G.ctx, G.d_ssp, (uint32_t)vrs->alpha, (uint32_t)vrs->beta, (uint32_t)vrs->s, G.d_sk, d_proofs, count, d_ok));
  HK(hipMemcpy(ok, d_ok, count, hipMemcpyDeviceToHost));
}

/* regev_decrypt (src/lwe.c:105-111) for `count` ciphertexts under one key: ms[k] initialised by the caller.  From 4096 ciphertexts the dot products run
 * as a Toeplitz int8 GEMM on the matrix cores (mfh_decrypt).  Not in the reference (src/benchmark_lwe.c:35-38 decrypts one at a time). */
void mfuoco_decrypt_batch(mpz_t *ms, sk_t sk, ct_t *cts, size_t count)
{
  KEEP_ERRNO;
  if (!count) return;
  sk_resident(sk);
  uint64_t *d_cts = up_reserve(count);
  uint32_t *d_m = (uint32_t *)(d_cts + G.up_cap * CTL), *hm = xmalloc(count * 4);
  cts_upload(d_cts, NULL, cts, count);
  CK(mfh_decrypt(G.ctx, G.d_sk, d_cts, count, d_m));
  HK(hipMemcpy(hm, d_m, count * 4, hipMemcpyDeviceToHost));
  for (size_t k = 0; k < count; k++) mpz_set_ui(ms[k], hm[k]);
  free(hm);
}
