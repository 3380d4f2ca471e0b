/*
 * mangiafuoco_api.h -- the C interface libmfuoco_gpu.so exports: the same function names, argument meaning and
 * ownership rules as the reference library's five headers (src/aes.h, src/entropy.h, src/lwe.h, src/ssp.h,
 * src/snark.h), so that a program written against those headers links against libmfuoco_gpu.so instead of the
 * reference objects.  One combined header, our own text; programs that already include the reference's headers do
 * not need it (the layouts below are the ones those headers define).
 *
 * Everything that the reference computes with AES/GMP on the CPU on this path is computed on the GPU through
 * libmfhip.so (include/mfhip.h); host code only marshals mpz_t <-> dense limbs and draws OS entropy.
 * Parameters follow the reference's compile-time switch: -DNDEBUG => D = 2^15, M = 21845, else D = 256, M = 64.
 */
#ifndef MANGIAFUOCO_API_H
#define MANGIAFUOCO_API_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <gmp.h>
#include <flint/nmod_poly.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- parameters (src/lwe.h:14-31) ---- */
#ifdef NDEBUG
#define GAMMA_D (1UL << 15)
#define GAMMA_M (21845)
#else
#define GAMMA_D (1UL << 8)
#define GAMMA_M (1UL << 6)
#endif
#define GAMMA_N 1470
#define GAMMA_LOGQ 736
#define GAMMA_P 0xfffffffbUL
#define GAMMA_LOG_SMUDGING 640
#define GAMMA_LOG_SIGMA 556
#define LOGQ_BYTES 92UL
#define CT_BYTES (LOGQ_BYTES)

/* ---- stream (src/aes.h:21-40, src/entropy.h:35-60) ---- */
struct aesctr {
  uint64_t nonce;
  void *key; /* opaque to callers (the reference stores an AES_KEY* here) */
  uint64_t ctr;
  uint8_t remb[16];
  size_t rem;
};
typedef struct aesctr *aesctr_ptr;
typedef struct aesctr aesctr_t[1];
typedef uint8_t rseed_t[32 + 8];
typedef aesctr_t rng_t[1];

void aesctr_init(aesctr_ptr stream, const uint8_t *key, const uint64_t nonce);
void aesctr_prg(aesctr_ptr stream, void *out, size_t count);
void aesctr_clear(aesctr_ptr stream);
void rng_init(rng_t rs, uint8_t *rseed);
void rng_clear(rng_t rs);
void rng_seek(rng_t rs, size_t count);
void mpz2_urandomb(mpz_ptr rop, rng_t rs, size_t nbits);
void mpz2_urandomb2(mpz_ptr rop, size_t nbits);

/* ---- LWE (src/lwe.h:33-74, src/lwe.c:141,160) ---- */
typedef mpz_t sk_t[GAMMA_N];
typedef mpz_t ct_t[GAMMA_N + 1];
void key_gen(sk_t sk);
void key_clear(sk_t sk);
void errdist_uniform(mpz_t e);
void ct_init(ct_t ct);
void ct_clear(ct_t ct);
void ct_export(uint8_t *buf, ct_t ct);
void ct_import(ct_t ct, rng_t rng, uint8_t *buf);
void decompress_encryption(ct_t c, rng_t rs, mpz_t b);
void regev_encrypt2(ct_t c, rng_t rs, sk_t sk, mpz_t m, void (*chi)(mpz_t));
void regev_decrypt(mpz_t m, sk_t sk, ct_t ct);
void mpz_add_dotp(mpz_t rop, mpz_t a[], mpz_t b[], size_t len);
void ct_smudge(ct_t ct);
void ct_add(ct_t rop, ct_t a, ct_t b);
void ct_mul_ui(ct_t rop, ct_t a, uint64_t b);
void ct_addmul_ui(ct_t rop, ct_t a, uint64_t b);
void ct_zero(ct_t rop);
void eval_poly(ct_t rop, rng_t rng, uint8_t (*c8)[CT_BYTES], nmod_poly_t coeffs, size_t d);

/* ---- SSP (src/ssp.h:6-14) ---- */
#define SSP_SIZE (GAMMA_D * 8 * (GAMMA_M + 3))
void nmod_poly_import(nmod_poly_t *pp, void *buf, size_t degree);
void nmod_poly_export(void *buf, nmod_poly_t *pp, size_t degree);
void random_ssp(mpz_t input, uint8_t *circuit);

/* ---- SNARK (src/snark.h:6-51) ---- */
struct proof { ct_t h, hat_h, hat_v, v_w, b_w; };
struct vrs { uint64_t alpha, beta, s; sk_t sk; };
struct crs { rseed_t seed; uint8_t (*s)[CT_BYTES]; uint8_t (*as)[CT_BYTES]; uint8_t (*v)[CT_BYTES]; uint8_t *t; };
typedef uint8_t *ssp_t;
typedef struct crs crs_t[1];
typedef struct proof proof_t[1];
typedef struct vrs vrs_t[1];
void crs_init(crs_t crs);
void crs_clear(crs_t crs);
void proof_init(proof_t pi);
void proof_clear(proof_t pi);
void setup(crs_t crs, vrs_t vrs, ssp_t ssp);
void prover(proof_t pi, crs_t crs, ssp_t ssp, mpz_t witness);
bool verifier(ssp_t ssp, vrs_t vrs, proof_t pi);

/* ---- the reference headers' inline helpers and macros: same names and meaning (src/lwe.h:57-67,76-104, src/entropy.h:40-72, src/snark.h:8-12,
 * src/ssp.h:8-9, src/aes.h:33-34).  modq (src/lwe.h:107-118) works on GMP internals and is applied by the device kernels to every value they produce:
 * a caller never needs it; CTR(x) / REM(x) read the stream state this header's struct aesctr keeps like the reference's. ---- */
#include <strings.h>
#include <sys/random.h>
#define CTR(x) ((*(x))->ctr)
#define REM(x) ((*(x))->rem)
#define CTR_CT (CT_BYTES * GAMMA_N)
#define CTR_S 0
#define CTR_AS (CTR_CT * GAMMA_D)
#define CTR_BT (2 * CTR_CT * GAMMA_D)
#define CTR_BV (2 * CTR_CT * GAMMA_D + CTR_CT)
#define ssp_t_offset 0
#define ssp_v_offset(i) (GAMMA_D * 8 * ((i) + 1))
#define RNG_INIT(rs) do { rseed_t rseed_; getrandom(&rseed_, sizeof(rseed_t), GRND_NONBLOCK); rng_init(rs, rseed_); bzero(&rseed_, sizeof(rseed_t)); } while (0)
static inline void rng_gen(rng_t prg, void *out, size_t count) { aesctr_prg((aesctr_ptr)prg, out, count); }
#define mpz2_urandommv(vs, rng, bits, len) do { for (size_t i_ = 0; i_ < (len); i_++) mpz2_urandomb((vs)[i_], rng, bits); } while (0)
#define mpz2_urandombv2(vs, bits, len) do { for (size_t i_ = 0; i_ < (len); i_++) mpz2_urandomb2((vs)[i_], bits); } while (0)
#define mpz_initv(vs, len) do { for (size_t i_ = 0; i_ < (len); i_++) mpz_init((vs)[i_]); } while (0)
#define mpz_clearv(vs, len) do { for (size_t i_ = 0; i_ < (len); i_++) mpz_clear((vs)[i_]); } while (0)
#define ct_clearv(vs, len) do { for (size_t i_ = 0; i_ < (len); i_++) ct_clear((vs)[i_]); } while (0)
static inline void mpz_dotp(mpz_t rop, mpz_t a[], mpz_t b[], size_t len) { mpz_set_ui(rop, 0); mpz_add_dotp(rop, a, b, len); }
static inline void regev_encrypt(ct_t c, rng_t rs, sk_t sk, mpz_t m) { regev_encrypt2(c, rs, sk, m, errdist_uniform); }
static inline uint64_t rand_modp(void) { uint64_t rop_ = 0; while (getrandom(&rop_, sizeof rop_, 0) != (ssize_t)sizeof rop_) { } return rop_ % GAMMA_P; }

/* ---- additions (not in the reference) ---- */
/* The shim keeps the last SSP it uploaded (keyed by host pointer) resident in HBM; call this after changing the
 * bytes of an SSP buffer in place.  (It also drops the expanded CRS images -- a CRS changed in place is noticed without it --, forgets the cached secret key and
 * frees the device scratch the batch calls keep between calls: 706 KB per proof of the largest batch so far.) */
void mfuoco_gpu_invalidate(void);
/* prover() for `count` statements under one CRS and SSP: rows expanded once per group of proofs, multiply-accumulate on the matrix
 * cores; every proof is what prover() would produce with the same randomness.  pis[k] initialised by proof_init. */
void mfuoco_prover_batch(proof_t *pis, crs_t crs, ssp_t ssp, mpz_t *witnesses, size_t count);
/* verifier() for `count` proofs under one SSP and key, on the device: ok[k] = 1 iff pis[k] is accepted (src/snark.c:192-250 per proof) */
void mfuoco_verifier_batch(ssp_t ssp, vrs_t vrs, proof_t *pis, size_t count, uint8_t *ok);
/* regev_decrypt for `count` ciphertexts under one key (src/lwe.c:105-111 per ciphertext); ms[k] initialised by the caller */
void mfuoco_decrypt_batch(mpz_t *ms, sk_t sk, ct_t *cts, size_t count);
/* regev_encrypt + ct_export (src/lwe.c:78-97,115-119; the loops of src/benchmark_lwe.c:28-33 and src/snark.c:75-110) for `count` messages under one key: row k is
 * encrypted with the stream at rs + k * CTR_CT and its own error draw, c8[k] receives the exported b (ct_import at that stream position restores the ciphertext),
 * rs ends count rows further.  mfuoco_encrypt_batch2 takes the error distribution like regev_encrypt2 (NULL = errdist_uniform, drawn in bulk). */
void mfuoco_encrypt_batch(uint8_t (*c8)[CT_BYTES], rng_t rs, sk_t sk, mpz_t *ms, size_t count);
void mfuoco_encrypt_batch2(uint8_t (*c8)[CT_BYTES], rng_t rs, sk_t sk, mpz_t *ms, size_t count, void (*chi)(mpz_t));
/* regev_decrypt of `count` seed-compressed ciphertexts (c8[k] = exported b of the row at rs + k * CTR_CT; the a part is regenerated on the device as ct_import
 * does, src/lwe.c:122-126): ms[k] initialised by the caller, rs ends count rows further */
void mfuoco_decrypt_rows_batch(mpz_t *ms, rng_t rs, sk_t sk, uint8_t (*c8)[CT_BYTES], size_t count);
/* select the GPU (default 0, or $MFUOCO_GPU); must precede the first call: returns 0, or -1 (with a message) once the shim runs on another GPU.
 * mfuoco_gpu_device: the GPU the shim runs on, -1 before its first call. */
int mfuoco_gpu_set_device(int device);
int mfuoco_gpu_device(void);
/* The shim keeps the CRS it expanded across prover calls, keyed by the seed and a device-side digest of the compressed CRS (SURVEY 8(d)'s materialised-CRS
 * regime behind the reference's types): mfuoco_prover_batch / _sharded stream the matrix-core image from the second call on (no AES in the call).  prover() streams the
 * single-proof row image from its FIRST call on when the CRS came out of setup() (which writes the rows as a by-product of its encryptions) or out of
 * mfuoco_crs_map() (which queues the expansion, mfuoco_gpu_prefetch_crs); under a CRS the caller filled in by hand, from the second call on (that call expands it).
 * Default on; $MFUOCO_GPU_RESIDENT_CRS=0 or mfuoco_gpu_set_resident_crs(0) turn it off and free the images; mfuoco_gpu_invalidate() drops them (and the resident SSP). */
void mfuoco_gpu_set_resident_crs(int on);
/* stage `crs` on the device and queue the expansion of its row image (returns after a few ms of host time, the GPU expands in the background): the first prover() under it
 * then streams the rows instead of regenerating them.  mfuoco_crs_map() calls it for read-only mappings; $MFUOCO_GPU_PREFETCH=0 turns that off. */
void mfuoco_gpu_prefetch_crs(crs_t crs);
/* what the last prover() ran on: 0 = keystream regenerated (src/lwe.c:122-126 as the reference does), 1 = the resident row image */
int mfuoco_gpu_last_prover_path(void);


/* ---- flat on-disk images (host only; host/mfuoco_files.c).  Sizes are the reference's: CRS_SIZE (src/snark.h:6),
 * SSP_SIZE (src/ssp.h:6), d * CT_BYTES for ciphertext-row files (src/benchmark_eval.c:44-66).  Return 0 / a valid
 * pointer on success, -1 / NULL with errno set otherwise (EINVAL: the file does not have the expected size). ---- */
#define CRS_SIZE (CT_BYTES * (2 * GAMMA_D + GAMMA_M + 1 + 2))
/* rows in keystream order: s[0..D) | as[0..D) | t | v[0..M) | 2 trailer rows = 40-byte seed + zeros */
int mfuoco_crs_save(const char *path, const struct crs *crs);
/* point crs->s/as/t/v into a mapping of the image and copy the seed out; release with mfuoco_crs_unmap, NOT crs_clear */
int mfuoco_crs_map(struct crs *crs, const char *path, int writable);
/* create the image (seed taken from crs->seed) and map it writable, so that setup() fills the file directly;
 * call mfuoco_crs_sync_seed before unmapping if crs->seed changed afterwards */
int mfuoco_crs_create(struct crs *crs, const char *path);
void mfuoco_crs_sync_seed(struct crs *crs);
void mfuoco_crs_unmap(struct crs *crs);
int mfuoco_ssp_save(const char *path, const uint8_t *ssp);
uint8_t *mfuoco_ssp_map(const char *path, int writable);
void mfuoco_ssp_unmap(uint8_t *ssp);
int mfuoco_rows_save(const char *path, uint8_t (*c8)[CT_BYTES], size_t rows);
uint8_t (*mfuoco_rows_map(const char *path, size_t *rows))[CT_BYTES];
void mfuoco_rows_unmap(uint8_t (*c8)[CT_BYTES], size_t rows);
/* 5 ciphertexts in struct order, each GAMMA_N + 1 values of CT_BYTES little-endian bytes (the reference has no proof format) */
int mfuoco_proof_save(const char *path, proof_t pi);
int mfuoco_proof_load(proof_t pi, const char *path);

#ifdef __cplusplus
}
#endif
#endif
