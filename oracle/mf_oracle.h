/*
 * mf_oracle.h -- CPU restatement ("oracle") of the mangiafuoco LWE/SSP-SNARK hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / CPU comparator.  The product path (c-lwe-snarks_amd/) never links it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - AES-256-CTR keystream, seek, byte-granular stateful reads and the byte->limb sampler
 *     (reference src/aes.c, src/entropy.c) are pinned against the REAL reference, compiled
 *     in place into oracle/_ref/ (oracle/build_ref.sh), and against committed goldens that
 *     were generated from it (tests/golden/, tests/golden/make_golden.py).
 *   - lwe.c / snark.c / ssp.c cannot be built here (they include <flint/nmod_poly.h>; FLINT
 *     is not in this image and no stand-in is written for it).  Their restatement below is
 *     pinned by (i) a second restatement that performs the same GMP calls the reference
 *     performs (oracle/gmp_check.c, real libgmp 6.2.1), and (ii) the reference's own test
 *     properties (src/test_lwe.c, src/test_snark.c, src/test_ssp.c) restated in tests/.
 *     No reference-generated KAT exists for those layers and the reference cannot be built for
 *     them here: in the task's vocabulary these layers are "PARITY UNPINNED" -- what stands in
 *     for the missing pin is (i) and (ii), nothing more is claimed.
 *
 * Conventions.  A "value" is L = ceil(logq/64) little-endian uint64 limbs (12 at logq=736,
 * 23 at logq=1472).  K = logq/64 limbs survive modq (11 -> the reference's effective
 * modulus 2^704, src/lwe.h:107-118; 23 -> 2^1472).  A ciphertext is (n+1) values, the
 * last one being b.  Everything is plain C11, no GMP, no OpenSSL.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFO_P 0xfffffffbULL /* GAMMA_P, src/lwe.h:25 */

typedef struct {
  uint32_t n;    /* GAMMA_N    (1470)             src/lwe.h:23 */
  uint32_t logq; /* GAMMA_LOGQ (736 | 1472)       src/lwe.h:24 */
  uint32_t d;    /* GAMMA_D    (256 | 2^15 | ...) src/lwe.h:15,19 */
  uint32_t m;    /* GAMMA_M    (64 | 21845 | ...) src/lwe.h:17,20 */
} mfo_params;

static inline uint32_t mfo_L(const mfo_params *P) { return (P->logq + 63) / 64; }
static inline uint32_t mfo_K(const mfo_params *P) { return P->logq / 64; }
static inline uint32_t mfo_ctb(const mfo_params *P) { return P->logq / 8; } /* CT_BYTES */
/* stream offsets, src/snark.h:8-12 */
static inline uint64_t mfo_ctr_ct(const mfo_params *P) { return (uint64_t)mfo_ctb(P) * P->n; }
static inline uint64_t mfo_ctr_s(const mfo_params *P) { (void)P; return 0; }
static inline uint64_t mfo_ctr_as(const mfo_params *P) { return mfo_ctr_ct(P) * P->d; }
static inline uint64_t mfo_ctr_bt(const mfo_params *P) { return 2 * mfo_ctr_ct(P) * P->d; }
static inline uint64_t mfo_ctr_bv(const mfo_params *P) { return 2 * mfo_ctr_ct(P) * P->d + mfo_ctr_ct(P); }

/* ---- L0/L1: AES-256-CTR stream (src/aes.c:49-144, src/entropy.c:46-61) ---- */
typedef struct {
  uint32_t rk[60]; /* expanded key, big-endian words as in FIPS-197 */
  uint64_t nonce;  /* seed[0..8) loaded natively (little-endian host) */
  uint64_t ctr;
  uint8_t remb[16];
  size_t rem;
} mfo_rng;

void mfo_aes256_expand_key(uint32_t rk[60], const uint8_t key[32]);
void mfo_aes256_encrypt_block(const uint32_t rk[60], const uint8_t in[16], uint8_t out[16]);

void mfo_rng_init(mfo_rng *r, const uint8_t seed[40]);          /* rng_init   src/entropy.c:58-61 */
void mfo_rng_seek(mfo_rng *r, uint64_t off);                    /* rng_seek   src/entropy.c:46-56 */
void mfo_rng_gen(mfo_rng *r, void *out, size_t n);              /* aesctr_prg src/aes.c:104-144   */
/* stateless view of the same stream: bytes [off, off+n) */
void mfo_keystream(const uint8_t seed[40], uint64_t off, void *out, size_t n);

/* ---- L1: sampler (mpz2_urandomb, src/entropy.c:11-26) ---- */
/* reads nbits/8 stream bytes into limbs (little-endian), masks to nbits; out has ceil(nbits/64) limbs.
 * Bits above 8*floor(nbits/8) are stale heap in the reference; here they are zero. */
void mfo_urandomb(uint64_t *out, mfo_rng *r, size_t nbits);

/* ---- L2: LWE (src/lwe.c, src/lwe.h) ---- */
void mfo_modq(const mfo_params *P, uint64_t *v);                                        /* src/lwe.h:107-118 */
/* rop(L limbs, treated mod 2^(64K)) += sum a[j]*b[j], then modq         src/lwe.c:20-28 */
void mfo_add_dotp(const mfo_params *P, uint64_t *rop, const uint64_t *a, const uint64_t *b, size_t len);
void mfo_sample_a(const mfo_params *P, uint64_t *ct, mfo_rng *r);                       /* mpz2_urandommv, src/lwe.c:90,101,124 */
/* regev_encrypt2 with an explicit error value e (L limbs): src/lwe.c:78-97 */
void mfo_encrypt(const mfo_params *P, uint64_t *ct, mfo_rng *r, const uint64_t *sk, uint64_t m, const uint64_t *e);
uint64_t mfo_decrypt(const mfo_params *P, const uint64_t *sk, const uint64_t *ct);      /* src/lwe.c:105-111 */
void mfo_ct_export(const mfo_params *P, uint8_t *buf, const uint64_t *ct);              /* src/lwe.c:115-119 */
void mfo_ct_import(const mfo_params *P, uint64_t *ct, mfo_rng *r, const uint8_t *buf);  /* src/lwe.c:122-126 */
void mfo_ct_mul_ui(const mfo_params *P, uint64_t *rop, const uint64_t *a, uint64_t b);  /* src/lwe.c:131-139 */
void mfo_ct_addmul_ui(const mfo_params *P, uint64_t *rop, const uint64_t *a, uint64_t b); /* src/lwe.c:141-149 */
void mfo_ct_add(const mfo_params *P, uint64_t *rop, const uint64_t *a, const uint64_t *b); /* src/lwe.c:151-157 */
/* ct_smudge with explicit entropy: mag = logsmudge/8 bytes, sign byte. Returns 1 if the reference
 * would have left a negative (unreduced) value (prob ~2^-32), in which case ours is reduced mod 2^(64K). */
int mfo_ct_smudge(const mfo_params *P, uint64_t *ct, const uint8_t *mag, size_t maglen, uint8_t sign); /* src/lwe.c:65-76 */
/* eval_poly: rop += sum_{i<d} coeff[i]*Import(c8[i]); coeff already reduced (<p)   src/lwe.c:176-186 */
void mfo_eval_poly(const mfo_params *P, uint64_t *rop, mfo_rng *r, const uint8_t *c8, const uint64_t *coeff, size_t d);

/* ---- L3: polynomials mod p and SSP layout (src/ssp.c, src/ssp.h) ---- */
void mfo_poly_import(uint64_t *poly, const void *buf, size_t d);      /* nmod_poly_import: reduces mod p, src/ssp.c:28-34 */
uint64_t mfo_poly_eval(const uint64_t *poly, size_t d, uint64_t x);   /* nmod_poly_evaluate_nmod (Horner) */
/* q = floor((v^2 - 1) / t) over F_p[x]; v,t have d coefficients; q gets d coefficients (zero padded). O(d^2). */
void mfo_poly_h(uint64_t *q, const uint64_t *v, const uint64_t *t, size_t d);
/* remainder of (v^2-1) mod t == 0 ? */
int mfo_poly_divides(const uint64_t *v, const uint64_t *t, size_t d);
/* random_ssp with an explicit byte tape (8*d bytes per v_i, i = 0..m-1) and explicit witness bits: src/ssp.c:37-77 */
void mfo_ssp_from_tape(const mfo_params *P, uint8_t *ssp, const uint8_t *tape, const uint8_t *witness_bits);

/* ---- L4: SNARK (src/snark.c) ---- */
typedef struct {
  const uint8_t *seed;  /* 40 B */
  const uint8_t *s;     /* d * ctb */
  const uint8_t *as;    /* d * ctb */
  const uint8_t *v;     /* m * ctb (m-1 used) */
  const uint8_t *t;     /* ctb */
} mfo_crs;

/* setup with explicit entropy: alpha,beta,s already reduced mod p; sk = n values; etape = (2d+m) error values of L limbs
 * in encryption order.  Writes crs->s/as/v/t (caller-allocated, non-const view).      src/snark.c:57-115 */
void mfo_setup(const mfo_params *P, uint8_t *s, uint8_t *as, uint8_t *v, uint8_t *t, const uint8_t seed[40],
               const uint8_t *ssp, uint64_t alpha, uint64_t beta, uint64_t spt, const uint64_t *sk, const uint64_t *etape);

/* prover with explicit entropy. delta already reduced. smudge tape: 5 x (maglen bytes magnitude, 1 byte sign), in the
 * reference's call order (h, hat_h, hat_v, v_w, v_w).  proof = 5 cts in struct order (h, hat_h, hat_v, v_w, b_w).
 * If pre is non-NULL the un-smudged proof is stored there too.  w_out/h_out (d coeffs) optional.  src/snark.c:117-190 */
void mfo_prover(const mfo_params *P, uint64_t *proof, uint64_t *pre, const mfo_crs *crs, const uint8_t *ssp,
                const uint8_t *witness_bits, uint64_t delta, const uint8_t *smudge_tape, size_t maglen,
                uint64_t *w_out, uint64_t *h_out);

/* verifier: src/snark.c:192-250. (The "test-error" bound is vacuous in the reference; restated faithfully.) */
int mfo_verifier(const mfo_params *P, const uint8_t *ssp, uint64_t alpha, uint64_t beta, uint64_t spt,
                 const uint64_t *sk, const uint64_t *proof);

/* ---- cpu_baseline helpers (bench.py): reference-faithful single-thread costs ---- */
/* one prover row-touch: ct_import + ct_addmul_ui, repeated `rows` times; returns a checksum */
uint64_t mfo_bench_eval_rows(const mfo_params *P, const uint8_t seed[40], size_t rows);
uint64_t mfo_bench_encrypt(const mfo_params *P, const uint8_t seed[40], size_t count);
uint64_t mfo_bench_decrypt(const mfo_params *P, const uint8_t seed[40], size_t count);

#ifdef __cplusplus
}
#endif
