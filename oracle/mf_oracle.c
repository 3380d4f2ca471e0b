/*
 * mf_oracle.c -- CPU restatement of the mangiafuoco hot path.  TEST INFRASTRUCTURE ONLY
 * (see mf_oracle.h for the contract and the pinning status).
 *
 * Written from the behaviour of the reference (file:line cited per function), not from its
 * text: no GMP, no OpenSSL, no FLINT.  Big integers are little-endian uint64 limb arrays.
 */
#include "mf_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * AES-256 (FIPS-197).  The reference calls OpenSSL's AES_set_encrypt_key/AES_encrypt
 * (src/aes.c:92,101); any correct AES-256 gives the same bytes.  Tables are generated, not typed.
 * ------------------------------------------------------------------------------------------ */
static uint8_t SBOX[256];
static uint32_t TE0[256], TE1[256], TE2[256], TE3[256]; /* big-endian column convention */
static int tables_ready;

static uint8_t gf_mul(uint8_t a, uint8_t b)
{
  uint8_t r = 0;
  while (b) {
    if (b & 1) r ^= a;
    a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
    b >>= 1;
  }
  return r;
}

static void make_tables(void)
{
  if (tables_ready) return;
  /* multiplicative inverse by brute force, then the affine map */
  for (int x = 0; x < 256; x++) {
    uint8_t inv = 0;
    if (x)
      for (int y = 1; y < 256; y++)
        if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
    uint8_t s = inv, r = inv;
    for (int k = 0; k < 4; k++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
    SBOX[x] = s ^ 0x63;
  }
  for (int x = 0; x < 256; x++) {
    uint8_t s = SBOX[x], s2 = gf_mul(s, 2), s3 = (uint8_t)(s2 ^ s);
    uint32_t w = ((uint32_t)s2 << 24) | ((uint32_t)s << 16) | ((uint32_t)s << 8) | s3;
    TE0[x] = w;
    TE1[x] = (w >> 8) | (w << 24);
    TE2[x] = (w >> 16) | (w << 16);
    TE3[x] = (w >> 24) | (w << 8);
  }
  tables_ready = 1;
}

static inline uint32_t ld_be32(const uint8_t *p)
{
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static inline void st_be32(uint8_t *p, uint32_t v)
{
  p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

void mfo_aes256_expand_key(uint32_t rk[60], const uint8_t key[32])
{
  make_tables();
  for (int i = 0; i < 8; i++) rk[i] = ld_be32(key + 4 * i);
  uint8_t rcon = 1;
  for (int i = 8; i < 60; i++) {
    uint32_t t = rk[i - 1];
    if (i % 8 == 0) {
      t = (t << 8) | (t >> 24);
      t = ((uint32_t)SBOX[t >> 24] << 24) | ((uint32_t)SBOX[(t >> 16) & 255] << 16) |
          ((uint32_t)SBOX[(t >> 8) & 255] << 8) | SBOX[t & 255];
      t ^= (uint32_t)rcon << 24;
      rcon = gf_mul(rcon, 2);
    } else if (i % 8 == 4) {
      t = ((uint32_t)SBOX[t >> 24] << 24) | ((uint32_t)SBOX[(t >> 16) & 255] << 16) |
          ((uint32_t)SBOX[(t >> 8) & 255] << 8) | SBOX[t & 255];
    }
    rk[i] = rk[i - 8] ^ t;
  }
}

void mfo_aes256_encrypt_block(const uint32_t rk[60], const uint8_t in[16], uint8_t out[16])
{
  uint32_t s0 = ld_be32(in) ^ rk[0], s1 = ld_be32(in + 4) ^ rk[1];
  uint32_t s2 = ld_be32(in + 8) ^ rk[2], s3 = ld_be32(in + 12) ^ rk[3];
  uint32_t t0, t1, t2, t3;
  for (int r = 1; r < 14; r++) {
    const uint32_t *k = rk + 4 * r;
    t0 = TE0[s0 >> 24] ^ TE1[(s1 >> 16) & 255] ^ TE2[(s2 >> 8) & 255] ^ TE3[s3 & 255] ^ k[0];
    t1 = TE0[s1 >> 24] ^ TE1[(s2 >> 16) & 255] ^ TE2[(s3 >> 8) & 255] ^ TE3[s0 & 255] ^ k[1];
    t2 = TE0[s2 >> 24] ^ TE1[(s3 >> 16) & 255] ^ TE2[(s0 >> 8) & 255] ^ TE3[s1 & 255] ^ k[2];
    t3 = TE0[s3 >> 24] ^ TE1[(s0 >> 16) & 255] ^ TE2[(s1 >> 8) & 255] ^ TE3[s2 & 255] ^ k[3];
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  const uint32_t *k = rk + 56;
#define SB4(a, b, c, d) (((uint32_t)SBOX[(a) >> 24] << 24) | ((uint32_t)SBOX[((b) >> 16) & 255] << 16) | \
                         ((uint32_t)SBOX[((c) >> 8) & 255] << 8) | SBOX[(d) & 255])
  t0 = SB4(s0, s1, s2, s3) ^ k[0];
  t1 = SB4(s1, s2, s3, s0) ^ k[1];
  t2 = SB4(s2, s3, s0, s1) ^ k[2];
  t3 = SB4(s3, s0, s1, s2) ^ k[3];
#undef SB4
  st_be32(out, t0); st_be32(out + 4, t1); st_be32(out + 8, t2); st_be32(out + 12, t3);
}

/* ------------------------------------------------------------------------------------------
 * CTR stream.  Block c = AES_K(nonce_le64 || c_le64), K = seed[8..40), nonce = seed[0..8)
 * (src/entropy.c:58-61, src/aes.c:108,127-128).  The stateful reader reproduces ctr/rem/remb.
 * ------------------------------------------------------------------------------------------ */
void mfo_rng_init(mfo_rng *r, const uint8_t seed[40])
{
  memset(r, 0, sizeof *r);
  memcpy(&r->nonce, seed, 8);
  mfo_aes256_expand_key(r->rk, seed + 8);
}

static inline void stream_block(const mfo_rng *r, uint64_t ctr, uint8_t out[16])
{
  uint8_t in[16];
  memcpy(in, &r->nonce, 8);
  memcpy(in + 8, &ctr, 8);
  mfo_aes256_encrypt_block(r->rk, in, out);
}

void mfo_rng_gen(mfo_rng *r, void *out_, size_t bytes)
{
  uint8_t *out = out_;
  if (r->rem >= bytes) { /* src/aes.c:110-114 */
    memmove(out, r->remb, bytes);
    r->rem -= bytes;
    memmove(r->remb, r->remb + bytes, r->rem);
    return;
  } else if (r->rem > 0) { /* src/aes.c:115-120 */
    memcpy(out, r->remb, r->rem);
    bytes -= r->rem;
    out += r->rem;
    r->rem = 0;
  }
  for (size_t blocks = bytes / 16; blocks; blocks--) { /* src/aes.c:122-133, one block per call */
    stream_block(r, r->ctr++, out);
    out += 16;
  }
  bytes %= 16;
  if (bytes) { /* src/aes.c:135-142 */
    stream_block(r, r->ctr++, r->remb);
    memcpy(out, r->remb, bytes);
    r->rem = 16 - bytes;
    memmove(r->remb, r->remb + bytes, r->rem);
  }
}

void mfo_rng_seek(mfo_rng *r, uint64_t off)
{
  r->rem = 0;
  r->ctr = off / 16;
  off -= r->ctr * 16;
  if (off) {
    uint8_t sink[16];
    mfo_rng_gen(r, sink, (size_t)off);
  }
}

void mfo_keystream(const uint8_t seed[40], uint64_t off, void *out, size_t n)
{
  mfo_rng r;
  mfo_rng_init(&r, seed);
  mfo_rng_seek(&r, off);
  mfo_rng_gen(&r, out, n);
}

/* mpz2_urandomb, src/entropy.c:11-26 */
void mfo_urandomb(uint64_t *out, mfo_rng *r, size_t nbits)
{
  size_t limbs = (nbits + 63) / 64, bytes = nbits / 8;
  memset(out, 0, limbs * 8);
  mfo_rng_gen(r, out, bytes);
  if (limbs) out[limbs - 1] &= ~0ULL >> (limbs * 64 - nbits);
}

/* ------------------------------------------------------------------------------------------
 * LWE layer
 * ------------------------------------------------------------------------------------------ */
void mfo_modq(const mfo_params *P, uint64_t *v)
{
  /* src/lwe.h:107-118: limbs at index >= logq/64 are dropped (the mask on limb `pos` is dead
   * code because SIZ is then set to <= pos) => value mod 2^(64*floor(logq/64)). */
  for (uint32_t i = mfo_K(P); i < mfo_L(P); i++) v[i] = 0;
}

/* acc[0..K) += a[0..K) * b[0..K)  (mod 2^(64K)) */
static void mac_trunc(uint64_t *acc, const uint64_t *a, const uint64_t *b, uint32_t K)
{
  for (uint32_t i = 0; i < K; i++) {
    u128 carry = 0;
    uint64_t ai = a[i];
    if (!ai) continue;
    for (uint32_t j = 0; i + j < K; j++) {
      u128 t = (u128)ai * b[j] + acc[i + j] + (uint64_t)carry;
      acc[i + j] = (uint64_t)t;
      carry = t >> 64;
    }
  }
}

void mfo_add_dotp(const mfo_params *P, uint64_t *rop, const uint64_t *a, const uint64_t *b, size_t len)
{
  uint32_t L = mfo_L(P), K = mfo_K(P);
  /* the reference accumulates unreduced and truncates once (src/lwe.c:20-28); truncation commutes */
  for (size_t j = 0; j < len; j++) mac_trunc(rop, a + j * L, b + j * L, K);
  mfo_modq(P, rop);
}

void mfo_sample_a(const mfo_params *P, uint64_t *ct, mfo_rng *r)
{
  uint32_t L = mfo_L(P);
  for (uint32_t j = 0; j < P->n; j++) mfo_urandomb(ct + (size_t)j * L, r, P->logq);
}

/* v[0..K) = (v + w * x) mod 2^(64K), x < 2^64; v,w may alias */
static void addmul1_trunc(uint64_t *v, const uint64_t *w, uint64_t x, uint32_t K)
{
  u128 carry = 0;
  for (uint32_t i = 0; i < K; i++) {
    u128 t = (u128)w[i] * x + v[i] + (uint64_t)carry;
    v[i] = (uint64_t)t;
    carry = t >> 64;
  }
}

void mfo_encrypt(const mfo_params *P, uint64_t *ct, mfo_rng *r, const uint64_t *sk, uint64_t m, const uint64_t *e)
{
  uint32_t L = mfo_L(P), K = mfo_K(P);
  uint64_t *b = ct + (size_t)P->n * L;
  /* c[N] = e * p (src/lwe.c:86); the later sign flip of e is a no-op for the output (src/lwe.c:87) */
  memset(b, 0, L * 8);
  addmul1_trunc(b, e, MFO_P, K);
  mfo_sample_a(P, ct, r);                /* src/lwe.c:90 */
  mfo_add_dotp(P, b, sk, ct, P->n);      /* src/lwe.c:92 */
  u128 c = (u128)b[0] + m;               /* src/lwe.c:93 */
  b[0] = (uint64_t)c;
  for (uint32_t i = 1; i < K && (c >> 64); i++) { c = (u128)b[i] + 1; b[i] = (uint64_t)c; }
  mfo_modq(P, b);                        /* src/lwe.c:94 */
}

/* x mod p for an L-limb value */
static uint64_t limbs_mod_p(const uint64_t *v, uint32_t L)
{
  u128 r = 0;
  for (int i = (int)L - 1; i >= 0; i--) r = ((r << 64) | v[i]) % MFO_P;
  return (uint64_t)r;
}

uint64_t mfo_decrypt(const mfo_params *P, const uint64_t *sk, const uint64_t *ct)
{
  uint32_t L = mfo_L(P);
  uint64_t dot[64];
  memset(dot, 0, sizeof dot);
  mfo_add_dotp(P, dot, ct, sk, P->n);             /* mpz_dotp(m, ct, sk, N), src/lwe.c:107 */
  uint64_t bm = limbs_mod_p(ct + (size_t)P->n * L, L); /* b is NOT reduced first: full L limbs */
  uint64_t dm = limbs_mod_p(dot, L);
  return (bm + MFO_P - dm) % MFO_P;               /* mpz_mod_ui of the signed difference */
}

void mfo_ct_export(const mfo_params *P, uint8_t *buf, const uint64_t *ct)
{
  uint32_t L = mfo_L(P);
  uint8_t tmp[8 * 64];
  memcpy(tmp, ct + (size_t)P->n * L, L * 8);
  memcpy(buf, tmp, mfo_ctb(P)); /* little-endian bytes, zero padded (src/lwe.c:117-118) */
}

void mfo_ct_import(const mfo_params *P, uint64_t *ct, mfo_rng *r, const uint8_t *buf)
{
  uint32_t L = mfo_L(P);
  mfo_sample_a(P, ct, r);
  uint64_t *b = ct + (size_t)P->n * L;
  memset(b, 0, L * 8);
  memcpy(b, buf, mfo_ctb(P)); /* all CT_BYTES are imported, no modq (src/lwe.c:125) */
}

void mfo_ct_mul_ui(const mfo_params *P, uint64_t *rop, const uint64_t *a, uint64_t x)
{
  uint32_t L = mfo_L(P), K = mfo_K(P);
  for (uint32_t j = 0; j <= P->n; j++) {
    uint64_t tmp[64];
    memset(tmp, 0, L * 8);
    addmul1_trunc(tmp, a + (size_t)j * L, x, K);
    memcpy(rop + (size_t)j * L, tmp, L * 8);
  }
}

void mfo_ct_addmul_ui(const mfo_params *P, uint64_t *rop, const uint64_t *a, uint64_t x)
{
  uint32_t L = mfo_L(P), K = mfo_K(P);
  for (uint32_t j = 0; j <= P->n; j++) {
    addmul1_trunc(rop + (size_t)j * L, a + (size_t)j * L, x, K);
    mfo_modq(P, rop + (size_t)j * L);
  }
}

void mfo_ct_add(const mfo_params *P, uint64_t *rop, const uint64_t *a, const uint64_t *b)
{
  uint32_t L = mfo_L(P), K = mfo_K(P);
  for (uint32_t j = 0; j <= P->n; j++) {
    const uint64_t *x = a + (size_t)j * L, *y = b + (size_t)j * L;
    uint64_t *z = rop + (size_t)j * L;
    unsigned carry = 0;
    for (uint32_t i = 0; i < K; i++) {
      u128 t = (u128)x[i] + y[i] + carry;
      z[i] = (uint64_t)t;
      carry = (unsigned)(t >> 64);
    }
    for (uint32_t i = K; i < L; i++) z[i] = 0;
  }
}

int mfo_ct_smudge(const mfo_params *P, uint64_t *ct, const uint8_t *mag, size_t maglen, uint8_t sign)
{
  uint32_t L = mfo_L(P), K = mfo_K(P);
  uint64_t *b = ct + (size_t)P->n * L;
  uint64_t u[64], up[64];
  memset(u, 0, sizeof u);
  memset(up, 0, sizeof up);
  memcpy(u, mag, maglen); /* mpz2_urandomb2(smudging, 640): src/lwe.c:69 */
  /* up = u * p : at most maglen*8+32 bits < 2^(64K) for the supported parameter sets */
  addmul1_trunc(up, u, MFO_P, K);
  int would_be_negative = 0;
  if (sign & 1) { /* src/lwe.c:51-58 */
    /* b - up; the reference leaves a negative mpz unreduced when b < up (assert compiled out) */
    unsigned borrow = 0;
    for (uint32_t i = 0; i < K; i++) {
      u128 t = (u128)b[i] - up[i] - borrow;
      b[i] = (uint64_t)t;
      borrow = (unsigned)((t >> 64) & 1);
    }
    would_be_negative = (int)borrow;
  } else {
    unsigned carry = 0;
    for (uint32_t i = 0; i < K; i++) {
      u128 t = (u128)b[i] + up[i] + carry;
      b[i] = (uint64_t)t;
      carry = (unsigned)(t >> 64);
    }
  }
  mfo_modq(P, b);
  return would_be_negative;
}

void mfo_eval_poly(const mfo_params *P, uint64_t *rop, mfo_rng *r, const uint8_t *c8, const uint64_t *coeff, size_t d)
{
  uint32_t L = mfo_L(P);
  uint64_t *ct = malloc((size_t)(P->n + 1) * L * 8);
  for (size_t i = 0; i < d; i++) {
    mfo_ct_import(P, ct, r, c8 + i * mfo_ctb(P));
    mfo_ct_addmul_ui(P, rop, ct, coeff[i]);
  }
  free(ct);
}

/* ------------------------------------------------------------------------------------------
 * Polynomials over F_p, p = 2^32-5, dense arrays of canonical coefficients
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % MFO_P); }
static uint64_t powmod(uint64_t a, uint64_t e)
{
  uint64_t r = 1;
  while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; }
  return r;
}

void mfo_poly_import(uint64_t *poly, const void *buf, size_t d)
{
  const uint8_t *b = buf;
  for (size_t i = 0; i < d; i++) {
    uint64_t x;
    memcpy(&x, b + 8 * i, 8);
    poly[i] = x % MFO_P; /* nmod_poly_set_coeff_ui reduces */
  }
}

uint64_t mfo_poly_eval(const uint64_t *poly, size_t d, uint64_t x)
{
  uint64_t r = 0;
  for (size_t i = d; i-- > 0;) r = (mulmod(r, x) + poly[i]) % MFO_P;
  return r;
}

/* num = v^2 - 1 (2d-1 coeffs); Euclidean division by t; quotient to q (d coeffs, zero padded),
 * returns 1 if the remainder is zero. */
static int poly_h_impl(uint64_t *q, const uint64_t *v, const uint64_t *t, size_t d)
{
  size_t nn = 2 * d - 1;
  uint64_t *num = calloc(nn, 8);
  for (size_t i = 0; i < d; i++) {
    if (!v[i]) continue;
    for (size_t j = 0; j < d; j++) num[i + j] = (num[i + j] + mulmod(v[i], v[j])) % MFO_P;
  }
  num[0] = (num[0] + MFO_P - 1) % MFO_P;
  long dt = -1;
  for (long i = (long)d - 1; i >= 0; i--) if (t[i]) { dt = i; break; }
  if (q) memset(q, 0, d * 8);
  int zero_rem = 1;
  if (dt >= 0) {
    uint64_t linv = powmod(t[dt], MFO_P - 2);
    for (long i = (long)nn - 1; i >= dt; i--) {
      uint64_t c = mulmod(num[i], linv);
      if (c) {
        for (long j = 0; j <= dt; j++) num[i - dt + j] = (num[i - dt + j] + MFO_P - mulmod(c, t[j])) % MFO_P;
      }
      if (q && (size_t)(i - dt) < d) q[i - dt] = c;
    }
    for (long i = 0; i < dt; i++) if (num[i]) zero_rem = 0;
  } else {
    zero_rem = 0; /* division by zero polynomial: undefined in the reference too */
  }
  free(num);
  return zero_rem;
}

void mfo_poly_h(uint64_t *q, const uint64_t *v, const uint64_t *t, size_t d) { poly_h_impl(q, v, t, d); }
int mfo_poly_divides(const uint64_t *v, const uint64_t *t, size_t d) { return poly_h_impl(NULL, v, t, d); }

static inline int bit_of(const uint8_t *bits, size_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

void mfo_ssp_from_tape(const mfo_params *P, uint8_t *ssp, const uint8_t *tape, const uint8_t *witness_bits)
{
  size_t d = P->d;
  uint64_t *t = calloc(d, 8), *vi = malloc(d * 8);
  for (size_t i = 0; i < P->m; i++) {
    mfo_poly_import(vi, tape + i * 8 * d, d);           /* src/ssp.c:56-57,62-63 */
    memcpy(ssp + 8 * d * (i + 1), vi, 8 * d);           /* ssp_v_offset(i), src/ssp.h:9 */
    if (i == 0 || bit_of(witness_bits, i - 1))          /* src/ssp.c:59,66-68 */
      for (size_t k = 0; k < d; k++) t[k] = (t[k] + vi[k]) % MFO_P;
  }
  t[0] = (t[0] + MFO_P - 1) % MFO_P;                    /* src/ssp.c:71 */
  memcpy(ssp, t, 8 * d);                                /* ssp_t_offset */
  free(t); free(vi);
}

/* ------------------------------------------------------------------------------------------
 * SNARK layer
 * ------------------------------------------------------------------------------------------ */
void mfo_setup(const mfo_params *P, uint8_t *s, uint8_t *as, uint8_t *v, uint8_t *t, const uint8_t seed[40],
               const uint8_t *ssp, uint64_t alpha, uint64_t beta, uint64_t spt, const uint64_t *sk, const uint64_t *etape)
{
  uint32_t L = mfo_L(P), ctb = mfo_ctb(P);
  size_t d = P->d;
  mfo_rng rng;
  mfo_rng_init(&rng, seed);
  uint64_t *ct = malloc((size_t)(P->n + 1) * L * 8), *poly = malloc(d * 8);
  size_t ei = 0;
  uint64_t si = 1;
  for (size_t i = 0; i < d; i++) { /* src/snark.c:75-82 */
    mfo_encrypt(P, ct, &rng, sk, si, etape + (ei++) * L);
    mfo_ct_export(P, s + i * ctb, ct);
    si = mulmod(si, spt);
  }
  uint64_t asi = alpha;
  for (size_t i = 0; i < d; i++) { /* src/snark.c:84-91 */
    mfo_encrypt(P, ct, &rng, sk, asi, etape + (ei++) * L);
    mfo_ct_export(P, as + i * ctb, ct);
    asi = mulmod(asi, spt);
  }
  mfo_poly_import(poly, ssp, d); /* beta t(s), src/snark.c:97-101 */
  mfo_encrypt(P, ct, &rng, sk, mulmod(mfo_poly_eval(poly, d, spt), beta), etape + (ei++) * L);
  mfo_ct_export(P, t, ct);
  for (size_t i = 1; i < P->m; i++) { /* src/snark.c:104-110 */
    mfo_poly_import(poly, ssp + 8 * d * (i + 1), d);
    mfo_encrypt(P, ct, &rng, sk, mulmod(mfo_poly_eval(poly, d, spt), beta), etape + (ei++) * L);
    mfo_ct_export(P, v + (i - 1) * ctb, ct);
  }
  free(ct); free(poly);
}

void mfo_prover(const mfo_params *P, uint64_t *proof, uint64_t *pre, const mfo_crs *crs, const uint8_t *ssp,
                const uint8_t *witness_bits, uint64_t delta, const uint8_t *smudge_tape, size_t maglen,
                uint64_t *w_out, uint64_t *h_out)
{
  uint32_t L = mfo_L(P), ctb = mfo_ctb(P);
  size_t d = P->d, ctl = (size_t)(P->n + 1) * L;
  uint64_t *pi_h = proof, *pi_hat_h = proof + ctl, *pi_hat_v = proof + 2 * ctl, *pi_v_w = proof + 3 * ctl,
           *pi_b_w = proof + 4 * ctl;
  memset(proof, 0, 5 * ctl * 8);
  mfo_rng rng;
  mfo_rng_init(&rng, crs->seed);
  uint64_t *t = malloc(d * 8), *vi = malloc(d * 8), *w = malloc(d * 8), *h = malloc(d * 8);
  uint64_t *ct = malloc(ctl * 8);

  mfo_poly_import(t, ssp, d);                                   /* src/snark.c:138 */
  for (size_t k = 0; k < d; k++) w[k] = mulmod(t[k], delta);    /* src/snark.c:141 */

  mfo_rng_seek(&rng, mfo_ctr_bt(P));                            /* src/snark.c:143-145 */
  mfo_ct_import(P, pi_b_w, &rng, crs->t);
  mfo_ct_mul_ui(P, pi_b_w, pi_b_w, delta);

  for (size_t i = 1; i < P->m; i++) {                           /* src/snark.c:147-155 */
    mfo_ct_import(P, ct, &rng, crs->v + (i - 1) * ctb);
    if (bit_of(witness_bits, i - 1)) {
      mfo_poly_import(vi, ssp + 8 * d * (i + 1), d);
      for (size_t k = 0; k < d; k++) w[k] = (w[k] + vi[k]) % MFO_P;
      mfo_ct_add(P, pi_b_w, pi_b_w, ct);
    }
  }
  if (w_out) memcpy(w_out, w, d * 8);

  mfo_rng_seek(&rng, mfo_ctr_s(P));                             /* src/snark.c:157-158 */
  mfo_eval_poly(P, pi_v_w, &rng, crs->s, w, d);

  mfo_poly_import(vi, ssp + 8 * d, d);                          /* v_0: src/snark.c:161-164 */
  for (size_t k = 0; k < d; k++) w[k] = (w[k] + vi[k]) % MFO_P;
  mfo_rng_seek(&rng, mfo_ctr_as(P));
  mfo_eval_poly(P, pi_hat_v, &rng, crs->as, w, d);

  mfo_poly_h(h, w, t, d);                                       /* src/snark.c:166-169 */
  if (h_out) memcpy(h_out, h, d * 8);

  mfo_rng_seek(&rng, mfo_ctr_s(P));                             /* src/snark.c:171-174 */
  mfo_eval_poly(P, pi_h, &rng, crs->s, h, d);
  mfo_rng_seek(&rng, mfo_ctr_as(P));
  mfo_eval_poly(P, pi_hat_h, &rng, crs->as, h, d);

  if (pre) memcpy(pre, proof, 5 * ctl * 8);

  /* src/snark.c:185-189: h, hat_h, hat_v, v_w, v_w (sic); b_w is never smudged */
  uint64_t *order[5] = { pi_h, pi_hat_h, pi_hat_v, pi_v_w, pi_v_w };
  for (int k = 0; k < 5; k++) {
    const uint8_t *e = smudge_tape + (size_t)k * (maglen + 1);
    mfo_ct_smudge(P, order[k], e, maglen, e[maglen]);
  }
  free(t); free(vi); free(w); free(h); free(ct);
}

int mfo_verifier(const mfo_params *P, const uint8_t *ssp, uint64_t alpha, uint64_t beta, uint64_t spt,
                 const uint64_t *sk, const uint64_t *proof)
{
  size_t d = P->d, ctl = (size_t)(P->n + 1) * mfo_L(P);
  uint64_t *poly = malloc(d * 8);
  mfo_poly_import(poly, ssp, d);
  uint64_t t_s = mfo_poly_eval(poly, d, spt);
  uint64_t h_s = mfo_decrypt(P, sk, proof);
  uint64_t hath_s = mfo_decrypt(P, sk, proof + ctl);
  uint64_t hatv_s = mfo_decrypt(P, sk, proof + 2 * ctl);
  uint64_t w_s = mfo_decrypt(P, sk, proof + 3 * ctl);
  uint64_t b_s = mfo_decrypt(P, sk, proof + 4 * ctl);
  mfo_poly_import(poly, ssp + 8 * d, d);
  uint64_t v_s = (mfo_poly_eval(poly, d, spt) + w_s) % MFO_P;
  free(poly);
  if (mulmod(h_s, alpha) != hath_s) return 0;                      /* eq-pke, src/snark.c:220-222 */
  if (mulmod(v_s, alpha) != hatv_s) return 0;                      /* src/snark.c:223-225 */
  uint64_t lhs = (mulmod(v_s, v_s) + MFO_P - 1) % MFO_P;           /* eq-div, src/snark.c:227-231 */
  if (lhs != mulmod(h_s, t_s)) return 0;
  if (mulmod(w_s, beta) != b_s) return 0;                          /* eq-lin, src/snark.c:233-235 */
  /* test-error (src/snark.c:238-241): -dot/p is <= 0 so SIZ(test) <= 0 < 80: never rejects. */
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * cpu_baseline helpers: the reference's per-row cost structure, one thread.
 * ------------------------------------------------------------------------------------------ */
uint64_t mfo_bench_eval_rows(const mfo_params *P, const uint8_t seed[40], size_t rows)
{
  uint32_t L = mfo_L(P);
  size_t ctl = (size_t)(P->n + 1) * L;
  uint64_t *acc = calloc(ctl, 8), *ct = malloc(ctl * 8);
  uint8_t b[256];
  memset(b, 0x5a, sizeof b);
  memset(b + mfo_K(P) * 8, 0, sizeof b - mfo_K(P) * 8);
  mfo_rng rng;
  mfo_rng_init(&rng, seed);
  for (size_t i = 0; i < rows; i++) {
    mfo_ct_import(P, ct, &rng, b);
    mfo_ct_addmul_ui(P, acc, ct, 0x9e3779b9u + (uint32_t)i);
  }
  uint64_t x = 0;
  for (size_t i = 0; i < ctl; i++) x ^= acc[i] * (2 * i + 1);
  free(acc); free(ct);
  return x;
}

uint64_t mfo_bench_encrypt(const mfo_params *P, const uint8_t seed[40], size_t count)
{
  uint32_t L = mfo_L(P);
  size_t ctl = (size_t)(P->n + 1) * L;
  uint64_t *sk = malloc((size_t)P->n * L * 8), *ct = malloc(ctl * 8), e[64];
  mfo_rng rng, krng;
  uint8_t kseed[40];
  for (int i = 0; i < 40; i++) kseed[i] = (uint8_t)(seed[i] ^ 0xa5);
  mfo_rng_init(&krng, kseed);
  mfo_sample_a(P, sk, &krng);
  mfo_rng_init(&rng, seed);
  uint64_t x = 0;
  for (size_t i = 0; i < count; i++) {
    memset(e, 0, sizeof e);
    mfo_rng_gen(&krng, e, 69);
    mfo_encrypt(P, ct, &rng, sk, (0x12345u * (i + 1)) % MFO_P, e);
    x ^= ct[(size_t)P->n * L] + i;
  }
  free(sk); free(ct);
  return x;
}

/* `count` regev_decrypt calls (src/lwe.c:105-111, what src/benchmark_lwe.c:35-38 times) on one sampled ciphertext and key */
uint64_t mfo_bench_decrypt(const mfo_params *P, const uint8_t seed[40], size_t count)
{
  uint32_t L = mfo_L(P);
  size_t ctl = (size_t)(P->n + 1) * L;
  uint64_t *sk = malloc((size_t)P->n * L * 8), *ct = calloc(ctl, 8);
  mfo_rng rng, krng;
  uint8_t kseed[40];
  for (int i = 0; i < 40; i++) kseed[i] = (uint8_t)(seed[i] ^ 0xa5);
  mfo_rng_init(&krng, kseed);
  mfo_sample_a(P, sk, &krng);
  mfo_rng_init(&rng, seed);
  mfo_sample_a(P, ct, &rng);
  uint64_t x = 0;
  for (size_t i = 0; i < count; i++) {
    ct[(size_t)P->n * L] = i; /* b */
    x ^= mfo_decrypt(P, sk, ct) + i;
  }
  free(sk); free(ct);
  return x;
}
