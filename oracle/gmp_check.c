/*
 * gmp_check.c -- second restatement of the LWE arithmetic that performs the SAME libgmp calls the
 * reference performs (mpz_addmul, mpz_addmul_ui, mpz_mul_ui, mpz_add, mpz_mod_ui, mpz_import/export),
 * so that the limb-level oracle (mf_oracle.c) is checked against real GMP 6.2.1 semantics.
 *
 * TEST INFRASTRUCTURE ONLY.  Our own code (the reference's lwe.c cannot be compiled here: FLINT is
 * absent).  Values cross the boundary as little-endian uint64 limb arrays of `L` limbs.
 * logq = 736 only (the reference's modq is `#error` for anything else, src/lwe.h:119-121).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <gmp.h>

#define P32 0xfffffffbUL
#define LOGQ 736
#define L 12

static void get(mpz_t z, const uint64_t *v, size_t limbs) { mpz_import(z, limbs, -1, 8, 0, 0, v); }
static void put(uint64_t *v, const mpz_t z, size_t limbs)
{
  memset(v, 0, limbs * 8);
  size_t cnt = 0;
  /* caller guarantees it fits; negative values export their magnitude */
  mpz_export(v, &cnt, -1, 8, 0, 0, z);
}

/* modq as src/lwe.h:107-118 behaves: if more than pos = LOGQ/64 limbs are in use, mask limb `pos`,
 * strip leading zero limbs starting from limb pos-1 downwards, and set the size to what is left. */
static void modq_like_ref(mpz_t a)
{
  int pos = LOGQ / 64;
  if (a->_mp_size > pos) {
    a->_mp_d[pos] &= (1UL << 32) - 1;
    while (pos > 0 && a->_mp_d[pos - 1] == 0) pos--;
    a->_mp_size = pos;
  }
}

void gx_modq(uint64_t *out, const uint64_t *in, size_t limbs)
{
  mpz_t a;
  mpz_init(a);
  get(a, in, limbs);
  modq_like_ref(a);
  put(out, a, limbs);
  mpz_clear(a);
}

/* rop += sum a[j]*b[j]; modq  (src/lwe.c:20-28) */
void gx_add_dotp(uint64_t *rop, const uint64_t *a, const uint64_t *b, size_t len)
{
  mpz_t r, x, y;
  mpz_inits(r, x, y, NULL);
  get(r, rop, L);
  for (size_t j = 0; j < len; j++) {
    get(x, a + j * L, L);
    get(y, b + j * L, L);
    mpz_addmul(r, x, y);
  }
  modq_like_ref(r);
  put(rop, r, L);
  mpz_clears(r, x, y, NULL);
}

/* one coordinate of ct_mul_ui / ct_addmul_ui / ct_add (src/lwe.c:131-157) */
void gx_mul_ui(uint64_t *rop, const uint64_t *a, uint64_t b)
{
  mpz_t r, x;
  mpz_inits(r, x, NULL);
  get(x, a, L);
  mpz_mul_ui(r, x, b);
  modq_like_ref(r);
  put(rop, r, L);
  mpz_clears(r, x, NULL);
}
void gx_addmul_ui(uint64_t *rop, const uint64_t *a, uint64_t b)
{
  mpz_t r, x;
  mpz_inits(r, x, NULL);
  get(r, rop, L);
  get(x, a, L);
  mpz_addmul_ui(r, x, b);
  modq_like_ref(r);
  put(rop, r, L);
  mpz_clears(r, x, NULL);
}
void gx_add(uint64_t *rop, const uint64_t *a, const uint64_t *b)
{
  mpz_t r, x, y;
  mpz_inits(r, x, y, NULL);
  get(x, a, L);
  get(y, b, L);
  mpz_add(r, x, y);
  modq_like_ref(r);
  put(rop, r, L);
  mpz_clears(r, x, y, NULL);
}

/* b of regev_encrypt2 given the sampled a's (src/lwe.c:78-97) */
void gx_encrypt_b(uint64_t *b, const uint64_t *a, const uint64_t *sk, size_t n, uint64_t m, const uint64_t *e)
{
  mpz_t c, x, y;
  mpz_inits(c, x, y, NULL);
  get(x, e, L);
  mpz_mul_ui(c, x, P32);
  for (size_t j = 0; j < n; j++) {
    get(x, sk + j * L, L);
    get(y, a + j * L, L);
    mpz_addmul(c, x, y);
  }
  modq_like_ref(c);
  mpz_add_ui(c, c, m);
  modq_like_ref(c);
  put(b, c, L);
  mpz_clears(c, x, y, NULL);
}

/* regev_decrypt (src/lwe.c:105-111) on explicit (a, b) */
uint64_t gx_decrypt(const uint64_t *a, const uint64_t *b, const uint64_t *sk, size_t n)
{
  mpz_t m, x, y;
  mpz_inits(m, x, y, NULL);
  for (size_t j = 0; j < n; j++) {
    get(x, a + j * L, L);
    get(y, sk + j * L, L);
    mpz_addmul(m, x, y);
  }
  modq_like_ref(m);
  mpz_neg(m, m);
  get(x, b, L);
  mpz_add(m, x, m);
  uint64_t r = mpz_mod_ui(m, m, P32);
  mpz_clears(m, x, y, NULL);
  return r;
}

/* ct_smudge on coordinate b (src/lwe.c:65-76).  Returns 1 if the result is negative (left unreduced by the
 * reference under NDEBUG); then `b` receives the value reduced mod 2^704 instead, as the oracle does. */
int gx_smudge(uint64_t *b, const uint8_t *mag, size_t maglen, uint8_t sign)
{
  mpz_t s, x;
  mpz_inits(s, x, NULL);
  mpz_import(s, maglen, -1, 1, 0, 0, mag);
  if (sign & 1) mpz_neg(s, s);
  mpz_mul_ui(s, s, P32);
  get(x, b, L);
  mpz_add(x, x, s);
  int neg = mpz_sgn(x) < 0;
  if (neg) {
    mpz_t q;
    mpz_init(q);
    mpz_ui_pow_ui(q, 2, 704);
    mpz_mod(x, x, q);
    mpz_clear(q);
  } else {
    modq_like_ref(x);
  }
  put(b, x, L);
  mpz_clears(s, x, NULL);
  return neg;
}
