/*
 * test_shim.c -- the properties the reference's own test programs assert (src/test_entropy.c, src/test_lwe.c,
 * src/test_snark.c), restated against libmfuoco_gpu (reference function names and signatures, debug parameters
 * D = 256, M = 64 like the reference's tests).  Our own text.  Exit code 0 = all properties hold.  Needs a GPU.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>

#include "mfuoco/mangiafuoco_api.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)
#define CTR_CT ((size_t)CT_BYTES * GAMMA_N)
#define CTR_AS (CTR_CT * GAMMA_D)

static uint64_t rnd_modp(void) { uint64_t r; getrandom(&r, 8, 0); return r % GAMMA_P; }

static void t_entropy(void)
{
  rseed_t seed;
  getrandom(seed, sizeof seed, 0);
  rng_t a, b;
  rng_init(a, seed);
  rng_init(b, seed);
  mpz_t x, y;
  mpz_inits(x, y, NULL);
  size_t widths[] = { 64, 1, 5, 32, 40, 520, 512, 736, 737, 743, 751 };
  for (size_t i = 0; i < sizeof widths / sizeof *widths; i++) { /* src/test_entropy.c:24-78 */
    mpz2_urandomb(x, a, widths[i]);
    mpz2_urandomb(y, b, widths[i]);
    CHECK(!mpz_cmp(x, y));
  }
  uint8_t bulk[92 * 40], chunk[92 * 40], sink[512];
  rng_seek(a, 0);
  rng_seek(b, 0);
  aesctr_prg((aesctr_ptr)a, bulk, sizeof bulk); /* :111-137 bulk == chunked */
  for (int i = 0; i < 40; i++) aesctr_prg((aesctr_ptr)b, chunk + 92 * i, 92);
  CHECK(!memcmp(bulk, chunk, sizeof bulk));
  rng_seek(a, 0);
  aesctr_prg((aesctr_ptr)a, sink, 512); /* :138-156 seek(512) == reading past 512 bytes */
  rng_seek(b, 512);
  uint64_t got, expected;
  aesctr_prg((aesctr_ptr)b, &got, 8);
  aesctr_prg((aesctr_ptr)a, &expected, 8);
  CHECK(got == expected);
  mpz_clears(x, y, NULL);
  rng_clear(a);
  rng_clear(b);
}

static void t_lwe(void)
{
  rseed_t seed;
  getrandom(seed, sizeof seed, 0);
  rng_t rng, twin;
  rng_init(rng, seed);
  rng_init(twin, seed);
  sk_t sk;
  key_gen(sk);
  ct_t c, c2, acc;
  ct_init(c); ct_init(c2); ct_init(acc);
  mpz_t m, m2, sum;
  mpz_inits(m, m2, sum, NULL);
  enum { d = 12 };
  uint8_t (*buf)[CT_BYTES] = calloc(d, CT_BYTES);
  nmod_poly_t coeffs;
  nmod_poly_init(coeffs, GAMMA_P);
  for (size_t i = 0; i < d; i++) {
    mpz_set_ui(m, rnd_modp());
    regev_encrypt2(c, rng, sk, m, errdist_uniform);
    regev_decrypt(m2, sk, c);
    CHECK(!mpz_cmp(m, m2)); /* src/test_lwe.c:74-95 */
    ct_export(buf[i], c);
    ct_import(c2, twin, buf[i]); /* :36-70 */
    for (size_t j = 0; j <= GAMMA_N; j += 97) CHECK(!mpz_cmp(c[j], c2[j]));
    CHECK(!mpz_cmp(c[GAMMA_N], c2[GAMMA_N]));
    ct_smudge(c); /* :183-205 */
    regev_decrypt(m2, sk, c);
    CHECK(!mpz_cmp(m, m2));
    mpz_add(sum, sum, m);
    nmod_poly_set_coeff_ui(coeffs, i, 1);
  }
  rng_seek(twin, 0);
  eval_poly(acc, twin, buf, coeffs, d); /* :105-181 */
  regev_decrypt(m2, sk, acc);
  mpz_mod_ui(sum, sum, GAMMA_P);
  CHECK(!mpz_cmp(sum, m2));
  /* ct_add / ct_mul_ui are homomorphic */
  rng_seek(twin, 0);
  ct_import(c, twin, buf[0]);
  ct_import(c2, twin, buf[1]);
  regev_decrypt(m, sk, c);
  regev_decrypt(m2, sk, c2);
  ct_add(acc, c, c2);
  mpz_add(sum, m, m2);
  mpz_mod_ui(sum, sum, GAMMA_P);
  regev_decrypt(m2, sk, acc);
  CHECK(!mpz_cmp(sum, m2));
  ct_mul_ui(acc, c, 12345);
  mpz_mul_ui(sum, m, 12345);
  mpz_mod_ui(sum, sum, GAMMA_P);
  regev_decrypt(m2, sk, acc);
  CHECK(!mpz_cmp(sum, m2));
  /* modq semantics: effective modulus 2^704 */
  ct_zero(c);
  mpz_ui_pow_ui(c[0], 2, 720);
  mpz_add_ui(c[0], c[0], 5);
  ct_mul_ui(acc, c, 1);
  CHECK(!mpz_cmp_ui(acc[0], 5));
  free(buf);
  nmod_poly_clear(coeffs);
  mpz_clears(m, m2, sum, NULL);
  ct_clear(c); ct_clear(c2); ct_clear(acc);
  key_clear(sk);
  rng_clear(rng);
  rng_clear(twin);
}

static void t_snark(void)
{
  crs_t crs;
  crs_init(crs);
  uint8_t *ssp = calloc(1, SSP_SIZE);
  mpz_t witness;
  mpz_init(witness);
  random_ssp(witness, ssp);
  vrs_t vrs;
  setup(crs, vrs, ssp);
  rng_t rng;
  rng_init(rng, crs->seed);
  ct_t ct_s, ct_as;
  ct_init(ct_s); ct_init(ct_as);
  mpz_t s, as;
  mpz_inits(s, as, NULL);
  rng_seek(rng, 0);
  ct_import(ct_s, rng, crs->s[0]);
  rng_seek(rng, CTR_AS);
  ct_import(ct_as, rng, crs->as[0]);
  regev_decrypt(s, vrs->sk, ct_s);
  regev_decrypt(as, vrs->sk, ct_as);
  CHECK(!mpz_cmp_ui(s, 1) && !mpz_cmp_ui(as, vrs->alpha)); /* src/test_snark.c:35-49 */
  size_t idx[2] = { 1, GAMMA_D - 1 };
  for (int k = 0; k < 2; k++) { /* :52-70 */
    rng_seek(rng, CTR_CT * idx[k]);
    ct_import(ct_s, rng, crs->s[idx[k]]);
    rng_seek(rng, CTR_AS + CTR_CT * idx[k]);
    ct_import(ct_as, rng, crs->as[idx[k]]);
    regev_decrypt(s, vrs->sk, ct_s);
    regev_decrypt(as, vrs->sk, ct_as);
    mpz_mul_ui(s, s, vrs->alpha);
    mpz_mod_ui(s, s, GAMMA_P);
    CHECK(!mpz_cmp(s, as));
  }
  proof_t pi;
  proof_init(pi);
  prover(pi, crs, ssp, witness);
  regev_decrypt(s, vrs->sk, pi->h);
  regev_decrypt(as, vrs->sk, pi->hat_h);
  mpz_mul_ui(s, s, vrs->alpha);
  mpz_mod_ui(s, s, GAMMA_P);
  CHECK(mpz_cmp_ui(s, 0) > 0 && !mpz_cmp(s, as)); /* :81-89 */
  CHECK(verifier(ssp, vrs, pi));                   /* :105-107 */
  mpz_add_ui(pi->v_w[GAMMA_N], pi->v_w[GAMMA_N], 1);
  CHECK(!verifier(ssp, vrs, pi));
  proof_clear(pi);
  crs_clear(crs);
  free(ssp);
  key_clear(vrs->sk);
  ct_clear(ct_s); ct_clear(ct_as);
  mpz_clears(s, as, witness, NULL);
  rng_clear(rng);
}

int main(void)
{
  t_entropy();
  puts("entropy ok");
  t_lwe();
  puts("lwe ok");
  t_snark();
  puts("snark ok");
  return 0;
}
