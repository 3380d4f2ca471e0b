/*
 * test_shim.c -- the properties the reference's own test programs assert (src/test_entropy.c, src/test_lwe.c,
 * src/test_snark.c), restated against libmfuoco_gpu (reference function names and signatures, debug parameters
 * D = 256, M = 64 like the reference's tests).  Our own text.  Exit code 0 = all properties hold.  Needs a GPU.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mfuoco/mangiafuoco_api.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)



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
  { /* three statements in one batch: the witness twice (accepted) and a corrupted witness (rejected) */
    proof_t pb[3];
    mpz_t wit[3];
    for (int k = 0; k < 3; k++) { proof_init(pb[k]); mpz_init_set(wit[k], witness); }
    mpz_combit(wit[1], 3);
    mfuoco_prover_batch(pb, crs, ssp, wit, 3);
    CHECK(verifier(ssp, vrs, pb[0]) && !verifier(ssp, vrs, pb[1]) && verifier(ssp, vrs, pb[2]));
    uint8_t okb[3] = { 9, 9, 9 };
    mfuoco_verifier_batch(ssp, vrs, pb, 3, okb); /* the same three through the device verifier */
    CHECK(okb[0] == 1 && okb[1] == 0 && okb[2] == 1);
    { /* and their h ciphertexts through the batched decryption: what regev_decrypt returns one by one */
      mpz_t mb[3], one;
      ct_t cb[3];
      mpz_init(one);
      for (int k = 0; k < 3; k++) {
        mpz_init(mb[k]);
        ct_init(cb[k]);
        for (size_t j = 0; j <= GAMMA_N; j++) mpz_set(cb[k][j], pb[k]->h[j]);
      }
      mfuoco_decrypt_batch(mb, vrs->sk, cb, 3);
      for (int k = 0; k < 3; k++) {
        regev_decrypt(one, vrs->sk, pb[k]->h);
        CHECK(!mpz_cmp(one, mb[k]));
        mpz_clear(mb[k]);
        ct_clear(cb[k]);
      }
      mpz_clear(one);
    }
    for (int k = 0; k < 3; k++) { proof_clear(pb[k]); mpz_clear(wit[k]); }
  }
  { /* the expanded CRS kept across calls (mangiafuoco_api.h: mfuoco_gpu_set_resident_crs): 40 statements per call so that the call streams an image.
     * Call 1 expands it, call 2 finds it (seed and device-side digest of the compressed CRS unchanged); then ONE BYTE of crs->s changes in place: the
     * image no longer serves that CRS, call 3 must notice without being told -- its proofs are computed from the corrupted row and are REJECTED (from
     * the stale image they would be accepted) --, and with the byte restored call 4 is accepted again.  The same for prover()'s single-proof image
     * (regenerates, expands, streams, notices). */
    enum { NB = 40 };
    proof_t pb[NB];
    mpz_t wit[NB];
    for (int k = 0; k < NB; k++) { proof_init(pb[k]); mpz_init_set(wit[k], witness); }
    for (int call = 1; call <= 4; call++) {
      if (call == 3) crs->s[5][40] ^= 0x10;
      if (call == 4) crs->s[5][40] ^= 0x10;
      mfuoco_prover_batch(pb, crs, ssp, wit, NB);
      uint8_t okb[NB];
      mfuoco_verifier_batch(ssp, vrs, pb, NB, okb);
      for (int k = 0; k < NB; k++) CHECK(okb[k] == (call == 3 ? 0 : 1));
    }
    for (int call = 1; call <= 5; call++) {
      if (call == 4) crs->as[7][12] ^= 0x01;
      if (call == 5) crs->as[7][12] ^= 0x01;
      prover(pi, crs, ssp, witness);
      CHECK(verifier(ssp, vrs, pi) == (call != 4));
    }
    mfuoco_gpu_set_resident_crs(0); /* off: images freed, every call regenerates as before */
    mfuoco_prover_batch(pb, crs, ssp, wit, NB);
    CHECK(verifier(ssp, vrs, pb[0]) && verifier(ssp, vrs, pb[NB - 1]));
    mfuoco_gpu_set_resident_crs(1);
    for (int k = 0; k < NB; k++) { proof_clear(pb[k]); mpz_clear(wit[k]); }
  }
  proof_clear(pi);
  crs_clear(crs);
  free(ssp);
  key_clear(vrs->sk);
  ct_clear(ct_s); ct_clear(ct_as);
  mpz_clears(s, as, witness, NULL);
  rng_clear(rng);
}

/* on-disk images: setup() straight into a mapped crs.mfuoco, SSP and proof through files, prover from the
 * re-mapped read-only images, verifier on the re-loaded proof; plus a "coeffs" row file driving eval_poly as
 * src/benchmark_eval.c:44-70 does. */
static void t_files(void)
{
  char dir[] = "/tmp/mfuoco_files_XXXXXX", path[4][64];
  CHECK(mkdtemp(dir) != NULL);
  const char *names[4] = { "crs.mfuoco", "ssp.mfuoco", "proof.mfuoco", "coeffs" };
  for (int i = 0; i < 4; i++) snprintf(path[i], sizeof path[i], "%s/%s", dir, names[i]);

  uint8_t *ssp = calloc(1, SSP_SIZE);
  mpz_t witness;
  mpz_init(witness);
  random_ssp(witness, ssp);
  CHECK(mfuoco_ssp_save(path[1], ssp) == 0);
  free(ssp);

  crs_t crs;
  CHECK(getrandom(crs->seed, sizeof(rseed_t), GRND_NONBLOCK) == sizeof(rseed_t));
  CHECK(mfuoco_crs_create(crs, path[0]) == 0);
  ssp = mfuoco_ssp_map(path[1], 0);
  CHECK(ssp != NULL);
  vrs_t vrs;
  setup(crs, vrs, ssp);
  rseed_t seed;
  memcpy(seed, crs->seed, sizeof seed);
  mfuoco_crs_unmap(crs);
  struct stat st;
  CHECK(stat(path[0], &st) == 0 && (size_t)st.st_size == CRS_SIZE);
  CHECK(stat(path[1], &st) == 0 && (size_t)st.st_size == SSP_SIZE);

  crs_t crs2;
  CHECK(mfuoco_crs_map(crs2, path[0], 0) == 0);
  CHECK(!memcmp(crs2->seed, seed, sizeof seed));
  CHECK((uint8_t *)crs2->as == (uint8_t *)crs2->s + CT_BYTES * GAMMA_D && crs2->t == (uint8_t *)crs2->s + 2 * CT_BYTES * GAMMA_D &&
        (uint8_t *)crs2->v == crs2->t + CT_BYTES);
  proof_t pi, pj;
  proof_init(pi);
  proof_init(pj);
  prover(pi, crs2, ssp, witness);
  CHECK(mfuoco_proof_save(path[2], pi) == 0);
  CHECK(mfuoco_proof_load(pj, path[2]) == 0);
  for (size_t j = 0; j <= GAMMA_N; j++) CHECK(!mpz_cmp(pi->h[j], pj->h[j]) && !mpz_cmp(pi->b_w[j], pj->b_w[j]));
  CHECK(verifier(ssp, vrs, pj));
  /* a saved copy of the heap-allocated form is the same image */
  crs_t crs3;
  crs_init(crs3);
  memcpy(crs3->seed, seed, sizeof seed);
  memcpy(crs3->s, crs2->s, CT_BYTES * GAMMA_D);
  memcpy(crs3->as, crs2->as, CT_BYTES * GAMMA_D);
  memcpy(crs3->t, crs2->t, CT_BYTES);
  memcpy(crs3->v, crs2->v, CT_BYTES * GAMMA_M);
  char copy[80];
  snprintf(copy, sizeof copy, "%s/copy.mfuoco", dir);
  CHECK(mfuoco_crs_save(copy, crs3) == 0);
  crs_t crs4;
  CHECK(mfuoco_crs_map(crs4, copy, 0) == 0);
  CHECK(!memcmp(crs4->s, crs2->s, CRS_SIZE));
  mfuoco_crs_unmap(crs4);
  crs_clear(crs3);
  /* wrong size is refused */
  CHECK(mfuoco_crs_map(crs4, path[1], 0) == -1 && errno == EINVAL);

  /* ciphertext-row file -> eval_poly over the mapping == eval_poly over the heap rows */
  {
    enum { ROWS = 19 };
    uint8_t (*rows)[CT_BYTES] = malloc(ROWS * CT_BYTES);
    CHECK(getrandom(rows, ROWS * CT_BYTES, GRND_NONBLOCK) == ROWS * CT_BYTES);
    CHECK(mfuoco_rows_save(path[3], rows, ROWS) == 0);
    size_t n = 0;
    uint8_t (*mapped)[CT_BYTES] = mfuoco_rows_map(path[3], &n);
    CHECK(mapped != NULL && n == ROWS);
    nmod_poly_t co;
    nmod_poly_init(co, GAMMA_P);
    for (size_t i = 0; i < ROWS; i++) nmod_poly_set_coeff_ui(co, i, 1000003u * (i + 1));
    rng_t rng;
    rng_init(rng, seed);
    ct_t a, b;
    ct_init(a);
    ct_init(b);
    eval_poly(a, rng, rows, co, ROWS);
    rng_seek(rng, 0);
    eval_poly(b, rng, mapped, co, ROWS);
    for (size_t j = 0; j <= GAMMA_N; j++) CHECK(!mpz_cmp(a[j], b[j]));
    ct_clear(a);
    ct_clear(b);
    rng_clear(rng);
    nmod_poly_clear(co);
    mfuoco_rows_unmap(mapped, n);
    free(rows);
  }

  mfuoco_crs_unmap(crs2);
  mfuoco_ssp_unmap(ssp);
  proof_clear(pi);
  proof_clear(pj);
  key_clear(vrs->sk);
  mpz_clear(witness);
  for (int i = 0; i < 4; i++) unlink(path[i]);
  unlink(copy);
  rmdir(dir);
}

int main(void)
{
  t_entropy();
  puts("entropy ok");
  t_lwe();
  puts("lwe ok");
  t_snark();
  puts("snark ok");
  t_files();
  puts("files ok");
  return 0;
}
