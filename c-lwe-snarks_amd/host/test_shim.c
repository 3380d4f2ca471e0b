/*
 * test_shim.c -- what libmfuoco_gpu adds to the reference's interface, tested through the reference's types at the debug parameters (D = 256, M = 64): the batch
 * entry points (encryption, decryption, prover, verifier), the images the shim keeps across calls, the on-disk formats.  The reference's own test programs run
 * unmodified against the library elsewhere (tests/test_gpu_reference_drivers.py).  Exit code 0 = everything holds.  Needs a GPU.
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

/* An entropy tape (SURVEY Appendix A's determinism recipe): this program's getrandom() comes before libc's in the lookup order of every object of the process, the shim
 * included.  Off (the default) it is the system call; between tape_start(seed) and tape_stop() it serves a splitmix64 stream, so that two prover() calls can be given the
 * same delta and smudging terms. */
#include <sys/syscall.h>
static int tape_on;
static uint64_t tape_state;
static void tape_start(uint64_t seed) { tape_on = 1; tape_state = seed; }
static void tape_stop(void) { tape_on = 0; }
ssize_t getrandom(void *buf, size_t len, unsigned int flags)
{
  if (!tape_on) return syscall(SYS_getrandom, buf, len, flags);
  uint8_t *p = buf;
  for (size_t i = 0; i < len; i += 8) {
    uint64_t z = (tape_state += 0x9e3779b97f4a7c15UL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9UL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebUL;
    z ^= z >> 31;
    memcpy(p + i, &z, len - i < 8 ? len - i : 8);
  }
  return (ssize_t)len;
}
static int proofs_equal(proof_t a, proof_t b)
{
  for (size_t j = 0; j <= GAMMA_N; j++)
    if (mpz_cmp(a->h[j], b->h[j]) || mpz_cmp(a->hat_h[j], b->hat_h[j]) || mpz_cmp(a->hat_v[j], b->hat_v[j]) || mpz_cmp(a->v_w[j], b->v_w[j]) || mpz_cmp(a->b_w[j], b->b_w[j])) return 0;
  return 1;
}


static uint64_t rnd_modp(void) { uint64_t r; getrandom(&r, 8, 0); return r % GAMMA_P; }

/* mfuoco_encrypt_batch / mfuoco_encrypt_batch2 / mfuoco_decrypt_rows_batch (the loops of src/benchmark_lwe.c:28-38 and src/snark.c:75-110 as one call):
 *   - with a deterministic error distribution, row k of a batch is bit for bit what regev_encrypt2 + ct_export give with the stream at k * CTR_CT and the k-th error;
 *   - the stream ends count rows further, exactly where a seek would put it;
 *   - with the bulk errdist_uniform draw every row decrypts to its message, through the seed-compressed batch decryption and through ct_import + mfuoco_decrypt_batch;
 *   - three chunks in flight (2 x 16384 + 37 rows), a one-row batch, and the key cache: a second key is noticed. */
static uint64_t chi_index;
static void chi_counter(mpz_t e)
{ /* a 552-bit value that depends on the draw's index only */
  mpz_set_ui(e, 0x9e3779b97f4a7c15UL * (chi_index + 1));
  mpz_mul_2exp(e, e, 480);
  mpz_add_ui(e, e, 1000003UL * chi_index + 7);
  chi_index++;
}
static void t_encrypt_batch(void)
{
  enum { COUNT = 2 * 16384 + 37 };
  rseed_t seed;
  getrandom(seed, sizeof seed, 0);
  rng_t rs, twin;
  rng_init(rs, seed);
  rng_init(twin, seed);
  sk_t sk, sk2;
  key_gen(sk);
  key_gen(sk2);
  mpz_t *ms = malloc(COUNT * sizeof *ms), *back = malloc(COUNT * sizeof *back);
  for (size_t k = 0; k < COUNT; k++) { mpz_init_set_ui(ms[k], rnd_modp()); mpz_init(back[k]); }
  mpz_set_ui(ms[0], 0);
  mpz_set_ui(ms[1], GAMMA_P - 1);
  uint8_t (*c8)[CT_BYTES] = malloc((size_t)COUNT * CT_BYTES), one[CT_BYTES];
  ct_t c;
  ct_init(c);

  /* deterministic errors: batch == one at a time */
  const uint64_t base = 3 * CTR_CT; /* (an odd row: the stream position is 8 mod 16) */
  rng_seek(rs, base);
  chi_index = 0;
  mfuoco_encrypt_batch2(c8, rs, sk, ms, COUNT, chi_counter);
  size_t probe[] = { 0, 1, 2, 16383, 16384, 16385, 32767, 32768, COUNT - 1 };
  for (size_t i = 0; i < sizeof probe / sizeof *probe; i++) {
    const size_t k = probe[i];
    rng_seek(twin, base + k * CTR_CT);
    chi_index = k;
    regev_encrypt2(c, twin, sk, ms[k], chi_counter);
    ct_export(one, c);
    CHECK(!memcmp(one, c8[k], CT_BYTES));
  }
  uint64_t got, want;
  rng_seek(twin, base + (uint64_t)COUNT * CTR_CT);
  aesctr_prg((aesctr_ptr)rs, &got, 8);
  aesctr_prg((aesctr_ptr)twin, &want, 8);
  CHECK(got == want);

  /* OS-entropy errors in bulk: everything decrypts */
  rng_seek(rs, 0);
  mfuoco_encrypt_batch(c8, rs, sk, ms, COUNT);
  rng_seek(twin, 0);
  mfuoco_decrypt_rows_batch(back, twin, sk, c8, COUNT);
  for (size_t k = 0; k < COUNT; k++) CHECK(!mpz_cmp(ms[k], back[k]));
  {
    enum { NCT = 5 };
    size_t rows[NCT] = { 0, 7, 16384, 20001, COUNT - 1 };
    ct_t cts[NCT];
    mpz_t out[NCT];
    for (int i = 0; i < NCT; i++) {
      ct_init(cts[i]);
      mpz_init(out[i]);
      rng_seek(twin, rows[i] * CTR_CT);
      ct_import(cts[i], twin, c8[rows[i]]);
    }
    mfuoco_decrypt_batch(out, sk, cts, NCT);
    for (int i = 0; i < NCT; i++) {
      CHECK(!mpz_cmp(out[i], ms[rows[i]]));
      regev_decrypt(back[0], sk, cts[i]);
      CHECK(!mpz_cmp(back[0], ms[rows[i]]));
      ct_clear(cts[i]);
      mpz_clear(out[i]);
    }
  }
  { /* a stream position that is no multiple of 8 (a caller that drew 5 bytes first): both batch calls still serve it (the encryption on the VALU kernel, the decryption row by row) */
    enum { NU = 9 };
    rng_seek(rs, 5);
    mfuoco_encrypt_batch(c8, rs, sk, ms, NU);
    rng_seek(twin, 5);
    mfuoco_decrypt_rows_batch(back, twin, sk, c8, NU);
    for (size_t k = 0; k < NU; k++) CHECK(!mpz_cmp(ms[k], back[k]));
    rng_seek(twin, 5 + (uint64_t)NU * CTR_CT);
    aesctr_prg((aesctr_ptr)rs, &got, 8);
    aesctr_prg((aesctr_ptr)twin, &want, 8);
    CHECK(got == want);
  }
  /* another key is noticed by the key cache: rows under sk2 decrypt under sk2 and (overwhelmingly) not under sk; one-row batch */
  rng_seek(rs, 0);
  mfuoco_encrypt_batch(c8, rs, sk2, ms, 1);
  rng_seek(twin, 0);
  mfuoco_decrypt_rows_batch(back, twin, sk2, c8, 1);
  CHECK(!mpz_cmp(ms[0], back[0]));
  rng_seek(twin, 0);
  mpz_set_ui(ms[2], 123456789);
  mfuoco_encrypt_batch(c8, twin, sk2, ms + 2, 1);
  rng_seek(twin, 0);
  mfuoco_decrypt_rows_batch(back, twin, sk, c8, 1);
  CHECK(mpz_cmp(ms[2], back[0]));
  rng_seek(twin, 0);
  ct_import(c, twin, c8[0]);
  regev_decrypt(back[0], sk2, c);
  CHECK(!mpz_cmp(ms[2], back[0]));

  for (size_t k = 0; k < COUNT; k++) { mpz_clear(ms[k]); mpz_clear(back[k]); }
  free(ms); free(back); free(c8);
  ct_clear(c);
  key_clear(sk);
  key_clear(sk2);
  rng_clear(rs);
  rng_clear(twin);
}

/* The batch entry points and the resident images behind the reference's types.  (The reference's own test programs -- src/test_entropy.c, test_lwe.c, test_snark.c,
 * test_ssp.c, test_aes.c -- run unmodified against this library in tests/test_gpu_reference_drivers.py; what they assert is not restated here.) */
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
  proof_t pi;
  proof_init(pi);
  { /* setup() leaves the expanded rows behind, so the FIRST prover() under its CRS streams them -- and what it computes is, on one entropy tape, bit for
     * bit the proof the regenerating path (the reference's ct_import, src/lwe.c:122-126) computes */
    proof_t pr;
    proof_init(pr);
    tape_start(0x5eed0001);
    prover(pi, crs, ssp, witness);
    tape_stop();
    CHECK(mfuoco_gpu_last_prover_path() == 1);
    CHECK(verifier(ssp, vrs, pi));
    mfuoco_gpu_set_resident_crs(0); /* images freed: the next call regenerates */
    tape_start(0x5eed0001);
    prover(pr, crs, ssp, witness);
    tape_stop();
    CHECK(mfuoco_gpu_last_prover_path() == 0);
    CHECK(proofs_equal(pi, pr));
    tape_start(0x5eed0002); /* (and the tape matters: other smudging terms, another proof) */
    prover(pr, crs, ssp, witness);
    tape_stop();
    CHECK(!proofs_equal(pi, pr) && verifier(ssp, vrs, pr));
    mfuoco_gpu_set_resident_crs(1);
    proof_clear(pr);
  }
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
  { /* more proofs than one pinned slab holds in either direction (64 proofs = 320 ciphertexts per slab): 150 statements down through the drain pipeline (two super-groups'
     * worth of events would need 256; here one, three slabs), up again through the upload pipeline of the batch verifier and of the batch decryption; one proof is damaged
     * on the host in between and must be the only one rejected */
    enum { NP = 150 };
    proof_t *pb = malloc(NP * sizeof *pb);
    mpz_t *wit = malloc(NP * sizeof *wit);
    uint8_t okb[NP];
    for (int k = 0; k < NP; k++) { proof_init(pb[k]); mpz_init_set(wit[k], witness); }
    mfuoco_prover_batch(pb, crs, ssp, wit, NP);
    mpz_add_ui(pb[97]->hat_h[GAMMA_N], pb[97]->hat_h[GAMMA_N], 1);
    mfuoco_verifier_batch(ssp, vrs, pb, NP, okb);
    for (int k = 0; k < NP; k++) CHECK(okb[k] == (k != 97));
    ct_t *cts = malloc(5 * NP * sizeof *cts);
    mpz_t *dec = malloc(5 * NP * sizeof *dec);
    mpz_t one;
    mpz_init(one);
    for (int k = 0; k < 5 * NP; k++) {
      ct_init(cts[k]);
      mpz_init(dec[k]);
      struct proof *pk = pb[k / 5];
      mpz_t *src = k % 5 == 0 ? pk->h : k % 5 == 1 ? pk->hat_h : k % 5 == 2 ? pk->hat_v : k % 5 == 3 ? pk->v_w : pk->b_w;
      for (size_t j = 0; j <= GAMMA_N; j++) mpz_set(cts[k][j], src[j]);
    }
    mfuoco_decrypt_batch(dec, vrs->sk, cts, 5 * NP);
    for (int k = 0; k < 5 * NP; k += 37) { /* against the one-at-a-time path */
      regev_decrypt(one, vrs->sk, cts[k]);
      CHECK(!mpz_cmp(one, dec[k]));
    }
    for (int k = 0; k < NP; k++) { /* eq-pke on the decrypted values: hat_h = alpha h (src/snark.c:213-216), except for the damaged proof */
      mpz_mul_ui(one, dec[5 * k], vrs->alpha);
      mpz_mod_ui(one, one, GAMMA_P);
      CHECK((mpz_cmp(one, dec[5 * k + 1]) == 0) == (k != 97));
    }
    for (int k = 0; k < 5 * NP; k++) { ct_clear(cts[k]); mpz_clear(dec[k]); }
    for (int k = 0; k < NP; k++) { proof_clear(pb[k]); mpz_clear(wit[k]); }
    mpz_clear(one);
    free(cts); free(dec); free(pb); free(wit);
  }
  proof_clear(pi);
  crs_clear(crs);
  free(ssp);
  key_clear(vrs->sk);
  mpz_clear(witness);
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
  mfuoco_gpu_invalidate(); /* (forget what setup() left on the device: the mapping below must bring its own row image) */
  CHECK(mfuoco_crs_map(crs2, path[0], 0) == 0);
  CHECK(!memcmp(crs2->seed, seed, sizeof seed));
  CHECK((uint8_t *)crs2->as == (uint8_t *)crs2->s + CT_BYTES * GAMMA_D && crs2->t == (uint8_t *)crs2->s + 2 * CT_BYTES * GAMMA_D &&
        (uint8_t *)crs2->v == crs2->t + CT_BYTES);
  proof_t pi, pj;
  proof_init(pi);
  proof_init(pj);
  tape_start(0x5eed0003);
  prover(pi, crs2, ssp, witness); /* the first prover() under a mapped CRS streams the rows mfuoco_crs_map() had expanded in the background ... */
  tape_stop();
  CHECK(mfuoco_gpu_last_prover_path() == 1);
  mfuoco_gpu_set_resident_crs(0);
  tape_start(0x5eed0003);
  prover(pj, crs2, ssp, witness); /* ... and computes what the regenerating path computes */
  tape_stop();
  CHECK(mfuoco_gpu_last_prover_path() == 0 && proofs_equal(pi, pj));
  mfuoco_gpu_set_resident_crs(1);
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
  t_encrypt_batch();
  puts("encrypt batch ok");
  t_snark();
  puts("snark ok");
  t_files();
  puts("files ok");
  return 0;
}
