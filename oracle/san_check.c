/*
 * san_check.c -- runs the oracle end to end at a small instance under AddressSanitizer + UndefinedBehaviorSanitizer
 * ("build CPU restatement under ASan/UBSan"; the reference itself has known UB in this area -- stale limb bits,
 * MPN_NORMALIZE underrun on zero, src/gmp-impl.h:16-24 -- which the restatement must not inherit).
 * TEST INFRASTRUCTURE ONLY.  Exit code 0 = every property below holds and no sanitizer report was raised.
 *
 * Instance: n = 1470, logq in {736, 1472}, D = 16, M = 9 (sizes are runtime values in the oracle).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mf_oracle.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "san_check FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static uint64_t x_ = 0x9e3779b97f4a7c15ULL;
static uint64_t rnd(void) { x_ ^= x_ << 13; x_ ^= x_ >> 7; x_ ^= x_ << 17; return x_; }
static void fill(void *p, size_t n) { uint8_t *b = p; for (size_t i = 0; i < n; i++) b[i] = (uint8_t)rnd(); }

static void run(uint32_t logq)
{
  const mfo_params P = { 1470, logq, 16, 9 };
  const uint32_t L = mfo_L(&P), ctb = mfo_ctb(&P);
  const size_t ctl = (size_t)(P.n + 1) * L;
  uint8_t seed[40];
  fill(seed, sizeof seed);

  /* stream: bulk == chunked, seek == skip (src/test_entropy.c:111-156) */
  {
    uint8_t a[1000], b[1000];
    mfo_keystream(seed, 3, a, sizeof a);
    mfo_rng r;
    mfo_rng_init(&r, seed);
    mfo_rng_seek(&r, 3);
    for (size_t i = 0; i < sizeof b; i += 7) mfo_rng_gen(&r, b + i, sizeof b - i < 7 ? sizeof b - i : 7);
    CHECK(!memcmp(a, b, sizeof a));
    uint64_t v[32];
    const size_t widths[] = { 1, 5, 32, 40, 64, 512, 520, 736, 737, 751, 1472 };
    for (size_t i = 0; i < sizeof widths / sizeof *widths; i++) mfo_urandomb(v, &r, widths[i]);
  }

  /* LWE: dec(enc(m)) = m, export/import round trip, homomorphic ops, smudging (src/test_lwe.c) */
  uint64_t *sk = calloc((size_t)P.n * L, 8), *e = calloc(L, 8), *ct = calloc(ctl, 8), *ct2 = calloc(ctl, 8), *acc = calloc(ctl, 8);
  uint8_t *buf = malloc(ctb);
  for (uint32_t i = 0; i < P.n; i++) { fill(sk + (size_t)i * L, 8 * (logq / 64)); }
  for (int it = 0; it < 3; it++) {
    memset(e, 0, L * 8);
    fill(e, 69);
    const uint64_t m = rnd() % MFO_P;
    mfo_rng r, r2;
    mfo_rng_init(&r, seed);
    mfo_rng_seek(&r, (uint64_t)it * mfo_ctr_ct(&P));
    r2 = r;
    mfo_encrypt(&P, ct, &r, sk, m, e);
    CHECK(mfo_decrypt(&P, sk, ct) == m);
    mfo_ct_export(&P, buf, ct);
    mfo_ct_import(&P, ct2, &r2, buf);
    CHECK(!memcmp(ct, ct2, ctl * 8));
    mfo_ct_mul_ui(&P, ct2, ct, 7);
    CHECK(mfo_decrypt(&P, sk, ct2) == m * 7 % MFO_P);
    mfo_ct_addmul_ui(&P, ct2, ct, 5);
    CHECK(mfo_decrypt(&P, sk, ct2) == m * 12 % MFO_P);
    mfo_ct_add(&P, acc, ct, ct2);
    CHECK(mfo_decrypt(&P, sk, acc) == m * 13 % MFO_P);
    uint8_t mag[80];
    fill(mag, sizeof mag);
    mfo_ct_smudge(&P, ct, mag, sizeof mag, (uint8_t)(it & 1));
    CHECK(mfo_decrypt(&P, sk, ct) == m);
  }

  /* SSP + SNARK: setup -> prover -> verifier accepts; a flipped witness bit is rejected (src/test_ssp.c, src/test_snark.c) */
  {
    const size_t ssp_bytes = (size_t)P.d * 8 * (P.m + 3);
    uint8_t *ssp = calloc(1, ssp_bytes), *tape = malloc((size_t)P.m * 8 * P.d), bits[2] = { 0xa5, 0x01 };
    fill(tape, (size_t)P.m * 8 * P.d);
    mfo_ssp_from_tape(&P, ssp, tape, bits);
    const size_t rows = 2 * (size_t)P.d + P.m;
    uint64_t *etape = calloc(rows * L, 8);
    for (size_t i = 0; i < rows; i++) fill(etape + i * L, 69);
    uint8_t *s = malloc((size_t)P.d * ctb), *as = malloc((size_t)P.d * ctb), *v = calloc(P.m, ctb), *t = malloc(ctb);
    const uint64_t alpha = rnd() % MFO_P, beta = rnd() % MFO_P, spt = rnd() % MFO_P;
    mfo_setup(&P, s, as, v, t, seed, ssp, alpha, beta, spt, sk, etape);
    const mfo_crs crs = { seed, s, as, v, t };
    uint64_t *proof = calloc(5 * ctl, 8);
    uint8_t stape[5 * 81];
    fill(stape, sizeof stape);
    mfo_prover(&P, proof, NULL, &crs, ssp, bits, rnd() % MFO_P, stape, 80, NULL, NULL);
    CHECK(mfo_verifier(&P, ssp, alpha, beta, spt, sk, proof) == 1);
    uint8_t bad[2] = { (uint8_t)(bits[0] ^ 2), bits[1] };
    mfo_prover(&P, proof, NULL, &crs, ssp, bad, rnd() % MFO_P, stape, 80, NULL, NULL);
    CHECK(mfo_verifier(&P, ssp, alpha, beta, spt, sk, proof) == 0);
    free(ssp); free(tape); free(etape); free(s); free(as); free(v); free(t); free(proof);
  }
  free(sk); free(e); free(ct); free(ct2); free(acc); free(buf);
}

int main(void)
{
  run(736);
  run(1472);
  puts("san_check ok");
  return 0;
}
