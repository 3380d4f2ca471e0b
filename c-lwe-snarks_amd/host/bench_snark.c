/*
 * bench_snark.c -- the measurement the reference's benchmark_snark / benchmark_lwe programs make (wall clock around
 * setup(), prover(), verifier(), regev_encrypt(), regev_decrypt(); `label\tseconds` lines on stdout), written against
 * the same function names, for libmfuoco_gpu.  Our own text; the reference's drivers also link unchanged
 * (tests/test_link_compat.py) but their sources do not travel to the GPU box.
 *
 * This is the DROP-IN path: every call crosses PCIe and converts mpz_t, so it measures the PCIe/host-inclusive cost of
 * using the GPU through the reference API, not the kernel throughput bench.py reports.
 *   usage: bench_snark [nproofs] [nenc] [nbatch] [eval] [nencbatch]
 *          nbatch statements through mfuoco_prover_batch, three calls (the first expands the CRS image, the others stream the image the shim kept:
 *          `prover_batch` lines); eval != 0: what src/benchmark_eval.c:30-86 measures (D encryptions written to a coeffs file, the file mapped, ONE
 *          eval_poly over its D rows timed: `eval` line); nencbatch messages
 *          through mfuoco_encrypt_batch (three calls) and back through mfuoco_decrypt_rows_batch (`encryption_batch` / `decryption_rows_batch` lines).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/random.h>
#include <sys/time.h>

#include "mfuoco/mangiafuoco_api.h"

static double now(void)
{
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return tv.tv_sec + tv.tv_usec * 1e-6;
}

int main(int argc, char **argv)
{
  int nproofs = argc > 1 ? atoi(argv[1]) : 3, nenc = argc > 2 ? atoi(argv[2]) : 200, nbatch = argc > 3 ? atoi(argv[3]) : 255, do_eval = argc > 4 ? atoi(argv[4]) : 1;
  int nencb = argc > 5 ? atoi(argv[5]) : 65536;
  ssp_t ssp = calloc(1, SSP_SIZE);
  mpz_t witness;
  mpz_init(witness);
  double t0 = now();
  random_ssp(witness, ssp);
  fprintf(stderr, "ssp generation (host, getrandom)\t%lf\n", now() - t0);

  crs_t crs;
  crs_init(crs);
  vrs_t vrs;
  t0 = now();
  setup(crs, vrs, ssp);
  printf("setup\t%lf\n", now() - t0);

  proof_t pi;
  proof_init(pi);
  int ok = 1;
  for (int k = 0; k < nproofs; k++) {
    t0 = now();
    prover(pi, crs, ssp, witness);
    printf("prover\t%lf\n", now() - t0);
    t0 = now();
    bool out = verifier(ssp, vrs, pi);
    printf("verifier\t%lf\n", now() - t0);
    ok = ok && out;
  }

  /* many statements under one CRS through the reference's types: the first call expands the CRS (AES on the CU) and the shim keeps the image, keyed by
   * the seed and a device-side digest of the compressed CRS; the following calls stream it (the materialised-CRS regime) */
  if (nbatch > 0) {
    proof_t *pb = malloc(nbatch * sizeof *pb);
    mpz_t *wit = malloc(nbatch * sizeof *wit);
    uint8_t *okb = malloc(nbatch);
    for (int k = 0; k < nbatch; k++) { proof_init(pb[k]); mpz_init_set(wit[k], witness); }
    for (int call = 0; call < 3; call++) {
      t0 = now();
      mfuoco_prover_batch(pb, crs, ssp, wit, nbatch);
      double dt = now() - t0;
      printf("prover_batch\t%lf\t(%d statements, %s: %.1f proofs/s incl. PCIe and mpz_t conversion)\n", dt, nbatch,
             call == 0 ? "cold: the call expands the CRS image" : "warm: the image kept by the shim is streamed", nbatch / dt);
    }
    t0 = now();
    mfuoco_verifier_batch(ssp, vrs, pb, nbatch, okb);
    printf("verifier_batch\t%lf\t(%d proofs)\n", now() - t0, nbatch);
    for (int k = 0; k < nbatch; k++) ok = ok && okb[k] == 1;
    for (int k = 0; k < nbatch; k++) { proof_clear(pb[k]); mpz_clear(wit[k]); }
    free(pb); free(wit); free(okb);
  }

  /* benchmark_eval (src/benchmark_eval.c:30-86): D encryptions exported to a coeffs file, the file mapped read-only, ONE eval_poly over its D rows timed.
   * (Here the stream is rewound before the evaluation, so the result is also checked: it decrypts to sum_i coeff_i m_i mod p.) */
  if (do_eval) {
    char dir[] = "/tmp/mfuoco_eval_XXXXXX", path[64];
    if (!mkdtemp(dir)) { perror("mkdtemp"); return 1; }
    snprintf(path, sizeof path, "%s/coeffs", dir);
    rng_t erng;
    rng_init(erng, crs->seed);
    ct_t ect, evaluated;
    ct_init(ect);
    ct_init(evaluated);
    nmod_poly_t coeffs;
    nmod_poly_init(coeffs, GAMMA_P);
    uint8_t (*rows)[CT_BYTES] = malloc((size_t)GAMMA_D * CT_BYTES);
    mpz_t em;
    mpz_init(em);
    unsigned __int128 expect = 0;
    t0 = now();
    for (size_t i = 0; i != GAMMA_D; i++) {
      uint64_t r[2];
      getrandom(r, 16, 0);
      nmod_poly_set_coeff_ui(coeffs, i, r[0] % GAMMA_P);
      mpz_set_ui(em, r[1] % GAMMA_P);
      regev_encrypt(ect, erng, vrs->sk, em);
      ct_export(rows[i], ect);
      expect = (expect + (unsigned __int128)(r[0] % GAMMA_P) * (r[1] % GAMMA_P)) % GAMMA_P;
    }
    fprintf(stderr, "eval: %d encryptions, one at a time\t%lf\n", (int)GAMMA_D, now() - t0);
    if (mfuoco_rows_save(path, rows, GAMMA_D)) { perror("coeffs"); return 1; }
    free(rows);
    size_t nrows = 0;
    uint8_t (*c8)[CT_BYTES] = mfuoco_rows_map(path, &nrows);
    if (!c8 || nrows != GAMMA_D) { perror("coeffs map"); return 1; }
    rng_seek(erng, 0);
    t0 = now();
    eval_poly(evaluated, erng, c8, coeffs, GAMMA_D);
    printf("eval\t%lf\n", now() - t0);
    regev_decrypt(em, vrs->sk, evaluated);
    ok = ok && !mpz_cmp_ui(em, (unsigned long)expect);
    mfuoco_rows_unmap(c8, nrows);
    unlink(path);
    rmdir(dir);
    mpz_clear(em);
    nmod_poly_clear(coeffs);
    ct_clear(ect);
    ct_clear(evaluated);
    rng_clear(erng);
  }

  /* benchmark_lwe's loop: one encryption and one decryption at a time */
  rng_t rng;
  rng_init(rng, crs->seed);
  ct_t c;
  ct_init(c);
  mpz_t m, m2;
  mpz_inits(m, m2, NULL);
  double te = 0, td = 0;
  for (int i = 0; i < nenc; i++) {
    uint64_t r;
    getrandom(&r, 8, 0);
    mpz_set_ui(m, r % GAMMA_P);
    t0 = now();
    regev_encrypt2(c, rng, vrs->sk, m, errdist_uniform);
    te += now() - t0;
    t0 = now();
    regev_decrypt(m2, vrs->sk, c);
    td += now() - t0;
    ok = ok && !mpz_cmp(m, m2);
  }
  if (nenc) printf("encryption\t%lf\ndecryption\t%lf\n", te / nenc, td / nenc);

  /* the same loop as ONE call per direction: nencb messages under one key (errors drawn from the OS in bulk, rows at consecutive stream positions), then the
   * seed-compressed rows decrypted again; three calls, the first one allocates the staging */
  if (nencb > 0) {
    mpz_t *bm = malloc((size_t)nencb * sizeof *bm), *bd = malloc((size_t)nencb * sizeof *bd);
    uint8_t (*bc)[CT_BYTES] = malloc((size_t)nencb * CT_BYTES);
    for (int i = 0; i < nencb; i++) {
      uint64_t r;
      getrandom(&r, 8, 0);
      mpz_init_set_ui(bm[i], r % GAMMA_P);
      mpz_init(bd[i]);
    }
    for (int call = 0; call < 3; call++) {
      rng_seek(rng, 0);
      t0 = now();
      mfuoco_encrypt_batch(bc, rng, vrs->sk, bm, nencb);
      double dt = now() - t0;
      printf("encryption_batch\t%lf\t(%d messages, %s: %.0f enc/s incl. OS entropy, PCIe and the export)\n", dt, nencb, call ? "warm" : "first call", nencb / dt);
    }
    rng_seek(rng, 0);
    t0 = now();
    mfuoco_decrypt_rows_batch(bd, rng, vrs->sk, bc, nencb);
    double dt = now() - t0;
    printf("decryption_rows_batch\t%lf\t(%d seed-compressed ciphertexts: %.0f dec/s incl. PCIe)\n", dt, nencb, nencb / dt);
    for (int i = 0; i < nencb; i++) { ok = ok && !mpz_cmp(bm[i], bd[i]); mpz_clear(bm[i]); mpz_clear(bd[i]); }
    free(bm); free(bd); free(bc);
  }
  fprintf(stderr, "%s\n", ok ? "all proofs verified, all decryptions correct" : "FAILURE");
  return ok ? 0 : 1;
}
