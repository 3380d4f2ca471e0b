/*
 * bench_snark.c -- the measurement the reference's benchmark_snark / benchmark_lwe programs make (wall clock around
 * setup(), prover(), verifier(), regev_encrypt(), regev_decrypt(); `label\tseconds` lines on stdout), written against
 * the same function names, for libmfuoco_gpu.  Our own text; the reference's drivers also link unchanged
 * (tests/test_link_compat.py) but their sources do not travel to the GPU box.
 *
 * This is the DROP-IN path: every call crosses PCIe and converts mpz_t, so it measures the PCIe/host-inclusive cost of
 * using the GPU through the reference API, not the kernel throughput bench.py reports.
 *   usage: bench_snark [nproofs] [nenc]
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
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
  int nproofs = argc > 1 ? atoi(argv[1]) : 3, nenc = argc > 2 ? atoi(argv[2]) : 200;
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
  fprintf(stderr, "%s\n", ok ? "all proofs verified, all decryptions correct" : "FAILURE");
  return ok ? 0 : 1;
}
