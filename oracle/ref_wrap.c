/*
 * ref_wrap.c -- plain-pointer entry points around the REAL reference AES/entropy layer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with the reference's
 * own src/aes.c and src/entropy.c *where they lie* under /root/reference (never copied) into
 * oracle/_ref/libmfref.so by oracle/Makefile (target `ref`).  It exists so that Python tests and
 * tests/golden/make_golden.py can drive the reference's rng_init / rng_seek / rng_gen /
 * mpz2_urandomb (src/entropy.h:38-60) without marshalling mpz_t through ctypes.
 *
 * lwe.c / ssp.c / snark.c are NOT part of this build: they include <flint/nmod_poly.h>, FLINT is
 * absent from the image, and no stand-in is written for it.
 */
#include "config.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <gmp.h>

#include "entropy.h"

/* bytes [off, off+n) of the stream defined by seed, via rng_init + rng_seek + rng_gen */
void ref_keystream(uint8_t *seed, uint64_t off, void *out, size_t n)
{
  rng_t rng;
  rng_init(rng, seed);
  rng_seek(rng, off);
  rng_gen(rng, out, n);
  rng_clear(rng);
}

/* a stateful sequence of reads of the given sizes from offset `off`; outputs concatenated */
void ref_gen_sequence(uint8_t *seed, uint64_t off, const uint32_t *sizes, size_t k, uint8_t *out)
{
  rng_t rng;
  rng_init(rng, seed);
  rng_seek(rng, off);
  for (size_t i = 0; i < k; i++) {
    rng_gen(rng, out, sizes[i]);
    out += sizes[i];
  }
  rng_clear(rng);
}

/* `count` consecutive mpz2_urandomb(nbits) draws from offset `off`; each written as
 * ceil(nbits/64) little-endian limbs (zero extended).  The mpz is pre-cleared to zero limbs so the
 * reference's stale-heap bits (nbits % 8 != 0, src/entropy.c:11-26) read as zero, like the oracle. */
void ref_urandomb(uint8_t *seed, uint64_t off, size_t nbits, size_t count, uint64_t *out)
{
  rng_t rng;
  rng_init(rng, seed);
  rng_seek(rng, off);
  size_t limbs = (nbits + 63) / 64;
  mpz_t a;
  mpz_init2(a, nbits + 64);
  for (size_t i = 0; i < count; i++) {
    memset(a->_mp_d, 0, (size_t)a->_mp_alloc * sizeof(mp_limb_t));
    mpz2_urandomb(a, rng, nbits);
    memset(out, 0, limbs * 8);
    size_t sz = (size_t)(a->_mp_size < 0 ? -a->_mp_size : a->_mp_size);
    memcpy(out, a->_mp_d, sz * 8);
    out += limbs;
  }
  mpz_clear(a);
  rng_clear(rng);
}

/* exposes the reference's stream position after a sequence of reads: ctr and rem (src/aes.h:21-30) */
void ref_state_after(uint8_t *seed, uint64_t off, const uint32_t *sizes, size_t k, uint64_t *ctr, uint64_t *rem)
{
  rng_t rng;
  rng_init(rng, seed);
  rng_seek(rng, off);
  uint8_t *sink = malloc(1 << 20);
  for (size_t i = 0; i < k; i++) rng_gen(rng, sink, sizes[i]);
  *ctr = CTR(rng);
  *rem = REM(rng);
  free(sink);
  rng_clear(rng);
}

/* time `n` bytes of the reference keystream generator (for the cpu_baseline "reference" leg) */
uint64_t ref_bench_keystream(uint8_t *seed, size_t n, size_t chunk)
{
  rng_t rng;
  rng_init(rng, seed);
  uint8_t *buf = malloc(chunk);
  uint64_t x = 0;
  for (size_t done = 0; done < n; done += chunk) {
    rng_gen(rng, buf, chunk);
    x ^= buf[0] | ((uint64_t)buf[chunk - 1] << 8);
  }
  free(buf);
  rng_clear(rng);
  return x;
}
