/*
 * flint/nmod_poly.h -- the part of FLINT's nmod_poly interface that the mangiafuoco API exposes, for hosts on
 * which FLINT is not installed.  The struct layout is FLINT's (coeffs, alloc, length, mod{n, ninv, norm}) so a caller
 * built against the real FLINT can hand its nmod_poly_t to libmfuoco_gpu unchanged; with real FLINT on the include
 * path this file is simply not picked up.
 *
 * Only the host-side harness pieces live here (dense schoolbook, fine for SSP generation and tests).  The prover's
 * polynomial step h = (v^2-1)/t does NOT use this: it runs on the GPU (csrc/poly.hip).
 */
#ifndef MFUOCO_FLINT_NMOD_POLY_H
#define MFUOCO_FLINT_NMOD_POLY_H

#include <gmp.h>
#include <stdlib.h>
#include <string.h>

typedef long slong;
typedef unsigned long ulong;
typedef struct { mp_limb_t n; mp_limb_t ninv; unsigned long norm; } nmod_t;
typedef struct { mp_ptr coeffs; slong alloc; slong length; nmod_t mod; } nmod_poly_struct;
typedef nmod_poly_struct nmod_poly_t[1];

static inline void nmod_poly_init(nmod_poly_t p, mp_limb_t n) { p->coeffs = NULL; p->alloc = 0; p->length = 0; p->mod.n = n; p->mod.ninv = 0; p->mod.norm = 0; }
static inline void nmod_poly_clear(nmod_poly_t p) { free(p->coeffs); p->coeffs = NULL; p->alloc = p->length = 0; }
static inline mp_limb_t nmod_poly_modulus(const nmod_poly_t p) { return p->mod.n; }
static inline slong nmod_poly_degree(const nmod_poly_t p) { return p->length - 1; }
static inline void mf_nmod_poly_fit(nmod_poly_t p, slong len) {
  if (len > p->alloc) {
    slong na = len > 2 * p->alloc ? len : 2 * p->alloc;
    p->coeffs = (mp_ptr)realloc(p->coeffs, (size_t)na * sizeof(mp_limb_t));
    memset(p->coeffs + p->alloc, 0, (size_t)(na - p->alloc) * sizeof(mp_limb_t));
    p->alloc = na;
  }
}
static inline void mf_nmod_poly_normalise(nmod_poly_t p) { while (p->length > 0 && p->coeffs[p->length - 1] == 0) p->length--; }
static inline mp_limb_t nmod_poly_get_coeff_ui(const nmod_poly_t p, slong i) { return i < p->length ? p->coeffs[i] : 0; }
static inline void nmod_poly_set_coeff_ui(nmod_poly_t p, slong i, mp_limb_t c) {
  c %= p->mod.n; /* FLINT reduces */
  mf_nmod_poly_fit(p, i + 1);
  if (i >= p->length) { memset(p->coeffs + p->length, 0, (size_t)(i + 1 - p->length) * sizeof(mp_limb_t)); p->length = i + 1; }
  p->coeffs[i] = c;
  mf_nmod_poly_normalise(p);
}
static inline void nmod_poly_set(nmod_poly_t r, const nmod_poly_t a) {
  if (r == a) return;
  mf_nmod_poly_fit(r, a->length);
  memcpy(r->coeffs, a->coeffs, (size_t)a->length * sizeof(mp_limb_t));
  r->length = a->length;
}
static inline void nmod_poly_add(nmod_poly_t r, const nmod_poly_t a, const nmod_poly_t b) {
  slong la = a->length, lb = b->length, l = la > lb ? la : lb;
  mf_nmod_poly_fit(r, l);
  for (slong i = 0; i < l; i++) {
    mp_limb_t x = i < la ? a->coeffs[i] : 0, y = i < lb ? b->coeffs[i] : 0, s = x + y;
    r->coeffs[i] = s >= r->mod.n ? s - r->mod.n : s;
  }
  r->length = l;
  mf_nmod_poly_normalise(r);
}
static inline void nmod_poly_sub(nmod_poly_t r, const nmod_poly_t a, const nmod_poly_t b) {
  slong la = a->length, lb = b->length, l = la > lb ? la : lb;
  mf_nmod_poly_fit(r, l);
  for (slong i = 0; i < l; i++) {
    mp_limb_t x = i < la ? a->coeffs[i] : 0, y = i < lb ? b->coeffs[i] : 0;
    r->coeffs[i] = x >= y ? x - y : x + r->mod.n - y;
  }
  r->length = l;
  mf_nmod_poly_normalise(r);
}
static inline void nmod_poly_scalar_mul_nmod(nmod_poly_t r, const nmod_poly_t a, mp_limb_t c) {
  mf_nmod_poly_fit(r, a->length);
  for (slong i = 0; i < a->length; i++) r->coeffs[i] = (mp_limb_t)(((unsigned __int128)a->coeffs[i] * c) % r->mod.n);
  r->length = a->length;
  mf_nmod_poly_normalise(r);
}
static inline mp_limb_t nmod_poly_evaluate_nmod(const nmod_poly_t p, mp_limb_t x) {
  mp_limb_t r = 0;
  for (slong i = p->length - 1; i >= 0; i--) r = (mp_limb_t)((((unsigned __int128)r * x) + p->coeffs[i]) % p->mod.n);
  return r;
}

/* schoolbook product / Euclidean division, harness sizes only (r may alias the inputs) */
static inline void nmod_poly_mul(nmod_poly_t r, const nmod_poly_t a, const nmod_poly_t b) {
  slong la = a->length, lb = b->length;
  if (!la || !lb) { r->length = 0; return; }
  mp_ptr t = (mp_ptr)calloc((size_t)(la + lb - 1), sizeof(mp_limb_t));
  for (slong i = 0; i < la; i++)
    for (slong j = 0; j < lb; j++)
      t[i + j] = (mp_limb_t)((t[i + j] + (unsigned __int128)a->coeffs[i] * b->coeffs[j]) % r->mod.n);
  mf_nmod_poly_fit(r, la + lb - 1);
  memcpy(r->coeffs, t, (size_t)(la + lb - 1) * sizeof(mp_limb_t));
  r->length = la + lb - 1;
  free(t);
  mf_nmod_poly_normalise(r);
}
static inline void nmod_poly_pow(nmod_poly_t r, const nmod_poly_t a, ulong e) {
  nmod_poly_t acc, base;
  nmod_poly_init(acc, a->mod.n);
  nmod_poly_init(base, a->mod.n);
  nmod_poly_set(base, a);
  nmod_poly_set_coeff_ui(acc, 0, 1);
  for (; e; e >>= 1) {
    if (e & 1) nmod_poly_mul(acc, acc, base);
    if (e > 1) nmod_poly_mul(base, base, base);
  }
  nmod_poly_set(r, acc);
  nmod_poly_clear(acc);
  nmod_poly_clear(base);
}
static inline mp_limb_t mf_nmod_inv(mp_limb_t a, mp_limb_t n) { /* n prime */
  mp_limb_t r = 1, e = n - 2;
  for (a %= n; e; e >>= 1) { if (e & 1) r = (mp_limb_t)((unsigned __int128)r * a % n); a = (mp_limb_t)((unsigned __int128)a * a % n); }
  return r;
}
static inline void mf_nmod_poly_divrem(nmod_poly_t q, nmod_poly_t rem, const nmod_poly_t a, const nmod_poly_t b) {
  slong la = a->length, lb = b->length;
  mp_limb_t n = a->mod.n;
  mp_ptr w = (mp_ptr)malloc((size_t)(la ? la : 1) * sizeof(mp_limb_t));
  memcpy(w, a->coeffs, (size_t)la * sizeof(mp_limb_t));
  slong lq = la >= lb ? la - lb + 1 : 0;
  mp_ptr qq = (mp_ptr)calloc((size_t)(lq ? lq : 1), sizeof(mp_limb_t));
  if (lb > 0) {
    mp_limb_t linv = mf_nmod_inv(b->coeffs[lb - 1], n);
    for (slong i = la - 1; i >= lb - 1; i--) {
      mp_limb_t c = (mp_limb_t)((unsigned __int128)w[i] * linv % n);
      qq[i - (lb - 1)] = c;
      if (c) for (slong j = 0; j < lb; j++) w[i - (lb - 1) + j] = (mp_limb_t)((w[i - (lb - 1) + j] + (unsigned __int128)(n - c) * b->coeffs[j]) % n);
    }
  }
  if (q) { mf_nmod_poly_fit(q, lq); memcpy(q->coeffs, qq, (size_t)lq * sizeof(mp_limb_t)); q->length = lq; mf_nmod_poly_normalise(q); }
  if (rem) { slong lr = lb > 0 && la >= lb ? lb - 1 : la; mf_nmod_poly_fit(rem, lr); memcpy(rem->coeffs, w, (size_t)lr * sizeof(mp_limb_t)); rem->length = lr; mf_nmod_poly_normalise(rem); }
  free(w); free(qq);
}
static inline void nmod_poly_div(nmod_poly_t q, const nmod_poly_t a, const nmod_poly_t b) { mf_nmod_poly_divrem(q, NULL, a, b); }
static inline void nmod_poly_rem(nmod_poly_t r, const nmod_poly_t a, const nmod_poly_t b) { mf_nmod_poly_divrem(NULL, r, a, b); }

#endif
