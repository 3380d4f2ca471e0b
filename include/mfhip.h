/*
 * mfhip.h -- C ABI of the MI355X-native LWE/SSP-SNARK prover core (libmfhip.so).
 *
 * This is the drop-in boundary for the hot path of mmaker/c-lwe-snarks ("mangiafuoco"):
 * every entry point is `extern "C"`, takes plain pointers and sizes, and cites the reference
 * interface it replaces (file:line relative to the reference tree).  The reference API is
 * one-ciphertext-at-a-time over GMP `mpz_t`; these are the batched, dense-limb equivalents that
 * the reference-signature shim (c-lwe-snarks_amd/host/) calls underneath.
 *
 * Data conventions
 *   value      : L = ceil(logq/64) little-endian uint64 limbs (12 @ logq=736, 23 @ 1472).  Results are
 *                reduced the way the reference's modq() reduces (src/lwe.h:107-118): only the low
 *                K = logq/64 limbs survive (effective modulus 2^704 @ 736, 2^1472 @ 1472).
 *   ciphertext : (n+1) values, coordinate-major: ct[j*L .. j*L+L), j = n is `b`   (ct_t, src/lwe.h:44)
 *   c8         : CT_BYTES = logq/8 little-endian bytes of `b` per ciphertext       (ct_export, src/lwe.c:115-119)
 *   seed       : 40 bytes: [0,8) nonce, [8,40) AES-256 key                         (rseed_t, src/entropy.h:35)
 *   stream     : byte x of the public stream = byte (x&15) of AES256_K(nonce_le64 || le64(x>>4))
 *                (src/aes.c:104-144, src/entropy.c:46-61)
 *   Pointers named d_* are DEVICE pointers (HBM); h_* are host pointers.  All launches go to the
 *   context's HIP stream and are asynchronous unless stated; mfh_sync() waits.
 *
 * Threading: like the reference (no function of which is re-entrant w.r.t. a shared rng_t), a context is NOT thread-safe; use
 * one context per thread / per GPU.  Different contexts are independent.
 *
 * Error behaviour: the reference API is void and asserts preconditions (debug builds only).  Here every
 * function returns 0 on success or a negative MFH_E* code; mfh_last_error() gives the text.  There is no
 * CPU fallback anywhere in the library: without a usable HIP device mfh_ctx_create fails.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFH_P 0xfffffffbu /* GAMMA_P, src/lwe.h:25 */

enum {
  MFH_OK = 0,
  MFH_EINVAL = -1,  /* bad argument (the reference would assert or misbehave) */
  MFH_EDEVICE = -2, /* HIP runtime error / no device */
  MFH_ENOMEM = -3,
  MFH_EUNSUPPORTED = -4 /* parameter set not compiled in (logq must be 736 or 1472) */
};

typedef struct {
  uint32_t n;    /* GAMMA_N,    src/lwe.h:23 (1470) */
  uint32_t logq; /* GAMMA_LOGQ, src/lwe.h:24 (736; 1472 for the doubled-modulus stress config) */
  uint32_t d;    /* GAMMA_D,    src/lwe.h:15,19 */
  uint32_t m;    /* GAMMA_M,    src/lwe.h:17,20 */
} mfh_params;

typedef struct mfh_ctx mfh_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
int mfh_ctx_create(mfh_ctx **out, int device, const mfh_params *P);
void mfh_ctx_destroy(mfh_ctx *ctx);
/* run on an existing hipStream_t (e.g. torch's current stream).  NULL means HIP's default (null) stream,
 * NOT the context's own stream (a fresh context runs on its own non-blocking stream until this is called). */
int mfh_set_stream(mfh_ctx *ctx, void *hip_stream);
int mfh_sync(mfh_ctx *ctx);
/* The prover entry points stage what the host contributes -- witness bits, deltas (src/snark.c:140) and the smudging terms u p (src/lwe.c:65-76) -- in pinned host
 * buffers the context keeps.  This waits for their copies and zeroes them: a caller whose proofs must stay zero-knowledge against a later reader of its memory calls
 * it when the proofs are out (the shim's prover() / mfuoco_prover_batch() do; the reference, single-threaded on the CPU, leaves delta and u in freed mpz limbs). */
int mfh_scrub_staging(mfh_ctx *ctx);
const char *mfh_last_error(const mfh_ctx *ctx);
/* rng_init (src/entropy.c:58-61) + aesctr_init (src/aes.c:49-95): expand the AES-256 key of `seed`
 * and make it the context's current public stream. */
int mfh_set_seed(mfh_ctx *ctx, const uint8_t h_seed[40]);

/* ---- L0/L1: keystream and sampler ---------------------------------------------------------------- */
/* rng_seek + rng_gen / aesctr_prg (src/entropy.c:46-56, src/aes.c:104-144): bytes [off, off+nbytes)
 * of the current stream into d_out. */
int mfh_keystream(mfh_ctx *ctx, uint64_t off, void *d_out, size_t nbytes);
/* mpz2_urandommv(c, rs, GAMMA_LOGQ, GAMMA_N) for `nrows` consecutive ciphertext rows starting at stream
 * offset `off` (src/entropy.h:62-66, src/lwe.c:90,101,124): d_out[row][j][L] limbs, j < n, masked to logq bits. */
int mfh_sample_rows(mfh_ctx *ctx, uint64_t off, size_t nrows, uint64_t *d_out);

/* ---- L2: ciphertext algebra ---------------------------------------------------------------------- */
/* ct_add / ct_mul_ui / ct_addmul_ui on `count` ciphertexts (src/lwe.c:131-157).  In-place allowed. */
int mfh_ct_add(mfh_ctx *ctx, uint64_t *d_rop, const uint64_t *d_a, const uint64_t *d_b, size_t count);
int mfh_ct_mul_ui(mfh_ctx *ctx, uint64_t *d_rop, const uint64_t *d_a, uint32_t b, size_t count);
int mfh_ct_addmul_ui(mfh_ctx *ctx, uint64_t *d_rop, const uint64_t *d_a, uint32_t b, size_t count);

/* eval_poly (src/lwe.c:176-186), fused over one or two coefficient vectors:
 *   rop_k (+)= sum_{i<nrows} coeff_k[i] * Import(c8[i])   with the stream pre-seeked to `off`
 * i.e. row i is expanded from stream bytes [off + i*n*CT_BYTES, ...) exactly as ct_import does
 * (src/lwe.c:122-126), once, and multiply-accumulated into both outputs.  Rows whose coefficients are
 * all zero are not expanded (the reference expands them only to advance its stream).
 * d_coeff1/d_rop1 may be NULL.  accumulate != 0 keeps the previous contents of d_rop* (the reference
 * always accumulates into rop); 0 overwrites.  Coefficients must be < p (the reference asserts). */
int mfh_eval_rows(mfh_ctx *ctx, uint64_t off, size_t nrows, const uint8_t *d_c8, const uint32_t *d_coeff0,
                  const uint32_t *d_coeff1, uint64_t *d_rop0, uint64_t *d_rop1, int accumulate);

/* ---- resident (materialised) CRS: SURVEY 8(d)'s second regime ---------------------------------------------------
 * The reference regenerates the a-vectors of every CRS row from the seed on every use (ct_import, src/lwe.c:122-126).
 * With 288 GB of HBM they can instead be expanded ONCE (11.3 GB for the default CRS) and streamed at HBM speed:
 * mfh_crs_expand writes rows [off/CTR_CT ...) in a limb-plane layout (mfh_resident_row_bytes() per row: only the words that
 * survive modq, coordinate n = the row's b); mfh_eval_rows_resident is eval_poly over rows [first_row, first_row+nrows)
 * of such an image (coefficient i belongs to row first_row + i).  mfh_crs_set_resident(ctx, image) makes
 * mfh_prove / mfh_prove_partial use the image (row index = position in stream order from CTR_S); NULL reverts. */
size_t mfh_resident_row_bytes(const mfh_ctx *ctx);
int mfh_crs_expand(mfh_ctx *ctx, uint64_t off, size_t nrows, const uint8_t *d_c8, void *d_rows_out);
/* d_c8 may be NULL in mfh_crs_expand: the a parts only (they depend on the seed alone, not on the ciphertexts), coordinate n left zero -- so that the 12 ms of AES can run
 * BEFORE the b's exist (the shim's setup() queues it beside the SSP upload); mfh_crs_image_set_b then fills coordinate n of rows [first_row, first_row + nrows)
 * of the image from their compressed ciphertexts (d_c8[0] = row first_row).  Expanding with d_c8 gives the same bytes as expanding without and setting b afterwards. */
int mfh_crs_image_set_b(mfh_ctx *ctx, size_t first_row, size_t nrows, const uint8_t *d_c8, void *d_rows);
int mfh_eval_rows_resident(mfh_ctx *ctx, const void *d_rows, size_t first_row, size_t nrows, const uint32_t *d_coeff0,
                           const uint32_t *d_coeff1, uint64_t *d_rop0, uint64_t *d_rop1, int accumulate);
int mfh_crs_set_resident(mfh_ctx *ctx, const void *d_rows);
/* Partial residency for a CRS whose expansion exceeds HBM (362 GB for the 2^20-constraint CRS): only stream rows
 * [0, nrows_resident) are in the image; mfh_prove* streams those and regenerates the remaining rows from the seed. */
int mfh_crs_set_resident_prefix(mfh_ctx *ctx, const void *d_rows, size_t nrows_resident);
/* Multi-GPU form ("GPU g ... keeps its slice of the expanded CRS resident", 45 GB per GPU for the 2^20-constraint
 * CRS on 8 GPUs): the image holds only rank `rank`'s contiguous shares, in the order S share | AS share | BT+BV share
 * (mfh_resident_share_rows() rows of mfh_resident_row_bytes() each).  mfh_prove_partial* with the same (rank, world) then streams it. */
size_t mfh_resident_share_rows(const mfh_ctx *ctx, uint32_t rank, uint32_t world);
int mfh_crs_expand_share(mfh_ctx *ctx, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, void *d_image);
int mfh_crs_set_resident_share(mfh_ctx *ctx, const void *d_image, uint32_t rank, uint32_t world);

/* Batched regev_encrypt2 + ct_export (src/lwe.c:78-97,115-119): for i < nrows
 *   b_i = (e_i*p + <sk, a_i> + m_i) mod 2^(64K),  a_i = row at stream offset off + i*n*CT_BYTES
 * d_sk: n values; d_msg: nrows uint32 (< p); d_err: nrows values (the sampled error e, any L-limb value;
 * the reference draws 559 bits from getrandom, src/lwe.c:60-63); d_c8_out: nrows*CT_BYTES bytes. */
int mfh_encrypt_rows(mfh_ctx *ctx, uint64_t off, size_t nrows, const uint64_t *d_sk, const uint32_t *d_msg,
                     const uint64_t *d_err, uint8_t *d_c8_out);
/* Which kernel mfh_encrypt_rows (hence mfh_setup) uses; the results are identical.  0 (default): batches of 4096 rows or more run <sk, a> on
 * the matrix cores -- the dot product as a (rows x keystream bytes) x Toeplitz(sk) int8 GEMM, each lane's AES output block being an MFMA
 * operand as it stands -- when off and n * CT_BYTES are multiples of 8, anything else the VALU kernel; 1: always the VALU kernel;
 * 2: always the matrix-core kernel (MFH_EINVAL from mfh_encrypt_rows if the alignment does not allow it). */
int mfh_set_encrypt_path(mfh_ctx *ctx, int path);

/* Batched regev_decrypt (src/lwe.c:105-111) of `count` explicit ciphertexts: d_out[i] = (b - <a,sk> mod 2^(64K)) mod p */
int mfh_decrypt(mfh_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_cts, size_t count, uint32_t *d_out);
/* mfh_decrypt: 0 (default) = by batch size (from 4096 ciphertexts <a, sk> runs as a Toeplitz int8 GEMM on the matrix cores, HBM-bound), 1 = VALU kernel,
 * 2 = matrix-core kernel.  Same results. */
int mfh_set_decrypt_path(mfh_ctx *ctx, int path);
/* regev_decrypt (src/lwe.c:105-111) of nrows SEED-COMPRESSED ciphertexts -- the form ct_export / the CRS hold (src/lwe.c:115-126): row i's a part is
 * regenerated from the public stream at off + i * n * CT_BYTES (ct_import), d_c8 holds its CT_BYTES little-endian b.  d_out[i] = (b - <a, sk>) mod p.
 * off and n * CT_BYTES must be multiples of 8 (MFH_EUNSUPPORTED otherwise).  AES-bound like mfh_encrypt_rows. */
int mfh_decrypt_rows(mfh_ctx *ctx, uint64_t off, size_t nrows, const uint64_t *d_sk, const uint8_t *d_c8, uint32_t *d_out);

/* mpz_add_dotp (src/lwe.c:20-28): rop = (rop + sum_{j<len} a[j]*b[j]) mod 2^(64K); rop is one value, a and b are len values */
int mfh_add_dotp(mfh_ctx *ctx, uint64_t *d_rop, const uint64_t *d_a, const uint64_t *d_b, size_t len);

/* ct_smudge (src/lwe.c:65-76) with caller-supplied entropy: b += (+/-) u*p on coordinate n of each of `count`
 * ciphertexts; h_mag = count * maglen little-endian bytes of u, h_sign[i]&1 selects the minus sign. */
int mfh_ct_smudge(mfh_ctx *ctx, uint64_t *d_cts, size_t count, const uint8_t *h_mag, size_t maglen, const uint8_t *h_sign);

/* ---- L3: SSP witness polynomial ------------------------------------------------------------------- */
/* Device SSP layout: uint32 d_ssp[(m+3)][d], slot 0 = t, slot i+1 = v_i, coefficients reduced mod p
 * (reference host layout: uint64 little-endian, src/ssp.h:6-9; mfh_ssp_upload converts and reduces as
 * nmod_poly_import does, src/ssp.c:28-34). */
int mfh_ssp_upload(mfh_ctx *ctx, const void *h_ssp_u64, uint32_t *d_ssp, size_t first_slot, size_t nslots);
/* w(x) = delta*t(x) + sum_{i=1..m-1, witness bit i-1 set} v_i(x) mod p   (src/snark.c:141,147-155);
 * h_witness_bits: little-endian bit string, bit i-1 <-> v_i (mpz_tstbit).  d_w: d uint32. */
int mfh_witness_poly(mfh_ctx *ctx, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta, uint32_t *d_w);

/* Generator-defined SSP (BASELINE configs 4/5; SURVEY 8(d): "SSP coefficients defined by a counter-based PRG (not stored)").
 * The dense buffer of src/ssp.h:6-9 is 5.9 TB at 2^20 constraints; here v_i[k] = f(seed, slot i+1, k) (a 32-bit integer hash
 * reduced into [0,p), csrc/ssp_prg.hpp) and only slot 0 (t, d uint32) is stored.  mfh_ssp_prg_make_t builds
 * t = v_0 + sum_{witness bit} v_i - 1 as random_ssp does (src/ssp.c:59-71); mfh_ssp_set_prg(ctx, seed, d_t) registers the SSP,
 * after which EVERY entry point that takes `d_ssp` accepts NULL to mean "the registered generator-defined SSP"
 * (mfh_witness_poly, mfh_witness_lanes, mfh_ssp_prepare, mfh_setup_messages, mfh_setup, mfh_prove*, mfh_verify).
 * mfh_ssp_prg_fill materialises slots [first_slot, first_slot+nslots) (slots >= 1) as a dense uint32 image (tests). */
int mfh_ssp_set_prg(mfh_ctx *ctx, uint64_t seed, const uint32_t *d_t);
int mfh_ssp_prg_make_t(mfh_ctx *ctx, uint64_t seed, const uint8_t *h_witness_bits, uint32_t *d_t);
int mfh_ssp_prg_fill(mfh_ctx *ctx, uint64_t seed, size_t first_slot, size_t nslots, uint32_t *d_out);

/* ---- L3/L4: polynomial step, setup, prover ------------------------------------------------------------ */
/* c = a*b over F_p[x] (la+lb-1 canonical coefficients).  What nmod_poly_mul/pow compute (src/snark.c:167). */
int mfh_poly_mul(mfh_ctx *ctx, const uint32_t *d_a, size_t la, const uint32_t *d_b, size_t lb, uint32_t *d_c);
int mfh_poly_add(mfh_ctx *ctx, const uint32_t *d_a, const uint32_t *d_b, size_t count, uint32_t *d_out);
/* Per-SSP precomputation for the quotient by t(x): power-series inverse of rev(t) (and its transform).
 * d_t = d coefficients (slot 0 of the device SSP).  Fails (MFH_EINVAL) for t = 0, where nmod_poly_div raises. */
int mfh_poly_prepare_t(mfh_ctx *ctx, const uint32_t *d_t);
int mfh_ssp_prepare(mfh_ctx *ctx, const uint32_t *d_ssp); /* = mfh_poly_prepare_t(slot 0) */
/* h = floor((v^2 - 1) / t), first d coefficients (nmod_poly_pow/sub/div, src/snark.c:166-169).  v: d coefficients. */
int mfh_poly_h(mfh_ctx *ctx, const uint32_t *d_v, uint32_t *d_h);
/* the same for nb polynomials side by side (v_k at d_v + k d, h_k at d_h + k d): one set of launches for the whole batch */
int mfh_poly_h_multi(mfh_ctx *ctx, const uint32_t *d_v, uint32_t *d_h, uint32_t nb);
/* A batch (nb >= 4) takes the exact-division path when the prepared t has degree d - 1 and is a unit modulo x^N - 1 (N = the power of two >= d): a prover with a valid
 * witness divides exactly (src/ssp.c:37-77), and then h = (v^2 - 1 mod x^N - 1) t^-1 mod x^N - 1 -- two cyclic products of length N instead of two linear ones of
 * length 2N.  Every result is CHECKED on the device (h(r) t(r) = v(r)^2 - 1 at four points); the statements that fail -- a witness that does not satisfy the SSP
 * -- are recomputed by the Euclidean division above, by kernels queued behind the check (no host round trip; a batch with k such statements pays that path for k):
 * the output is nmod_poly_div's in every case.  mode 1 (the default): as described, except that after a batch in which the check has failed the next 64 batches take
 * the Euclidean path alone before the exact path is tried again (a caller whose statements do not satisfy the SSP would pay both; the failure is noticed through a
 * word in host memory whenever the device gets there -- a hint, nothing is waited for).  mode 2: every batch tries the exact path (A/B, tests).  mode 0: never.
 * Setting a mode forgets what earlier batches have taught. */
int mfh_set_poly_exact(mfh_ctx *ctx, int mode);
/* statements that failed that check (and were recomputed) since the last call; waits for the stream.  -1: the prepared t has no exact-division path */
long mfh_poly_exact_fallbacks(mfh_ctx *ctx);

/* The 2d+m plaintexts setup() encrypts, in stream order: s^i | alpha s^i | beta t(s) | beta v_i(s), i=1..m-1
 * (src/snark.c:73-110; the Horner values nmod_poly_evaluate_nmod are computed as dot products with the powers of s). */
int mfh_setup_messages(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, uint32_t *d_msg);
/* setup() (src/snark.c:57-115) with caller-supplied secrets: d_sk = n values (key_gen, src/lwe.c:30-34),
 * d_err = (2d+m) error values in encryption order (errdist_uniform, src/lwe.c:60-63).  The public seed is the one
 * given to mfh_set_seed (crs->seed).  Output: the device CRS, (2d+m)*CT_BYTES bytes in stream order
 * s[0..d) | as[0..d) | t | v[0..m-1)  (struct crs, src/snark.h:27-33, keeps them in four arrays). */
int mfh_setup(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, const uint64_t *d_sk,
              const uint64_t *d_err, uint8_t *d_crs_c8);
/* The same, leaving the expanded rows behind as a by-product ("writing the expanded rows to HBM as a by-product, so the prover starts with a materialised
 * CRS"): d_rows_image (may be NULL: then this is mfh_setup) receives all 2d+m rows in the layout of mfh_crs_expand -- (2d+m) x mfh_resident_row_bytes() bytes, what
 * mfh_crs_set_resident registers and mfh_prove streams -- so that the first proof under the new CRS does not regenerate a single a-vector.  The rows are written by a second
 * pass over the stream (the expansion kernel of mfh_crs_expand, queued behind the encryptions); the encryption kernel itself keeps the keystream in its MFMA operand
 * registers (csrc/encmm.hip) and would have to scatter it in 4-byte pieces to lay it out. */
int mfh_setup_image(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, const uint64_t *d_sk,
                    const uint64_t *d_err, uint8_t *d_crs_c8, void *d_rows_image);
/* prover() (src/snark.c:117-190) with caller-supplied entropy: delta (< p; the reference draws 8 bytes % p) and the
 * five smudging draws in call order h, hat_h, hat_v, v_w, v_w (sic: v_w twice, b_w never; src/snark.c:185-189):
 * h_smudge_mag = 5*maglen bytes, h_smudge_sign = 5 bytes.  Needs mfh_set_seed(crs seed) and mfh_ssp_prepare.
 * d_proof = 5 ciphertexts in struct order h | hat_h | hat_v | v_w | b_w (struct proof, src/snark.h:14-20).
 * Every CRS row is expanded once: S rows feed v_w and h, AS rows feed hat_v and hat_h (the reference expands each twice). */
int mfh_prove(mfh_ctx *ctx, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
              const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign, uint64_t *d_proof);

/* verifier() (src/snark.c:192-250) for `count` proofs (5 ciphertexts each) entirely on the device: t(s), v_0(s), the 5*count
 * decryptions and the eq-pke / eq-div / eq-lin checks.  d_ok[i] = 1 iff proof i is accepted.  (The reference's final
 * "test-error" bound never rejects: SIZ of a non-positive value is <= 0, src/snark.c:238-241.) */
int mfh_verify(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, const uint64_t *d_sk,
               const uint64_t *d_proofs, size_t count, uint8_t *d_ok);

/* The CRS expanded ONCE for the matrix-core path (the resident regime of the batch prover): mfh_crs_expand_mm writes the S, AS and
 * BT+BV regions in MFMA A-fragment order (mfh_crs_mm_image_bytes bytes: 11.3 GB at the default instance); while an image is
 * registered with mfh_crs_set_resident_mm (NULL clears it), mfh_eval_rows_multi over exactly one of those regions -- hence
 * mfh_prove_batch -- streams it from HBM instead of regenerating the keystream.  Results are identical. */
size_t mfh_crs_mm_image_bytes(const mfh_ctx *ctx);
int mfh_crs_expand_mm(mfh_ctx *ctx, const uint8_t *d_crs_c8, uint8_t *d_image);
int mfh_crs_set_resident_mm(mfh_ctx *ctx, const uint8_t *d_image);
/* Which kernel writes the image (same bytes for every row tile and row that exists; tuning / A-B): 0 (default) the barrier-free writer
 * (lane = row, 16 x 16 byte transposition on the matrix cores, csrc/expandmm.hip), 1 the LDS-tile writer (k_evalmm16<MODE 1>). */
int mfh_set_expand_path(mfh_ctx *ctx, int path);
/* prover() for nproofs statements under ONE CRS and SSP.  The S and AS regions are expanded (or streamed from the image, below) once per group of up to 31 proofs, the
 * BT+BV region once per up to 255 (the S / AS launches of the streaming regime: 255 proofs per two passes over a region), and the multiply-accumulate of the coefficient vectors runs on the matrix cores (mfh_eval_rows_multi); proof b is bit-identical to
 * mfh_prove(witness b, delta b, smudging b).  h_witness_bits: nproofs bit strings, bits_stride bytes apart; h_delta: nproofs values
 * < p; h_smudge_mag: nproofs x 5 x maglen bytes; h_smudge_sign: nproofs x 5 bytes; d_proofs: nproofs x 5 ciphertexts.
 * The single-proof resident image (mfh_crs_set_resident) must not be set; the matrix-core image (mfh_crs_set_resident_mm) may.
 * With more than 31 proofs and no image registered, the call first expands the CRS into a transient image of
 * mfh_crs_mm_image_bytes bytes (scratch kept by the context; expanded again by every call) and streams it for every group instead of
 * running AES once per group; if that scratch cannot be allocated, or after mfh_set_batch_image(ctx, 0) (which also frees it), every
 * group regenerates the keystream.  Same proofs either way.  The call only QUEUES its work (about 2.5 ms of host time per 1020 statements at the default instance, for calls of up to 8
 * super-groups = 2040 statements: the witness pass stages through a ring of 8 pinned buffers, and a longer call waits for the copy of super-group k - 8 before it stages k): it returns
 * long before the proofs exist -- mfh_sync, or mfh_prove_batch_stream_wait below, before d_proofs is read. */
int mfh_set_batch_image(mfh_ctx *ctx, int enabled);
/* Row slabs of mfh_prove_batch.  A call whose matrix-core image does not fit the GPU's memory (363 GB at 2^20 constraints) cuts the CRS
 * rows into slabs -- the row shares of the multi-GPU prover, one after the other on one GPU --, expands a slab's image once and streams it
 * for every group of the call, accumulating mod 2^(64K): the keystream is generated once per call instead of once per group of 31 proofs.
 * 0 (default): slabs only when needed, their number from the free memory; n: always n slabs (tests, tuning).  Same proofs. */
int mfh_set_batch_slabs(mfh_ctx *ctx, uint32_t nslabs);
/* tuning knobs (results do not depend on them; MFH_EINVAL outside the range): column chunks per row of the matrix-core encryption kernel
 * (0 = picked from the batch size, at most 64); statements per witness GEMM pass of the batch chain (0 = one pass per super-group of up to 255, else 32..256). */
int mfh_set_encrypt_chunks(mfh_ctx *ctx, uint32_t chunks);
/* mfh_eval_rows / mfh_prove*: 0 (default) = tile kernel (k_eval: two 512-coordinate row tiles per workgroup), 1 = at logq = 736 the wave-autonomous
 * kernel (k_eval_w: a wave owns 64 coordinates and a private keystream tile, no workgroup barriers; measured 4 % slower).  Same results (A/B, tests). */
int mfh_set_eval_path(mfh_ctx *ctx, int path);
int mfh_set_witness_per(mfh_ctx *ctx, uint32_t statements);
/* launch shape of the streaming regime of mfh_prove_batch* (tuning; results do not depend on it): groups of 31 proofs served by one pass
 * over a region's image (1..8, default 4), and whether the S and AS groups of a round share ONE launch (default) or run as two
 * launches on two streams. */
int mfh_set_batch_launch(mfh_ctx *ctx, uint32_t groups_per_launch, int merge_regions);
/* b_w (src/snark.c:143-155) of all super-groups of a streaming mfh_prove_batch call in ONE launch per up to 8 super-groups over the BT+BV image (default), or one
 * launch per super-group (0; rounds 1 - 3).  Tuning; results do not depend on it. */
int mfh_set_batch_bw(mfh_ctx *ctx, int merged);
/* how a streaming launch with several groups is laid out on the chip (tuning; results do not depend on it).  map: 0 = a tile group's workgroups for all
 * groups of both regions are neighbours, 1 = the 32 workgroups an XCD runs at a time are 32 / g tile groups x the g groups of ONE region (a group's digit
 * fragments are then shared by twice as many workgroups of the XCD).  persistent: 1 = one workgroup per CU looping over its XCD's items instead of one workgroup
 * per item (default), 2 = the same grid with the one-wave-per-SIMD body (k_mmstream_w: 256 accumulators per wave in AccVGPRs, half the LDS reads; measured the same time); sync_mode (persistent only): 0 = none, 1 = the workgroups that stream the same fragments begin every item together, 2 = all workgroups of an XCD
 * do -- a speed-only rendezvous bounded by spin_max polls (a workgroup never waits longer, so the grid drains whatever is resident). */
int mfh_set_mm_stream(mfh_ctx *ctx, int map, int persistent, int sync_mode, uint32_t spin_max);
/* Width of the persistent S / AS launch: workgroups per XCD (1..32; 32 = one per CU, the default).  Fewer leave CUs of every XCD free; with early_chain != 0
 * mfh_prove_batch then queues the chain of super-group k + 1 and the epilogue of super-group k on side streams beside the streaming launches instead of between them
 * (the round-5 experiment on the power finding, EXPERIMENTS.md).  Tuning; results do not depend on it. */
int mfh_set_mm_width(mfh_ctx *ctx, uint32_t per_xcd, int early_chain);
/* rows per row chunk of the matrix-core launches (mfh_eval_rows_multi, mfh_prove_batch): the int32 accumulators hold at most
 * 131071 rows (the default; 0 restores it); smaller values split every region into more chunks -- same results (tuning, tests). */
int mfh_set_mm_chunk_rows(mfh_ctx *ctx, uint32_t rows);
/* How the streaming kernels (k_mmstream*) hand the partial products sum_i A'[i][m] C'[i][n] to the epilogue: on (default) recombined in the kernel -- the four byte
 * positions a lane holds and, for four-byte coefficient vectors, the vector's four digit columns: a 16-byte record per (byte-position quad, vector) instead of sixteen
 * int32, a quarter of the bytes written and read back (2.1 GB per super-group launch otherwise) --; off: the int32 layout of rounds 1 - 5.  Exact integer arithmetic either
 * way: same results (tuning, tests). */
int mfh_set_mm_pack(mfh_ctx *ctx, int on);
int mfh_prove_batch(mfh_ctx *ctx, const uint8_t *d_crs_c8, const uint32_t *d_ssp, uint32_t nproofs, const uint8_t *h_witness_bits,
                    size_t bits_stride, const uint32_t *h_delta, const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign,
                    uint64_t *d_proofs);
/* Draining a call while it runs.  mfh_prove_batch queues its work super-group by super-group (mfh_prove_batch_supergroup() statements each: 255 when the
 * call streams an image, 248 when every group regenerates the keystream; 0 before the first call) and statement k's five ciphertexts are final in d_proofs
 * once super-group k / size has been smudged.  mfh_prove_batch_stream_wait makes `hip_stream` (a stream of the caller's, not the context's) wait until statements
 * [0, upto) of the LAST mfh_prove_batch call are final: a device-to-host copy queued on it afterwards runs under the following super-groups' kernels (the host shim's
 * mfuoco_prover_batch converts super-group k to mpz_t while the GPU works on k + 1).  Row-slab calls complete all statements together. */
uint32_t mfh_prove_batch_supergroup(const mfh_ctx *ctx);
int mfh_prove_batch_stream_wait(mfh_ctx *ctx, uint32_t upto, void *hip_stream);
/* ---- multi-GPU: row-sharded BATCH prover (BASELINE configs 3/4: "ciphertexts sharded across 8 x MI355X + RCCL reduce") ------
 * The loops that shard are src/snark.c:147-155 (b_w over the BT+BV rows) and :157-174 (the four eval_poly passes over the S / AS rows):
 * rank r of `world` owns rows [R r / world, R (r+1) / world) of each region (R = d, d, m) and, for the steps that do not touch the CRS,
 * a contiguous slice of the STATEMENTS.  One step over nstmt statements:
 *   1. mfh_batch_chain on the rank's own statements: w = delta t + sum_bits v_i, v = w + v_0, h = (v^2 - 1) / t   (src/snark.c:141-169)
 *   2. exchange (all-to-all): rank r receives rows [d r / world, ...) of w | h | v of ALL statements
 *   3. mfh_prove_batch_partial: the rank's row shares of all five ciphertexts of all statements (no delta ct_t term, un-smudged),
 *      streamed from the rank's share of the matrix-core image (mfh_crs_expand_mm_share: 45 GB per GPU for the 2^20-constraint CRS on 8)
 *   4. mfh_ct_to_lanes -> ONE reduce-scatter (sum) of uint64 lanes per step: every rank receives the summed lanes of its own statements
 *   5. mfh_ct_from_lanes (carries + modq) -> mfh_prove_batch_finish: + delta ct_t on b_w, smudging.
 * Proofs are bit-identical to mfh_prove_batch's (sums mod 2^(64K) do not depend on the order).  world = 1 degenerates to mfh_prove_batch.
 * c-lwe-snarks_amd/dist.py (prove_batch_sharded) drives the sequence over torch.distributed (backend nccl = RCCL); for a generator-defined
 * SSP it runs step 1 in two halves (mfh_batch_witness_cols / mfh_batch_chain_from_w below, one more all-to-all). */
size_t mfh_crs_mm_share_bytes(const mfh_ctx *ctx, uint32_t rank, uint32_t world);
int mfh_crs_expand_mm_share(mfh_ctx *ctx, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, uint8_t *d_image);
int mfh_crs_set_resident_mm_share(mfh_ctx *ctx, const uint8_t *d_image, uint32_t rank, uint32_t world);
/* step 1: d_w, d_h, d_v = nstmt x d coefficients each (statement-major) */
int mfh_batch_chain(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_witness_bits, size_t bits_stride, const uint32_t *h_delta,
                    uint32_t *d_w, uint32_t *d_h, uint32_t *d_v);
/* step 1 in two halves, for SSPs whose witness pass dominates the chain (generator-defined: 2^20 constraints): (1a) every rank computes
 * the coefficients [col0, col0 + ncols) of w of ALL nstmt statements -- 1 / world of the generation (or read) of the selected rows, no
 * reduction -- into d_w[b * w_stride + (k - col0)] (col0, ncols multiples of 128, else MFH_EUNSUPPORTED: use mfh_batch_chain);
 * the slices are exchanged (all-to-all by statement owner); (1b) the owner finishes the chain of its statements from their whole w:
 * d_v = d_w + v_0, d_h = (d_v^2 - 1) / t. */
int mfh_batch_witness_cols(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_witness_bits, size_t bits_stride, const uint32_t *h_delta,
                           uint32_t col0, uint32_t ncols, uint32_t *d_w, size_t w_stride);
int mfh_batch_chain_from_w(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t nstmt, const uint32_t *d_w, uint32_t *d_h, uint32_t *d_v);
/* step 3: statement b's coefficients for the rank's S / AS rows [d rank / world, d (rank+1) / world) at d_w / d_h / d_v + b * coef_stride
 * (uint32 words; with the whole polynomials in memory: d_w = W + d rank / world, coef_stride = d).  The witness bits select the rank's
 * share of the BT+BV rows.  d_partial = nstmt x 5 partial ciphertexts (struct proof order).  With more than 31 statements and no image
 * registered the call expands the rank's shares into a transient image first (see mfh_prove_batch). */
int mfh_prove_batch_partial(mfh_ctx *ctx, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, uint32_t nstmt, const uint8_t *h_witness_bits,
                            size_t bits_stride, const uint32_t *d_w, const uint32_t *d_h, const uint32_t *d_v, size_t coef_stride, uint64_t *d_partial);
/* step 5: d_proofs = nstmt summed proofs (after mfh_ct_from_lanes): b_w += delta_b ct_t (src/snark.c:143-145), then the five smudging
 * draws per proof in the order of mfh_prove (src/snark.c:185-189).  Queue-only: the host's contribution (deltas, smudging terms, signs) is staged once in pinned memory and
 * the call returns without waiting for the GPU to reach it (a pipelined caller queues it behind work that has not run yet: host/mfuoco_dist.c). */
int mfh_prove_batch_finish(mfh_ctx *ctx, const uint8_t *d_crs_c8, uint32_t nstmt, const uint32_t *h_delta, const uint8_t *h_smudge_mag, size_t maglen,
                           const uint8_t *h_smudge_sign, uint64_t *d_proofs);

/* ---- multi-GPU: row-sharded prover -------------------------------------------------------------
 * Every proof element is sum_i coeff_i * row_i and the public stream is seekable, so the CRS rows of each region are
 * split into `world` contiguous shares.  mfh_prove_partial computes rank `rank`'s share of the five (un-smudged)
 * ciphertexts; the shares are summed with ONE all-reduce of uint64 lanes (mfh_ct_to_lanes -> RCCL sum ->
 * mfh_ct_from_lanes propagates the carries and applies modq), then mfh_prove_finish smudges.  world = 1 is mfh_prove. */
int mfh_prove_partial(mfh_ctx *ctx, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
                      uint32_t rank, uint32_t world, uint64_t *d_partial);
int mfh_prove_finish(mfh_ctx *ctx, uint64_t *d_proof, const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign);
/* mfh_witness_poly for nstmt <= 12 statements in ONE pass over the SSP: every selected v_i is read (d_ssp == NULL: generated) once and
 * added into the polynomials of the statements whose bit selects it.  h_bits: nstmt bit strings bits_stride bytes apart; d_w: nstmt x d coefficients. */
int mfh_witness_poly_multi(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                           uint32_t *d_w);
/* The same for nstmt <= 256 statements (dense and generator-defined SSP alike) in ONE read of the SSP, as a GEMM on the matrix cores (bits x SSP bytes, exact); d a multiple
 * of 128 (d_ssp == NULL, the generator-defined SSP: the SSP bytes are generated in the kernel).  Keeps a second image of the SSP in MFMA fragment order (same size), built on first use and rebuilt after
 * mfh_ssp_upload / mfh_ssp_prepare.  Used by mfh_prove_batch. */
int mfh_witness_poly_mm(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                        uint32_t *d_w);
/* The same restricted to the coefficients [col0, col0 + ncols) of the polynomials: d_w[b * w_stride + (k - col0)] (col0 and ncols
 * multiples of 128, else MFH_EUNSUPPORTED).  The cost is the range's share of the read / generation of the selected rows. */
int mfh_witness_poly_mm_cols(mfh_ctx *ctx, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                             uint32_t col0, uint32_t ncols, uint32_t *d_w, size_t w_stride);
/* Optional second exchange that also shards the witness polynomial (the SSP pass, 1.4 GB per proof at the default size):
 * mfh_witness_lanes = this rank's share of sum_{bit} v_i as d uint64 lanes (each < p) -> all-reduce (sum) ->
 * mfh_prove_partial_w takes the summed lanes instead of recomputing w on every rank.  mfh_witness_from_lanes: w = delta t + lanes mod p. */
int mfh_witness_lanes(mfh_ctx *ctx, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t rank, uint32_t world, uint64_t *d_lanes);
int mfh_witness_from_lanes(mfh_ctx *ctx, const uint32_t *d_ssp, const uint64_t *d_lanes, uint32_t delta, uint32_t *d_w);
int mfh_prove_partial_w(mfh_ctx *ctx, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
                        uint32_t rank, uint32_t world, const uint64_t *d_wlanes, uint64_t *d_partial);
/* count ciphertexts <-> count * (n+1) * mfh_lanes_per_value uint64 lanes: lane j of a value = its bits [56 j, 56 j + 56) (13 lanes per 704-bit value, 27 at
 * logq 1472): sums of up to 256 ranks' lanes fit 64 bits; mfh_ct_from_lanes propagates the carries and applies modq (src/lwe.h:107-118). */
uint32_t mfh_lanes_per_value(const mfh_ctx *ctx);
int mfh_ct_to_lanes(mfh_ctx *ctx, const uint64_t *d_cts, size_t count, uint64_t *d_lanes);
int mfh_ct_from_lanes(mfh_ctx *ctx, const uint64_t *d_lanes, size_t count, uint64_t *d_cts);

/* ---- library info -------------------------------------------------------------------------------- */
const char *mfh_version(void);
/* 128-bit digest of nbytes at d_buf (4-byte aligned), copied to the host: a cache key for "has this device buffer changed" (the host shim keeps the expanded
 * CRS image across prover calls while the compressed CRS it was made from is unchanged), not a cryptographic hash.  Synchronises the context's stream. */
int mfh_digest128(mfh_ctx *ctx, const void *d_buf, size_t nbytes, uint64_t h_digest[2]);
/* size in bytes the context's scratch currently occupies on the device */
size_t mfh_workspace_bytes(const mfh_ctx *ctx);
/* eval_poly (src/lwe.c:160-178) for MANY coefficient vectors over the same nrows CRS rows -- the S / AS / BV regions of a batch of
 * proofs under one CRS: the rows are expanded once and the multiply-accumulate runs on the matrix cores (i8 MFMA over 8-bit
 * keystream bytes x coefficient bytes; exact integer arithmetic, same result as nvec calls of mfh_eval_rows).
 * d_coeffs = nvec vectors of nrows uint32, vector-major; coeff_bytes = 4: any uint32 value, nvec <= 63; coeff_bytes = 1: the caller
 * guarantees every coefficient < 256 (only the low byte is used), nvec <= 255.  More than 128 digit columns (nvec * coeff_bytes + 1)
 * select the 256-column kernel, which needs off and the row length n * CT_BYTES to be multiples of 8 (true for every CRS region).  d_rops = nvec ciphertexts, vector-major.
 * At logq = 1472 every call uses the 256-column kernel. */
int mfh_eval_rows_multi(mfh_ctx *ctx, uint64_t off, size_t nrows, const uint8_t *d_c8, const uint32_t *d_coeffs, uint32_t nvec,
                        uint32_t coeff_bytes, uint64_t *d_rops, int accumulate);

/* Kernel timing for the roofline leg of bench.py.  With timing enabled every launch of a hot kernel is bracketed by
 * HIP events on the context's stream (no synchronisation is added).  mfh_timing_drain waits for the stream, then
 * reports and forgets the launches of kind `which`: "eval2" / "eval1" (k_eval with 2 / 1 coefficient vectors),
 * "eval" (both), "encrypt", "keystream", "expand", "mac2" / "mac1" (resident MAC), "evalmm" / "evalmm_resident" (mfh_eval_rows_multi from the seed / from the image), "mmstream_rounds" (those of "evalmm_resident" that serve several groups of a batch: the S / AS rounds of mfh_prove_batch; drain it first), "mmstream_bw" (b_w of several super-groups in one launch), "mmstream_rounds_persistent" / "mmstream_bw_persistent" (those of the two that ran the persistent one-workgroup-per-CU grid; drain them before their supersets), "expandmm" (mfh_crs_expand_mm, one launch per region).  total_rows = rows handed to those launches (AES blocks for "keystream"). */
int mfh_set_timing(mfh_ctx *ctx, int enabled);
/* prover scheduling: mfh_prove* run the witness pass + polynomial step on an internal stream beside the evaluation of
 * b_w's rows and join before the S / AS regions; results are identical in every mode.  0 = one stream, 1 (default) = two
 * streams, queueing order picked from the size of b_w's share, 2 = b_w's rows queued first, 3 = the chain queued first. */
int mfh_set_overlap(mfh_ctx *ctx, int mode);
int mfh_timing_drain(mfh_ctx *ctx, const char *which, uint64_t *count, double *total_ms, uint64_t *total_rows, float *last_ms);
float mfh_last_kernel_ms(mfh_ctx *ctx, const char *which); /* = last_ms of mfh_timing_drain; < 0 if none */
/* Union of the [start, end) spans of the launches the last mfh_timing_drain matched, in ms: equals total_ms when the launches
 * ran one after the other, less when launches of two streams overlapped (mfh_prove_batch). */
double mfh_timing_busy_ms(const mfh_ctx *ctx);
/* rows x evaluations served by the launches the last mfh_timing_drain matched: a streaming launch of mfh_prove_batch serves 4 groups of
 * coefficient vectors from one pass over its rows (= total_rows for every other kind) */
uint64_t mfh_timing_work_rows(const mfh_ctx *ctx);

#ifdef __cplusplus
}
#endif
