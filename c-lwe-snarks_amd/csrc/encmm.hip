// encmm.hip -- batched regev_encrypt2 + ct_export (reference src/lwe.c:78-97,115-119) with the dot product <sk, a> on the matrix cores.
//
// For one row i the hot part is b_i = sum_j sk_j * a_ij mod 2^(64K) (mpz_add_dotp, src/lwe.c:20-28,92): 1470 truncated 704 x 704-bit
// products.  With sk fixed for a batch this is a GEMM in byte digits over the row's keystream:
//
//     G[i][t] = sum_k A[i][k] * B[k][t]          M = rows, K = the row's keystream bytes (1470 x 92), N = 88 result byte positions
//     k = 92 j + u :  B[k][t] = S_j[t - u]  for u <= t (u < 88), else 0        (a Toeplitz band per coordinate)
//     b_i = sum_t G[i][t] 256^t  mod 2^(64K)
//
// and -- the point of this orientation -- the A operand of v_mfma_i32_16x16x64_i8 is "16 consecutive K bytes of one row per lane": exactly
// one AES-CTR output block.  Lane (r = l & 15, g = l >> 4) of a wave computes stream block 4 ks + g of row r and feeds its four output
// words to the MFMA as they are: no LDS tile, no transposition, no barrier in the loop.  The kernel is the bare AES of aes_dev.hpp plus
// 6 (12 at logq = 1472) MFMAs per block; the 253 (1081) v_mad_u64_u32 per coordinate of the VALU kernel (k_encrypt, mfhip.hip) are gone.
//
// Signedness.  The MFMA is signed.  The keystream comes out as A' = A - 128 for free (0x80808080 folded into the last round key); the key
// is recoded ONCE per call into balanced digits S_j = sum_w Sb_j[w] 256^w with Sb in [-128, 127] (the carry out of the top digit is a
// multiple of 2^(64K)).  Then  sum A Sb = sum A' Sb + 128 PS[t],  PS[t] = sum_j sum_{w <= t} Sb_j[w]  -- a per-key constant, no per-row
// correction.  |A' Sb| <= 2^14 and a result position sums at most 1470 x 88 = 129 360 products < 2^31 / 2^14: int32 accumulation is exact
// (at logq = 1472 the columns are split so that no accumulator sees more than 131 071 products).
//
// Row geometry.  A row starts at stream byte off + i * n * CT_BYTES; with off a multiple of 8 that is byte 0 or 8 of an AES block and,
// n * CT_BYTES being 8 mod 16 at logq = 736, alternates with the row's parity.  K is counted from the row's first block, so B exists in two
// versions (head 0 / head 8) and a 16-row MFMA tile holds rows of one parity: workgroup (sb, par) takes rows 512 sb + par + 2 i, i < 256.
#include <algorithm>
#include <cstdlib>

#include "ctx.hpp"

namespace {

using mf::AesKey;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int LOGQ> struct EG;
template <> struct EG<736> { static constexpr int CTB = 92, SBY = 88, NQ = 6, L = 12, KW = 22; };
template <> struct EG<1472> { static constexpr int CTB = 184, SBY = 184, NQ = 12, L = 23, KW = 46; };

// balanced digits of the key: sb[j][w] in [-128, 127], sk_j = sum_w sb[j][w] 256^w mod 2^(8 sby)
__global__ void k_sk_digits(const uint64_t *__restrict__ sk, uint32_t n, uint32_t L, uint32_t sby, int8_t *__restrict__ sb) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint8_t *b = reinterpret_cast<const uint8_t *>(sk + (uint64_t)j * L);
  int carry = 0;
  for (uint32_t w = 0; w < sby; w++) {
    int v = (int)b[w] + carry;
    carry = v >= 128;
    sb[(uint64_t)j * sby + w] = (int8_t)(v - 256 * carry);
  }
}
// col[t] += sum over 64 coordinates of sb[j][t] (col zeroed by the caller); then ps[t] = sum_{w <= t} col[w] = sum_j sum_{w <= t} sb[j][w]
__global__ void k_sk_colsum(const int8_t *__restrict__ sb, uint32_t n, uint32_t sby, long long *__restrict__ col) {
  const uint32_t t = threadIdx.x;
  if (t >= sby) return;
  long long s = 0;
  for (uint32_t j = blockIdx.x * 64; j < min(n, (blockIdx.x + 1) * 64); j++) s += sb[(uint64_t)j * sby + t];
  atomicAdd(reinterpret_cast<unsigned long long *>(col + t), (unsigned long long)s);
}
__global__ void k_sk_prefix(const long long *__restrict__ col, uint32_t sby, int64_t *__restrict__ ps) {
  const uint32_t t = threadIdx.x;
  if (t >= sby) return;
  int64_t p = 0;
  for (uint32_t w = 0; w <= t; w++) p += col[w];
  ps[t] = p;
}
// B in MFMA B-fragment order for one head value: bf[(ks * NQ + q) * 64 + lane] = the 16 bytes B[64 ks + 16 g + e][16 q + c], e = 0..15,
// lane = 16 g + c.  B[k][t]: x = k - head (byte of the row), j = x / ctb, u = x % ctb; sb[j][t - u] if 0 <= x < rowlen, u <= t < sby.
__global__ void k_toeplitz_frag(const int8_t *__restrict__ sb, uint32_t ctb, uint32_t sby, uint32_t NQ, uint32_t ksteps, uint32_t rowlen, uint32_t head,
                                v4i *__restrict__ bf) {
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = idx & 63, q = (uint32_t)((idx >> 6) % NQ), ks = (uint32_t)((idx >> 6) / NQ);
  if (ks >= ksteps) return;
  const uint32_t c16 = lane & 15, g = lane >> 4, t = 16 * q + c16;
  uint32_t pk[4] = {0, 0, 0, 0};
  if (t < sby) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int64_t x = (int64_t)64 * ks + 16 * g + e - head;
      if (x < 0 || x >= (int64_t)rowlen) continue;
      const uint32_t j = (uint32_t)x / ctb, u = (uint32_t)x % ctb;
      if (u > t) continue;  // (u <= t < sby)
      pk[e >> 2] |= (uint32_t)(uint8_t)sb[(uint64_t)j * sby + (t - u)] << (8 * (e & 3));
    }
  }
  bf[idx] = v4i{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
}

// grid = (column chunks, 2 x ceil(nrows / 512)); block = 16 waves, wave w = rows base + 2 (16 w + r), r = lane & 15.
// part[(row * gridDim.x + chunk) * 16 NQ + t] = this chunk's sum_k A'[row][k] B[k][t]
// Two workgroups per CU (2 x 64 KiB of table, 8 waves per SIMD) when the kernel fits 64 VGPRs: it does at logq = 736 (24 accumulator
// registers), not at 1472 (48).  ENCMM_WPE_736=4 builds the one-workgroup variant for A/B timing.
#ifndef ENCMM_WPE_736
#define ENCMM_WPE_736 8
#endif
template <int LOGQ>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(LOGQ == 736 ? ENCMM_WPE_736 : 4, LOGQ == 736 ? ENCMM_WPE_736 : 4))) void k_encrypt_mm(AesKey key /* rk[56..59] ^ 0x80808080 */, const uint32_t *__restrict__ g_t0, uint64_t off, uint32_t rowlen,
                                                     uint32_t nrows, uint32_t ksteps, uint32_t ks_per_chunk, const v4i *__restrict__ bf, int *__restrict__ part) {
  constexpr int NQ = EG<LOGQ>::NQ;
  __shared__ __attribute__((aligned(16))) uint32_t lt[mf::kTabBytes / 4];  // the only LDS object: address 0 (aes_dev.hpp)
  mf::lds_fill_tab(lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t r = lane & 15, g = lane >> 4;
  const uint32_t base = 512 * (blockIdx.y >> 1) + (blockIdx.y & 1);
  const uint32_t row = base + 2 * (16 * wave + r);
  const uint64_t rowstart = off + (uint64_t)row * rowlen;
  // all rows of the workgroup have the parity of `base`: one head value, one B version (wave-uniform pointer)
  const uint32_t head = __builtin_amdgcn_readfirstlane((uint32_t)((off + (uint64_t)base * rowlen) & 15));
  const v4i *__restrict__ bfh = bf + (uint64_t)(head >> 3) * ksteps * NQ * 64;
  const uint32_t ks0 = blockIdx.x * ks_per_chunk, ks1 = min(ksteps, ks0 + ks_per_chunk);
  v4i acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = v4i{0, 0, 0, 0};
  __syncthreads();
  if (base >= nrows || ks0 >= ks1) return;  // (uniform)
  // Counter-mode shortcut (aes_dev.hpp): rounds 1-2 depend on the counter's span of 256 blocks only.  A lane walks 4 blocks per k-step, so
  // for 64 k-steps after a refresh its span is span_a or span_a + 1: both constant sets are kept and the refresh is wave-uniform.
  uint64_t ctr = (rowstart >> 4) + 4ull * ks0 + g;
  uint64_t span_a = ctr >> 8;
  uint32_t sca[5], scb[5];
  mf::aes_span_consts(tab, L, key, span_a, sca);
  mf::aes_span_consts(tab, L, key, span_a + 1, scb);
  for (uint32_t ks = ks0; ks < ks1; ks++, ctr += 4) {
    if (((ks - ks0) & 63) == 0 && ks != ks0) {
      const uint64_t cur = ctr >> 8;
      if (cur != span_a) {
#pragma unroll
        for (int i = 0; i < 5; i++) sca[i] = scb[i];
      }
      span_a = cur;
      mf::aes_span_consts(tab, L, key, span_a + 1, scb);
    }
    // the k-step's B fragments: NQ x 1 KiB, the same for every wave of the workgroup (L1 / L2); in flight under the AES
    v4i b[NQ];
    const v4i *bp = bfh + (uint64_t)ks * NQ * 64 + lane;
#pragma unroll
    for (int q = 0; q < NQ; q++) b[q] = bp[q * 64];
    const bool crossed = (ctr >> 8) != span_a;
    uint32_t sc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) sc[i] = crossed ? scb[i] : sca[i];
    uint32_t w[4];
    mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
    const v4i a = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[q], acc[q], 0, 0, 0);
  }
  // D: register e of lane (c, g) = row 4 g + e of the tile, column c
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t orow = base + 2 * (16 * wave + 4 * g + e);
    if (orow >= nrows) continue;
    int *p = part + ((uint64_t)orow * gridDim.x + blockIdx.x) * (16 * NQ) + r;
#pragma unroll
    for (int q = 0; q < NQ; q++) p[16 * q] = acc[q][e];
  }
}

// b = (sum_t (G[t] + 128 PS[t]) 256^t + e p + m) mod 2^(64K)  ->  CT_BYTES little-endian bytes (ct_export, src/lwe.c:115-119).  The digit
// sums are signed: the carry chain runs in two's complement, which is arithmetic mod 2^(64K) all the same.
// One wave per row: the lanes sum the column chunks of "their" byte positions (coalesced), form the 32-bit-word values W_l = sum_k x[4l+k]
// 256^k (|x| < 2^40: W fits 64 bits) and one lane runs the KW-step carry chain together with e p + m.
template <int LOGQ>
__global__ __launch_bounds__(256) void k_encrypt_finish_mm(const int *__restrict__ part, uint32_t nchunks, uint32_t nrows, const int64_t *__restrict__ ps,
                                                           const uint32_t *__restrict__ msg, const uint64_t *__restrict__ err, uint8_t *__restrict__ c8) {
  using G = EG<LOGQ>;
  constexpr int NC = 16 * G::NQ;
  __shared__ int64_t xs[4][NC];
  __shared__ int64_t ws[4][G::KW];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t row = blockIdx.x * 4 + wv;
  if (row < nrows) {
    const int *p = part + (uint64_t)row * nchunks * NC;
    for (uint32_t t = lane; t < (uint32_t)G::SBY; t += 64) {
      int64_t x = 128 * ps[t];
      for (uint32_t ch = 0; ch < nchunks; ch++) x += p[(uint64_t)ch * NC + t];
      xs[wv][t] = x;
    }
  }
  __syncthreads();
  if (row < nrows && lane < (uint32_t)G::KW) {
    int64_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) w += xs[wv][4 * lane + k] * ((int64_t)1 << (8 * k));
    ws[wv][lane] = w;
  }
  __syncthreads();
  if (row >= nrows || lane != 0) return;
  const uint32_t *e = reinterpret_cast<const uint32_t *>(err + (uint64_t)row * G::L);
  uint32_t *o = reinterpret_cast<uint32_t *>(c8 + (uint64_t)row * G::CTB);
  int64_t dcarry = 0;          // carry of the signed digit chain
  uint64_t carry = msg[row];   // carry of the word chain (+ m)
  uint64_t mulc = 0;
  for (int l = 0; l < G::KW; l++) {
    const int64_t x = ws[wv][l] + dcarry;
    dcarry = x >> 32;  // arithmetic
    const uint64_t ep = (uint64_t)e[l] * MFH_P + mulc;  // e * p, word l
    mulc = ep >> 32;
    const uint64_t tt = carry + (uint32_t)ep + (uint32_t)x;
    o[l] = (uint32_t)tt;
    carry = tt >> 32;
  }
  for (int l = G::KW; l < G::CTB / 4; l++) o[l] = 0;
}

template <int LOGQ>
int encrypt_rows_mm_t(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint32_t *msg, const uint64_t *err, uint8_t *c8) {
  using G = EG<LOGQ>;
  const uint32_t n = c->P.n;
  const uint64_t rowlen64 = (uint64_t)n * G::CTB;
  if (rowlen64 > 0x7fffffffu) return MFH_EUNSUPPORTED;
  const uint32_t rowlen = (uint32_t)rowlen64;
  const uint32_t ksteps = (rowlen + 8 + 63) / 64;  // from the row's first block, head <= 8
  // Column chunks.  No int32 accumulator may see more than 131 071 products: a chunk of ks k-steps holds at most 64 ks keystream bytes of a
  // row, of which at most n * SBY ever meet a non-zero B entry in one column -- one chunk is enough at logq = 736 (129 360 products).
  const uint32_t nblk = 2 * (((uint32_t)nrows + 511) / 512);
  uint32_t kc_min = 1;
  if ((uint64_t)n * G::SBY > 131071) kc_min = (ksteps * 64 + 131070) / 131071;
  // Column chunks also set the grain of the dispatch: with about four workgroups per workgroup slot (256 CUs x 1 or 2 workgroups of 64 KiB
  // table) the CUs stay evenly loaded to the end (65 536 rows: 60 Gblock/s of AES with one chunk, 66 with 2, 74 with 8 -- measured), while
  // a chunk stays long against the workgroup's start-up (the table fill is worth about half a k-step)
  const uint32_t slots = 256 * ((LOGQ == 736 && ENCMM_WPE_736 == 8) ? 2 : 1);
  // -- and among such counts the one whose last round of workgroups is fullest (87 381 rows: 65 Gblock/s with 6 chunks = 4.008 rounds, 73
  // with 16 = 10.7 rounds)
  const uint32_t k_lo = std::max(kc_min, (4 * slots + nblk - 1) / nblk), k_hi = std::max(k_lo, std::min(3 * k_lo, std::max(1u, ksteps / 32)));
  uint32_t kc = std::min(k_lo, k_hi);
  double best = 0;
  for (uint32_t k = kc; k <= k_hi; k++) {
    const uint64_t wg = (uint64_t)nblk * k, rounds = (wg + slots - 1) / slots;
    const double eff = (double)wg / ((double)slots * (double)rounds);
    if (eff > best + 1e-9) { best = eff; kc = k; }
  }
  kc = std::max(kc_min, kc);
  if (c->enc_chunks) kc = std::max<uint32_t>(kc_min, c->enc_chunks);  // tuning override (mfh_set_encrypt_chunks: at most 64)
  const uint32_t kpc = (ksteps + kc - 1) / kc;
  kc = (ksteps + kpc - 1) / kpc;
  const size_t sb_b = ((size_t)n * G::SBY + 255) & ~(size_t)255, ps_b = 2 * 256 * 8;  // ps | column sums
  const size_t bf_b = (size_t)2 * ksteps * G::NQ * 1024;
  const size_t part_b = (size_t)nrows * kc * 16 * G::NQ * 4;
  int rc = ws_reserve(c, sb_b + ps_b + bf_b + part_b);
  if (rc) return rc;
  uint8_t *w = (uint8_t *)c->ws;
  int8_t *sb = (int8_t *)w;
  int64_t *ps = (int64_t *)(w + sb_b);
  v4i *bf = (v4i *)(w + sb_b + ps_b);
  int *part = (int *)(w + sb_b + ps_b + bf_b);
  hipLaunchKernelGGL(k_sk_digits, dim3((n + 255) / 256), dim3(256), 0, c->stream, sk, n, (uint32_t)G::L, (uint32_t)G::SBY, sb);
  long long *col = (long long *)(ps + 256);
  HIP_TRY(c, hipMemsetAsync(col, 0, 256 * 8, c->stream));
  hipLaunchKernelGGL(k_sk_colsum, dim3((n + 63) / 64), dim3(256), 0, c->stream, sb, n, (uint32_t)G::SBY, col);
  hipLaunchKernelGGL(k_sk_prefix, dim3(1), dim3(256), 0, c->stream, col, (uint32_t)G::SBY, ps);
  const uint64_t nfr = (uint64_t)ksteps * G::NQ * 64;
  for (uint32_t h = 0; h < 2; h++)
    hipLaunchKernelGGL(k_toeplitz_frag, dim3((uint32_t)((nfr + 255) / 256)), dim3(256), 0, c->stream, sb, (uint32_t)G::CTB, (uint32_t)G::SBY, (uint32_t)G::NQ, ksteps,
                       rowlen, 8 * h, bf + (size_t)h * nfr);
  AesKey keyx = c->key;
  for (int i = 56; i < 60; i++) keyx.rk[i] ^= 0x80808080u;  // the keystream bytes come out as A - 128
  {
    Timer t(c, 3, nrows);
    hipLaunchKernelGGL(k_encrypt_mm<LOGQ>, dim3(kc, nblk), dim3(1024), 0, c->stream, keyx, c->d_t0, off, rowlen, (uint32_t)nrows, ksteps, kpc, bf, part);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL(k_encrypt_finish_mm<LOGQ>, dim3(((uint32_t)nrows + 3) / 4), dim3(256), 0, c->stream, part, kc, (uint32_t)nrows, ps, msg, err, c8);
  HIP_TRY(c, hipGetLastError());
  // the balanced digits of the secret key, their prefix sums and Toeplitz fragments do not outlive the call in the shared scratch
  HIP_TRY(c, hipMemsetAsync(w, 0, sb_b + ps_b + bf_b, c->stream));
  return MFH_OK;
}

}  // namespace

// the matrix-core form of mfh_encrypt_rows; needs off to be a multiple of 8 (every row then starts at byte 0 or 8 of an AES block)
int encrypt_rows_mm(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint32_t *msg, const uint64_t *err, uint8_t *c8) {
  if (c->P.logq == 736) return encrypt_rows_mm_t<736>(c, off, nrows, sk, msg, err, c8);
  return encrypt_rows_mm_t<1472>(c, off, nrows, sk, msg, err, c8);
}
