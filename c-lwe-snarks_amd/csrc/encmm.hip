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
#ifdef MF_AES_GL_ENC
  const mf::AesGl GLT = mf::aes_gl(g_t0 + 256);
#endif
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
#ifdef MF_AES_GL_ENC
    mf::aes256_ctr_block_sc<true>(tab, L, key, ctr, sc, w, &GLT);
#else
    mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
#endif
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

// ---- regev_decrypt (src/lwe.c:105-111) for batches: m = (b - (<a, sk> mod 2^(64K))) mod p, the dot product as the same Toeplitz int8 GEMM ---------------
// (1) FULL ciphertexts resident in HBM ((n + 1) values of L limbs): k_decrypt_mm.  G[i][t] = sum_k A[i][k] B[k][t] with K = the bytes of the
//     ciphertext's a part as they lie in memory (n x 8L: 96-byte elements at logq 736, the top 8 bytes of each meeting zero rows of B) -- the A operand of
//     v_mfma_i32_16x16x64_i8 is "16 consecutive K bytes of one row per lane" = one 16-byte global load.  HBM-bound by construction (141 KB
//     per decryption against 1.7 10^7 int8 multiply-adds); what has to be kept off the memory pipe is B: 6 KiB per 64 K-bytes, the same for every
//     row, staged through LDS once per workgroup and group of 4 k-steps and read from there by its 16 waves (one row tile = 16 rows per wave).
//     Vector-memory loads return in order, so every global load of the loop is consumed exactly one group after its issue: the A fragments
//     of group s + 1 and the B staging words of group s + 2 are issued at the top of group s.  Measured (65 536 ciphertexts, 9.25 GB): 1.89 ms =
//     4.9 TB/s with 8 waves per workgroup, 1.82 - 1.87 ms with 16 (4 waves: 2.29: the more rows share a staged B, the better); row tiles per wave x k-steps per group = 2 x 2: 2.10, 2 x 4: 1.95, 1 x 4: 1.89 ms; non-temporal loads 2.6-2.9 ms (a lane's
//     64-byte half lines want the cache to keep the other half); without the ds_bpermute below 2.05 ms.
//     Signedness: A' = A ^ 0x80 (= A - 128), Sb balanced; sum A Sb = sum A' Sb + 128 PS[t] as in k_encrypt_mm (the pad bytes meet B = 0).
// (2) SEED-COMPRESSED ciphertexts (stream offset + the 92-byte b of ct_export, what the CRS holds): the a part is regenerated -- that is
//     k_encrypt_mm as it stands (AES-bound), with a finishing kernel that subtracts instead of adding e p + m.
#ifndef DEC_RT  /* timing variants: tools/build_variant.sh encmm "-DDEC_RT=1 -DDEC_GK=4" */
#define DEC_RT 1
#endif
#ifndef DEC_GK
#define DEC_GK 4
#endif
#ifndef DEC_WAVES
#define DEC_WAVES 16
#endif
template <int LOGQ>
struct DG {
  static constexpr int ELB = EG<LOGQ>::L * 8;          // bytes of a value in memory (96 | 184)
  static constexpr int RT = LOGQ == 736 ? DEC_RT : 1;  // row tiles (16 rows) per wave
  static constexpr int WAVES = DEC_WAVES, ROWS = WAVES * RT * 16, GK = LOGQ == 736 || DEC_GK <= 4 ? DEC_GK : 4;  // k-steps per group (logq 1472: twice the fragments per k-step, 4 at most fit the LDS)
};

template <int LOGQ>
__global__ __launch_bounds__(DG<LOGQ>::WAVES * 64) void k_decrypt_mm(const uint8_t *__restrict__ cts, uint64_t ct_stride, uint32_t nrows, uint32_t ksteps,
                                                                     uint32_t ks_per_chunk, const v4i *__restrict__ bf, int *__restrict__ part) {
  using D = DG<LOGQ>;
  constexpr int NQ = EG<LOGQ>::NQ, RT = D::RT, GK = D::GK, NT = D::WAVES * 64;
  constexpr int BPG = GK * NQ * 64;                 // v4i of B per group
  constexpr int SPT = (BPG + NT - 1) / NT;          // staging words per thread and group
  __shared__ v4i bs[2][BPG];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t r = lane & 15, g = lane >> 4;
  const uint32_t row0 = blockIdx.y * D::ROWS + wave * (RT * 16);
  const uint32_t ks0 = blockIdx.x * ks_per_chunk, ks1 = min(ksteps, ks0 + ks_per_chunk);
  if (ks0 >= ks1) return;  // (uniform; the host launches no empty chunk)
  const uint32_t ngroups = (ks1 - ks0 + GK - 1) / GK;
  const uint8_t *ap[RT];
#pragma unroll
  for (int t = 0; t < RT; t++) ap[t] = cts + (uint64_t)min(row0 + 16 * t + r, nrows - 1) * ct_stride + 16 * g;  // (rows past the end: clamped, dropped at the end)
  // k-steps past ks1 - 1 are clamped to it for the loads (no branches around loads: s_waitcnt counts stay static) and skipped in the MFMAs
  // The MFMA wants lane (g, r) to hold K bytes 16 g .. 16 g + 15 of row r: neighbouring lanes are neighbouring ROWS, 141 KB apart, and a load in that
  // shape is 64 separate 16-byte requests (measured: 2.05 ms per 65 536 ciphertexts whatever the prefetch depth).  So the load is issued in a coalesced
  // shape -- lane l takes segment l & 3 of row l >> 2: a quad reads 64 contiguous bytes -- and the four dwords are moved to their MFMA lanes with
  // ds_bpermute_b32 (lane 16 g + r reads lane 16 (r >> 2) + 4 (r & 3) + g): a wave-local pass through the LDS crossbar, no LDS memory, no barrier.
#ifdef DEC_NO_BPERM
  constexpr bool kBperm = false;
#else
  constexpr bool kBperm = true;
#endif
  if (kBperm) {
#pragma unroll
    for (int t = 0; t < RT; t++) ap[t] = cts + (uint64_t)min(row0 + 16 * t + (lane >> 2), nrows - 1) * ct_stride + 16 * (lane & 3);
  }
  const int bperm_src = (int)(4 * (16 * (r >> 2) + 4 * (r & 3) + g));
  auto to_mfma_lanes = [&](v4i x) -> v4i {
    x ^= v4i{(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};  // A - 128
    if (!kBperm) return x;
    return v4i{__builtin_amdgcn_ds_bpermute(bperm_src, x[0]), __builtin_amdgcn_ds_bpermute(bperm_src, x[1]), __builtin_amdgcn_ds_bpermute(bperm_src, x[2]),
               __builtin_amdgcn_ds_bpermute(bperm_src, x[3])};
  };
  auto a_load = [&](int t, uint32_t ks) -> v4i { return *reinterpret_cast<const v4i *>(ap[t] + (uint64_t)min(ks, ks1 - 1) * 64); };
  auto b_load = [&](uint32_t grp, int i) -> v4i {
    const uint32_t w = tid + NT * i;  // word of the group: (k-step of the group, q, lane)
    const uint32_t ks = min(ks0 + grp * GK + w / (NQ * 64), ks1 - 1);
    return bf[(uint64_t)ks * NQ * 64 + w % (NQ * 64)];
  };
  v4i acc[RT][NQ];
#pragma unroll
  for (int t = 0; t < RT; t++)
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[t][q] = v4i{0, 0, 0, 0};
  v4i a[GK][RT], bst[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++)
    if (tid + NT * i < BPG) bs[0][tid + NT * i] = b_load(0, i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < SPT; i++) bst[i] = b_load(1 < ngroups ? 1 : 0, i);
#pragma unroll
  for (int k = 0; k < GK; k++)
#pragma unroll
    for (int t = 0; t < RT; t++) a[k][t] = a_load(t, ks0 + k);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  for (uint32_t grp = 0; grp < ngroups; grp++) {
    const v4i *bcur = bs[grp & 1];
    v4i acur[GK][RT];
#pragma unroll
    for (int k = 0; k < GK; k++)
#pragma unroll
      for (int t = 0; t < RT; t++) acur[k][t] = to_mfma_lanes(a[k][t]);
    // the staging words of group grp + 1 (loaded a group ago) go to the other buffer, which group grp - 1 has finished with (barrier below)
#pragma unroll
    for (int i = 0; i < SPT; i++)
      if (tid + NT * i < BPG) bs[(grp & 1) ^ 1][tid + NT * i] = bst[i];
    // issue: staging words of group grp + 2, then the A fragments of group grp + 1 -- all consumed one group from now
    const uint32_t g2 = min(grp + 2, ngroups - 1), g1 = grp + 1;
#pragma unroll
    for (int i = 0; i < SPT; i++) bst[i] = b_load(g2, i);
#pragma unroll
    for (int k = 0; k < GK; k++)
#pragma unroll
      for (int t = 0; t < RT; t++) a[k][t] = a_load(t, ks0 + g1 * GK + k);
#pragma unroll
    for (int k = 0; k < GK; k++) {
      if (ks0 + grp * GK + k < ks1) {  // (uniform)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const v4i b = bcur[(k * NQ + q) * 64 + lane];
#pragma unroll
          for (int t = 0; t < RT; t++) acc[t][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(acur[k][t], b, acc[t][q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // D: register e of lane (c = r, g) = row 4 g + e of the tile, column c
#pragma unroll
  for (int t = 0; t < RT; t++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t orow = row0 + 16 * t + 4 * g + e;
      if (orow >= nrows) continue;
      int *p = part + ((uint64_t)orow * gridDim.x + blockIdx.x) * (16 * NQ) + r;
#pragma unroll
      for (int q = 0; q < NQ; q++) p[16 * q] = acc[t][q][e];
    }
}

__device__ __forceinline__ uint32_t words_mod_p_(const uint32_t *w, int nw) {  // 2^32 = 5 (mod p): Horner from the top word
  uint64_t r = 0;
  for (int l = nw - 1; l >= 0; l--) r = (r * 5 + w[l]) % MFH_P;
  return (uint32_t)r;
}
// out[row] = (b - dot) mod p with dot = sum_t (G[t] + 128 PS[t]) 256^t mod 2^(64K) and b the row's last value: FULL = 1: the L limbs at
// bsrc + row * bstride (taken whole: ct_import does not reduce it, src/lwe.c:125), FULL = 0: the CT_BYTES of ct_export.  One wave per row.
template <int LOGQ, int FULL>
__global__ __launch_bounds__(256) void k_decrypt_finish_mm(const int *__restrict__ part, uint32_t nchunks, uint32_t nrows, const int64_t *__restrict__ ps,
                                                           const uint8_t *__restrict__ bsrc, uint64_t bstride, uint32_t *__restrict__ out) {
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
  uint32_t dot[G::KW];
  int64_t dcarry = 0;  // signed digit chain: two's complement is arithmetic mod 2^(64K) all the same
  for (int l = 0; l < G::KW; l++) {
    const int64_t x = ws[wv][l] + dcarry;
    dcarry = x >> 32;  // arithmetic
    dot[l] = (uint32_t)x;
  }
  constexpr int BW = FULL ? 2 * G::L : G::CTB / 4;
  uint32_t b[BW];
  const uint32_t *bp = reinterpret_cast<const uint32_t *>(bsrc + (uint64_t)row * bstride);
  for (int l = 0; l < BW; l++) b[l] = bp[l];
  const uint32_t bm = words_mod_p_(b, BW), dm = words_mod_p_(dot, G::KW);
  out[row] = (uint32_t)(((uint64_t)bm + MFH_P - dm) % MFH_P);
}

// per-key operands shared by the three matrix-core forms: balanced digits, prefix sums, Toeplitz fragments for element stride `ctb` (heads 0 and, for
// stream rows, 8).  Returns the scratch layout in w / sb / ps / bf / part; the caller's KeyWipe zeroes [w, w + total) when the call ends.
struct KeyOps { uint8_t *w; int8_t *sb; int64_t *ps; v4i *bf; int *part; size_t keyed, total; };
// Everything in the shared scratch that is derived from the secret key is zeroed on EVERY way out of the call, early error returns included: the balanced
// digits, prefix sums and Toeplitz fragments ([w, w + keyed)) and the partial products behind them -- those hold the exact digits of <a_i, sk> for every row,
// which with the public a (about n rows) determine sk, and for an encryption also expose e p + m.
struct KeyWipe {
  mfh_ctx *c;
  KeyOps &K;
  KeyWipe(mfh_ctx *c_, KeyOps &K_) : c(c_), K(K_) { K.w = nullptr; K.total = 0; }
  ~KeyWipe() {
    if (K.w && K.total) (void)hipMemsetAsync(K.w, 0, K.total, c->stream);
  }
};
template <int LOGQ>
int key_operands(mfh_ctx *c, const uint64_t *sk, uint32_t ctb, uint32_t rowlen, uint32_t ksteps, int nheads, size_t part_b, KeyOps &K) {
  using G = EG<LOGQ>;
  const uint32_t n = c->P.n;
  const size_t sb_b = ((size_t)n * G::SBY + 255) & ~(size_t)255, ps_b = 2 * 256 * 8;  // ps | column sums
  const size_t bf_b = (size_t)nheads * ksteps * G::NQ * 1024;
  int rc = ws_reserve(c, sb_b + ps_b + bf_b + part_b);
  if (rc) return rc;
  K.w = (uint8_t *)c->ws;
  K.sb = (int8_t *)K.w;
  K.ps = (int64_t *)(K.w + sb_b);
  K.bf = (v4i *)(K.w + sb_b + ps_b);
  K.part = (int *)(K.w + sb_b + ps_b + bf_b);
  K.keyed = sb_b + ps_b + bf_b;
  K.total = K.keyed + part_b;
  hipLaunchKernelGGL(k_sk_digits, dim3((n + 255) / 256), dim3(256), 0, c->stream, sk, n, (uint32_t)G::L, (uint32_t)G::SBY, K.sb);
  long long *col = (long long *)(K.ps + 256);
  HIP_TRY(c, hipMemsetAsync(col, 0, 256 * 8, c->stream));
  hipLaunchKernelGGL(k_sk_colsum, dim3((n + 63) / 64), dim3(256), 0, c->stream, K.sb, n, (uint32_t)G::SBY, col);
  hipLaunchKernelGGL(k_sk_prefix, dim3(1), dim3(256), 0, c->stream, col, (uint32_t)G::SBY, K.ps);
  const uint64_t nfr = (uint64_t)ksteps * G::NQ * 64;
  for (int h = 0; h < nheads; h++)
    hipLaunchKernelGGL(k_toeplitz_frag, dim3((uint32_t)((nfr + 255) / 256)), dim3(256), 0, c->stream, K.sb, ctb, (uint32_t)G::SBY, (uint32_t)G::NQ, ksteps, rowlen,
                       8u * h, K.bf + (size_t)h * nfr);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

template <int LOGQ>
int decrypt_mm_t(mfh_ctx *c, const uint64_t *sk, const uint64_t *cts, size_t count, uint32_t *out) {
  using G = EG<LOGQ>;
  using D = DG<LOGQ>;
  const uint32_t n = c->P.n;
  const uint64_t klen = (uint64_t)n * D::ELB;  // K: the bytes of the a part as they lie in memory
  if (klen > 0x7fffffffu || count > 0x7fffffffu) return MFH_EUNSUPPORTED;
  const uint32_t ksteps = (uint32_t)((klen + 63) / 64);
  // K chunks: no int32 accumulator may see more than 131 071 products of magnitude <= 2^14 (n x SBY non-zero B entries per column at most), and
  // the chunks set the dispatch grain: about four workgroups per workgroup slot (2 per CU) keep the CUs evenly loaded to the end
  uint32_t kc = 1;
  if ((uint64_t)n * G::SBY > 131071) kc = (ksteps * 64 + 131070) / 131071;
  const uint32_t nblk = ((uint32_t)count + D::ROWS - 1) / D::ROWS;
  kc = std::max(kc, std::min(std::max(1u, ksteps / 64), (8u * c->ncu + nblk - 1) / nblk));
  uint32_t kpc = (ksteps + kc - 1) / kc;
  kpc = (kpc + D::GK - 1) / D::GK * D::GK;
  kc = (ksteps + kpc - 1) / kpc;
  KeyOps K;
  KeyWipe wipe(c, K);
  int rc = key_operands<LOGQ>(c, sk, (uint32_t)D::ELB, (uint32_t)klen, ksteps, 1, (size_t)count * kc * 16 * G::NQ * 4, K);
  if (rc) return rc;
  const uint64_t ct_stride = (uint64_t)(n + 1) * D::ELB;
  {
    Timer t(c, 11, count);
    hipLaunchKernelGGL(k_decrypt_mm<LOGQ>, dim3(kc, nblk), dim3(D::WAVES * 64), 0, c->stream, reinterpret_cast<const uint8_t *>(cts), ct_stride, (uint32_t)count, ksteps,
                       kpc, K.bf, K.part);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL((k_decrypt_finish_mm<LOGQ, 1>), dim3(((uint32_t)count + 3) / 4), dim3(256), 0, c->stream, K.part, kc, (uint32_t)count, K.ps,
                     reinterpret_cast<const uint8_t *>(cts) + (uint64_t)n * D::ELB, ct_stride, out);
  HIP_TRY(c, hipGetLastError());
  // (KeyWipe zeroes the key's digits, prefix sums, Toeplitz fragments and the partial products on every way out)
  return MFH_OK;
}

// column chunks of a k_encrypt_mm launch over nrows stream rows
struct EncPlan { uint32_t rowlen, ksteps, nblk, kc, kpc; };
template <int LOGQ>
int enc_plan(mfh_ctx *c, size_t nrows, EncPlan &P) {
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
  P = EncPlan{rowlen, ksteps, nblk, kc, kpc};
  return MFH_OK;
}

template <int LOGQ>
int encrypt_rows_mm_t(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint32_t *msg, const uint64_t *err, uint8_t *c8) {
  using G = EG<LOGQ>;
  EncPlan P;
  int rc = enc_plan<LOGQ>(c, nrows, P);
  if (rc) return rc;
  KeyOps K;
  KeyWipe wipe(c, K);
  rc = key_operands<LOGQ>(c, sk, (uint32_t)G::CTB, P.rowlen, P.ksteps, 2, (size_t)nrows * P.kc * 16 * G::NQ * 4, K);
  if (rc) return rc;
  AesKey keyx = c->key;
  for (int i = 56; i < 60; i++) keyx.rk[i] ^= 0x80808080u;  // the keystream bytes come out as A - 128
  {
    Timer t(c, 3, nrows);
    hipLaunchKernelGGL(k_encrypt_mm<LOGQ>, dim3(P.kc, P.nblk), dim3(1024), 0, c->stream, keyx, c->d_t0, off, P.rowlen, (uint32_t)nrows, P.ksteps, P.kpc, K.bf, K.part);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL(k_encrypt_finish_mm<LOGQ>, dim3(((uint32_t)nrows + 3) / 4), dim3(256), 0, c->stream, K.part, P.kc, (uint32_t)nrows, K.ps, msg, err, c8);
  HIP_TRY(c, hipGetLastError());
  // (KeyWipe zeroes the key's digits, prefix sums, Toeplitz fragments and the partial products on every way out)
  return MFH_OK;
}

// regev_decrypt of nrows SEED-COMPRESSED ciphertexts (row i: the a part at stream offset off + i * n * CT_BYTES, b = the CT_BYTES at c8): the same launch
template <int LOGQ>
int decrypt_rows_mm_t(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint8_t *c8, uint32_t *out) {
  using G = EG<LOGQ>;
  EncPlan P;
  int rc = enc_plan<LOGQ>(c, nrows, P);
  if (rc) return rc;
  KeyOps K;
  KeyWipe wipe(c, K);
  rc = key_operands<LOGQ>(c, sk, (uint32_t)G::CTB, P.rowlen, P.ksteps, 2, (size_t)nrows * P.kc * 16 * G::NQ * 4, K);
  if (rc) return rc;
  AesKey keyx = c->key;
  for (int i = 56; i < 60; i++) keyx.rk[i] ^= 0x80808080u;
  {
    Timer t(c, 13, nrows);
    hipLaunchKernelGGL(k_encrypt_mm<LOGQ>, dim3(P.kc, P.nblk), dim3(1024), 0, c->stream, keyx, c->d_t0, off, P.rowlen, (uint32_t)nrows, P.ksteps, P.kpc, K.bf, K.part);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL((k_decrypt_finish_mm<LOGQ, 0>), dim3(((uint32_t)nrows + 3) / 4), dim3(256), 0, c->stream, K.part, P.kc, (uint32_t)nrows, K.ps, c8, (uint64_t)G::CTB, out);
  HIP_TRY(c, hipGetLastError());
  // (KeyWipe zeroes the key's digits, prefix sums, Toeplitz fragments and the partial products on every way out)
  return MFH_OK;
}

}  // namespace

// the matrix-core form of mfh_encrypt_rows; needs off to be a multiple of 8 (every row then starts at byte 0 or 8 of an AES block)
int encrypt_rows_mm(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint32_t *msg, const uint64_t *err, uint8_t *c8) {
  if (c->P.logq == 736) return encrypt_rows_mm_t<736>(c, off, nrows, sk, msg, err, c8);
  return encrypt_rows_mm_t<1472>(c, off, nrows, sk, msg, err, c8);
}

// the matrix-core form of mfh_decrypt: count full ciphertexts in HBM
int decrypt_mm(mfh_ctx *c, const uint64_t *sk, const uint64_t *cts, size_t count, uint32_t *out) {
  if (c->P.logq == 736) return decrypt_mm_t<736>(c, sk, cts, count, out);
  return decrypt_mm_t<1472>(c, sk, cts, count, out);
}
// regev_decrypt of seed-compressed ciphertexts (needs off and the row length to be multiples of 8, like encrypt_rows_mm)
int decrypt_rows_mm(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint8_t *c8, uint32_t *out) {
  if (c->P.logq == 736) return decrypt_rows_mm_t<736>(c, off, nrows, sk, c8, out);
  return decrypt_rows_mm_t<1472>(c, off, nrows, sk, c8, out);
}
