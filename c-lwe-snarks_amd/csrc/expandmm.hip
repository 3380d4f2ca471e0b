// expandmm.hip -- the CRS expanded into the MFMA A-fragment image of the batch prover (evalmm.hip: k_mmstream reads it), without an LDS
// tile and without barriers.  Replaces k_evalmm16<MODE 1> as the writer of mfh_crs_expand_mm*.
//
// What is written (unchanged layout): for row tile P (16 byte positions of the row's SIGNIFICANT bytes, the b coordinate included) and
// 64-row k-step ks, the 1 KiB fragment  image[((P * KS + ks) * 64 + lane) * 16 + e] = byte (position 16 P + (lane & 15)) of row
// 64 ks + 16 (lane >> 4) + e, offset by 128.  That is a 16 x 16 byte transposition of what AES-CTR produces (16 consecutive bytes of ONE
// row per block) -- the old kernel did it with an LDS tile, 16 ds_read_u8 per lane and fragment, and five barriers per 256 rows, at 45-50
// Gblock/s of AES.  Here:
//   * lane = row.  A wave takes the 64 rows of a k-step and walks along them, one AES block per lane and step (the free-running AES of
//     k_encrypt_mm: the 64 KiB table is the only LDS object, 8 waves per SIMD).  Rows start at byte 0 or 8 of a block (row length 8 mod
//     16): a lane whose row starts at byte 8 shifts its block stream by two dwords (two v_cndmask per dword), after which every lane of
//     the wave sees the same row-relative dword stream.
//   * significant bytes.  At logq = 736 a value is 23 stream dwords of which 22 survive modq; at 1472 all 46 do, but a coordinate's 11.5
//     fragments are padded to 12.  Both are static patterns with a period of 23 blocks (92 dwords = 4 | 2 coordinates): which of the four
//     new dwords join which pending ones to form the next 16-byte piece is decided at compile time per block position (a 23-way
//     wave-uniform switch of register moves; the AES stays rolled).
//   * the transposition runs on the matrix cores: with the 64 lanes' pieces as the A operand of v_mfma_i32_16x16x64_i8 and a selection
//     matrix B_s[k][n] = [k == 16 s + n], D_s[m][n] = byte n of the piece of lane 16 s + m: four MFMAs (64 of the matrix pipe's cycles,
//     otherwise idle) turn "16 bytes of one row per lane" into "4 rows of one byte position per lane"; v_perm packs the four int32s into a
//     dword and v_permlane32_swap / v_permlane16_swap transpose the 4 x 4 dwords across the wave's 16-lane rows (4 instructions).  The
//     lane then holds its fragment element: one 16-byte store per lane, 1 KiB contiguous per wave.
//   * the b coordinate (from the compressed CRS) and the padding follow the keystream in the same dword stream (tail of the last period).
#include <algorithm>

#include "ctx.hpp"

namespace {

using mf::AesKey;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr uint32_t PADW = 0x80808080u;  // A' = -128, i.e. A = 0
constexpr int MACB = 23;                // blocks (= 92 dwords) per period

template <int LOGQ> struct XG;
template <> struct XG<736> { static constexpr int DW = 23, SIG = 22, CPM = 4, PPM = 22, MT = 11, CT = 2; };   // dwords per value, significant, coords / pieces per period
template <> struct XG<1472> { static constexpr int DW = 46, SIG = 46, CPM = 2, PPM = 24, MT = 12, CT = 1; };

// Compile-time plan of block position T of a period: which registers form the pieces emitted after this block and what stays pending.
// Codes: 0..2 = pending dword, 3..6 = new dword 0..3, 7 = padding.
struct Plan {
  int nemit;
  int src[2][4];
  int piece[2];   // piece index inside the period
  int npend;
  int pend[3];
};
template <int LOGQ>
constexpr Plan make_plan(int T) {
  using G = XG<LOGQ>;
  int np = 0, piece = 0;
  Plan cur{};
  for (int t = 0; t <= T; t++) {
    int list[8] = {7, 7, 7, 7, 7, 7, 7, 7}, nl = 0;
    for (int i = 0; i < np; i++) list[nl++] = i;  // what the previous block left pending sits in slots 0..np-1
    cur = Plan{};
    for (int i = 0; i < 4; i++) {
      const int within = (4 * t + i) % G::DW;
      if (within < G::SIG) list[nl++] = 3 + i;
      // a full piece; at logq = 1472 also a coordinate's last fragment: 2 dwords + padding (11.5 -> 12 row tiles per coordinate)
      if (nl == 4 || (LOGQ == 1472 && within == G::DW - 1 && nl > 0)) {
        for (int k = 0; k < 4; k++) cur.src[cur.nemit][k] = k < nl ? list[k] : 7;
        cur.piece[cur.nemit] = piece++;
        cur.nemit++;
        nl = 0;
      }
    }
    cur.npend = nl;
    for (int k = 0; k < 3; k++) cur.pend[k] = k < nl ? list[k] : 7;
    np = nl;
  }
  return cur;
}
static_assert(make_plan<736>(22).npend == 0 && make_plan<736>(22).piece[make_plan<736>(22).nemit - 1] == 21, "22 pieces per period at logq 736");
static_assert(make_plan<1472>(22).npend == 0 && make_plan<1472>(22).nemit == 2 && make_plan<1472>(22).piece[1] == 23, "24 pieces per period at logq 1472");
static_assert(make_plan<736>(5).nemit == 0 && make_plan<736>(5).npend == 3 && make_plan<736>(6).nemit == 1 && make_plan<736>(6).npend == 3, "the block holding a value's insignificant dword yields 3 dwords");

__device__ __forceinline__ uint32_t pick(int code, const uint32_t (&p)[3], const uint32_t (&v)[4]) {
  return code < 3 ? p[code] : code < 7 ? v[code - 3] : PADW;
}

// grid = (period chunks, ceil(k-steps / 16)); block = 16 waves = 16 consecutive k-steps of 64 rows.
template <int LOGQ>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_expand_mm(
    AesKey key /* rk[56..59] ^ 0x80808080: the keystream comes out as A - 128 */, const uint32_t *__restrict__ g_t0, uint64_t off, uint32_t n, uint32_t nrows,
    uint32_t periods_per_chunk, const uint8_t *__restrict__ c8, v4i *__restrict__ image, uint32_t KS, uint32_t mtiles) {
  using G = XG<LOGQ>;
  __shared__ __attribute__((aligned(16))) uint32_t lt[mf::kTabBytes / 4];  // the only LDS object: address 0 (aes_dev.hpp)
  mf::lds_fill_tab(lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ks = blockIdx.y * 16 + wave;
  const uint32_t row = 64 * ks + lane;
  const uint32_t rowdw = n * G::DW;                        // keystream dwords of a row; then DW dwords of b, then padding
  const uint32_t nper = (n + 1 + G::CPM - 1) / G::CPM;     // periods of a row
  const uint32_t m0 = blockIdx.x * periods_per_chunk, m1 = min(nper, m0 + periods_per_chunk);
  __syncthreads();
  if (64 * ks >= nrows || m0 >= m1) return;  // (wave-uniform)
  const uint64_t rowstart = off + (uint64_t)row * rowdw * 4;
  const bool shifted = (rowstart & 8) != 0;  // the row starts at byte 8 of a block: its dword stream lags the block stream by two dwords
  const uint32_t *__restrict__ bw = reinterpret_cast<const uint32_t *>(c8 + (uint64_t)min(row, nrows - 1) * (G::DW * 4));
  // selection operands of the transposing MFMAs: B_s[k][n] = [k == 16 s + n] -> lane (n, g): byte n of its 16 is 1 iff g == s
  const uint32_t c16 = lane & 15, g4 = lane >> 4;
  // (kept as ONE fragment with the byte set in every lane row and masked per use: sixteen registers less than four fragments)
  const uint32_t one = 1u << (8 * (c16 & 3));
  const v4i selb = {(int)((c16 >> 2) == 0 ? one : 0u), (int)((c16 >> 2) == 1 ? one : 0u), (int)((c16 >> 2) == 2 ? one : 0u), (int)((c16 >> 2) == 3 ? one : 0u)};
  auto emit = [&](uint32_t P, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
    // P is wave-uniform: the fragment's address is a scalar base + the lane's 16 bytes.  (Opaque to the optimiser, or it hoists the 22
    // pieces' addresses of a period out of the block loop as 64-bit VGPR pairs and spills them.)
    asm volatile("" : "+s"(P));
    const v4i a = {(int)d0, (int)d1, (int)d2, (int)d3};
    const v4i zero = {0, 0, 0, 0};
    uint32_t w[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const bool mine = g4 == (uint32_t)s;
      const v4i bs = {mine ? selb[0] : 0, mine ? selb[1] : 0, mine ? selb[2] : 0, mine ? selb[3] : 0};
      const v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bs, zero, 0, 0, 0);  // d[e] = byte c16 of the piece of lane 16 s + 4 g4 + e
      const uint32_t lo = __builtin_amdgcn_perm((uint32_t)d[1], (uint32_t)d[0], 0x0c0c0400u);  // {d0.b0, d1.b0, 0, 0}
      const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d[3], (uint32_t)d[2], 0x04000c0cu);  // {0, 0, d2.b0, d3.b0}
      w[s] = lo | hi;
    }
    // 4 x 4 transposition of (register s) x (16-lane row g4): afterwards register j of lane row G holds what register G held in lane row j
    {
      auto r02 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
      auto r13 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
      auto r01 = __builtin_amdgcn_permlane16_swap(r02[0], r13[0], false, false);
      auto r23 = __builtin_amdgcn_permlane16_swap(r02[1], r13[1], false, false);
      w[0] = r01[0]; w[1] = r01[1]; w[2] = r23[0]; w[3] = r23[1];
    }
    if (P < mtiles) {
      char *frag = reinterpret_cast<char *>(image) + ((uint64_t)P * KS + ks) * 1024;  // (scalar)
      // (the lane's 16-byte offset is RECOMPUTED for every piece, from an opaque zero: kept live -- alone or, hoisted by the optimiser, as
      // the 64-bit pair image + 16 lane -- it is what the register allocator spills at 64 VGPRs, and the reload sat before every store
      // behind an s_waitcnt vmcnt(0): every piece waited for all stores before it)
      uint32_t z = 0;
      asm volatile("" : "+s"(z));
      const uint32_t l16 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)) << 4;
      *reinterpret_cast<v4i *>(frag + l16) = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    }
  };

  // the lane's block stream: v-block B (row-relative dwords 4B .. 4B+3) = stream block cb0 + B, or for a shifted row the upper half of
  // stream block cb0 + B followed by the lower half of cb0 + B + 1
  const uint64_t cb0 = rowstart >> 4;
  uint64_t ctr = cb0 + (uint64_t)m0 * MACB;
  // counter-mode shortcut: a chunk is at most 10 periods = 230 (+1) consecutive blocks per lane: they lie in the span of the first one or the next
  const uint64_t span_a = ctr >> 8;
  uint32_t sca[5], scb[5];
#ifdef MF_AES_GL_EXPAND
  const mf::AesGl GLT = mf::aes_gl(g_t0 + 256);
#endif
  mf::aes_span_consts(tab, L, key, span_a, sca);
  mf::aes_span_consts(tab, L, key, span_a + 1, scb);
  auto block = [&](uint64_t c, uint32_t (&o)[4]) {
    const bool crossed = (c >> 8) != span_a;
    uint32_t sc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) sc[i] = crossed ? scb[i] : sca[i];
#ifdef MF_AES_GL_EXPAND
    mf::aes256_ctr_block_sc<true>(tab, L, key, c, sc, o, &GLT);
#else
    mf::aes256_ctr_block_sc(tab, L, key, c, sc, o);
#endif
  };
  uint32_t x[4];
  block(ctr, x);
  uint32_t pend[3] = {PADW, PADW, PADW};
  for (uint32_t m = m0; m < m1; m++) {
    const uint32_t pbase = m * G::PPM;
    const uint32_t qbase = m * (MACB * 4);  // row-relative dword index of the period's first dword
#pragma unroll 1
    for (uint32_t t = 0; t < (uint32_t)MACB; t++) {
      const uint32_t q0 = qbase + 4 * t;
      uint32_t v[4];
      if (q0 < rowdw) {  // (uniform) keystream: one more stream block
        ctr++;
        uint32_t y[4];
        block(ctr, y);
        v[0] = shifted ? x[2] : x[0];
        v[1] = shifted ? x[3] : x[1];
        v[2] = shifted ? y[0] : x[2];
        v[3] = shifted ? y[1] : x[3];
        // an unshifted row consumes x whole and continues with y; a shifted one keeps y's upper half for the next v-block: both "x = y"
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = y[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = PADW;
      }
      if (q0 + 3 >= rowdw) {  // (uniform) the row's tail: b from the compressed CRS (offset by 128 like the keystream), then padding
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t q = q0 + i;
          if (q >= rowdw) v[i] = q - rowdw < (uint32_t)G::DW ? bw[q - rowdw] ^ PADW : PADW;
        }
      }
      // compaction of the significant dwords into 16-byte pieces: static per block position
      switch (t) {
#define XCASE(T)                                                                                                                        \
  case T: {                                                                                                                             \
    constexpr Plan pl = make_plan<LOGQ>(T);                                                                                             \
    const uint32_t p0[3] = {pend[0], pend[1], pend[2]};                                                                                 \
    if constexpr (pl.nemit >= 1) emit(pbase + pl.piece[0], pick(pl.src[0][0], p0, v), pick(pl.src[0][1], p0, v), pick(pl.src[0][2], p0, v), pick(pl.src[0][3], p0, v)); \
    if constexpr (pl.nemit >= 2) emit(pbase + pl.piece[1], pick(pl.src[1][0], p0, v), pick(pl.src[1][1], p0, v), pick(pl.src[1][2], p0, v), pick(pl.src[1][3], p0, v)); \
    pend[0] = pick(pl.pend[0], p0, v);                                                                                                  \
    pend[1] = pick(pl.pend[1], p0, v);                                                                                                  \
    pend[2] = pick(pl.pend[2], p0, v);                                                                                                  \
  } break;
        XCASE(0) XCASE(1) XCASE(2) XCASE(3) XCASE(4) XCASE(5) XCASE(6) XCASE(7) XCASE(8) XCASE(9) XCASE(10) XCASE(11) XCASE(12) XCASE(13) XCASE(14) XCASE(15)
        XCASE(16) XCASE(17) XCASE(18) XCASE(19) XCASE(20) XCASE(21) XCASE(22)
#undef XCASE
        default: break;
      }
    }
  }
}

}  // namespace

// expands rows [row0, row0 + nrows) of the stream (row0 * n * CT_BYTES + off0 = their offset; c8 = their compressed ciphertexts) into
// `image` in MFMA A-fragment order: the region layout of mfh_crs_expand_mm* (evalmm.hip)
int expand_mm_region(mfh_ctx *c, uint64_t off, uint32_t nrows, const uint8_t *c8, uint8_t *image) {
  const uint32_t n = c->P.n;
  const bool q736 = c->P.logq == 736;
  const uint32_t ct = q736 ? XG<736>::CT : XG<1472>::CT, mt = q736 ? XG<736>::MT : XG<1472>::MT, cpm = q736 ? XG<736>::CPM : XG<1472>::CPM;
  const uint32_t mtiles = (n + 1 + ct - 1) / ct * mt;
  const uint32_t KS = (nrows + 255) / 256 * 4;  // 64-row k-steps of the region, padded to the streaming kernel's 256-row stages
  const uint32_t ksteps = (nrows + 63) / 64, nper = (n + 1 + cpm - 1) / cpm;
  // chunks of at most 10 periods (the counter-mode shortcut's two spans), sized for about eight rounds of the 512 workgroup slots
  const uint32_t gy = (ksteps + 15) / 16;
  uint32_t ppc = 10;
  while (ppc > 2 && (uint64_t)gy * ((nper + ppc - 1) / ppc) < 4096) ppc--;
  const uint32_t gx = (nper + ppc - 1) / ppc;
  mf::AesKey keyx = c->key;
  for (int i = 56; i < 60; i++) keyx.rk[i] ^= 0x80808080u;
  if (q736)
    hipLaunchKernelGGL(k_expand_mm<736>, dim3(gx, gy), dim3(1024), 0, c->stream, keyx, c->d_t0, off, n, nrows, ppc, c8, (v4i *)image, KS, mtiles);
  else
    hipLaunchKernelGGL(k_expand_mm<1472>, dim3(gx, gy), dim3(1024), 0, c->stream, keyx, c->d_t0, off, n, nrows, ppc, c8, (v4i *)image, KS, mtiles);
  return MFH_OK;
}
