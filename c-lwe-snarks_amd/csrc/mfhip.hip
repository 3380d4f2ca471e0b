// mfhip.hip -- MI355X (gfx950) kernels and the C ABI of include/mfhip.h.
//
// Hot path of mmaker/c-lwe-snarks re-designed for CDNA4 (citations: reference file:line):
//   eval_poly  (src/lwe.c:176-186)  = ct_import (regenerate 1470 x 92 B of AES-256-CTR per row, src/lwe.c:122-126)
//                                     + ct_addmul_ui (1471 x [704-bit += 736-bit x 32-bit], src/lwe.c:141-149)
//   regev_encrypt2 (src/lwe.c:78-97) = same row expansion + <sk, a> (1470 truncated 704x704-bit products)
// The reference spends ~97 % of a prover row in the keystream.  Here the keystream never leaves the CU:
// a workgroup expands a 512-coordinate tile of one ciphertext row into LDS (T-table AES, table replicated
// 32x in LDS so lookups are bank-conflict free), then every thread multiply-accumulates "its" coordinate
// into register-resident 32-bit limbs.  HBM sees only the 92-byte b's, the coefficients and one partial
// accumulator per workgroup.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "aes_dev.hpp"
#include "mfhip.h"

using mf::AesKey;

// ------------------------------------------------------------------------------------------------------
// Compile-time parameter sets.  logq = 736 is the reference's (src/lwe.h:24); 1472 is BASELINE config 5.
// ------------------------------------------------------------------------------------------------------
template <int LOGQ>
struct PS {
  static constexpr int CTB = LOGQ / 8;         // CT_BYTES
  static constexpr int EW = CTB / 4;           // stream words per element (23 | 46)
  static constexpr int L = (LOGQ + 63) / 64;   // limbs of a value (12 | 23)
  static constexpr int K = LOGQ / 64;          // limbs surviving modq (11 | 23)
  static constexpr int KW = 2 * K;             // 32-bit words surviving modq (22 | 46)
  static constexpr int TILE = LOGQ == 736 ? 512 : 256;  // coordinates per row tile
  static constexpr int ROWS = 2;                        // rows a workgroup expands per iteration
  static constexpr int THREADS = TILE * ROWS;           // 1024 | 512
  static constexpr int KS_BYTES = TILE * CTB + 16;      // one row tile of keystream (+1 block when misaligned)
};

#include "ctx.hpp"

static void upload_free(mfh_ctx *c);  // (mfh_ssp_upload's staging lanes, below)

// ------------------------------------------------------------------------------------------------------
// keystream kernel: aesctr_prg / rng_seek (src/aes.c:104-144, src/entropy.c:46-56), stateless form
// ------------------------------------------------------------------------------------------------------
// 128-bit digest of a device buffer (mfh_digest128): sum over the 32-bit words of two different 64-bit mixes of (position, word).  The mix is a
// bijection of the pair, so ANY change of one word changes both sums; changes of several words cancel with probability 2^-128.  A cache key
// (has the caller rewritten this buffer since it was expanded?), not a cryptographic hash.
__device__ __forceinline__ uint64_t digest_mix(uint64_t x, uint64_t m1, uint64_t m2) {
  x = (x ^ (x >> 30)) * m1;
  x = (x ^ (x >> 27)) * m2;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_digest128(const uint8_t *__restrict__ buf, uint64_t nbytes, unsigned long long *__restrict__ acc) {
  const uint64_t nfull = nbytes / 4;
  uint64_t h0 = 0, h1 = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i * 4 < nbytes; i += (uint64_t)gridDim.x * 256) {
    uint32_t w = 0;
    if (i < nfull) w = reinterpret_cast<const uint32_t *>(buf)[i];
    else
      for (uint64_t b = 4 * i; b < nbytes; b++) w |= (uint32_t)buf[b] << (8 * (b - 4 * i));
    const uint64_t x = ((i + 1) << 32) ^ w ^ ((i + 1) >> 32 << 40);  // (position, word) -> one 64-bit value, injective below 2^56 words
    h0 += digest_mix(x + 0x9e3779b97f4a7c15ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull);
    h1 += digest_mix(x ^ 0xd6e8feb86659fd93ull, 0xff51afd7ed558ccdull, 0xc4ceb9fe1a85ec53ull);
  }
  for (int o = 32; o; o >>= 1) {
    h0 += __shfl_xor((unsigned long long)h0, o);
    h1 += __shfl_xor((unsigned long long)h1, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(acc, (unsigned long long)h0);
    atomicAdd(acc + 1, (unsigned long long)h1);
  }
}

__global__ __launch_bounds__(1024) void k_keystream(AesKey key, const uint32_t *__restrict__ g_t0, uint64_t off,
                                                    uint8_t *__restrict__ out, uint64_t nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lt[mf::kTabBytes / 4];
  mf::lds_fill_tab(lt, g_t0);
  __syncthreads();
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lt);
  const mf::AesLane L = mf::aes_lane();
  const uint64_t cb0 = off >> 4;
  const uint32_t head = (uint32_t)(off & 15);
  const uint64_t nblk = (head + nbytes + 15) >> 4;
  const bool aligned = head == 0 && (((uintptr_t)out) & 15) == 0;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t w[4];
    mf::aes256_ctr_block(tab, L, key, cb0 + b, w);
    int64_t o = (int64_t)(b * 16) - head;  // output index of this block's byte 0
    if (aligned && (uint64_t)o + 16 <= nbytes) {
      *reinterpret_cast<uint4 *>(out + o) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      for (int i = 0; i < 16; i++) {
        int64_t idx = o + i;
        if (idx >= 0 && (uint64_t)idx < nbytes) out[idx] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
      }
    }
  }
}

// repack raw stream elements (CTB bytes each) into L-limb values, masked to logq bits
template <int LOGQ>
__global__ void k_repack_values(const uint32_t *__restrict__ ks, uint32_t *__restrict__ out, uint64_t nelem) {
  using S = PS<LOGQ>;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t total = nelem * (2 * S::L);
  if (i >= total) return;
  uint64_t e = i / (2 * S::L);
  uint32_t w = (uint32_t)(i % (2 * S::L));
  out[i] = w < S::EW ? ks[e * S::EW + w] : 0u;
}

// ------------------------------------------------------------------------------------------------------
// Active-row compaction: idx[] = rows with a non-zero coefficient, cnt[0] = how many.  One workgroup.
// (The reference expands zero-coefficient rows only to advance its stream, src/snark.c:147-155.)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_rows(const uint32_t *__restrict__ c0, const uint32_t *__restrict__ c1, uint32_t nrows,
                                                      uint32_t *__restrict__ idx, uint32_t *__restrict__ cnt) {
  // cnt[0] must be zero on entry.  The order of idx[] depends on scheduling; every consumer only sums over it, and the sums
  // are exact integers mod 2^(64K), so results do not.
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const bool act = r < nrows && (c0[r] != 0 || (c1 && c1[r] != 0));
  const unsigned long long m = __ballot(act);
  if (!m) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(cnt, (uint32_t)__popcll(m));
  base = __shfl(base, 0);
  if (act) idx[base + __popcll(m & ((1ull << lane) - 1))] = r;
}

// ------------------------------------------------------------------------------------------------------
// Row-tile expansion into LDS, shared by eval and encrypt.
// Tile = coordinates [j0, j0+TILE) of the row whose element 0 sits at stream byte `rowoff`; `t` is the
// thread's index inside the TILE threads that serve this row.  Whole waves take whole 64-block rounds
// (block b -> thread b % TILE of round b / TILE), so a round in which a wave has no block is skipped by
// that wave entirely: no half-empty issue slots.
// ------------------------------------------------------------------------------------------------------
constexpr int kSpanSlots = 16;  // a row tile covers <= 2945 blocks = at most 13 spans of 256 counters

struct TileGeom {
  uint64_t cb0;   // first AES block of the tile
  uint32_t head;  // byte offset of the tile's first element inside that block
  uint32_t nblk;
};
template <int LOGQ>
__device__ __forceinline__ TileGeom tile_geom(uint64_t rowoff, uint32_t j0, uint32_t nelem) {
  using S = PS<LOGQ>;
  const uint64_t B0 = rowoff + (uint64_t)j0 * S::CTB;
  TileGeom g;
  g.cb0 = B0 >> 4;
  g.head = (uint32_t)(B0 & 15);
  g.nblk = (g.head + nelem * S::CTB + 15) >> 4;
  return g;
}
// threads t < (number of spans) fill spanc[t][0..5) for the tile (rounds 1-2 constants, aes_dev.hpp)
__device__ __forceinline__ void fill_span_table(const uint8_t *tab, const mf::AesLane &L, const AesKey &key, const TileGeom &g, uint32_t t,
                                                uint32_t (*spanc)[8]) {
  const uint64_t sp0 = g.cb0 >> 8;
  const uint32_t nsp = g.nblk ? (uint32_t)(((g.cb0 + g.nblk - 1) >> 8) - sp0 + 1) : 0;
  if (t < nsp) {
    uint32_t sc[5];
    mf::aes_span_consts(tab, L, key, sp0 + t, sc);
#pragma unroll
    for (int i = 0; i < 5; i++) spanc[t][i] = sc[i];
  }
}
template <int LOGQ>
__device__ __forceinline__ void expand_tile_to_lds(const uint8_t *tab, const mf::AesLane &L, const AesKey &key, uint8_t *ks, const TileGeom &g,
                                                   uint32_t t, const uint32_t (*spanc)[8]) {
  using S = PS<LOGQ>;
  const uint64_t sp0 = g.cb0 >> 8;
  for (uint32_t b = t; b < g.nblk; b += S::TILE) {
    const uint64_t ctr = g.cb0 + b;
    const uint32_t *scp = spanc[(uint32_t)((ctr >> 8) - sp0)];
    uint32_t sc[5] = {scp[0], scp[1], scp[2], scp[3], scp[4]};
    uint32_t w[4];
    mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
    *reinterpret_cast<uint4 *>(ks + 16 * b) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// the same for ONE wave and its private tile: lane l takes blocks l, l + 64, ...
__device__ __forceinline__ void expand_tile_wave(const uint8_t *tab, const mf::AesLane &L, const AesKey &key, uint8_t *ks, const TileGeom &g, uint32_t lane,
                                                 const uint32_t (*spanc)[8]) {
  const uint64_t sp0 = g.cb0 >> 8;
  for (uint32_t b = lane; b < g.nblk; b += 64) {
    const uint64_t ctr = g.cb0 + b;
    const uint32_t *scp = spanc[(uint32_t)((ctr >> 8) - sp0)];
    uint32_t sc[5] = {scp[0], scp[1], scp[2], scp[3], scp[4]};
    uint32_t w[4];
    mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
    *reinterpret_cast<uint4 *>(ks + 16 * b) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ------------------------------------------------------------------------------------------------------
// eval kernel: fused ct_import + ct_addmul_ui over the active rows, 1 or 2 coefficient vectors.
// grid = (ntiles, nchunks); block = ROWS x TILE threads; thread (rs, t) owns coordinate j0 + t for the rows
// idx[k], k = k0 + rs, k0 + rs + ROWS, ... of its chunk.
// partials (the two row halves folded first): part[((chunk*NACC + a)*KW + l)*NJ + j]  (uint32), NJ = ntiles*TILE
// ------------------------------------------------------------------------------------------------------
template <int LOGQ, int NACC>
__global__ __launch_bounds__(PS<LOGQ>::THREADS) void k_eval(AesKey key, const uint32_t *__restrict__ g_t0, uint64_t off, uint32_t n,
                                                            const uint32_t *__restrict__ idx, const uint32_t *__restrict__ cnt,
                                                            uint32_t nrows_dense, const uint8_t *__restrict__ c8,
                                                            const uint32_t *__restrict__ coeff0, const uint32_t *__restrict__ coeff1,
                                                            uint32_t *__restrict__ part) {
  using S = PS<LOGQ>;
  // One LDS object with the AES table FIRST: it then sits at LDS address 0 and a lookup's address is exactly (entry << 8) | replica
  // offset.  As separate __shared__ arrays the compiler put the table behind the tiles (0x17020), beyond ds_read's 16-bit immediate,
  // and paid one v_add_u32 per lookup (203 per block, a quarter of the VALU work).
  struct __attribute__((aligned(16))) Lds {
    uint32_t lt[mf::kTabBytes / 4];
    uint8_t ksbuf[S::ROWS][S::KS_BYTES];
    uint32_t spanc[S::ROWS][kSpanSlots][8];
  };
  __shared__ Lds lds;
  auto &ksbuf = lds.ksbuf;
  auto &spanc = lds.spanc;
  mf::lds_fill_tab(lds.lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lds.lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t rs = threadIdx.x / S::TILE, t = threadIdx.x % S::TILE;
  uint8_t *ks = ksbuf[rs];
  const uint32_t j0 = blockIdx.x * S::TILE;
  const uint32_t j = j0 + t;
  const uint32_t nelem = j0 >= n ? 0u : min((uint32_t)S::TILE, n - j0);  // keystream-backed coordinates in this tile
  const uint32_t nact = idx ? cnt[0] : nrows_dense;  // idx == nullptr: every row is active (row k is row k)
  // chunks of a multiple of ROWS active rows
  uint32_t per = (nact + gridDim.y - 1) / gridDim.y;
  per = (per + S::ROWS - 1) / S::ROWS * S::ROWS;
  const uint32_t k0 = blockIdx.y * per;
  const uint32_t k1 = min(nact, k0 + per);
  const uint32_t NJ = gridDim.x * S::TILE;

  uint32_t acc[NACC][S::KW];
#pragma unroll
  for (int a = 0; a < NACC; a++)
#pragma unroll
    for (int l = 0; l < S::KW; l++) acc[a][l] = 0;

  // span constants of the first row, then one barrier; afterwards they are refreshed during the previous row's MAC phase
  TileGeom g{};
  bool have = k0 + rs < k1;
  uint32_t row = 0;
  if (have) {
    row = idx ? idx[k0 + rs] : k0 + rs;
    g = tile_geom<LOGQ>(off + (uint64_t)row * n * S::CTB, j0, nelem);
  }
  __syncthreads();
  if (have) fill_span_table(tab, L, key, g, t, spanc[rs]);
  __syncthreads();
  for (uint32_t kk = k0; kk < k1; kk += S::ROWS) {  // uniform trip count for the whole workgroup (barriers inside)
    uint32_t c[NACC];
    if (have) {
      c[0] = coeff0[row];
      if constexpr (NACC > 1) c[1] = coeff1[row];
      expand_tile_to_lds<LOGQ>(tab, L, key, ks, g, t, spanc[rs]);
    }
    const uint32_t head = g.head;
    const uint32_t cur_row = row;
    const bool cur_have = have;
    __syncthreads();
    // next row of this half: geometry + span constants (nobody reads spanc until after the next barrier)
    const uint32_t kn = kk + S::ROWS + rs;
    have = kn < k1;
    if (have) {
      row = idx ? idx[kn] : kn;
      g = tile_geom<LOGQ>(off + (uint64_t)row * n * S::CTB, j0, nelem);
      fill_span_table(tab, L, key, g, t, spanc[rs]);
    }
    if (cur_have && j <= n) {
      uint32_t a[S::KW];
      if (j < n) {
        const uint32_t *kw = reinterpret_cast<const uint32_t *>(ks + head) + t * S::EW;
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = kw[l];
      } else {
        const uint32_t *bw = reinterpret_cast<const uint32_t *>(c8 + (uint64_t)cur_row * S::CTB);
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = bw[l];
      }
#pragma unroll
      for (int q = 0; q < NACC; q++) {
        uint32_t carry = 0;
#pragma unroll
        for (int l = 0; l < S::KW; l++) {
          uint64_t tt = (uint64_t)a[l] * c[q] + acc[q][l] + carry;
          acc[q][l] = (uint32_t)tt;
          carry = (uint32_t)(tt >> 32);
        }
      }
    }
    __syncthreads();
  }
  // Fold the second row half into the first through the (now free) keystream tiles, so that one partial sum per workgroup and
  // coordinate goes to HBM instead of two.  Sums are mod 2^(32 KW): the carry out of the top word is dropped, as in the MAC.
  static_assert(S::ROWS == 2 && (size_t)S::ROWS * S::KS_BYTES >= (size_t)2 * S::KW * S::TILE * 4, "fold buffer");
  uint32_t *xch = reinterpret_cast<uint32_t *>(&ksbuf[0][0]);
  if (rs == 1) {
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
      for (int l = 0; l < S::KW; l++) xch[(a * S::KW + l) * S::TILE + t] = acc[a][l];
  }
  __syncthreads();
  if (rs == 0) {
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      uint32_t carry = 0;
#pragma unroll
      for (int l = 0; l < S::KW; l++) {
        const uint64_t tt = (uint64_t)acc[a][l] + xch[(a * S::KW + l) * S::TILE + t] + carry;
        part[(((uint64_t)blockIdx.y * NACC + a) * S::KW + l) * NJ + j] = (uint32_t)tt;
        carry = (uint32_t)(tt >> 32);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// k_eval_w: the same fused ct_import + ct_addmul_ui (src/lwe.c:122-126,141-149,176-186) with WAVE-autonomous work: no workgroup
// barrier after the table fill.  An EXPERIMENT that answers "do k_eval's barriers and phases cost its 66.6 Gblock/s against the bare loop's
// 73 at the same occupancy?" -- they do not: this kernel, bit-identical, runs at 63.7 Gblock/s (single proof 10.25 against 9.83 ms, same box,
// tools/eval_ab.py).  k_eval's tile of 2944 blocks is exactly 46 wave-rounds of 64 blocks (whole waves take whole rounds: 92 per iteration and
// CU); a wave-private chunk of 368 blocks is 5.75 rounds, i.e. 6 with the last three quarters full: 96 per CU -- 4.3 % more LDS lookups, and the
// kernel is 4.4 % slower: the AES kernels are bound by the LDS pipe (32 bank accesses per clock and CU; k_eval sustains 22.9 LDS operations per
// clock, the bare loop at 8 waves per SIMD 29.7), not by their barriers.  Kept behind mfh_set_eval_path(ctx, 1) as the measured alternative.
// A wave owns 64 coordinates (64 x 92 B = exactly 368 AES blocks) of a row chunk and a PRIVATE 5.9 KB tile: it computes its 2-3
// span constants (3 lanes), expands its 368 (369 when the row starts mid-block) blocks, reads its coordinates back (stride 23 words:
// conflict-free) and multiplies, row after row, never waiting for another wave; the 16 waves of a CU drift apart, so one wave's
// multiply-accumulate (VALU) runs under the others' table lookups (LDS).  LDS operations of one wave complete in order, so the
// tile needs no barrier, only s_waitcnt.  23 column chunks (1472 = n + 1 coordinates + 1 pad) x nrc row chunks = one wave each;
// 16 waves per workgroup share the table: 64 KiB + 16 x 5904 B + span constants = 158.3 KB, one workgroup per CU, 4 waves per SIMD.
// logq = 736 only (at 1472 a wave's tile is 11.8 KB: the LDS holds 8 of them; k_eval stays).
// partials: part[((row chunk * NACC + a) * KW + l) * NJ + j], NJ = 64 x column chunks: what k_eval_reduce_sum / _carry read.
// ------------------------------------------------------------------------------------------------------
constexpr int kEvalwWaves = 16;
template <int NACC>
__global__ __launch_bounds__(kEvalwWaves * 64) void k_eval_w(AesKey key, const uint32_t *__restrict__ g_t0, uint64_t off, uint32_t n,
                                                             const uint32_t *__restrict__ idx, const uint32_t *__restrict__ cnt, uint32_t nrows_dense,
                                                             const uint8_t *__restrict__ c8, const uint32_t *__restrict__ coeff0,
                                                             const uint32_t *__restrict__ coeff1, uint32_t ncc /* column chunks */, uint32_t nrc /* row chunks */,
                                                             uint32_t *__restrict__ part) {
  using S = PS<736>;
  constexpr int WBLK = 64 * S::CTB / 16 + 1;  // 369 blocks: 64 coordinates, +1 when the row starts at byte 8 of a block
  struct __attribute__((aligned(16))) Lds {   // AES table first => LDS address 0 (see k_eval)
    uint32_t lt[mf::kTabBytes / 4];
    uint8_t tile[kEvalwWaves][WBLK * 16];
    uint32_t spanc[kEvalwWaves][4][8];
  };
  __shared__ Lds lds;
  mf::lds_fill_tab(lds.lt, g_t0);
  __syncthreads();  // the only workgroup barrier
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lds.lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t gw = blockIdx.x * kEvalwWaves + wave;  // wave of the launch: column chunk fastest, so that a workgroup's waves work on the same rows
  const uint32_t cc = gw % ncc, rc = gw / ncc;
  if (rc >= nrc) return;
  uint8_t *ks = lds.tile[wave];
  uint32_t(*spanc)[8] = lds.spanc[wave];
  const uint32_t j0 = cc * 64, j = j0 + lane;
  const uint32_t nelem = j0 >= n ? 0u : min(64u, n - j0);  // keystream-backed coordinates of this chunk
  const uint32_t nact = idx ? cnt[0] : nrows_dense;
  const uint32_t per = (nact + nrc - 1) / nrc;
  const uint32_t k0 = rc * per, k1 = min(nact, k0 + per);
  const uint32_t NJ = ncc * 64;
  uint32_t acc[NACC][S::KW];
#pragma unroll
  for (int a = 0; a < NACC; a++)
#pragma unroll
    for (int l = 0; l < S::KW; l++) acc[a][l] = 0;
  for (uint32_t k = k0; k < k1; k++) {
    const uint32_t row = idx ? idx[k] : k;  // (wave-uniform)
    const TileGeom g = tile_geom<736>(off + (uint64_t)row * n * S::CTB, j0, nelem);
    uint32_t c[NACC];
    c[0] = coeff0[row];
    if constexpr (NACC > 1) c[1] = coeff1[row];
    // span constants of this row's blocks (at most 3 spans of 256 counters), one lane each; every lane then reads them back
    fill_span_table(tab, L, key, g, lane, spanc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    expand_tile_wave(tab, L, key, ks, g, lane, spanc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (j <= n) {
      uint32_t a[S::KW];
      if (j < n) {
        const uint32_t *kw = reinterpret_cast<const uint32_t *>(ks + g.head) + lane * S::EW;
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = kw[l];
      } else {
        const uint32_t *bw = reinterpret_cast<const uint32_t *>(c8 + (uint64_t)row * S::CTB);
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = bw[l];
      }
#pragma unroll
      for (int q = 0; q < NACC; q++) {
        uint32_t carry = 0;
#pragma unroll
        for (int l = 0; l < S::KW; l++) {
          uint64_t tt = (uint64_t)a[l] * c[q] + acc[q][l] + carry;
          acc[q][l] = (uint32_t)tt;
          carry = (uint32_t)(tt >> 32);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the tile and the span constants are rewritten by the next row
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int a = 0; a < NACC; a++)
#pragma unroll
    for (int l = 0; l < S::KW; l++) part[(((uint64_t)rc * NACC + a) * S::KW + l) * NJ + j] = acc[a][l];
}

// ------------------------------------------------------------------------------------------------------
// Resident (materialised) CRS: SURVEY 8(d)'s second regime.  mfh_crs_expand writes every row once in a limb-plane
// layout that the streaming MAC kernel reads with fully coalesced 16-byte loads:
//   row r = [plane 0..NP16) of RS x uint4 | last plane of RS x uint2 (736) ]   RS = row stride in coordinates (n+1 rounded to 64)
//   plane k holds words 4k..4k+3 of every coordinate; only the KW words that survive modq are stored (88 of 92 bytes).
// Coordinate n of a row is its `b` (copied from the compressed CRS) so the MAC kernel has no special case.
// ------------------------------------------------------------------------------------------------------
template <int LOGQ>
struct RL {  // resident layout
  using S = PS<LOGQ>;
  static constexpr int NP16 = S::KW / 4;           // full 16-byte planes (5 | 11)
  static constexpr int TAIL = S::KW % 4;           // words in the last partial plane (2 | 2)
  __host__ __device__ static constexpr uint32_t rs(uint32_t n) { return (n + 1 + 63) / 64 * 64; }
  __host__ __device__ static constexpr uint64_t row_bytes(uint32_t n) { return (uint64_t)rs(n) * S::KW * 4; }
};

template <int LOGQ>
__global__ __launch_bounds__(PS<LOGQ>::THREADS) void k_expand(AesKey key, const uint32_t *__restrict__ g_t0, uint64_t off, uint32_t n,
                                                              uint32_t nrows, uint32_t rows_per_chunk, const uint8_t *__restrict__ c8,
                                                              uint8_t *__restrict__ out) {
  using S = PS<LOGQ>;
  using R = RL<LOGQ>;
  // One LDS object with the AES table FIRST: it then sits at LDS address 0 and a lookup's address is exactly (entry << 8) | replica
  // offset.  As separate __shared__ arrays the compiler put the table behind the tiles (0x17020), beyond ds_read's 16-bit immediate,
  // and paid one v_add_u32 per lookup (203 per block, a quarter of the VALU work).
  struct __attribute__((aligned(16))) Lds {
    uint32_t lt[mf::kTabBytes / 4];
    uint8_t ksbuf[S::ROWS][S::KS_BYTES];
    uint32_t spanc[S::ROWS][kSpanSlots][8];
  };
  __shared__ Lds lds;
  auto &ksbuf = lds.ksbuf;
  auto &spanc = lds.spanc;
  mf::lds_fill_tab(lds.lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lds.lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t rs = threadIdx.x / S::TILE, t = threadIdx.x % S::TILE;
  uint8_t *ks = ksbuf[rs];
  const uint32_t j0 = blockIdx.x * S::TILE, j = j0 + t;
  const uint32_t nelem = j0 >= n ? 0u : min((uint32_t)S::TILE, n - j0);
  const uint32_t r0 = blockIdx.y * rows_per_chunk, r1 = min(nrows, r0 + rows_per_chunk);
  const uint32_t RS = R::rs(n);
  TileGeom g{};
  uint32_t row = r0 + rs;
  bool have = row < r1;
  if (have) g = tile_geom<LOGQ>(off + (uint64_t)row * n * S::CTB, j0, nelem);
  __syncthreads();
  if (have) fill_span_table(tab, L, key, g, t, spanc[rs]);
  __syncthreads();
  for (uint32_t rr = r0; rr < r1; rr += S::ROWS) {
    if (have) expand_tile_to_lds<LOGQ>(tab, L, key, ks, g, t, spanc[rs]);
    const uint32_t head = g.head, cur_row = row;
    const bool cur_have = have;
    __syncthreads();
    row = rr + S::ROWS + rs;
    have = row < r1;
    if (have) {
      g = tile_geom<LOGQ>(off + (uint64_t)row * n * S::CTB, j0, nelem);
      fill_span_table(tab, L, key, g, t, spanc[rs]);
    }
    if (cur_have && j < RS) {
      uint32_t a[S::KW];
      if (j < n) {
        const uint32_t *kw = reinterpret_cast<const uint32_t *>(ks + head) + t * S::EW;
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = kw[l];
      } else if (j == n && c8) {  // (c8 == nullptr: the a parts only -- they depend on the seed alone --, b left zero for k_rows_set_b)
        const uint32_t *bw = reinterpret_cast<const uint32_t *>(c8 + (uint64_t)cur_row * S::CTB);
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = bw[l];
      } else {
#pragma unroll
        for (int l = 0; l < S::KW; l++) a[l] = 0;
      }
      uint8_t *rowp = out + (uint64_t)cur_row * R::row_bytes(n);
#pragma unroll
      for (int k = 0; k < R::NP16; k++)
        reinterpret_cast<uint4 *>(rowp + (uint64_t)k * RS * 16)[j] = make_uint4(a[4 * k], a[4 * k + 1], a[4 * k + 2], a[4 * k + 3]);
      if (R::TAIL == 2)
        reinterpret_cast<uint2 *>(rowp + (uint64_t)R::NP16 * RS * 16)[j] = make_uint2(a[4 * R::NP16], a[4 * R::NP16 + 1]);
    }
    __syncthreads();
  }
}

// coordinate n (the row's b) of rows [0, nrows) of an image from the compressed ciphertexts: what k_expand writes when it is given them
template <int LOGQ>
__global__ void k_rows_set_b(uint32_t n, uint32_t nrows, const uint8_t *__restrict__ c8, uint8_t *__restrict__ out) {
  using S = PS<LOGQ>;
  using R = RL<LOGQ>;
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  const uint32_t RS = R::rs(n);
  const uint32_t *bw = reinterpret_cast<const uint32_t *>(c8 + (uint64_t)row * S::CTB);
  uint8_t *rowp = out + (uint64_t)row * R::row_bytes(n);
#pragma unroll
  for (int k = 0; k < R::NP16; k++)
    reinterpret_cast<uint4 *>(rowp + (uint64_t)k * RS * 16)[n] = make_uint4(bw[4 * k], bw[4 * k + 1], bw[4 * k + 2], bw[4 * k + 3]);
  if (R::TAIL == 2) reinterpret_cast<uint2 *>(rowp + (uint64_t)R::NP16 * RS * 16)[n] = make_uint2(bw[4 * R::NP16], bw[4 * R::NP16 + 1]);
}

// streaming MAC over resident rows: grid = (RS/64 column groups x nchunks), 256 threads = 4 waves; each wave owns 64
// coordinates and walks its share of the active rows, two rows in flight.
// part layout as k_eval's: part[((chunk*NACC + a)*KW + l)*NJ + j], one slab per row chunk (blockIdx.y).
template <int LOGQ, int NACC>
__global__ __launch_bounds__(256) void k_mac_resident(const uint8_t *__restrict__ rows, uint32_t n, const uint32_t *__restrict__ idx,
                                                      const uint32_t *__restrict__ cnt, uint32_t nrows_dense, uint32_t row_base,
                                                      const uint32_t *__restrict__ coeff0, const uint32_t *__restrict__ coeff1,
                                                      uint32_t *__restrict__ part, uint32_t NJ) {
  using S = PS<LOGQ>;
  using R = RL<LOGQ>;
  const uint32_t RS = R::rs(n);
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  const uint32_t nact = idx ? cnt[0] : nrows_dense;
  const uint32_t per = (nact + gridDim.y - 1) / gridDim.y;
  const uint32_t k0 = blockIdx.y * per, k1 = min(nact, k0 + per);
  uint32_t acc[NACC][S::KW];
#pragma unroll
  for (int a = 0; a < NACC; a++)
#pragma unroll
    for (int l = 0; l < S::KW; l++) acc[a][l] = 0;
  if (j < RS) {
    for (uint32_t k = k0; k < k1; k++) {
      const uint32_t row = idx ? idx[k] : k;
      uint32_t c[NACC];
      c[0] = coeff0[row];
      if constexpr (NACC > 1) c[1] = coeff1[row];
      const uint8_t *rowp = rows + (uint64_t)(row_base + row) * R::row_bytes(n);
      uint32_t a[S::KW];
#pragma unroll
      for (int q = 0; q < R::NP16; q++) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rowp + (uint64_t)q * RS * 16) + j);
        a[4 * q] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3];
      }
      if (R::TAIL == 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(rowp + (uint64_t)R::NP16 * RS * 16) + j);
        a[4 * R::NP16] = v[0]; a[4 * R::NP16 + 1] = v[1];
      }
#pragma unroll
      for (int q = 0; q < NACC; q++) {
        uint32_t carry = 0;
#pragma unroll
        for (int l = 0; l < S::KW; l++) {
          uint64_t tt = (uint64_t)a[l] * c[q] + acc[q][l] + carry;
          acc[q][l] = (uint32_t)tt;
          carry = (uint32_t)(tt >> 32);
        }
      }
    }
  }
  if (j < NJ) {
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
      for (int l = 0; l < S::KW; l++) part[(((uint64_t)blockIdx.y * NACC + a) * S::KW + l) * NJ + j] = j < RS ? acc[a][l] : 0u;
  }
}

// Partial-sum reduction, stage 1: lazy[(a*KW + l)*NJ + j] = sum over slabs of the 32-bit partial words (uint64, no carries yet).
// grid = (NJ/256, KW, nacc): every load is a coalesced run over j.
template <int LOGQ>
__global__ __launch_bounds__(256) void k_eval_reduce_sum(const uint32_t *__restrict__ part, uint32_t nslabs, uint32_t nacc, uint32_t NJ,
                                                         uint64_t *__restrict__ lazy) {
  // grid = (NJ/256, KW * nacc, NG): slab group g sums slabs g, g+NG, ... and adds into lazy[(a*KW + l)*NJ + j] (zeroed beforehand;
  // integer atomics: the result does not depend on the order)
  using S = PS<LOGQ>;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t l = blockIdx.y % S::KW, a = blockIdx.y / S::KW;
  const uint32_t g = blockIdx.z, NG = gridDim.z;
  if (j >= NJ) return;
  const uint32_t *p = part + ((uint64_t)a * S::KW + l) * NJ + j;
  const uint64_t stride = (uint64_t)nacc * S::KW * NJ;
  uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  uint32_t ch = g;
  for (; ch + 3 * NG < nslabs; ch += 4 * NG) {
    s0 += p[(uint64_t)ch * stride];
    s1 += p[(uint64_t)(ch + NG) * stride];
    s2 += p[(uint64_t)(ch + 2 * NG) * stride];
    s3 += p[(uint64_t)(ch + 3 * NG) * stride];
  }
  for (; ch < nslabs; ch += NG) s0 += p[(uint64_t)ch * stride];
  atomicAdd(reinterpret_cast<unsigned long long *>(&lazy[((uint64_t)a * S::KW + l) * NJ + j]), (unsigned long long)(s0 + s1 + s2 + s3));
}
// stage 2: propagate carries over the KW words of each coordinate, write natural-layout values (optionally += previous)
// The lazy image and the active-row counter are zero between launches: this kernel, their last reader, clears what it read
// (that replaces two memset launches per evaluation).
template <int LOGQ>
__global__ void k_eval_reduce_carry(uint64_t *__restrict__ lazy, uint32_t *__restrict__ cnt, uint32_t nacc, uint32_t NJ, uint32_t n,
                                    uint64_t *__restrict__ rop0, uint64_t *__restrict__ rop1, int accumulate) {
  using S = PS<LOGQ>;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = blockIdx.y;
  if (j == 0 && a == 0) cnt[0] = 0;
  if (j >= NJ) return;
  if (j > n) {  // padding coordinates: their sums are zero by construction; keep the invariant explicit all the same
    for (int l = 0; l < S::KW; l++) lazy[((uint64_t)a * S::KW + l) * NJ + j] = 0;
    return;
  }
  uint64_t *rop = a == 0 ? rop0 : rop1;
  uint32_t *out = reinterpret_cast<uint32_t *>(rop + (uint64_t)j * S::L);
  uint64_t carry = 0;
  for (int l = 0; l < S::KW; l++) {
    uint64_t s = lazy[((uint64_t)a * S::KW + l) * NJ + j] + carry;  // < 2^32 * slabs + carry: no overflow
    lazy[((uint64_t)a * S::KW + l) * NJ + j] = 0;
    if (accumulate) s += out[l];
    out[l] = (uint32_t)s;
    carry = s >> 32;
  }
  for (int l = S::KW; l < 2 * S::L; l++) out[l] = 0;  // modq: limbs >= K dropped (src/lwe.h:107-118)
}

// (hi:lo) += a * b  as a 96-bit accumulator: v_mad_u64_u32 (carry out in vcc) + v_addc_co_u32.  Two instructions per product;
// the compiler's own lowering of `lo += p; hi += lo < p` takes four (64-bit add, 64-bit compare, select, add).
__device__ __forceinline__ void mac96(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}

// ------------------------------------------------------------------------------------------------------
// encrypt kernel: per row, <sk, a> over a coordinate tile (truncated KW-word products), reduced in LDS.
// grid = (ntiles, nchunks); thread (rs, t): rows r0 + rs, r0 + rs + ROWS, ...
// pb[(row*ntiles + tile)*KW + l] = tile partial of <sk,a> mod 2^(32 KW)
// ------------------------------------------------------------------------------------------------------
template <int LOGQ>
__global__ __launch_bounds__(PS<LOGQ>::THREADS) void k_encrypt(AesKey key, const uint32_t *__restrict__ g_t0, uint64_t off, uint32_t n,
                                                               uint32_t nrows, uint32_t rows_per_chunk,
                                                               const uint64_t *__restrict__ sk, uint32_t *__restrict__ pb) {
  using S = PS<LOGQ>;
  constexpr int NW = S::TILE / 64;
  struct __attribute__((aligned(16))) Lds {  // AES table first => LDS address 0 (see k_eval)
    uint32_t lt[mf::kTabBytes / 4];
    uint8_t ksbuf[S::ROWS][S::KS_BYTES];
    uint64_t sums[S::ROWS][S::KW];
    uint32_t spanc[S::ROWS][kSpanSlots][8];
  };
  __shared__ Lds lds;
  auto &ksbuf = lds.ksbuf;
  auto &sums = lds.sums;
  auto &spanc = lds.spanc;
  mf::lds_fill_tab(lds.lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lds.lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t rs = threadIdx.x / S::TILE, t = threadIdx.x % S::TILE;
  uint8_t *ks = ksbuf[rs];
  const uint32_t j0 = blockIdx.x * S::TILE;
  const uint32_t j = j0 + t;
  const uint32_t nelem = j0 >= n ? 0u : min((uint32_t)S::TILE, n - j0);
  const uint32_t r0 = blockIdx.y * rows_per_chunk;
  const uint32_t r1 = min(nrows, r0 + rows_per_chunk);
  const uint32_t lane = t & 63, wave = t >> 6;

  uint32_t s[S::KW];
  {
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(sk + (uint64_t)(j < n ? j : 0) * S::L);
#pragma unroll
    for (int l = 0; l < S::KW; l++) s[l] = j < n ? sw[l] : 0u;
  }
  uint32_t *red = reinterpret_cast<uint32_t *>(ks);  // [KW][TILE] words, reuses this half's keystream tile

  TileGeom g{};
  uint32_t nrow = r0 + rs;
  bool nhave = nrow < r1;
  if (nhave) g = tile_geom<LOGQ>(off + (uint64_t)nrow * n * S::CTB, j0, nelem);
  __syncthreads();
  if (nhave) fill_span_table(tab, L, key, g, t, spanc[rs]);
  __syncthreads();
  for (uint32_t rr = r0; rr < r1; rr += S::ROWS) {
    const uint32_t row = nrow;
    const bool have = nhave;
    if (have) expand_tile_to_lds<LOGQ>(tab, L, key, ks, g, t, spanc[rs]);
    const uint32_t head = g.head;
    __syncthreads();
    nrow = rr + S::ROWS + rs;
    nhave = nrow < r1;
    if (nhave) {  // next row's span constants (nobody reads spanc again before three more barriers)
      g = tile_geom<LOGQ>(off + (uint64_t)nrow * n * S::CTB, j0, nelem);
      fill_span_table(tab, L, key, g, t, spanc[rs]);
    }
    uint32_t prod[S::KW];
#pragma unroll
    for (int l = 0; l < S::KW; l++) prod[l] = 0;
    if (have && j < n) {
      uint32_t a[S::KW];
      const uint32_t *kw = reinterpret_cast<const uint32_t *>(ks + head) + t * S::EW;
#pragma unroll
      for (int l = 0; l < S::KW; l++) a[l] = kw[l];
      // prod = low KW words of a * s: product scanning, one 96-bit column accumulator (v_mad_u64_u32 + carry count),
      // no per-term carry chain; columns >= KW are never formed (mod 2^(32 KW))
      uint64_t lo = 0;
      uint32_t hi = 0;
#pragma unroll
      for (int k = 0; k < S::KW; k++) {
#pragma unroll
        for (int u = 0; u <= k; u++) mac96(lo, hi, a[u], s[k - u]);
        prod[k] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
      }
    }
    __syncthreads();  // everyone has read its keystream words
#pragma unroll
    for (int l = 0; l < S::KW; l++) red[l * S::TILE + t] = prod[l];
    __syncthreads();
    for (int l = wave; l < S::KW; l += NW) {
      uint64_t tt = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) tt += red[l * S::TILE + k * 64 + lane];
#pragma unroll
      for (int o = 32; o; o >>= 1) tt += __shfl_xor(tt, o);
      if (lane == 0) sums[rs][l] = tt;
    }
    __syncthreads();
    if (have && t == 0) {
      uint64_t carry = 0;
      uint32_t *o = pb + ((uint64_t)row * gridDim.x + blockIdx.x) * S::KW;
      for (int l = 0; l < S::KW; l++) {
        uint64_t tt = sums[rs][l] + carry;
        o[l] = (uint32_t)tt;
        carry = tt >> 32;
      }
    }
    // the next iteration's expansion overwrites ks/red: every read of red finished before the barrier above;
    // sums[] is rewritten only after three more barriers.
  }
}

// b = (e*p + sum_tiles pb + m) mod 2^(64K)  ->  CT_BYTES little-endian bytes (ct_export, src/lwe.c:115-119)
template <int LOGQ>
__global__ void k_encrypt_finish(const uint32_t *__restrict__ pb, uint32_t ntiles, uint32_t nrows, const uint32_t *__restrict__ msg,
                                 const uint64_t *__restrict__ err, uint8_t *__restrict__ c8) {
  using S = PS<LOGQ>;
  uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  const uint32_t *e = reinterpret_cast<const uint32_t *>(err + (uint64_t)row * S::L);
  uint32_t *o = reinterpret_cast<uint32_t *>(c8 + (uint64_t)row * S::CTB);
  uint64_t carry = msg[row];
  uint64_t mulc = 0;
  for (int l = 0; l < S::KW; l++) {
    uint64_t ep = (uint64_t)e[l] * MFH_P + mulc;  // e*p, word l
    mulc = ep >> 32;
    uint64_t t = carry + (uint32_t)ep;
    for (uint32_t k = 0; k < ntiles; k++) t += pb[((uint64_t)row * ntiles + k) * S::KW + l];
    o[l] = (uint32_t)t;
    carry = t >> 32;
  }
  for (int l = S::KW; l < S::EW; l++) o[l] = 0;
}

// ------------------------------------------------------------------------------------------------------
// small kernels: ct_add / ct_mul_ui / ct_addmul_ui (src/lwe.c:131-157), decrypt (src/lwe.c:105-111), smudge
// ------------------------------------------------------------------------------------------------------
template <int LOGQ, int OP>  // 0 add, 1 mul_ui, 2 addmul_ui
__global__ void k_ct_elementwise(uint64_t *rop, const uint64_t *a, const uint64_t *b, uint32_t x, uint64_t nvalues) {
  using S = PS<LOGQ>;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvalues) return;
  const uint32_t *aw = reinterpret_cast<const uint32_t *>(a + i * S::L);
  uint32_t *rw = reinterpret_cast<uint32_t *>(rop + i * S::L);
  uint64_t carry = 0;
  if (OP == 0) {
    const uint32_t *bw = reinterpret_cast<const uint32_t *>(b + i * S::L);
    for (int l = 0; l < S::KW; l++) {
      uint64_t t = (uint64_t)aw[l] + bw[l] + carry;
      rw[l] = (uint32_t)t;
      carry = t >> 32;
    }
  } else {
    for (int l = 0; l < S::KW; l++) {
      uint64_t t = (uint64_t)aw[l] * x + carry + (OP == 2 ? rw[l] : 0u);
      rw[l] = (uint32_t)t;
      carry = t >> 32;
    }
  }
  for (int l = S::KW; l < 2 * S::L; l++) rw[l] = 0;
}

__device__ __forceinline__ uint32_t words_mod_p(const uint32_t *w, int nw) {
  // 2^32 = 5 (mod p): Horner from the top word
  uint64_t r = 0;
  for (int l = nw - 1; l >= 0; l--) r = (r * 5 + w[l]) % MFH_P;  // r < p, r*5 + w < 2^35: no overflow... (r<2^32)*5+2^32 fits
  return (uint32_t)r;
}

// one workgroup (256 threads) per ciphertext
template <int LOGQ>
__global__ __launch_bounds__(256) void k_decrypt(const uint64_t *__restrict__ sk, const uint64_t *__restrict__ cts, uint32_t n,
                                                 uint32_t *__restrict__ out) {
  using S = PS<LOGQ>;
  __shared__ uint32_t red[S::KW * 256];
  __shared__ uint64_t sums[S::KW];
  const uint64_t *ct = cts + (uint64_t)blockIdx.x * (n + 1) * S::L;
  uint32_t acc[S::KW];
#pragma unroll
  for (int l = 0; l < S::KW; l++) acc[l] = 0;
  for (uint32_t j = threadIdx.x; j < n; j += 256) {
    const uint32_t *a = reinterpret_cast<const uint32_t *>(ct + (uint64_t)j * S::L);
    const uint32_t *s = reinterpret_cast<const uint32_t *>(sk + (uint64_t)j * S::L);
    uint32_t sw[S::KW];
#pragma unroll
    for (int l = 0; l < S::KW; l++) sw[l] = s[l];
#pragma unroll
    for (int u = 0; u < S::KW; u++) {
      uint32_t au = a[u], carry = 0;
#pragma unroll
      for (int v = 0; u + v < S::KW; v++) {
        uint64_t t = (uint64_t)au * sw[v] + acc[u + v] + carry;
        acc[u + v] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
      }
    }
  }
#pragma unroll
  for (int l = 0; l < S::KW; l++) red[l * 256 + threadIdx.x] = acc[l];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = wave; l < S::KW; l += 4) {
    uint64_t t = 0;
    for (int k = 0; k < 4; k++) t += red[l * 256 + k * 64 + lane];
    for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) sums[l] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t dot[S::KW];
    uint64_t carry = 0;
    for (int l = 0; l < S::KW; l++) {
      uint64_t t = sums[l] + carry;
      dot[l] = (uint32_t)t;
      carry = t >> 32;
    }
    // b is taken at full L limbs (ct_import does not reduce it, src/lwe.c:125)
    const uint32_t *b = reinterpret_cast<const uint32_t *>(ct + (uint64_t)n * S::L);
    uint32_t bm = words_mod_p(b, 2 * S::L), dm = words_mod_p(dot, S::KW);
    out[blockIdx.x] = (uint32_t)(((uint64_t)bm + MFH_P - dm) % MFH_P);
  }
}

// rop = (rop + sum_j a[j]*b[j]) mod 2^(64K): mpz_add_dotp (src/lwe.c:20-28).  One workgroup of 256 threads.
template <int LOGQ>
__global__ __launch_bounds__(256) void k_add_dotp(uint64_t *__restrict__ rop, const uint64_t *__restrict__ a, const uint64_t *__restrict__ b,
                                                  uint32_t len) {
  using S = PS<LOGQ>;
  __shared__ uint32_t red[S::KW * 256];
  __shared__ uint64_t sums[S::KW];
  uint32_t acc[S::KW];
#pragma unroll
  for (int l = 0; l < S::KW; l++) acc[l] = 0;
  for (uint32_t j = threadIdx.x; j < len; j += 256) {
    const uint32_t *x = reinterpret_cast<const uint32_t *>(a + (uint64_t)j * S::L);
    const uint32_t *y = reinterpret_cast<const uint32_t *>(b + (uint64_t)j * S::L);
    uint32_t yw[S::KW];
#pragma unroll
    for (int l = 0; l < S::KW; l++) yw[l] = y[l];
#pragma unroll
    for (int u = 0; u < S::KW; u++) {
      uint32_t xu = x[u], carry = 0;
#pragma unroll
      for (int v = 0; u + v < S::KW; v++) {
        uint64_t t = (uint64_t)xu * yw[v] + acc[u + v] + carry;
        acc[u + v] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
      }
    }
  }
#pragma unroll
  for (int l = 0; l < S::KW; l++) red[l * 256 + threadIdx.x] = acc[l];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = wave; l < S::KW; l += 4) {
    uint64_t t = 0;
    for (int k = 0; k < 4; k++) t += red[l * 256 + k * 64 + lane];
    for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) sums[l] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t *r = reinterpret_cast<uint32_t *>(rop);
    uint64_t carry = 0;
    for (int l = 0; l < S::KW; l++) {
      uint64_t t = sums[l] + carry + r[l];
      r[l] = (uint32_t)t;
      carry = t >> 32;
    }
    for (int l = S::KW; l < 2 * S::L; l++) r[l] = 0;
  }
}

template <int LOGQ>
__global__ void k_smudge(uint64_t *cts, uint32_t n, const uint32_t *__restrict__ up /*count x KW words*/, const uint8_t *__restrict__ sign,
                         uint32_t count) {
  using S = PS<LOGQ>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t *b = reinterpret_cast<uint32_t *>(cts + ((uint64_t)i * (n + 1) + n) * S::L);
  const uint32_t *u = up + (uint64_t)i * S::KW;
  if (sign[i] & 1) {
    uint32_t borrow = 0;
    for (int l = 0; l < S::KW; l++) {
      uint64_t t = (uint64_t)b[l] - u[l] - borrow;
      b[l] = (uint32_t)t;
      borrow = (uint32_t)(t >> 63);
    }
  } else {
    uint32_t carry = 0;
    for (int l = 0; l < S::KW; l++) {
      uint64_t t = (uint64_t)b[l] + u[l] + carry;
      b[l] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
  }
  for (int l = S::KW; l < 2 * S::L; l++) b[l] = 0;
}

// ------------------------------------------------------------------------------------------------------
// SSP: layout conversion and the witness polynomial (src/ssp.c:28-34, src/snark.c:141,147-155)
// ------------------------------------------------------------------------------------------------------
__global__ void k_ssp_reduce(const uint64_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t ncoef) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncoef; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = (uint32_t)(in[i] % MFH_P);
}

// partial[g][k] = sum over the g-th share of selected rows of v_row[k]   (uint64, no reduction needed: < 2^32 * rows)
__global__ __launch_bounds__(256) void k_witness_partial(const uint32_t *__restrict__ ssp, const uint32_t *__restrict__ rows, uint32_t nsel,
                                                         uint32_t d, uint64_t *__restrict__ partial) {
  const uint32_t k4 = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 coefficients
  if (k4 * 4 >= d) return;
  const uint32_t G = gridDim.y, g = blockIdx.y;
  uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (uint32_t i = g; i < nsel; i += G) {
    const uint4 v = *reinterpret_cast<const uint4 *>(ssp + (uint64_t)rows[i] * d + (uint64_t)k4 * 4);
    s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
  }
  uint64_t *o = partial + (uint64_t)g * d + (uint64_t)k4 * 4;
  o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
}

// the same partial sums with generator-defined rows: every thread makes 4 consecutive coefficients of each selected row
__global__ __launch_bounds__(256) void k_witness_partial_prg(uint64_t seed, const uint32_t *__restrict__ rows, uint32_t nsel, uint32_t d,
                                                             uint64_t *__restrict__ partial) {
  const uint32_t k4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k4 * 4 >= d) return;
  const uint32_t G = gridDim.y, g = blockIdx.y;
  const uint32_t k = k4 * 4;
  uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (uint32_t i = g; i < nsel; i += G) {
    const uint32_t rk = mf::ssp_prg_rowkey(seed, rows[i]);
    s0 += mf::ssp_prg_raw(rk, k);  // raw values: congruent to the coefficients mod p, the sums are reduced by the finish kernels
    s1 += mf::ssp_prg_raw(rk, k + 1);
    s2 += mf::ssp_prg_raw(rk, k + 2);
    s3 += mf::ssp_prg_raw(rk, k + 3);
  }
  uint64_t *o = partial + (uint64_t)g * d + k;
  o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
}
// The witness pass for NB statements at once: every selected SSP row is read ONCE and added into the accumulators of the statements
// whose bit selects it.  list[i] = {slot, mask}: bit b of mask = statement b selects the row (uniform per row: scalar branches).
// partial[(b * G + g) * d + k]: the per-statement layout k_witness_finish reads.
template <int NB>
__global__ __launch_bounds__(256) void k_witness_partial_multi(const uint32_t *__restrict__ ssp, const uint2 *__restrict__ list, uint32_t nsel,
                                                               uint32_t d, uint64_t *__restrict__ partial) {
  const uint32_t k4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k4 * 4 >= d) return;
  const uint32_t G = gridDim.y, g = blockIdx.y;
  uint64_t acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0;
  auto add = [&](const uint4 &v, uint32_t mask) {
#pragma unroll
    for (int b = 0; b < NB; b++)
      if ((mask >> b) & 1) { acc[b][0] += v.x; acc[b][1] += v.y; acc[b][2] += v.z; acc[b][3] += v.w; }
  };
  const uint32_t *col = ssp + (uint64_t)k4 * 4;
  uint32_t i = g;
  for (; i + 3 * G < nsel; i += 4 * G) {  // four rows in flight
    const uint2 e0 = list[i], e1 = list[i + G], e2 = list[i + 2 * G], e3 = list[i + 3 * G];
    const uint4 v0 = *reinterpret_cast<const uint4 *>(col + (uint64_t)e0.x * d), v1 = *reinterpret_cast<const uint4 *>(col + (uint64_t)e1.x * d);
    const uint4 v2 = *reinterpret_cast<const uint4 *>(col + (uint64_t)e2.x * d), v3 = *reinterpret_cast<const uint4 *>(col + (uint64_t)e3.x * d);
    add(v0, e0.y); add(v1, e1.y); add(v2, e2.y); add(v3, e3.y);
  }
  for (; i < nsel; i += G) {
    const uint2 e = list[i];
    add(*reinterpret_cast<const uint4 *>(col + (uint64_t)e.x * d), e.y);
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    uint64_t *o = partial + ((uint64_t)b * G + g) * d + (uint64_t)k4 * 4;
    o[0] = acc[b][0]; o[1] = acc[b][1]; o[2] = acc[b][2]; o[3] = acc[b][3];
  }
}
// the same with generator-defined rows (csrc/ssp_prg.hpp): every selected row is GENERATED once per NB statements
template <int NB>
__global__ __launch_bounds__(256) void k_witness_partial_multi_prg(uint64_t seed, const uint2 *__restrict__ list, uint32_t nsel, uint32_t d,
                                                                   uint64_t *__restrict__ partial) {
  const uint32_t k4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k4 * 4 >= d) return;
  const uint32_t G = gridDim.y, g = blockIdx.y, k = k4 * 4;
  uint64_t acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0;
  for (uint32_t i = g; i < nsel; i += G) {
    const uint2 e = list[i];
    const uint32_t rk = mf::ssp_prg_rowkey(seed, e.x);
    const uint32_t v0 = mf::ssp_prg_raw(rk, k), v1 = mf::ssp_prg_raw(rk, k + 1), v2 = mf::ssp_prg_raw(rk, k + 2), v3 = mf::ssp_prg_raw(rk, k + 3);
#pragma unroll
    for (int b = 0; b < NB; b++)
      if ((e.y >> b) & 1) { acc[b][0] += v0; acc[b][1] += v1; acc[b][2] += v2; acc[b][3] += v3; }
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    uint64_t *o = partial + ((uint64_t)b * G + g) * d + k;
    o[0] = acc[b][0]; o[1] = acc[b][1]; o[2] = acc[b][2]; o[3] = acc[b][3];
  }
}
// materialise generator-defined slots [first, first+nslots) as a dense uint32 image (tests; small instances)
__global__ void k_ssp_prg_fill(uint64_t seed, uint32_t first_slot, uint32_t d, uint64_t total, uint32_t *__restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t slot = first_slot + (uint32_t)(i / d), k = (uint32_t)(i % d);
    out[i] = mf::ssp_prg_coeff(mf::ssp_prg_rowkey(seed, slot), k);
  }
}
// t = v_0 + (summed selected rows) - 1: random_ssp's definition (src/ssp.c:59-71), for a generator-defined SSP
__global__ void k_ssp_prg_make_t(uint64_t seed, const uint64_t *__restrict__ partial, uint32_t G, uint32_t d, uint32_t *__restrict__ t) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  uint64_t s = mf::ssp_prg_coeff(mf::ssp_prg_rowkey(seed, 1), k);  // v_0 = slot 1
  for (uint32_t g = 0; g < G; g++) s = (s + partial[(uint64_t)g * d + k] % MFH_P) % MFH_P;
  if (k == 0) s = (s + MFH_P - 1) % MFH_P;
  t[k] = (uint32_t)s;
}

__global__ void k_witness_finish(const uint32_t *__restrict__ ssp, const uint64_t *__restrict__ partial, uint32_t G, uint32_t d, uint32_t delta,
                                 uint32_t *__restrict__ w) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  uint64_t s = ((uint64_t)ssp[k] * delta) % MFH_P;  // slot 0 = t
  for (uint32_t g = 0; g < G; g++) s = (s + partial[(uint64_t)g * d + k] % MFH_P) % MFH_P;
  w[k] = (uint32_t)s;
}

__global__ void k_witness_lanes(const uint64_t *__restrict__ partial, uint32_t G, uint32_t d, uint64_t *__restrict__ lanes) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  uint64_t s = 0;
  for (uint32_t g = 0; g < G; g++) s = (s + partial[(uint64_t)g * d + k] % MFH_P) % MFH_P;
  lanes[k] = s;
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
extern "C" {

const char *mfh_version(void) { return "mfhip 0.1 (gfx950)"; }

int mfh_ctx_create(mfh_ctx **out, int device, const mfh_params *P) {
  if (!out || !P) return MFH_EINVAL;
  *out = nullptr;
  if (P->logq != 736 && P->logq != 1472) return MFH_EUNSUPPORTED;
  if (P->n == 0 || P->n > (1u << 20)) return MFH_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MFH_EDEVICE;
  if (hipSetDevice(device) != hipSuccess) return MFH_EDEVICE;
  mfh_ctx *c = new mfh_ctx();
  c->P = *P;
  c->device = device;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->ncu = (uint32_t)ncu;
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return MFH_EDEVICE;
  }
  c->stream = c->own_stream;
  uint32_t t0[512];  // T0, then T3 = rotl24(T0) for the one-lookup-in-sixteen that the AES of the 8-waves-per-SIMD kernels gathers from global memory (aes_dev.hpp)
  mf::make_t0_le(t0);
  for (int a = 0; a < 256; a++) t0[256 + a] = (t0[a] << 24) | (t0[a] >> 8);
  if (hipMalloc(&c->d_t0, sizeof t0) != hipSuccess || hipMemcpy(c->d_t0, t0, sizeof t0, hipMemcpyHostToDevice) != hipSuccess) {
    mfh_ctx_destroy(c);
    return MFH_EDEVICE;
  }
  *out = c;
  return MFH_OK;
}

void mfh_ctx_destroy(mfh_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  mfh_poly_destroy(c);
  upload_free(c);
  if (c->sample_tmp) hipFree(c->sample_tmp);
  if (c->ev_sample) hipEventDestroy(c->ev_sample);
  pin_free(c->pin_rows);
  for (auto &b : c->pin_wring) pin_free(b);
  pin_free(c->pin_cw);
  pin_free(c->pin_smudge);
  if (c->side) hipStreamSynchronize(c->side);
  if (c->ws) hipFree(c->ws);
  if (c->wws) hipFree(c->wws);
  if (c->ws2) hipFree(c->ws2);
  if (c->ws3) hipFree(c->ws3);
  if (c->mm_sync) hipFree(c->mm_sync);
  for (hipEvent_t e : c->ev_round) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_sgdone) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_cdone) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_rdone) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_wdone) hipEventDestroy(e);
  if (c->lazy) hipFree(c->lazy);
  if (c->aux) hipFree(c->aux);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->ev_chain) hipEventDestroy(c->ev_chain);
  if (c->ev_chain_done) hipEventDestroy(c->ev_chain_done);
  if (c->side) hipStreamDestroy(c->side);
  if (c->side2) hipStreamDestroy(c->side2);
  if (c->d_msg) hipFree(c->d_msg);
  if (c->d_prover) hipFree(c->d_prover);
  if (c->d_batch) hipFree(c->d_batch);
  if (c->batch_img) hipFree(c->batch_img);
  if (c->ssp_frag) hipFree(c->ssp_frag);
  if (c->d_t0) hipFree(c->d_t0);
  for (auto &t : c->timed) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
  for (auto e : c->ev_pool) hipEventDestroy(e);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}

int mfh_set_stream(mfh_ctx *c, void *s) {
  if (!c) return MFH_EINVAL;
  c->stream = (hipStream_t)s;  // NULL is HIP's default (null) stream, which torch uses by default
  return MFH_OK;
}

int mfh_sync(mfh_ctx *c) {
  if (!c) return MFH_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MFH_OK;
}

int mfh_scrub_staging(mfh_ctx *c) {
  if (!c) return MFH_EINVAL;
  pin_scrub(c->pin_rows);
  pin_scrub(c->pin_cw);
  pin_scrub(c->pin_smudge);
  for (auto &b : c->pin_wring) pin_scrub(b);
  return MFH_OK;
}

const char *mfh_last_error(const mfh_ctx *c) { return c ? c->err.c_str() : "null context"; }
size_t mfh_workspace_bytes(const mfh_ctx *c) { return c ? c->ws_bytes : 0; }
int mfh_set_overlap(mfh_ctx *c, int en) {
  if (!c) return MFH_EINVAL;
  c->overlap = en != 0;
  c->overlap_mode = en > 1 ? en : 1;
  return MFH_OK;
}

int mfh_set_expand_path(mfh_ctx *c, int path) {
  if (!c || path < 0 || path > 1) return MFH_EINVAL;
  c->expand_path = path;
  return MFH_OK;
}

int mfh_set_encrypt_path(mfh_ctx *c, int path) {
  if (!c || path < 0 || path > 2) return MFH_EINVAL;
  c->enc_path = path;
  return MFH_OK;
}

int mfh_set_eval_path(mfh_ctx *c, int path) {
  if (!c || path < 0 || path > 1) return MFH_EINVAL;
  c->eval_path = path;
  return MFH_OK;
}
int mfh_set_encrypt_chunks(mfh_ctx *c, uint32_t chunks) {
  if (!c || chunks > 64) return MFH_EINVAL;
  c->enc_chunks = chunks;
  return MFH_OK;
}
int mfh_set_witness_per(mfh_ctx *c, uint32_t statements) {
  if (!c || (statements && (statements < 32 || statements > 256))) return MFH_EINVAL;
  c->witness_per = statements;
  return MFH_OK;
}
int mfh_set_batch_slabs(mfh_ctx *c, uint32_t nslabs) {
  if (!c || nslabs > 256) return MFH_EINVAL;
  c->batch_slabs = nslabs;
  return MFH_OK;
}

int mfh_set_batch_launch(mfh_ctx *c, uint32_t groups_per_launch, int merge_regions) {
  if (!c || groups_per_launch < 1 || groups_per_launch > 8) return MFH_EINVAL;
  c->batch_ngl = groups_per_launch;
  c->batch_merge = merge_regions != 0;
  return MFH_OK;
}

int mfh_set_batch_bw(mfh_ctx *c, int merged) {
  if (!c) return MFH_EINVAL;
  c->batch_bw_merged = merged != 0;
  return MFH_OK;
}

int mfh_set_batch_image(mfh_ctx *c, int en) {
  if (!c) return MFH_EINVAL;
  c->batch_image = en != 0;
  if (!en && c->batch_img) {
    HIP_TRY(c, hipDeviceSynchronize());
    hipFree(c->batch_img);
    c->batch_img = nullptr;
    c->batch_img_bytes = 0;
  }
  return MFH_OK;
}

int mfh_set_timing(mfh_ctx *c, int en) {
  if (!c) return MFH_EINVAL;
  c->timing = en != 0;
  return MFH_OK;
}
static int timing_kind(const char *which) {
  if (!strcmp(which, "keystream")) return 0;
  if (!strcmp(which, "eval1")) return 1;
  if (!strcmp(which, "eval2")) return 2;
  if (!strcmp(which, "encrypt")) return 3;
  if (!strcmp(which, "eval")) return 12;  // either eval flavour
  if (!strcmp(which, "expand")) return 4;
  if (!strcmp(which, "mac1")) return 5;
  if (!strcmp(which, "mac2")) return 6;
  if (!strcmp(which, "evalmm")) return 7;
  if (!strcmp(which, "evalmm_resident")) return 8;
  if (!strcmp(which, "expandmm")) return 9;
  if (!strcmp(which, "decrypt")) return 11;       // k_decrypt_mm (full ciphertexts from HBM)
  if (!strcmp(which, "decrypt_rows")) return 13;  // k_encrypt_mm run for mfh_decrypt_rows (seed-compressed ciphertexts)
  if (!strcmp(which, "mmstream_bw")) return 14;      // the streaming launch that serves b_w of several super-groups (mfh_set_batch_bw)
  if (!strcmp(which, "mmstream_rounds")) return 10;  // the streaming launches that serve several groups (a subset of "evalmm_resident")
  if (!strcmp(which, "mmstream_rounds_persistent")) return 110;  // ... those of them that ran the persistent one-workgroup-per-CU grid (k_mmstream_p / k_mmstream_w)
  if (!strcmp(which, "mmstream_bw_persistent")) return 114;      // ... and of "mmstream_bw" (k_mmstream_pb)
  return -1;
}

int mfh_timing_drain(mfh_ctx *c, const char *which, uint64_t *count, double *total_ms, uint64_t *total_rows, float *last_ms) {
  if (!c || !which) return MFH_EINVAL;
  const int kind = timing_kind(which);
  if (kind < 0) return MFH_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->side) HIP_TRY(c, hipStreamSynchronize(c->side));
  uint64_t n = 0, rows = 0, work = 0;
  double tot = 0;
  float last = -1.f;
  std::vector<mfh_ctx::Timed> keep;
  std::vector<std::pair<float, float>> spans;  // [start, end) of every matching launch, relative to the first one's start event
  hipEvent_t base = nullptr;
  for (auto &t : c->timed) {
    const bool match = t.kind == kind || (kind == 12 && (t.kind == 1 || t.kind == 2)) || (kind == 8 && (t.kind == 10 || t.kind == 14)) || (kind >= 100 && t.kind == kind - 100 && t.sub == 1);
    if (!match) { keep.push_back(t); continue; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
      n++; tot += ms; rows += t.rows; work += t.work; last = ms;
      if (!base) base = t.e0;
      float s0 = 0;
      if (hipEventElapsedTime(&s0, base, t.e0) == hipSuccess) spans.emplace_back(s0, s0 + ms);
    }
  }
  // launches of one kind may overlap (mfh_prove_batch runs two streams): the busy time is the union of their spans
  std::sort(spans.begin(), spans.end());
  double busy = 0;
  float hi = -1e30f;
  for (auto &sp : spans) {
    if (sp.second <= hi) continue;
    busy += sp.second - std::max(sp.first, hi);
    hi = sp.second;
  }
  c->last_busy_ms = busy;
  c->last_work_rows = work;
  for (auto &t : c->timed) {
    const bool match = t.kind == kind || (kind == 12 && (t.kind == 1 || t.kind == 2)) || (kind == 8 && (t.kind == 10 || t.kind == 14)) || (kind >= 100 && t.kind == kind - 100 && t.sub == 1);
    if (!match) continue;
    c->ev_pool.push_back(t.e0);
    c->ev_pool.push_back(t.e1);
  }
  c->timed.swap(keep);
  if (count) *count = n;
  if (total_ms) *total_ms = tot;
  if (total_rows) *total_rows = rows;
  if (last_ms) *last_ms = last;
  return MFH_OK;
}

double mfh_timing_busy_ms(const mfh_ctx *c) { return c ? c->last_busy_ms : -1.0; }
uint64_t mfh_timing_work_rows(const mfh_ctx *c) { return c ? c->last_work_rows : 0; }

float mfh_last_kernel_ms(mfh_ctx *c, const char *which) {
  float last = -1.f;
  if (mfh_timing_drain(c, which, nullptr, nullptr, nullptr, &last) != MFH_OK) return -1.f;
  return last;
}

int mfh_set_seed(mfh_ctx *c, const uint8_t seed[40]) {
  if (!c || !seed) return MFH_EINVAL;
  mf::expand_key(c->key, seed);
  c->have_seed = true;
  return MFH_OK;
}

#define NEED_SEED(c)                                  \
  do {                                                \
    if (!(c)->have_seed) {                            \
      (c)->err = "mfh_set_seed has not been called";  \
      return MFH_EINVAL;                              \
    }                                                 \
  } while (0)

#define DISPATCH_LOGQ(c, CALL736, CALL1472) \
  do {                                      \
    if ((c)->P.logq == 736) {               \
      CALL736;                              \
    } else {                                \
      CALL1472;                             \
    }                                       \
  } while (0)

int mfh_digest128(mfh_ctx *c, const void *d_buf, size_t nbytes, uint64_t h_digest[2]) {
  if (!c || (!d_buf && nbytes) || !h_digest || (reinterpret_cast<uintptr_t>(d_buf) & 3)) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = aux_reserve(c, 16);
  if (rc) return rc;
  HIP_TRY(c, hipMemsetAsync(c->aux, 0, 16, c->stream));
  const uint64_t nwords = (nbytes + 3) / 4;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (nwords + 255) / 256);
  if (grid) hipLaunchKernelGGL(k_digest128, dim3(grid), dim3(256), 0, c->stream, (const uint8_t *)d_buf, (uint64_t)nbytes, (unsigned long long *)c->aux);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(h_digest, c->aux, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MFH_OK;
}

int mfh_keystream(mfh_ctx *c, uint64_t off, void *d_out, size_t nbytes) {
  if (!c || (!d_out && nbytes)) return MFH_EINVAL;
  NEED_SEED(c);
  if (!nbytes) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t nblk = ((off & 15) + nbytes + 15) >> 4;
  uint32_t grid = (uint32_t)std::min<uint64_t>((nblk + 1023) / 1024, 256 * 2);
  {
    Timer t(c, 0, nblk);
    hipLaunchKernelGGL(k_keystream, dim3(grid), dim3(1024), 0, c->stream, c->key, c->d_t0, off, (uint8_t *)d_out, (uint64_t)nbytes);
  }
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_sample_rows(mfh_ctx *c, uint64_t off, size_t nrows, uint64_t *d_out) {
  if (!c || (!d_out && nrows)) return MFH_EINVAL;
  NEED_SEED(c);
  if (!nrows) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t nelem = (uint64_t)nrows * c->P.n;
  const uint64_t bytes = nelem * (c->P.logq / 8);
  // the raw stream bytes of the rows: up to 16 MiB (118 rows at logq = 736; the shim's ct_import / regev_encrypt2 ask for ONE row per call) in a buffer the context keeps --
  // no allocation, no synchronisation, the call stays asynchronous like every other --; larger requests allocate and wait as before
  const bool kept = bytes <= ((uint64_t)16 << 20);
  void *tmp = nullptr;
  if (kept) {
    if (c->sample_bytes < bytes) {
      if (c->sample_tmp) { hipDeviceSynchronize(); hipFree(c->sample_tmp); c->sample_tmp = nullptr; c->sample_bytes = 0; }
      HIP_TRY(c, hipMalloc(&c->sample_tmp, bytes));
      c->sample_bytes = bytes;
    }
    tmp = c->sample_tmp;
    // the buffer is the context's, the stream is whatever the caller set for THIS call: a previous call on another stream may still be repacking out of it
    if (!c->ev_sample) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_sample, hipEventDisableTiming));
    else HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_sample, 0));
  } else {
    HIP_TRY(c, hipMalloc(&tmp, bytes));
  }
  int rc = mfh_keystream(c, off, tmp, bytes);
  if (rc == MFH_OK) {
    const uint64_t total = nelem * 2 * ((c->P.logq + 63) / 64);
    dim3 grid((uint32_t)((total + 255) / 256));
    DISPATCH_LOGQ(c, hipLaunchKernelGGL(k_repack_values<736>, grid, dim3(256), 0, c->stream, (const uint32_t *)tmp, (uint32_t *)d_out, nelem),
                  hipLaunchKernelGGL(k_repack_values<1472>, grid, dim3(256), 0, c->stream, (const uint32_t *)tmp, (uint32_t *)d_out, nelem));
    if (hipGetLastError() != hipSuccess) rc = MFH_EDEVICE;
  }
  if (kept) {
    if (hipEventRecord(c->ev_sample, c->stream) != hipSuccess && rc == MFH_OK) rc = MFH_EDEVICE;  // (the last reader of the kept buffer)
  } else {
    hipStreamSynchronize(c->stream);
    hipFree(tmp);
  }
  return rc;
}

}  // extern "C"

template <int LOGQ, int OP>
static int ct_elementwise(mfh_ctx *c, uint64_t *rop, const uint64_t *a, const uint64_t *b, uint32_t x, size_t count) {
  const uint64_t nvalues = (uint64_t)count * (c->P.n + 1);
  if (!nvalues) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL((k_ct_elementwise<LOGQ, OP>), dim3((uint32_t)((nvalues + 255) / 256)), dim3(256), 0, c->stream, rop, a, b, x, nvalues);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

extern "C" {

int mfh_ct_add(mfh_ctx *c, uint64_t *rop, const uint64_t *a, const uint64_t *b, size_t count) {
  if (!c || !rop || !a || !b) return MFH_EINVAL;
  DISPATCH_LOGQ(c, return (ct_elementwise<736, 0>(c, rop, a, b, 0, count)), return (ct_elementwise<1472, 0>(c, rop, a, b, 0, count)));
}
int mfh_ct_mul_ui(mfh_ctx *c, uint64_t *rop, const uint64_t *a, uint32_t x, size_t count) {
  if (!c || !rop || !a) return MFH_EINVAL;
  if (x >= MFH_P) { c->err = "ct_mul_ui: scalar must be < p (src/lwe.c:133)"; return MFH_EINVAL; }
  DISPATCH_LOGQ(c, return (ct_elementwise<736, 1>(c, rop, a, nullptr, x, count)), return (ct_elementwise<1472, 1>(c, rop, a, nullptr, x, count)));
}
int mfh_ct_addmul_ui(mfh_ctx *c, uint64_t *rop, const uint64_t *a, uint32_t x, size_t count) {
  if (!c || !rop || !a) return MFH_EINVAL;
  if (x >= MFH_P) { c->err = "ct_addmul_ui: scalar must be < p (src/lwe.c:143)"; return MFH_EINVAL; }
  DISPATCH_LOGQ(c, return (ct_elementwise<736, 2>(c, rop, a, nullptr, x, count)), return (ct_elementwise<1472, 2>(c, rop, a, nullptr, x, count)));
}

}  // extern "C"

// number of row chunks: one workgroup per CU per tile row is the sweet spot (64 KiB table + 2 keystream tiles fill the LDS)
static uint32_t pick_chunks(uint32_t nrows, uint32_t ntiles, uint32_t rows_per_iter) {
  uint32_t target = std::max(1u, 255u / ntiles);  // one workgroup per CU for the whole launch (LDS allows only one at a time anyway)
  uint32_t maxc = std::max(1u, (nrows + rows_per_iter * 4 - 1) / (rows_per_iter * 4));  // >= 4 iterations per chunk to amortise the table fill
  return std::min(target, maxc);
}

template <int LOGQ>
static int eval_rows(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *c8, const uint32_t *c0, const uint32_t *c1, uint64_t *rop0,
                     uint64_t *rop1, int accumulate) {
  using S = PS<LOGQ>;
  const uint32_t n = c->P.n;
  const int nacc = c1 ? 2 : 1;
  // the tile kernel (k_eval) by default; mfh_set_eval_path(ctx, 1) at logq 736: the wave-autonomous kernel (k_eval_w) -- 64-coordinate column chunks x
  // row chunks, one wave each, 16 waves per workgroup, the launch sized to one workgroup per CU (the LDS admits one).  Measured 4 % slower (see k_eval_w).
  const bool wavek = LOGQ == 736 && c->eval_path == 1;
  const uint32_t ncc = (n + 1 + 63) / 64;
  uint32_t nrc = std::max(1u, (uint32_t)((uint64_t)c->ncu * kEvalwWaves / ncc));
  nrc = std::min(nrc, std::max(1u, ((uint32_t)nrows + 7) / 8));  // >= 8 rows per wave
  const uint32_t ntiles = (n + 1 + S::TILE - 1) / S::TILE;
  const uint32_t nchunks = pick_chunks((uint32_t)nrows, ntiles, S::ROWS);
  const uint32_t nslabs = wavek ? nrc : nchunks;  // (k_eval folds the two row halves of a workgroup before they leave the kernel)
  const uint32_t NJ = wavek ? ncc * 64 : ntiles * S::TILE;
  const size_t part_bytes = (size_t)nslabs * nacc * S::KW * NJ * 4;
  const uint32_t NG = 8;  // slab groups of the first reduction stage
  const size_t idx_bytes = (((size_t)nrows + 1) * 4 + 255) & ~(size_t)255;
  int rc = ws_reserve(c, part_bytes + idx_bytes);
  if (rc) return rc;
  rc = lazy_reserve(c, (size_t)nacc * S::KW * (size_t)std::max(NJ, ntiles * S::TILE) * 8);
  if (rc) return rc;
  uint32_t *part = (uint32_t *)c->ws;
  uint64_t *lazy = c->lazy;
  uint32_t *cnt = c->lazy_cnt;
  uint32_t *idx = nullptr;
  if (!c->eval_dense) {  // drop the rows whose coefficients are all zero (the reference expands them only to advance its stream)
    idx = (uint32_t *)((uint8_t *)c->ws + part_bytes);
    hipLaunchKernelGGL(k_compact_rows, dim3(((uint32_t)nrows + 255) / 256), dim3(256), 0, c->stream, c0, c1, (uint32_t)nrows, idx, cnt);
    HIP_TRY(c, hipGetLastError());
  }
  {
    Timer t(c, nacc, nrows);
    if (wavek) {
      const dim3 grid((ncc * nrc + kEvalwWaves - 1) / kEvalwWaves);
      if (nacc == 2)
        hipLaunchKernelGGL(k_eval_w<2>, grid, dim3(kEvalwWaves * 64), 0, c->stream, c->key, c->d_t0, off, n, idx, cnt, (uint32_t)nrows, c8, c0, c1, ncc, nrc, part);
      else
        hipLaunchKernelGGL(k_eval_w<1>, grid, dim3(kEvalwWaves * 64), 0, c->stream, c->key, c->d_t0, off, n, idx, cnt, (uint32_t)nrows, c8, c0, c1, ncc, nrc, part);
    } else if (nacc == 2)
      hipLaunchKernelGGL((k_eval<LOGQ, 2>), dim3(ntiles, nchunks), dim3(S::THREADS), 0, c->stream, c->key, c->d_t0, off, n, idx, cnt,
                         (uint32_t)nrows, c8, c0, c1, part);
    else
      hipLaunchKernelGGL((k_eval<LOGQ, 1>), dim3(ntiles, nchunks), dim3(S::THREADS), 0, c->stream, c->key, c->d_t0, off, n, idx, cnt,
                         (uint32_t)nrows, c8, c0, c1, part);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL(k_eval_reduce_sum<LOGQ>, dim3((NJ + 255) / 256, S::KW * nacc, NG), dim3(256), 0, c->stream, part, nslabs, (uint32_t)nacc, NJ, lazy);
  hipLaunchKernelGGL(k_eval_reduce_carry<LOGQ>, dim3((NJ + 255) / 256, nacc), dim3(256), 0, c->stream, lazy, cnt, (uint32_t)nacc, NJ, n, rop0, rop1,
                     accumulate);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

template <int LOGQ>
static int crs_expand(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *c8, uint8_t *out) {
  using S = PS<LOGQ>;
  const uint32_t n = c->P.n;
  const uint32_t RS = RL<LOGQ>::rs(n);
  const uint32_t ntiles = (RS + S::TILE - 1) / S::TILE;
  const uint32_t nchunks = pick_chunks((uint32_t)nrows, ntiles, S::ROWS);
  uint32_t rpc = ((uint32_t)nrows + nchunks - 1) / nchunks;
  rpc = (rpc + S::ROWS - 1) / S::ROWS * S::ROWS;
  const uint32_t gy = ((uint32_t)nrows + rpc - 1) / rpc;
  {
    Timer t(c, 4, nrows);
    hipLaunchKernelGGL(k_expand<LOGQ>, dim3(ntiles, gy), dim3(S::THREADS), 0, c->stream, c->key, c->d_t0, off, n, (uint32_t)nrows, rpc, c8, out);
  }
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

template <int LOGQ>
static int eval_rows_resident(mfh_ctx *c, const uint8_t *rows, size_t row_base, size_t nrows, const uint32_t *c0, const uint32_t *c1,
                              uint64_t *rop0, uint64_t *rop1, int accumulate) {
  using S = PS<LOGQ>;
  const uint32_t n = c->P.n;
  const uint32_t RS = RL<LOGQ>::rs(n);
  const int nacc = c1 ? 2 : 1;
  const uint32_t gx = (RS + 255) / 256;
  const uint32_t NJ = gx * 256;
  const uint32_t nchunks = std::max(1u, std::min((uint32_t)nrows, (256u * 8u) / gx));  // ~8 workgroups of 4 waves per CU
  const size_t part_bytes = (size_t)nchunks * nacc * S::KW * NJ * 4;
  const uint32_t NG = 8;  // slab groups of the first reduction stage
  const size_t idx_bytes = (((size_t)nrows + 1) * 4 + 255) & ~(size_t)255;
  int rc = ws_reserve(c, part_bytes + idx_bytes);
  if (rc) return rc;
  rc = lazy_reserve(c, (size_t)nacc * S::KW * NJ * 8);
  if (rc) return rc;
  uint32_t *part = (uint32_t *)c->ws;
  uint64_t *lazy = c->lazy;
  uint32_t *cnt = c->lazy_cnt;
  uint32_t *idx = nullptr;
  if (!c->eval_dense) {
    idx = (uint32_t *)((uint8_t *)c->ws + part_bytes);
    hipLaunchKernelGGL(k_compact_rows, dim3(((uint32_t)nrows + 255) / 256), dim3(256), 0, c->stream, c0, c1, (uint32_t)nrows, idx, cnt);
    HIP_TRY(c, hipGetLastError());
  }
  {
    Timer t(c, 4 + nacc, nrows);
    if (nacc == 2)
      hipLaunchKernelGGL((k_mac_resident<LOGQ, 2>), dim3(gx, nchunks), dim3(256), 0, c->stream, rows, n, idx, cnt, (uint32_t)nrows, (uint32_t)row_base,
                         c0, c1, part, NJ);
    else
      hipLaunchKernelGGL((k_mac_resident<LOGQ, 1>), dim3(gx, nchunks), dim3(256), 0, c->stream, rows, n, idx, cnt, (uint32_t)nrows, (uint32_t)row_base,
                         c0, c1, part, NJ);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL(k_eval_reduce_sum<LOGQ>, dim3((NJ + 255) / 256, S::KW * nacc, NG), dim3(256), 0, c->stream, part, nchunks, (uint32_t)nacc, NJ, lazy);
  hipLaunchKernelGGL(k_eval_reduce_carry<LOGQ>, dim3((NJ + 255) / 256, nacc), dim3(256), 0, c->stream, lazy, cnt, (uint32_t)nacc, NJ, n, rop0, rop1,
                     accumulate);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

extern "C" {

int mfh_eval_rows(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, const uint32_t *d_coeff0, const uint32_t *d_coeff1,
                  uint64_t *d_rop0, uint64_t *d_rop1, int accumulate) {
  if (!c || !d_rop0 || (nrows && (!d_c8 || !d_coeff0)) || ((d_coeff1 == nullptr) != (d_rop1 == nullptr))) return MFH_EINVAL;
  NEED_SEED(c);
  if (nrows > 0xffffffffu) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  if (nrows == 0) {
    if (!accumulate) {
      size_t bytes = (size_t)(c->P.n + 1) * ((c->P.logq + 63) / 64) * 8;
      HIP_TRY(c, hipMemsetAsync(d_rop0, 0, bytes, c->stream));
      if (d_rop1) HIP_TRY(c, hipMemsetAsync(d_rop1, 0, bytes, c->stream));
    }
    return MFH_OK;
  }
  DISPATCH_LOGQ(c, return eval_rows<736>(c, off, nrows, d_c8, d_coeff0, d_coeff1, d_rop0, d_rop1, accumulate),
                return eval_rows<1472>(c, off, nrows, d_c8, d_coeff0, d_coeff1, d_rop0, d_rop1, accumulate));
}

size_t mfh_resident_row_bytes(const mfh_ctx *c) {
  if (!c) return 0;
  return c->P.logq == 736 ? (size_t)RL<736>::row_bytes(c->P.n) : (size_t)RL<1472>::row_bytes(c->P.n);
}

int mfh_crs_image_set_b(mfh_ctx *c, size_t first_row, size_t nrows, const uint8_t *d_c8, void *d_rows) {
  if (!c || (nrows && (!d_c8 || !d_rows)) || nrows > 0xffffffffu) return MFH_EINVAL;
  if (!nrows) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t *out = (uint8_t *)d_rows + first_row * mfh_resident_row_bytes(c);
  const dim3 grid((uint32_t)((nrows + 255) / 256));
  DISPATCH_LOGQ(c, hipLaunchKernelGGL(k_rows_set_b<736>, grid, dim3(256), 0, c->stream, c->P.n, (uint32_t)nrows, d_c8, out),
                hipLaunchKernelGGL(k_rows_set_b<1472>, grid, dim3(256), 0, c->stream, c->P.n, (uint32_t)nrows, d_c8, out));
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_crs_expand(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, void *d_rows_out) {
  if (!c || (nrows && !d_rows_out) || nrows > 0xffffffffu) return MFH_EINVAL;
  NEED_SEED(c);
  if (!nrows) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  DISPATCH_LOGQ(c, return crs_expand<736>(c, off, nrows, d_c8, (uint8_t *)d_rows_out), return crs_expand<1472>(c, off, nrows, d_c8, (uint8_t *)d_rows_out));
}

int mfh_eval_rows_resident(mfh_ctx *c, const void *d_rows, size_t first_row, size_t nrows, const uint32_t *d_coeff0, const uint32_t *d_coeff1,
                           uint64_t *d_rop0, uint64_t *d_rop1, int accumulate) {
  if (!c || !d_rop0 || (nrows && (!d_rows || !d_coeff0)) || ((d_coeff1 == nullptr) != (d_rop1 == nullptr)) || nrows > 0xffffffffu ||
      first_row > 0xffffffffu)
    return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  if (nrows == 0) {
    if (!accumulate) {
      size_t bytes = (size_t)(c->P.n + 1) * ((c->P.logq + 63) / 64) * 8;
      HIP_TRY(c, hipMemsetAsync(d_rop0, 0, bytes, c->stream));
      if (d_rop1) HIP_TRY(c, hipMemsetAsync(d_rop1, 0, bytes, c->stream));
    }
    return MFH_OK;
  }
  DISPATCH_LOGQ(c, return eval_rows_resident<736>(c, (const uint8_t *)d_rows, first_row, nrows, d_coeff0, d_coeff1, d_rop0, d_rop1, accumulate),
                return eval_rows_resident<1472>(c, (const uint8_t *)d_rows, first_row, nrows, d_coeff0, d_coeff1, d_rop0, d_rop1, accumulate));
}

int mfh_crs_set_resident(mfh_ctx *c, const void *d_rows) {
  if (!c) return MFH_EINVAL;
  c->resident_rows = (const uint8_t *)d_rows;
  c->resident_sharded = false;
  c->resident_nrows = ~0ull;
  return MFH_OK;
}

int mfh_crs_set_resident_prefix(mfh_ctx *c, const void *d_rows, size_t nrows_resident) {
  if (!c) return MFH_EINVAL;
  c->resident_rows = (const uint8_t *)d_rows;
  c->resident_sharded = false;
  c->resident_nrows = nrows_resident;
  return MFH_OK;
}

static void row_share(uint32_t rows, uint32_t rank, uint32_t world, uint32_t &lo, uint32_t &cnt) {
  lo = (uint32_t)((uint64_t)rows * rank / world);
  cnt = (uint32_t)((uint64_t)rows * (rank + 1) / world) - lo;
}

size_t mfh_resident_share_rows(const mfh_ctx *c, uint32_t rank, uint32_t world) {
  if (!c || !world || rank >= world) return 0;
  uint32_t lo, cs, cb;
  row_share(c->P.d, rank, world, lo, cs);
  row_share(c->P.m, rank, world, lo, cb);
  return (size_t)2 * cs + cb;
}

int mfh_crs_expand_share(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, void *d_image) {
  if (!c || !d_crs_c8 || !d_image || !world || rank >= world) return MFH_EINVAL;
  const uint32_t d = c->P.d, m = c->P.m, ctb = c->P.logq / 8;
  const uint64_t ctr_ct = (uint64_t)ctb * c->P.n;
  const size_t rb = mfh_resident_row_bytes(c);
  uint32_t loS, cS, loB, cB;
  row_share(d, rank, world, loS, cS);
  row_share(m, rank, world, loB, cB);
  uint8_t *img = (uint8_t *)d_image;
  int rc = mfh_crs_expand(c, ctr_ct * loS, cS, d_crs_c8 + (size_t)loS * ctb, img);                                   // S share
  if (rc) return rc;
  rc = mfh_crs_expand(c, ctr_ct * ((uint64_t)d + loS), cS, d_crs_c8 + ((size_t)d + loS) * ctb, img + (size_t)cS * rb);  // AS share
  if (rc) return rc;
  return mfh_crs_expand(c, ctr_ct * ((uint64_t)2 * d + loB), cB, d_crs_c8 + ((size_t)2 * d + loB) * ctb, img + (size_t)2 * cS * rb);  // BT+BV share
}

int mfh_crs_set_resident_share(mfh_ctx *c, const void *d_image, uint32_t rank, uint32_t world) {
  if (!c || !world || rank >= world) return MFH_EINVAL;
  c->resident_rows = (const uint8_t *)d_image;
  c->resident_sharded = d_image != nullptr;
  c->res_rank = rank;
  c->res_world = world;
  return MFH_OK;
}

}  // extern "C"

template <int LOGQ>
static int encrypt_rows(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint32_t *msg, const uint64_t *err, uint8_t *c8) {
  using S = PS<LOGQ>;
  const uint32_t n = c->P.n;
  const uint32_t ntiles = (n + S::TILE - 1) / S::TILE;
  const uint32_t nchunks = pick_chunks((uint32_t)nrows, ntiles, S::ROWS);
  uint32_t rpc = ((uint32_t)nrows + nchunks - 1) / nchunks;
  rpc = (rpc + S::ROWS - 1) / S::ROWS * S::ROWS;
  const uint32_t gy = ((uint32_t)nrows + rpc - 1) / rpc;
  const size_t pb_bytes = (size_t)nrows * ntiles * S::KW * 4;
  int rc = ws_reserve(c, pb_bytes);
  if (rc) return rc;
  uint32_t *pb = (uint32_t *)c->ws;
  {
    Timer t(c, 3, nrows);
    hipLaunchKernelGGL(k_encrypt<LOGQ>, dim3(ntiles, gy), dim3(S::THREADS), 0, c->stream, c->key, c->d_t0, off, n, (uint32_t)nrows, rpc, sk, pb);
  }
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL(k_encrypt_finish<LOGQ>, dim3(((uint32_t)nrows + 255) / 256), dim3(256), 0, c->stream, pb, ntiles, (uint32_t)nrows, msg, err, c8);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

extern "C" {

int mfh_encrypt_rows(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *d_sk, const uint32_t *d_msg, const uint64_t *d_err,
                     uint8_t *d_c8_out) {
  if (!c || (nrows && (!d_sk || !d_msg || !d_err || !d_c8_out))) return MFH_EINVAL;
  NEED_SEED(c);
  if (!nrows) return MFH_OK;
  if (nrows > 0xffffffffu) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  // Batches: the dot product on the matrix cores (encmm.hip), one AES block per lane straight into the MFMA.  Needs every row to start at
  // byte 0 or 8 of an AES block with one parity pattern (off and the row length multiples of 8); small batches are cheaper on the VALU
  // kernel (no per-key operand preparation: 0.13 ms; measured break-even near 3000 rows).
  const uint64_t rowlen = (uint64_t)c->P.n * (c->P.logq / 8);
  const bool mm_ok = (off & 7) == 0 && (rowlen & 7) == 0;
  if (c->enc_path == 2 && !mm_ok) { c->err = "mfh_encrypt_rows: the matrix-core kernel needs off and the row length to be multiples of 8"; return MFH_EINVAL; }
  if (mm_ok && (c->enc_path == 2 || (c->enc_path == 0 && nrows >= 4096))) return encrypt_rows_mm(c, off, nrows, d_sk, d_msg, d_err, d_c8_out);
  DISPATCH_LOGQ(c, return encrypt_rows<736>(c, off, nrows, d_sk, d_msg, d_err, d_c8_out),
                return encrypt_rows<1472>(c, off, nrows, d_sk, d_msg, d_err, d_c8_out));
}

int mfh_decrypt(mfh_ctx *c, const uint64_t *d_sk, const uint64_t *d_cts, size_t count, uint32_t *d_out) {
  if (!c || (count && (!d_sk || !d_cts || !d_out))) return MFH_EINVAL;
  if (!count) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  // batches: <a, sk> as a Toeplitz int8 GEMM on the matrix cores (encmm.hip), HBM-bound; small counts stay on the VALU kernel (no per-key operand
  // preparation: 0.13 ms)
  if (c->dec_path == 2 || (c->dec_path == 0 && count >= 4096)) return decrypt_mm(c, d_sk, d_cts, count, d_out);
  DISPATCH_LOGQ(c, hipLaunchKernelGGL(k_decrypt<736>, dim3((uint32_t)count), dim3(256), 0, c->stream, d_sk, d_cts, c->P.n, d_out),
                hipLaunchKernelGGL(k_decrypt<1472>, dim3((uint32_t)count), dim3(256), 0, c->stream, d_sk, d_cts, c->P.n, d_out));
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_decrypt_rows(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *d_sk, const uint8_t *d_c8, uint32_t *d_out) {
  if (!c || (nrows && (!d_sk || !d_c8 || !d_out)) || nrows > 0x7fffffffu) return MFH_EINVAL;
  NEED_SEED(c);
  if (!nrows) return MFH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t rowlen = (uint64_t)c->P.n * (c->P.logq / 8);
  if ((off & 7) != 0 || (rowlen & 7) != 0) { c->err = "mfh_decrypt_rows: off and the row length must be multiples of 8"; return MFH_EUNSUPPORTED; }
  return decrypt_rows_mm(c, off, nrows, d_sk, d_c8, d_out);
}

int mfh_set_decrypt_path(mfh_ctx *c, int path) {
  if (!c || path < 0 || path > 2) return MFH_EINVAL;
  c->dec_path = path;
  return MFH_OK;
}

int mfh_add_dotp(mfh_ctx *c, uint64_t *d_rop, const uint64_t *d_a, const uint64_t *d_b, size_t len) {
  if (!c || !d_rop || (len && (!d_a || !d_b)) || len > 0xffffffffu) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  DISPATCH_LOGQ(c, hipLaunchKernelGGL(k_add_dotp<736>, dim3(1), dim3(256), 0, c->stream, d_rop, d_a, d_b, (uint32_t)len),
                hipLaunchKernelGGL(k_add_dotp<1472>, dim3(1), dim3(256), 0, c->stream, d_rop, d_a, d_b, (uint32_t)len));
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_ct_smudge(mfh_ctx *c, uint64_t *d_cts, size_t count, const uint8_t *h_mag, size_t maglen, const uint8_t *h_sign) {
  if (!c || (count && (!d_cts || !h_mag || !h_sign))) return MFH_EINVAL;
  if (!count) return MFH_OK;
  const uint32_t KW = 2 * (c->P.logq / 64);
  if (maglen + 4 > (size_t)KW * 4) { c->err = "smudge magnitude too wide"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  // u*p on the host (count is 5 per proof): schoolbook by 32-bit words, staged through pinned memory
  const size_t ub = (size_t)count * KW * 4;
  uint8_t *stage = (uint8_t *)pin_acquire(c, c->pin_smudge, ub + count);
  if (!stage) return MFH_ENOMEM;
  uint32_t *up = (uint32_t *)stage;
  for (size_t i = 0; i < count; i++) {
    uint64_t carry = 0;
    for (uint32_t l = 0; l < KW; l++) {
      uint32_t w = 0;
      const size_t o = (size_t)l * 4;
      if (o < maglen) memcpy(&w, h_mag + i * maglen + o, std::min<size_t>(4, maglen - o));
      uint64_t t = (uint64_t)w * MFH_P + carry;
      up[i * KW + l] = (uint32_t)t;
      carry = t >> 32;
    }
  }
  memcpy(stage + ub, h_sign, count);
  int rc = aux_reserve(c, ub + count);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->aux, stage, ub + count, hipMemcpyHostToDevice, c->stream));
  pin_release(c, c->pin_smudge);
  DISPATCH_LOGQ(c, hipLaunchKernelGGL(k_smudge<736>, dim3(((uint32_t)count + 63) / 64), dim3(64), 0, c->stream, d_cts, c->P.n, (const uint32_t *)c->aux,
                                      (const uint8_t *)c->aux + ub, (uint32_t)count),
                hipLaunchKernelGGL(k_smudge<1472>, dim3(((uint32_t)count + 63) / 64), dim3(64), 0, c->stream, d_cts, c->P.n, (const uint32_t *)c->aux,
                                   (const uint8_t *)c->aux + ub, (uint32_t)count));
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

// The SSP comes from pageable host memory (a calloc'ed or mmap'ed buffer of the caller's: 5.7 GB at the default instance).  Handed to hipMemcpy as it is, the runtime
// stages it through its own pinned buffer on ONE thread (about 12 GB/s: 0.45 s of setup()'s 0.55 s).  Here NT host threads (8, $MFH_UPLOAD_THREADS) each take every NT-th chunk of
// UP_CHUNK bytes: memcpy into the thread's own pinned pair, an asynchronous copy and the uint64 -> uint32 reduction (k_ssp_reduce) on the thread's own stream, the
// second buffer being filled while the first one crosses PCIe.  Host work is byte moving only; the reduction mod p stays on the GPU.
namespace {
constexpr size_t UP_CHUNK = (size_t)4 << 20;
constexpr int UP_T = 16;  // lanes that exist; a call uses min(UP_T, $MFH_UPLOAD_THREADS or 8) of them
struct UpLane {
  hipStream_t st = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  uint8_t *pin[2] = {nullptr, nullptr};
  uint8_t *dev[2] = {nullptr, nullptr};
};
struct Uploader {
  UpLane lane[UP_T];
};
}  // namespace
static void upload_free(mfh_ctx *c) {
  Uploader *u = (Uploader *)c->uploader;
  if (!u) return;
  for (auto &l : u->lane) {
    for (int i = 0; i < 2; i++) {
      if (l.pin[i]) hipHostFree(l.pin[i]);
      if (l.dev[i]) hipFree(l.dev[i]);
      if (l.ev[i]) hipEventDestroy(l.ev[i]);
    }
    if (l.st) hipStreamDestroy(l.st);
  }
  delete u;
  c->uploader = nullptr;
}
int mfh_ssp_upload(mfh_ctx *c, const void *h_ssp_u64, uint32_t *d_ssp, size_t first_slot, size_t nslots) {
  if (!c || !h_ssp_u64 || !d_ssp) return MFH_EINVAL;
  c->ssp_frag_src = nullptr;  // derived images of the SSP are stale
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t d = c->P.d;
  const size_t total = nslots * d * 8;  // bytes of uint64 coefficients
  if (!total) return MFH_OK;
  const uint8_t *src = (const uint8_t *)h_ssp_u64 + first_slot * d * 8;
  uint32_t *dst = d_ssp + first_slot * d;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // what the caller queued before (e.g. readers of the old image) is over
  const size_t nchunks = (total + UP_CHUNK - 1) / UP_CHUNK;
  if (nchunks < 16) {  // small images (the debug SSP is 137 KB): one copy on the caller's stream
    void *stage = nullptr;
    HIP_TRY(c, hipMalloc(&stage, total));
    int rc = MFH_OK;
    if (hipMemcpyAsync(stage, src, total, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MFH_EDEVICE;
    if (!rc) hipLaunchKernelGGL(k_ssp_reduce, dim3(2048), dim3(256), 0, c->stream, (const uint64_t *)stage, dst, (uint64_t)(total / 8));
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = MFH_EDEVICE;
    hipFree(stage);
    if (rc) c->err = "ssp upload failed";
    return rc;
  }
  static int nthreads = 0;
  if (!nthreads) {
    const char *e = getenv("MFH_UPLOAD_THREADS");
    const long want = e && *e ? atol(e) : 8;
    nthreads = (int)(want < 1 ? 1 : want > UP_T ? UP_T : want);
  }
  const int NT = nthreads;
  Uploader *u = (Uploader *)c->uploader;
  if (!u) c->uploader = u = new Uploader();
  for (int t = 0; t < NT; t++) {  // the lanes this call uses (made once, kept by the context)
    UpLane &l = u->lane[t];
    if (l.st) continue;  // (a lane with a stream is complete: a lane that fails half-way takes the whole uploader down with it, below)
    bool ok = hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 2; i++)
      ok = hipEventCreateWithFlags(&l.ev[i], hipEventDisableTiming) == hipSuccess && hipHostMalloc((void **)&l.pin[i], UP_CHUNK, hipHostMallocDefault) == hipSuccess &&
           hipMalloc((void **)&l.dev[i], UP_CHUNK) == hipSuccess;
    if (!ok) {  // no half-built lane is left for the next call to find (it would skip it at `if (l.st)` and copy into a null staging buffer)
      (void)hipGetLastError();
      upload_free(c);
      c->err = "ssp upload: no memory for the staging lanes";
      return MFH_ENOMEM;
    }
  }
  int failed[UP_T] = {};
  auto work = [&](int t) {
    if (hipSetDevice(c->device) != hipSuccess) { failed[t] = 1; return; }
    UpLane &l = u->lane[t];
    int slot = 0;
    bool used[2] = {false, false};
    for (size_t k = (size_t)t; k < nchunks; k += (size_t)NT, slot ^= 1) {
      const size_t off = k * UP_CHUNK, nb = std::min(UP_CHUNK, total - off);
      if (used[slot] && hipEventSynchronize(l.ev[slot]) != hipSuccess) { failed[t] = 1; return; }
      memcpy(l.pin[slot], src + off, nb);
      if (hipMemcpyAsync(l.dev[slot], l.pin[slot], nb, hipMemcpyHostToDevice, l.st) != hipSuccess) { failed[t] = 1; return; }
      hipLaunchKernelGGL(k_ssp_reduce, dim3(256), dim3(256), 0, l.st, (const uint64_t *)l.dev[slot], dst + off / 8, (uint64_t)(nb / 8));
      if (hipEventRecord(l.ev[slot], l.st) != hipSuccess) { failed[t] = 1; return; }
      used[slot] = true;
    }
    if (hipStreamSynchronize(l.st) != hipSuccess) failed[t] = 1;
  };
  std::thread th[UP_T];
  int started = 0;
  for (int t = 1; t < NT; t++) {
    try { th[t] = std::thread(work, t); started++; } catch (...) { break; }
  }
  work(0);
  for (int t = started + 1; t < NT; t++) work(t);  // lanes whose thread could not be had run here, one after the other
  for (int t = 1; t <= started; t++) th[t].join();
  for (int t = 0; t < NT; t++)
    if (failed[t] || hipGetLastError() != hipSuccess) { c->err = "ssp upload failed"; return MFH_EDEVICE; }
  return MFH_OK;
}

// shared body: partial[g][k] sums over this rank's share of the selected SSP rows
static int witness_partials(mfh_ctx *c, const mf::SspSrc &src, const uint8_t *h_bits, uint32_t rank, uint32_t world, uint32_t *G_out,
                            uint64_t **partial_out) {
  const uint32_t d = c->P.d, m = c->P.m;
  if (d % 4) { c->err = "d must be a multiple of 4"; return MFH_EINVAL; }
  uint32_t *rows = (uint32_t *)pin_acquire(c, c->pin_rows, (size_t)m * 4 + 4);
  if (!rows) return MFH_ENOMEM;
  uint32_t nall = 0;
  for (uint32_t i = 1; i < m; i++)
    if ((h_bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1) rows[nall++] = i + 1;  // slot of v_i
  // contiguous share of the selected rows
  const uint32_t lo = (uint32_t)((uint64_t)nall * rank / world), hi = (uint32_t)((uint64_t)nall * (rank + 1) / world);
  const uint32_t nsel = hi - lo;
  const uint32_t G = std::max(1u, std::min(64u, nsel / 8 + 1));
  const size_t rows_b = ((size_t)m * 4 + 255) & ~(size_t)255;
  int rc = wws_reserve(c, rows_b + (size_t)G * d * 8);
  if (rc) return rc;
  uint32_t *d_rows = (uint32_t *)c->wws;
  uint64_t *partial = (uint64_t *)((uint8_t *)c->wws + rows_b);
  if (nsel) HIP_TRY(c, hipMemcpyAsync(d_rows, rows + lo, (size_t)nsel * 4, hipMemcpyHostToDevice, c->stream));
  pin_release(c, c->pin_rows);
  if (src.dense)
    hipLaunchKernelGGL(k_witness_partial, dim3((d / 4 + 255) / 256, G), dim3(256), 0, c->stream, src.dense, d_rows, nsel, d, partial);
  else
    hipLaunchKernelGGL(k_witness_partial_prg, dim3((d / 4 + 255) / 256, G), dim3(256), 0, c->stream, src.seed, d_rows, nsel, d, partial);
  HIP_TRY(c, hipGetLastError());
  *G_out = G;
  *partial_out = partial;
  return MFH_OK;
}

int mfh_ssp_set_prg(mfh_ctx *c, uint64_t seed, const uint32_t *d_t) {
  if (!c) return MFH_EINVAL;
  c->prg_on = d_t != nullptr;
  c->prg_seed = seed;
  c->prg_t = d_t;
  return MFH_OK;
}

int mfh_ssp_prg_fill(mfh_ctx *c, uint64_t seed, size_t first_slot, size_t nslots, uint32_t *d_out) {
  if (!c || !d_out) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t total = (uint64_t)nslots * c->P.d;
  if (!total) return MFH_OK;
  hipLaunchKernelGGL(k_ssp_prg_fill, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 4096)), dim3(256), 0, c->stream, seed, (uint32_t)first_slot,
                     c->P.d, total, d_out);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_ssp_prg_make_t(mfh_ctx *c, uint64_t seed, const uint8_t *h_bits, uint32_t *d_t) {
  if (!c || !h_bits || !d_t) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  mf::SspSrc src{nullptr, d_t, seed};
  uint32_t G;
  uint64_t *partial;
  int rc = witness_partials(c, src, h_bits, 0, 1, &G, &partial);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ssp_prg_make_t, dim3((c->P.d + 255) / 256), dim3(256), 0, c->stream, seed, partial, G, c->P.d, d_t);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_witness_poly(mfh_ctx *c, const uint32_t *d_ssp, const uint8_t *h_bits, uint32_t delta, uint32_t *d_w) {
  if (!c || !h_bits || !d_w) return MFH_EINVAL;
  if (delta >= MFH_P) { c->err = "delta must be < p"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  uint32_t G;
  uint64_t *partial;
  rc = witness_partials(c, src, h_bits, 0, 1, &G, &partial);
  if (rc) return rc;
  hipLaunchKernelGGL(k_witness_finish, dim3((c->P.d + 255) / 256), dim3(256), 0, c->stream, src.t, partial, G, c->P.d, delta, d_w);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

// mfh_witness_poly for up to 12 statements in one pass over the SSP (d_ssp == NULL: the rows are generated once per pass): d_w = nstmt
// polynomials of d coefficients
int mfh_witness_poly_multi(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                           uint32_t *d_w) {
  constexpr int NB = 12;
  if (!c || !h_bits || !h_delta || !d_w || nstmt == 0 || nstmt > NB) return MFH_EINVAL;
  mf::SspSrc src;  // d_ssp == NULL: the registered generator-defined SSP
  {
    int rc0 = ssp_src(c, d_ssp, src);
    if (rc0) return rc0;
  }
  const uint32_t d = c->P.d, m = c->P.m;
  if (d % 4) { c->err = "d must be a multiple of 4"; return MFH_EINVAL; }
  for (uint32_t b = 0; b < nstmt; b++)
    if (h_delta[b] >= MFH_P) { c->err = "delta must be < p"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  uint2 *list = (uint2 *)pin_acquire(c, c->pin_rows, (size_t)m * 8 + 8);
  if (!list) return MFH_ENOMEM;
  uint32_t nsel = 0;
  for (uint32_t i = 1; i < m; i++) {
    uint32_t mask = 0;
    for (uint32_t b = 0; b < nstmt; b++) mask |= (uint32_t)((h_bits[b * bits_stride + ((i - 1) >> 3)] >> ((i - 1) & 7)) & 1) << b;
    if (mask) list[nsel++] = make_uint2(i + 1, mask);  // slot of v_i
  }
  const uint32_t G = std::max(1u, std::min(16u, nsel / 8 + 1));
  const size_t list_b = ((size_t)m * 8 + 255) & ~(size_t)255;
  int rc = wws_reserve(c, list_b + (size_t)NB * G * d * 8);
  if (rc) return rc;
  uint2 *d_list = (uint2 *)c->wws;
  uint64_t *partial = (uint64_t *)((uint8_t *)c->wws + list_b);
  if (nsel) HIP_TRY(c, hipMemcpyAsync(d_list, list, (size_t)nsel * 8, hipMemcpyHostToDevice, c->stream));
  pin_release(c, c->pin_rows);
  if (src.dense)
    hipLaunchKernelGGL(k_witness_partial_multi<NB>, dim3((d / 4 + 255) / 256, G), dim3(256), 0, c->stream, src.dense, d_list, nsel, d, partial);
  else
    hipLaunchKernelGGL(k_witness_partial_multi_prg<NB>, dim3((d / 4 + 255) / 256, G), dim3(256), 0, c->stream, src.seed, d_list, nsel, d, partial);
  HIP_TRY(c, hipGetLastError());
  for (uint32_t b = 0; b < nstmt; b++)
    hipLaunchKernelGGL(k_witness_finish, dim3((d + 255) / 256), dim3(256), 0, c->stream, src.t, partial + (size_t)b * G * d, G, d, h_delta[b],
                       d_w + (size_t)b * d);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

// rank's share of sum_{bit} v_i as d uint64 lanes, each already reduced mod p (so `world` of them sum without overflow)
int mfh_witness_lanes(mfh_ctx *c, const uint32_t *d_ssp, const uint8_t *h_bits, uint32_t rank, uint32_t world, uint64_t *d_lanes) {
  if (!c || !h_bits || !d_lanes || world == 0 || rank >= world) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  uint32_t G;
  uint64_t *partial;
  rc = witness_partials(c, src, h_bits, rank, world, &G, &partial);
  if (rc) return rc;
  hipLaunchKernelGGL(k_witness_lanes, dim3((c->P.d + 255) / 256), dim3(256), 0, c->stream, partial, G, c->P.d, d_lanes);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

// w = delta*t + (summed lanes) mod p
int mfh_witness_from_lanes(mfh_ctx *c, const uint32_t *d_ssp, const uint64_t *d_lanes, uint32_t delta, uint32_t *d_w) {
  if (!c || !d_lanes || !d_w) return MFH_EINVAL;
  if (delta >= MFH_P) { c->err = "delta must be < p"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  mf::SspSrc src;
  int rc0 = ssp_src(c, d_ssp, src);
  if (rc0) return rc0;
  hipLaunchKernelGGL(k_witness_finish, dim3((c->P.d + 255) / 256), dim3(256), 0, c->stream, src.t, d_lanes, 1u, c->P.d, delta, d_w);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

}  // extern "C"
