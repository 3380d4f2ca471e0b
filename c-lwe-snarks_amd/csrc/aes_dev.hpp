// aes_dev.hpp -- AES-256-CTR keystream for CDNA4 (gfx950), device side + host key schedule.
//
// Replaces the reference's L0 layer: aesctr_init / aesctr_prg (src/aes.c:49-144), which call
// OpenSSL one block at a time.  gfx950 has no AES instruction, so a block is 13 T-table rounds
// + 1 S-box round.  Design for the CU:
//   * Two tables, T0 (little-endian column convention: T0[a] = {2S, S, S, 3S} as bytes 0..3) and
//     T2 = rotl16(T0); T1/T3 are rotl8 of those (one v_alignbit_b32 per output column).
//   * Each is replicated 32x in LDS, entry-major (256-byte entries: 32 x T0[a], then 32 x T2[a]).
//     Lane l reads replica l & 31, so the 32 lanes of each ds_read_b32 lane group hit 32 distinct
//     banks whatever bytes they look up: every lookup is conflict-free (a single 1 KiB copy is
//     ~3.5-way conflicted on random data).  64 KiB of the CU's 160 KiB.
//   * Columns are little-endian words, so the nonce/counter words are the input columns as they
//     stand and the output words are the keystream's uint32 words as they stand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mf {

struct AesKey {
  uint32_t rk[60];  // round keys, little-endian column words
  uint32_t nonce_lo, nonce_hi;
};


// ---- host: S-box, T0, key schedule (FIPS-197) ----------------------------------------------------
inline uint8_t gf_mul2(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
inline uint8_t gf_mul(uint8_t a, uint8_t b) {
  uint8_t r = 0;
  for (; b; b >>= 1, a = gf_mul2(a))
    if (b & 1) r ^= a;
  return r;
}
inline void make_sbox(uint8_t sbox[256]) {
  // x -> x^254 (the inverse, 0 -> 0), then the affine map
  for (int x = 0; x < 256; x++) {
    uint8_t y = (uint8_t)x, acc = 1;
    for (int e = 254; e; e >>= 1, y = gf_mul(y, y))
      if (e & 1) acc = gf_mul(acc, y);
    uint8_t inv = x ? acc : 0, s = inv, r = inv;
    for (int k = 0; k < 4; k++) {
      r = (uint8_t)((r << 1) | (r >> 7));
      s ^= r;
    }
    sbox[x] = (uint8_t)(s ^ 0x63);
  }
}
inline void make_t0_le(uint32_t t0[256]) {
  uint8_t sbox[256];
  make_sbox(sbox);
  for (int a = 0; a < 256; a++) {
    uint32_t s = sbox[a], s2 = gf_mul2((uint8_t)s), s3 = s2 ^ s;
    t0[a] = s2 | (s << 8) | (s << 16) | (s3 << 24);
  }
}
inline uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
inline void expand_key(AesKey &k, const uint8_t seed[40]) {
  uint8_t sbox[256];
  make_sbox(sbox);
  const uint8_t *key = seed + 8;
  uint32_t w[60];
  auto subword = [&](uint32_t t) {
    return ((uint32_t)sbox[t >> 24] << 24) | ((uint32_t)sbox[(t >> 16) & 255] << 16) |
           ((uint32_t)sbox[(t >> 8) & 255] << 8) | sbox[t & 255];
  };
  for (int i = 0; i < 8; i++)
    w[i] = ((uint32_t)key[4 * i] << 24) | ((uint32_t)key[4 * i + 1] << 16) | ((uint32_t)key[4 * i + 2] << 8) | key[4 * i + 3];
  uint8_t rcon = 1;
  for (int i = 8; i < 60; i++) {
    uint32_t t = w[i - 1];
    if (i % 8 == 0) {
      t = subword((t << 8) | (t >> 24)) ^ ((uint32_t)rcon << 24);
      rcon = gf_mul2(rcon);
    } else if (i % 8 == 4) {
      t = subword(t);
    }
    w[i] = w[i - 8] ^ t;
  }
  for (int i = 0; i < 60; i++) k.rk[i] = bswap32(w[i]);
  // nonce = first 8 seed bytes loaded natively on a little-endian host (src/entropy.c:60)
  k.nonce_lo = (uint32_t)seed[0] | ((uint32_t)seed[1] << 8) | ((uint32_t)seed[2] << 16) | ((uint32_t)seed[3] << 24);
  k.nonce_hi = (uint32_t)seed[4] | ((uint32_t)seed[5] << 8) | ((uint32_t)seed[6] << 16) | ((uint32_t)seed[7] << 24);
}

// ---- device ----------------------------------------------------------------------------------------
// LDS table image (64 KiB): word a*64 + r holds T0[a] for r < 32 and T2[a] = rotl16(T0[a]) for r >= 32.
// A lane uses replica r = lane & 31 (+32 for T2): bank = r for every entry, so the 32 lanes of each
// ds_read_b32 lane group never conflict, and the byte address is (a << 8) | (r << 2): the entry index sits in
// byte 1 of the address, so one v_perm_b32 (or, for state byte 1, one v_bitop3 and-or) forms it.
// T1 = rotl8(T0) and T3 = rotl8(T2) share one v_alignbit per output column:
//   t_j = T0[b0(s_j)] ^ T2[b2(s_j+2)] ^ rotl8(T0[b1(s_j+1)] ^ T2[b3(s_j+3)]) ^ rk
// Measured on MI355X (tools/rate*_ubench.hip): ds_read_b32 ~2.15 CU-clk per wave-instruction, v_xor/v_bitop3 ~2.5
// and v_perm/v_alignbit ~4.3 SIMD-clk: 27 SIMD-clk of VALU per column against 34 of LDS: the round is LDS-bound.
constexpr int kTabBytes = 256 * 64 * 4;

__device__ __forceinline__ void lds_fill_tab(uint32_t *lt, const uint32_t *__restrict__ g_t0) {
  for (int i = threadIdx.x; i < 256 * 64; i += blockDim.x) {
    uint32_t v = g_t0[i >> 6];
    lt[i] = (i & 32) ? ((v << 16) | (v >> 16)) : v;
  }
}

struct AesLane {
  uint32_t lo0, lo2;  // byte offsets of this lane's T0 / T2 replica inside a 256-byte entry
  uint32_t m1;        // 0x0000ff00, kept in a VGPR so v_bitop3 runs at full rate
};
__device__ __forceinline__ AesLane aes_lane() {
  AesLane l;
  l.lo0 = (threadIdx.x & 31) * 4;
  l.lo2 = l.lo0 + 128;
  l.m1 = 0xff00u;
  asm volatile("" : "+v"(l.m1));  // keep it a VGPR
  return l;
}
// (Round 5's address-formation experiments -- all four tables in LDS, SDWA byte moves into persistent address registers, v_perm selectors in VGPRs -- lived here behind
// #ifdefs; none was faster (profiles/r05_aes_address_bound.txt, EXPERIMENTS.md section 5) and they are gone from the product header: git show 774b9bf:c-lwe-snarks_amd/csrc/aes_dev.hpp.)

#define MF_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define MF_ANDOR(a, m, c) __builtin_amdgcn_bitop3_b32((a), (m), (c), 0xEA) /* (a & m) | c */
#ifndef MF_LD  /* tools/aes3_ubench.hip overrides this to time the VALU stream alone */
#define MF_LD(addr) (*reinterpret_cast<const uint32_t *>(tab + (addr)))
#endif
// address of entry byte_k(s) in the T0 (lo0) or T2 (lo2) half
#ifndef MF_A  /* tools/aes3_ubench.hip overrides this too (cheap-address build: how much does the address VALU cost?  The ceiling bench.py's aes_ceiling objects quote) */
#define MF_A(s, lo, k) ((k) == 1 ? MF_ANDOR((s), L.m1, (lo)) : __builtin_amdgcn_perm((s), (lo), 0x0c0c0400u + ((k) << 8)))
#endif

__device__ __forceinline__ uint32_t aes_col(const uint8_t *tab, const AesLane &L, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t rk) {
  uint32_t x0 = MF_LD(MF_A(a, L.lo0, 0)), x1 = MF_LD(MF_A(b, L.lo0, 1));
  uint32_t x2 = MF_LD(MF_A(c, L.lo2, 2)), x3 = MF_LD(MF_A(d, L.lo2, 3));
  uint32_t y = x1 ^ x3;
  return MF_XOR3(x0 ^ rk, x2, __builtin_amdgcn_alignbit(y, y, 24));
}
// last round (SubBytes+ShiftRows+AddRoundKey): S = T2.byte0 = T0.byte1 = T0.byte2 = T2.byte3
__device__ __forceinline__ uint32_t aes_last(const uint8_t *tab, const AesLane &L, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t rk) {
  uint32_t x0 = MF_LD(MF_A(a, L.lo2, 0)), x1 = MF_LD(MF_A(b, L.lo0, 1));
  uint32_t x2 = MF_LD(MF_A(c, L.lo0, 2)), x3 = MF_LD(MF_A(d, L.lo2, 3));
  uint32_t lo = __builtin_amdgcn_perm(x1, x0, 0x0c0c0500u);  // {x0.b0, x1.b1, 0, 0}
  uint32_t hi = __builtin_amdgcn_perm(x3, x2, 0x07020c0cu);  // {0, 0, x2.b2, x3.b3}
  return MF_XOR3(lo, hi, rk);
}

// ---- counter-mode shortcut for rounds 1-2 (Bernstein-Schwabe style counter caching) ---------------------------
// Input block = nonce || ctr: inside a span of 256 consecutive counters only byte 0 of column 2 changes.  Round 1 then
// changes only through T0[b0(s2)] in column 2, and round 2 through the four bytes of that column:
//   t2 = T0[b0(s2)] ^ C2 ;  u0 = D0 ^ T2[b2(t2)], u1 = D1 ^ T1[b1(t2)], u2 = D2 ^ T0[b0(t2)], u3 = D3 ^ T3[b3(t2)]
// {C2, D0..D3} depend on ctr >> 8 only: 27 lookups once per span instead of 28 extra lookups per block.
#define MF_T0AT(s, k) MF_LD(MF_A(s, L.lo0, k))
#define MF_T2AT(s, k) MF_LD(MF_A(s, L.lo2, k))
__device__ __forceinline__ uint32_t rotl8_(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 24); }

__device__ __forceinline__ void aes_span_consts(const uint8_t *tab, const AesLane &L, const AesKey &k, uint64_t span /* ctr >> 8 */, uint32_t sc[5]) {
  const uint64_t ctr = span << 8;
  const uint32_t s0 = k.nonce_lo ^ k.rk[0], s1 = k.nonce_hi ^ k.rk[1];
  const uint32_t s2 = (uint32_t)ctr ^ k.rk[2], s3 = (uint32_t)(ctr >> 32) ^ k.rk[3];  // byte 0 of s2 is not used below
  const uint32_t t0 = aes_col(tab, L, s0, s1, s2, s3, k.rk[4]);
  const uint32_t t1 = aes_col(tab, L, s1, s2, s3, s0, k.rk[5]);
  const uint32_t t3 = aes_col(tab, L, s3, s0, s1, s2, k.rk[7]);
  sc[0] = rotl8_(MF_T0AT(s3, 1) ^ MF_T2AT(s1, 3)) ^ MF_T2AT(s0, 2) ^ k.rk[6];                     // C2
  sc[1] = MF_T0AT(t0, 0) ^ rotl8_(MF_T0AT(t1, 1) ^ MF_T2AT(t3, 3)) ^ k.rk[8];                     // D0
  sc[2] = MF_T0AT(t1, 0) ^ MF_T2AT(t3, 2) ^ rotl8_(MF_T2AT(t0, 3)) ^ k.rk[9];                     // D1
  sc[3] = rotl8_(MF_T0AT(t3, 1) ^ MF_T2AT(t1, 3)) ^ MF_T2AT(t0, 2) ^ k.rk[10];                    // D2
  sc[4] = MF_T0AT(t3, 0) ^ rotl8_(MF_T0AT(t0, 1)) ^ MF_T2AT(t1, 2) ^ k.rk[11];                    // D3
}

// One of a middle round's 16 lookups through the vector-memory path instead of LDS (tools/aes7_ubench.hip: the LDS pipe is the limiter of every AES kernel and
// the texture-addresser / L1 path is idle; one gather in sixteen lookups is what that path takes -- a dword gather costs ~23 CU-clk per wave-instruction against
// 2.15 for ds_read_b32 -- : +4.9 % at 8 waves per SIMD, +2.3 % at 4; two in sixteen already lose).  gt3 = T3 = rotl24(T0), 256 words in global memory
// (L1-resident after the first touch), addressed through a buffer resource so that the address is one VALU instruction.
struct AesGl {
  __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ AesGl aes_gl(const uint32_t *gt3) {
  return AesGl{__builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(gt3), 0, 1024, 0x00020000)};
}
__device__ __forceinline__ uint32_t aes_col_g(const uint8_t *tab, const AesLane &L, const AesGl &G, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t rk) {
  const uint32_t x3 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(G.rs, (int)((d >> 22) & 0x3fcu), 0, 0);
  const uint32_t x0 = MF_LD(MF_A(a, L.lo0, 0)), x1 = MF_LD(MF_A(b, L.lo0, 1)), x2 = MF_LD(MF_A(c, L.lo2, 2));
  return MF_XOR3(x0 ^ rk, x2, __builtin_amdgcn_alignbit(x1, x1, 24)) ^ x3;
}

// the same block as aes256_ctr_block, entering at round 3 with the span constants of ctr >> 8
template <bool GL = false>
__device__ __forceinline__ void aes256_ctr_block_sc(const uint8_t *tab, const AesLane &L, const AesKey &k, uint64_t ctr, const uint32_t sc[5],
                                                    uint32_t out[4], const AesGl *G = nullptr) {
  const uint32_t s2 = (uint32_t)ctr ^ k.rk[2];
  const uint32_t t2 = MF_T0AT(s2, 0) ^ sc[0];
  uint32_t s0 = sc[1] ^ MF_T2AT(t2, 2);
  uint32_t s1 = sc[2] ^ rotl8_(MF_T0AT(t2, 1));
  uint32_t sB = sc[3] ^ MF_T0AT(t2, 0);
  uint32_t s3 = sc[4] ^ rotl8_(MF_T2AT(t2, 3));
#pragma unroll
  for (int r = 3; r < 14; r++) {
    uint32_t u0 = GL ? aes_col_g(tab, L, *G, s0, s1, sB, s3, k.rk[4 * r]) : aes_col(tab, L, s0, s1, sB, s3, k.rk[4 * r]);
    uint32_t u1 = aes_col(tab, L, s1, sB, s3, s0, k.rk[4 * r + 1]);
    uint32_t u2 = aes_col(tab, L, sB, s3, s0, s1, k.rk[4 * r + 2]);
    uint32_t u3 = aes_col(tab, L, s3, s0, s1, sB, k.rk[4 * r + 3]);
    s0 = u0; s1 = u1; sB = u2; s3 = u3;
  }
  out[0] = aes_last(tab, L, s0, s1, sB, s3, k.rk[56]);
  out[1] = aes_last(tab, L, s1, sB, s3, s0, k.rk[57]);
  out[2] = aes_last(tab, L, sB, s3, s0, s1, k.rk[58]);
  out[3] = aes_last(tab, L, s3, s0, s1, sB, k.rk[59]);
}

// One stream block: AES256_K(nonce_le64 || le64(ctr)) as 4 little-endian words.
__device__ __forceinline__ void aes256_ctr_block(const uint8_t *tab, const AesLane &L, const AesKey &k, uint64_t ctr, uint32_t out[4]) {
  uint32_t s0 = k.nonce_lo ^ k.rk[0], s1 = k.nonce_hi ^ k.rk[1];
  uint32_t s2 = (uint32_t)ctr ^ k.rk[2], s3 = (uint32_t)(ctr >> 32) ^ k.rk[3];
#pragma unroll
  for (int r = 1; r < 14; r++) {
    uint32_t t0 = aes_col(tab, L, s0, s1, s2, s3, k.rk[4 * r]);
    uint32_t t1 = aes_col(tab, L, s1, s2, s3, s0, k.rk[4 * r + 1]);
    uint32_t t2 = aes_col(tab, L, s2, s3, s0, s1, k.rk[4 * r + 2]);
    uint32_t t3 = aes_col(tab, L, s3, s0, s1, s2, k.rk[4 * r + 3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  out[0] = aes_last(tab, L, s0, s1, s2, s3, k.rk[56]);
  out[1] = aes_last(tab, L, s1, s2, s3, s0, k.rk[57]);
  out[2] = aes_last(tab, L, s2, s3, s0, s1, k.rk[58]);
  out[3] = aes_last(tab, L, s3, s0, s1, s2, k.rk[59]);
}

}  // namespace mf
