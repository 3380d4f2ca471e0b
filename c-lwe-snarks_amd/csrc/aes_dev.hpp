// aes_dev.hpp -- AES-256-CTR keystream for CDNA4 (gfx950), device side + host key schedule.
//
// Replaces the reference's L0 layer: aesctr_init / aesctr_prg (src/aes.c:49-144), which call
// OpenSSL one block at a time.  gfx950 has no AES instruction, so a block is 13 T-table rounds
// + 1 S-box round.  Design for the CU:
//   * ONE table, T0 (little-endian column convention: T0[a] = {2S, S, S, 3S} as bytes 0..3), the
//     other three are rotations (v_alignbit_b32): 1 KiB of distinct data.
//   * The table is replicated 32x in LDS, entry-major: word a*32 + r holds T0[a].  Lane l reads
//     replica r = l & 31, so the 32 lanes of each ds_read_b32 lane group hit 32 distinct banks
//     whatever bytes they look up: every lookup is conflict-free (a single 1 KiB copy is ~3.5-way
//     conflicted on random data).  32 KiB of the CU's 160 KiB.
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

constexpr int kT0Words = 256 * 32;  // replicated table, words

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
__device__ __forceinline__ void lds_fill_t0(uint32_t *lt, const uint32_t *__restrict__ g_t0) {
  for (int i = threadIdx.x; i < kT0Words; i += blockDim.x) lt[i] = g_t0[i >> 5];
}

__device__ __forceinline__ uint32_t rotl8(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 24); }
__device__ __forceinline__ uint32_t rotl16(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 16); }
__device__ __forceinline__ uint32_t rotl24(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 8); }

// tl = replicated table base + (lane & 31); entry a lives at tl[a << 5]
#define MF_T(x) tl[(x) << 5]

// One stream block: AES256_K(nonce_le64 || le64(ctr)) as 4 little-endian words.
__device__ __forceinline__ void aes256_ctr_block(const uint32_t *tl, const AesKey &k, uint64_t ctr, uint32_t out[4]) {
  uint32_t s0 = k.nonce_lo ^ k.rk[0], s1 = k.nonce_hi ^ k.rk[1];
  uint32_t s2 = (uint32_t)ctr ^ k.rk[2], s3 = (uint32_t)(ctr >> 32) ^ k.rk[3];
#pragma unroll
  for (int r = 1; r < 14; r++) {
    uint32_t t0 = MF_T(s0 & 255) ^ rotl8(MF_T((s1 >> 8) & 255)) ^ rotl16(MF_T((s2 >> 16) & 255)) ^ rotl24(MF_T(s3 >> 24)) ^ k.rk[4 * r];
    uint32_t t1 = MF_T(s1 & 255) ^ rotl8(MF_T((s2 >> 8) & 255)) ^ rotl16(MF_T((s3 >> 16) & 255)) ^ rotl24(MF_T(s0 >> 24)) ^ k.rk[4 * r + 1];
    uint32_t t2 = MF_T(s2 & 255) ^ rotl8(MF_T((s3 >> 8) & 255)) ^ rotl16(MF_T((s0 >> 16) & 255)) ^ rotl24(MF_T(s1 >> 24)) ^ k.rk[4 * r + 2];
    uint32_t t3 = MF_T(s3 & 255) ^ rotl8(MF_T((s0 >> 8) & 255)) ^ rotl16(MF_T((s1 >> 16) & 255)) ^ rotl24(MF_T(s2 >> 24)) ^ k.rk[4 * r + 3];
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  // last round: SubBytes + ShiftRows + AddRoundKey.  S[a] is byte 1 (and byte 2) of T0[a].
#define MF_LAST(a, b, c, d)                                                                                      \
  (((MF_T((a) & 255) >> 8) & 0xffu) | (MF_T(((b) >> 8) & 255) & 0xff00u) | (MF_T(((c) >> 16) & 255) & 0xff0000u) | \
   ((MF_T((d) >> 24) << 8) & 0xff000000u))
  out[0] = MF_LAST(s0, s1, s2, s3) ^ k.rk[56];
  out[1] = MF_LAST(s1, s2, s3, s0) ^ k.rk[57];
  out[2] = MF_LAST(s2, s3, s0, s1) ^ k.rk[58];
  out[3] = MF_LAST(s3, s0, s1, s2) ^ k.rk[59];
#undef MF_LAST
}

}  // namespace mf
