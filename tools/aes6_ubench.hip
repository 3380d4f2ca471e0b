// aes6_ubench.hip -- bitsliced AES-256-CTR on gfx950 (no LDS, no tables) against the T-table AES of csrc/aes_dev.hpp.  (dev tool;
// SURVEY section 7 asked for the choice between the two to be MEASURED)
//
// Representation: a lane holds 32 blocks.  st[b][i] = one 32-bit register whose bit j is bit i of state byte b of block j
// (128 registers).  SubBytes = the Boyar-Peralta 113-gate circuit on the 8 bit planes of each of the 16 bytes (plain C: hipcc fuses
// gate pairs into v_bitop3_b32 where it can); ShiftRows = register renaming; MixColumns = xors of planes; AddRoundKey = xor with the
// round key in bitsliced form (bit (b, i) of round r broadcast to a whole word: 0 or ~0, read with scalar loads).  The counter input is
// free in this form: counter bits 0..4 are the constant patterns 0xAAAAAAAA ... 0xFFFF0000, all higher bits broadcasts of the lane's base.
// Two rounds per loop iteration (A -> B -> A) so that ShiftRows stays a renaming inside a rolled loop (a fully unrolled kernel is
// ~250 KB of code).  Output: either the xor of all planes (rate of the cipher alone) or -DTRANSPOSE: the four 32 x 32 bit transpositions
// that turn planes back into keystream words, which every consumer of bytes needs.
//
// Modes:  ./aes6_ubench            standalone rates (bitsliced at 1 and 2 waves per SIMD; table AES at 4 and 8)
//         ./aes6_ubench mix        a table kernel limited to 4 waves per SIMD (one 1024-thread workgroup per CU through its LDS footprint)
//                                  and a bitsliced kernel (one wave per SIMD) launched on two streams: do they share the CUs, and what is
//                                  the sum?
// Every run first checks the bitsliced keystream against the table kernel's on 64 Ki blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "aes_dev.hpp"
using mf::AesKey;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- bitsliced S-box: Boyar & Peralta, "A new combinational logic minimization technique with applications to cryptology" (2009),
// 32 AND + 81 XOR/XNOR.  q[i] = plane of bit i (bit 0 = least significant). -----------------------------------------------------------
__device__ __host__ __forceinline__ void bs_sbox(uint32_t q[8]) {
  const uint32_t x0 = q[7], x1 = q[6], x2 = q[5], x3 = q[4], x4 = q[3], x5 = q[2], x6 = q[1], x7 = q[0];
  const uint32_t y14 = x3 ^ x5, y13 = x0 ^ x6, y9 = x0 ^ x3, y8 = x0 ^ x5, t0 = x1 ^ x2, y1 = t0 ^ x7, y4 = y1 ^ x3, y12 = y13 ^ y14, y2 = y1 ^ x0,
                 y5 = y1 ^ x6, y3 = y5 ^ y8, t1 = x4 ^ y12, y15 = t1 ^ x5, y20 = t1 ^ x1, y6 = y15 ^ x7, y10 = y15 ^ t0, y11 = y20 ^ y9, y7 = x7 ^ y11,
                 y17 = y10 ^ y11, y19 = y10 ^ y8, y16 = t0 ^ y11, y21 = y13 ^ y16, y18 = x0 ^ y16;
  const uint32_t t2 = y12 & y15, t3 = y3 & y6, t4 = t3 ^ t2, t5 = y4 & x7, t6 = t5 ^ t2, t7 = y13 & y16, t8 = y5 & y1, t9 = t8 ^ t7, t10 = y2 & y7,
                 t11 = t10 ^ t7, t12 = y9 & y11, t13 = y14 & y17, t14 = t13 ^ t12, t15 = y8 & y10, t16 = t15 ^ t12, t17 = t4 ^ t14, t18 = t6 ^ t16,
                 t19 = t9 ^ t14, t20 = t11 ^ t16, t21 = t17 ^ y20, t22 = t18 ^ y19, t23 = t19 ^ y21, t24 = t20 ^ y18;
  const uint32_t t25 = t21 ^ t22, t26 = t21 & t23, t27 = t24 ^ t26, t28 = t25 & t27, t29 = t28 ^ t22, t30 = t23 ^ t24, t31 = t22 ^ t26, t32 = t31 & t30,
                 t33 = t32 ^ t24, t34 = t23 ^ t33, t35 = t27 ^ t33, t36 = t24 & t35, t37 = t36 ^ t34, t38 = t27 ^ t36, t39 = t29 & t38, t40 = t25 ^ t39;
  const uint32_t t41 = t40 ^ t37, t42 = t29 ^ t33, t43 = t29 ^ t40, t44 = t33 ^ t37, t45 = t42 ^ t41;
  const uint32_t z0 = t44 & y15, z1 = t37 & y6, z2 = t33 & x7, z3 = t43 & y16, z4 = t40 & y1, z5 = t29 & y7, z6 = t42 & y11, z7 = t45 & y17, z8 = t41 & y10,
                 z9 = t44 & y12, z10 = t37 & y3, z11 = t33 & y4, z12 = t43 & y13, z13 = t40 & y5, z14 = t29 & y2, z15 = t42 & y9, z16 = t45 & y14,
                 z17 = t41 & y8;
  const uint32_t t46 = z15 ^ z16, t47 = z10 ^ z11, t48 = z5 ^ z13, t49 = z9 ^ z10, t50 = z2 ^ z12, t51 = z2 ^ z5, t52 = z7 ^ z8, t53 = z0 ^ z3, t54 = z6 ^ z7,
                 t55 = z16 ^ z17, t56 = z12 ^ t48, t57 = t50 ^ t53, t58 = z4 ^ t46, t59 = z3 ^ t54, t60 = t46 ^ t57, t61 = z14 ^ t57, t62 = t52 ^ t58,
                 t63 = t49 ^ t58, t64 = z4 ^ t59, t65 = t61 ^ t62, t66 = z1 ^ t63;
  const uint32_t s0 = t59 ^ t63, s6 = t56 ^ ~t62, s7 = t48 ^ ~t60, t67 = t64 ^ t65, s3 = t53 ^ t66, s4 = t51 ^ t66, s5 = t47 ^ t65, s1 = t64 ^ ~s3,
                 s2 = t55 ^ ~t67;
  q[7] = s0; q[6] = s1; q[5] = s2; q[4] = s3; q[3] = s4; q[2] = s5; q[1] = s6; q[0] = s7;
}

// one round A -> B: SubBytes in place on A, then B[row r of column c] = MixColumns(ShiftRows(A)) ^ rk.  State byte index = 4 c + r.
// bk = the round key, 128 broadcast words [byte][bit].  LAST: no MixColumns.
// plane (byte 4c + r, bit i) of a round key = bit 8r + i of its column word c, broadcast: one s_bfe_i32 on the scalar unit
#define KBIT(rk4, c, r, i) ((uint32_t)(((int32_t)((rk4)[c] << (31 - (8 * (r) + (i))))) >> 31))
#define OPAQUE(x) asm volatile("" : "+v"(x))  /* stops the optimizer from re-associating xor trees across S-box / MixColumns boundaries */
template <bool LAST>
__device__ __forceinline__ void bs_round(uint32_t (&A)[16][8], uint32_t (&B)[16][8], const uint32_t *__restrict__ bk) {
  // Column by column, one S-box at a time, with scheduling barriers between the pieces: the state is 128 registers and an S-box in
  // flight ~45 more; left to itself the scheduler interleaves all sixteen S-boxes (and hoists all 128 scalar key loads), which spills.
  // A byte of A is consumed by exactly one column of B, so A shrinks as B grows: ~128 + 45 + 40 registers live.
#pragma unroll
  for (int c = 0; c < 4; c++) {
    // after ShiftRows, row r of column c comes from column (c + r) % 4
    uint32_t(&a0)[8] = A[4 * ((c + 0) & 3) + 0];
    uint32_t(&a1)[8] = A[4 * ((c + 1) & 3) + 1];
    uint32_t(&a2)[8] = A[4 * ((c + 2) & 3) + 2];
    uint32_t(&a3)[8] = A[4 * ((c + 3) & 3) + 3];
    bs_sbox(a0);
#pragma unroll
    for (int i = 0; i < 8; i++) OPAQUE(a0[i]);
    __builtin_amdgcn_sched_barrier(0);
    bs_sbox(a1);
#pragma unroll
    for (int i = 0; i < 8; i++) OPAQUE(a1[i]);
    __builtin_amdgcn_sched_barrier(0);
    bs_sbox(a2);
#pragma unroll
    for (int i = 0; i < 8; i++) OPAQUE(a2[i]);
    __builtin_amdgcn_sched_barrier(0);
    bs_sbox(a3);
#pragma unroll
    for (int i = 0; i < 8; i++) OPAQUE(a3[i]);
    __builtin_amdgcn_sched_barrier(0);
    if (LAST) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        B[4 * c + 0][i] = a0[i] ^ KBIT(bk, c, 0, i);
        B[4 * c + 1][i] = a1[i] ^ KBIT(bk, c, 1, i);
        B[4 * c + 2][i] = a2[i] ^ KBIT(bk, c, 2, i);
        B[4 * c + 3][i] = a3[i] ^ KBIT(bk, c, 3, i);
      }
    } else {
      uint32_t t[4][8], all[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        t[0][i] = a0[i] ^ a1[i];
        t[1][i] = a1[i] ^ a2[i];
        t[2][i] = a2[i] ^ a3[i];
        t[3][i] = a3[i] ^ a0[i];
        all[i] = t[0][i] ^ t[2][i];
      }
      uint32_t(*ar[4])[8] = {&a0, &a1, &a2, &a3};
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          // out_r = xtime(a_r ^ a_(r+1)) ^ a_(r+1) ^ a_(r+2) ^ a_(r+3) ^ k = xtime(t_r) ^ all ^ a_r ^ k;  xtime(t)[i] = t[i-1] ^ (i in {0,1,3,4} ? t[7] : 0), t[-1] = 0
          uint32_t x = all[i] ^ (*ar[r])[i] ^ KBIT(bk, c, r, i);
          if (i > 0) x ^= t[r][i - 1];
          if (i == 0 || i == 1 || i == 3 || i == 4) x ^= t[r][7];
          OPAQUE(x);
          B[4 * c + r][i] = x;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// planes of 32 consecutive counter blocks starting at base (a multiple of 32), round key 0 already added
__device__ __forceinline__ void bs_ctr_input(uint32_t (&A)[16][8], uint32_t nonce_lo, uint32_t nonce_hi, uint64_t base, const uint32_t *__restrict__ bk0) {
  const uint32_t p0 = 0xAAAAAAAAu, p1 = 0xCCCCCCCCu, p2 = 0xF0F0F0F0u, p3 = 0xFF00FF00u, p4 = 0xFFFF0000u;
#pragma unroll
  for (int b = 0; b < 16; b++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t v;
      if (b < 4) v = 0u - ((nonce_lo >> (8 * b + i)) & 1u);
      else if (b < 8) v = 0u - ((nonce_hi >> (8 * (b - 4) + i)) & 1u);
      else {
        const int bit = 8 * (b - 8) + i;
        v = bit == 0 ? p0 : bit == 1 ? p1 : bit == 2 ? p2 : bit == 3 ? p3 : bit == 4 ? p4 : 0u - (uint32_t)((base >> bit) & 1u);
      }
      A[b][i] = v ^ KBIT(bk0, b / 4, b % 4, i);
    }
}

// 32 x 32 bit transposition in registers: w[k] bit j  <->  w[j] bit k
template <int S>
__device__ __forceinline__ void transpose_stage(uint32_t w[32]) {
  constexpr uint32_t m = S == 16 ? 0x0000FFFFu : S == 8 ? 0x00FF00FFu : S == 4 ? 0x0F0F0F0Fu : S == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
  for (int k = 0; k < 32; k++)
    if (!(k & S)) {
      const uint32_t a = w[k], b = w[k + S];
      w[k] = (a & m) | ((b & m) << S);
      w[k + S] = ((a >> S) & m) | (b & ~m);
    }
}
__device__ __forceinline__ void transpose32(uint32_t w[32]) {
  transpose_stage<16>(w);
  transpose_stage<8>(w);
  transpose_stage<4>(w);
  transpose_stage<2>(w);
  transpose_stage<1>(w);
}

// Each lane does `niter` batches of 32 blocks.  OUT = 0: xor of all planes into out[lane] (the cipher alone); 1: planes transposed to keystream
// words, xored into out[lane]; 2: keystream words stored block-major to `words` (check mode).
template <int MINB, int OUT>
__global__ __launch_bounds__(256, MINB) void k_bitsliced(AesKey key, uint32_t niter, uint32_t *__restrict__ out, uint32_t *__restrict__ words) {
  const uint64_t lane_id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < niter; it++) {
    const uint64_t base = (lane_id * niter + it) * 32;
    uint32_t A[16][8], B[16][8];
    bs_ctr_input(A, key.nonce_lo, key.nonce_hi, base, key.rk);
    // rounds 1..12 in pairs, 13, then the last
    for (int r = 1; r <= 11; r += 2) {
      bs_round<false>(A, B, key.rk + 4 * r);
      bs_round<false>(B, A, key.rk + 4 * (r + 1));
    }
    bs_round<false>(A, B, key.rk + 4 * 13);
    bs_round<true>(B, A, key.rk + 4 * 14);
    if (OUT) {
      // word c of block j = bytes 4c..4c+3: planes A[4c + k][i] -> bit 8k + i
#pragma unroll
      for (int c = 0; c < 4; c++) {
        uint32_t w[32];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int i = 0; i < 8; i++) w[8 * k + i] = A[4 * c + k][i];
        transpose32(w);
        if (OUT == 2) {
#pragma unroll
          for (int j = 0; j < 32; j++) words[(base + j) * 4 + c] = w[j];
        } else {
#pragma unroll
          for (int j = 0; j < 32; j++) acc ^= w[j] + j;  // (the + j keeps the transposition from being optimised into a plane xor)
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < 16; b++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc ^= A[b][i];
    }
  }
  if (OUT != 2) out[lane_id] = acc;
}

// ---- the table AES as the product kernels run it (table at LDS address 0).  PAD bytes of extra LDS limit the workgroups per CU. ----------
template <int MINW, int PAD>
__global__ __launch_bounds__(1024, MINW) void k_table(AesKey key, const uint32_t *g_t0, uint32_t nb, uint32_t *out, uint32_t *words) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[65536 + PAD];
  if (PAD && nb == 0xffffffffu) smem[65536 + threadIdx.x % (PAD ? PAD : 1)] = 1;
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const mf::AesLane L = mf::aes_lane();
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < nb; i++) {
    uint32_t w[4];
    mf::aes256_ctr_block(smem, L, key, base + i, w);
    if (words) { words[(base + i) * 4 + 0] = w[0]; words[(base + i) * 4 + 1] = w[1]; words[(base + i) * 4 + 2] = w[2]; words[(base + i) * 4 + 3] = w[3]; }
    acc ^= w[0] ^ w[1] ^ w[2] ^ w[3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void bitslice_key(const AesKey &key, std::vector<uint32_t> &bk) {
  bk.resize(15 * 128);
  for (int r = 0; r < 15; r++)
    for (int b = 0; b < 16; b++)
      for (int i = 0; i < 8; i++) bk[r * 128 + b * 8 + i] = 0u - ((key.rk[4 * r + b / 4] >> (8 * (b % 4) + i)) & 1u);  // rk words are little-endian columns
}

static float time_ms(hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)s;
  return ms;
}

int main(int argc, char **argv) {
  const bool mix = argc > 1 && !strcmp(argv[1], "mix");
  uint8_t seed[40];
  for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key;
  mf::expand_key(key, seed);
  uint32_t t0[256];
  mf::make_t0_le(t0);
  // host check of the S-box circuit against the table
  {
    uint8_t sbox[256];
    mf::make_sbox(sbox);
    for (int x0 = 0; x0 < 256; x0 += 32) {
      uint32_t q[8];
      for (int i = 0; i < 8; i++) {
        q[i] = 0;
        for (int j = 0; j < 32; j++) q[i] |= (uint32_t)(((x0 + j) >> i) & 1) << j;
      }
      bs_sbox(q);
      for (int j = 0; j < 32; j++) {
        int y = 0;
        for (int i = 0; i < 8; i++) y |= ((q[i] >> j) & 1) << i;
        if (y != sbox[x0 + j]) { printf("S-box circuit wrong at %d: %02x != %02x\n", x0 + j, y, sbox[x0 + j]); return 1; }
      }
    }
  }
  std::vector<uint32_t> bk;
  bitslice_key(key, bk);
  uint32_t *d_t0, *d_bk, *d_out, *d_out2, *d_w1, *d_w2;
  CHECK(hipMalloc(&d_t0, sizeof t0));
  CHECK(hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_bk, bk.size() * 4));
  CHECK(hipMemcpy(d_bk, bk.data(), bk.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_out, 256 * 8 * 1024 * 4));
  CHECK(hipMalloc(&d_out2, 256 * 8 * 1024 * 4));
  const uint32_t NCHK = 65536;
  CHECK(hipMalloc(&d_w1, NCHK * 16));
  CHECK(hipMalloc(&d_w2, NCHK * 16));
  // correctness: 65536 blocks from both
  hipLaunchKernelGGL((k_table<1, 0>), dim3(NCHK / 1024 / 4), dim3(1024), 0, 0, key, d_t0, 4u, d_out, d_w1);
  hipLaunchKernelGGL((k_bitsliced<1, 2>), dim3(NCHK / 32 / 256 / 2), dim3(256), 0, 0, key, 2u, d_out2, d_w2);
  CHECK(hipDeviceSynchronize());
  {
    std::vector<uint32_t> a(NCHK * 4), b(NCHK * 4);
    CHECK(hipMemcpy(a.data(), d_w1, NCHK * 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), d_w2, NCHK * 16, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
    printf("bitsliced keystream vs table keystream over %u blocks: %s (%zu words differ); block 0 = %08x %08x %08x %08x\n", NCHK, bad ? "MISMATCH" : "identical", bad,
           b[0], b[1], b[2], b[3]);
    if (bad) return 1;
  }
  hipEvent_t e0, e1, f0, f1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));
  auto report = [&](const char *name, double blocks, float ms) {
    printf("%-58s %8.3f ms %7.2f Gblk/s %6.2f clk/blk/CU@2.4GHz\n", name, ms, blocks / ms / 1e6, 256.0 * 2.4e9 / (blocks / (ms * 1e-3)));
  };
  if (!mix) {
    // ---- standalone
    for (int tr = 0; tr < 2; tr++)
      for (int wps = 1; wps <= 2; wps++) {  // waves per SIMD: one 256-thread workgroup = 4 waves = one per SIMD
        const uint32_t niter = 24, grid = 256 * wps * 4;  // 4 rounds of workgroups per CU slot
        float best = 1e30f;
        for (int it = 0; it < 3; it++) {
          CHECK(hipEventRecord(e0, 0));
          if (tr == 0 && wps == 1) hipLaunchKernelGGL((k_bitsliced<1, 0>), dim3(grid), dim3(256), 0, 0, key, niter, d_out, nullptr);
          if (tr == 0 && wps == 2) hipLaunchKernelGGL((k_bitsliced<2, 0>), dim3(grid), dim3(256), 0, 0, key, niter, d_out, nullptr);
          if (tr == 1 && wps == 1) hipLaunchKernelGGL((k_bitsliced<1, 1>), dim3(grid), dim3(256), 0, 0, key, niter, d_out, nullptr);
          if (tr == 1 && wps == 2) hipLaunchKernelGGL((k_bitsliced<2, 1>), dim3(grid), dim3(256), 0, 0, key, niter, d_out, nullptr);
          CHECK(hipEventRecord(e1, 0));
          const float ms = time_ms(0, e0, e1);
          if (it && ms < best) best = ms;
        }
        char nm[128];
        snprintf(nm, sizeof nm, "bitsliced%s, %d wave(s)/SIMD requested", tr ? " + transposition to words" : " (planes only)", wps);
        report(nm, (double)grid * 256 * niter * 32, best);
      }
    for (int v = 0; v < 2; v++) {
      const uint32_t nb = 256;
      float best = 1e30f;
      for (int it = 0; it < 3; it++) {
        CHECK(hipEventRecord(e0, 0));
        if (v == 0) hipLaunchKernelGGL((k_table<1, 94240>), dim3(256 * 2), dim3(1024), 0, 0, key, d_t0, nb, d_out, nullptr);
        else hipLaunchKernelGGL((k_table<2, 0>), dim3(256 * 4), dim3(1024), 0, 0, key, d_t0, nb, d_out, nullptr);
        CHECK(hipEventRecord(e1, 0));
        const float ms = time_ms(0, e0, e1);
        if (it && ms < best) best = ms;
      }
      report(v == 0 ? "table AES, 4 waves/SIMD (1 workgroup of 1024 per CU)" : "table AES, 8 waves/SIMD (2 workgroups of 1024 per CU)", 256.0 * (v ? 4 : 2) * 1024 * nb, best);
    }
    return 0;
  }
  // ---- mix: table kernel at 4 waves/SIMD (LDS-limited to one workgroup per CU, 64 VGPRs x 4 = half the register file) on stream A,
  // bitsliced kernel (one wave per SIMD per workgroup) on stream B
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const uint32_t nb = 512, tgrid = 256 * 3;    // table: 3 rounds of one workgroup per CU
  const double tblocks = (double)tgrid * 1024 * nb;
  auto run_table = [&](hipStream_t s) { hipLaunchKernelGGL((k_table<1, 94240>), dim3(tgrid), dim3(1024), 0, s, key, d_t0, nb, d_out, nullptr); };
  float t_alone = 1e30f, b_alone = 1e30f;
  for (int it = 0; it < 3; it++) {
    CHECK(hipEventRecord(e0, sa)); run_table(sa); CHECK(hipEventRecord(e1, sa));
    const float ms = time_ms(sa, e0, e1);
    if (it && ms < t_alone) t_alone = ms;
  }
  report("table alone (4 waves/SIMD)", tblocks, t_alone);
  for (int tr = 0; tr < 2; tr++) {
    // size the bitsliced launch so that alone it takes about as long as the table launch
    uint32_t niter = 16, bgrid = 256 * 4;
    auto run_bs = [&](hipStream_t s) {
      if (tr) hipLaunchKernelGGL((k_bitsliced<1, 1>), dim3(bgrid), dim3(256), 0, s, key, niter, d_out2, nullptr);
      else hipLaunchKernelGGL((k_bitsliced<1, 0>), dim3(bgrid), dim3(256), 0, s, key, niter, d_out2, nullptr);
    };
    for (int cal = 0; cal < 2; cal++) {
      b_alone = 1e30f;
      for (int it = 0; it < 3; it++) {
        CHECK(hipEventRecord(f0, sb)); run_bs(sb); CHECK(hipEventRecord(f1, sb));
        const float ms = time_ms(sb, f0, f1);
        if (it && ms < b_alone) b_alone = ms;
      }
      if (cal == 0) niter = (uint32_t)(niter * t_alone / b_alone + 0.5f);
    }
    const double bblocks = (double)bgrid * 256 * niter * 32;
    report(tr ? "bitsliced + transposition alone (1 wave/SIMD per workgroup)" : "bitsliced (planes) alone (1 wave/SIMD per workgroup)", bblocks, b_alone);
    float best_wall = 1e30f, tb = 0, bb = 0;
    for (int it = 0; it < 4; it++) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, sa)); CHECK(hipEventRecord(f0, sb));
      run_table(sa); run_bs(sb);
      CHECK(hipEventRecord(e1, sa)); CHECK(hipEventRecord(f1, sb));
      const float ta = time_ms(sa, e0, e1), tbm = time_ms(sb, f0, f1);
      float w1, w2;
      CHECK(hipEventElapsedTime(&w1, e0, f1));
      CHECK(hipEventElapsedTime(&w2, f0, e1));
      const float wall = fmaxf(fmaxf(ta, tbm), fmaxf(w1, w2));
      if (it && wall < best_wall) { best_wall = wall; tb = ta; bb = tbm; }
    }
    printf("  both at once: table %.3f ms, bitsliced %.3f ms, wall %.3f ms (sum alone %.3f)\n", tb, bb, best_wall, t_alone + b_alone);
    report(tr ? "table + bitsliced(+transposition) side by side, total" : "table + bitsliced(planes) side by side, total", tblocks + bblocks, best_wall);
  }
  return 0;
}
