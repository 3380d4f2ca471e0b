// aes7_ubench.hip -- T-table lookups through the idle vector-memory path (round 4).  (dev tool)
// The product AES (aes_dev.hpp, table at LDS address 0, 32 replicas) with NG of the 16 lookups of every middle round issued as buffer_load_dword gathers from a
// 4 KiB table {T0, T1, T2, T3} in global memory (L1-resident after the first touch; the rotated copies spare the v_alignbit) instead of ds_read_b32, at 4 and
// 8 waves per SIMD.  aes5_ubench measured the same idea in round 2 with dynamic LDS (table not at address 0) at 4 waves per SIMD only.
//   NG = 1, 2, 3  ->  f = 1/16, 1/8, 3/16 of a round's lookups; the global lookups are spread over the four columns (column j takes its T3 term first).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "aes_dev.hpp"
using mf::AesKey;
using mf::AesLane;
typedef __amdgpu_buffer_rsrc_t v4i_;
__device__ __forceinline__ v4i_ make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);  // raw dword buffer (gfx9 DATA_FORMAT = 32)
}
#define GLD(rs, off) ((uint32_t)__builtin_amdgcn_raw_buffer_load_b32((rs), (int)(off), 0, 0))
// column = T0[b0(a)] ^ T1[b1(b)] ^ T2[b2(c)] ^ T3[b3(d)] ^ rk, the last G terms (T3, then T2, then T1) from the global table
template <int G>
__device__ __forceinline__ uint32_t col_h(const uint8_t *tab, const AesLane &L, v4i_ rs, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t rk) {
  if (G == 0) return mf::aes_col(tab, L, a, b, c, d, rk);
  const uint32_t x3 = GLD(rs, 3072 + ((d >> 22) & 0x3fc));
  const uint32_t x0 = MF_LD(MF_A(a, L.lo0, 0));
  if (G == 1) {
    const uint32_t x1 = MF_LD(MF_A(b, L.lo0, 1)), x2 = MF_LD(MF_A(c, L.lo2, 2));
    return MF_XOR3(x0 ^ rk, x2, __builtin_amdgcn_alignbit(x1, x1, 24)) ^ x3;
  }
  const uint32_t x2 = GLD(rs, 2048 + ((c >> 14) & 0x3fc));
  if (G == 2) {
    const uint32_t x1 = MF_LD(MF_A(b, L.lo0, 1));
    return MF_XOR3(x0 ^ rk, x2, __builtin_amdgcn_alignbit(x1, x1, 24)) ^ x3;
  }
  const uint32_t x1 = GLD(rs, 1024 + ((b >> 6) & 0x3fc));
  return MF_XOR3(x0 ^ rk, x2, x1) ^ x3;
}
// NG global lookups per round: column j gets G_j with sum G_j = NG (NG <= 4: one per column, T3 term; 5..8: a second one, ...)
template <int NG>
__device__ __forceinline__ void aes_hybrid(const uint8_t *tab, const AesLane &L, v4i_ rs, const AesKey &k, uint64_t ctr, uint32_t out[4]) {
  uint32_t s0 = k.nonce_lo ^ k.rk[0], s1 = k.nonce_hi ^ k.rk[1], s2 = (uint32_t)ctr ^ k.rk[2], s3 = (uint32_t)(ctr >> 32) ^ k.rk[3];
  constexpr int G0 = (NG + 3) / 4, G1 = (NG + 2) / 4, G2 = (NG + 1) / 4, G3 = NG / 4;
#pragma unroll
  for (int r = 1; r < 14; r++) {
    uint32_t t0 = col_h<G0>(tab, L, rs, s0, s1, s2, s3, k.rk[4 * r]);
    uint32_t t1 = col_h<G1>(tab, L, rs, s1, s2, s3, s0, k.rk[4 * r + 1]);
    uint32_t t2 = col_h<G2>(tab, L, rs, s2, s3, s0, s1, k.rk[4 * r + 2]);
    uint32_t t3 = col_h<G3>(tab, L, rs, s3, s0, s1, s2, k.rk[4 * r + 3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  out[0] = mf::aes_last(tab, L, s0, s1, s2, s3, k.rk[56]);
  out[1] = mf::aes_last(tab, L, s1, s2, s3, s0, k.rk[57]);
  out[2] = mf::aes_last(tab, L, s2, s3, s0, s1, k.rk[58]);
  out[3] = mf::aes_last(tab, L, s3, s0, s1, s2, k.rk[59]);
}
template <int NG, int MINW, int PAD>
__global__ __launch_bounds__(1024, MINW) void k_bench(AesKey key, const uint32_t *g_t0, const uint32_t *gt, uint32_t nb, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[65536 + PAD];
  if (PAD && nb == 0xffffffffu) smem[65536 + threadIdx.x % (PAD ? PAD : 1)] = 1;
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const AesLane L = mf::aes_lane();
  const v4i_ rs = make_rsrc(gt, 4096);
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < nb; i++) { uint32_t w[4]; aes_hybrid<NG>(smem, L, rs, key, base + i, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int NG, int MINW, int PAD>
static void run(const AesKey &key, const uint32_t *d_t0, const uint32_t *d_gt, uint32_t *d_out, int threads, int wgcu, std::vector<uint32_t> *res) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t nb = 256; float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_bench<NG, MINW, PAD>), dim3(256 * wgcu), dim3(threads), 0, 0, key, d_t0, d_gt, nb, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
  }
  res->resize(256); hipMemcpy(res->data(), d_out, 1024, hipMemcpyDeviceToHost);
  double blocks = 256.0 * wgcu * threads * nb;
  printf("f = %d/16 of the middle rounds' lookups from global memory, %2d waves/SIMD: %7.3f ms %7.2f Gblk/s  %s\n", NG, threads / 64 * wgcu / 4, best, blocks / best / 1e6,
         hipGetErrorString(hipGetLastError()));
}
int main() {
  uint8_t seed[40]; for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key; mf::expand_key(key, seed);
  uint32_t t0[256], gt[1024]; mf::make_t0_le(t0);
  for (int a = 0; a < 256; a++) { uint32_t v = t0[a]; gt[a] = v; gt[256 + a] = (v << 8) | (v >> 24); gt[512 + a] = (v << 16) | (v >> 16); gt[768 + a] = (v << 24) | (v >> 8); }
  uint32_t *d_t0, *d_gt, *d_out; hipMalloc(&d_t0, sizeof t0); hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice);
  hipMalloc(&d_gt, sizeof gt); hipMemcpy(d_gt, gt, sizeof gt, hipMemcpyHostToDevice); hipMalloc(&d_out, 256 * 2 * 1024 * 4);
  std::vector<uint32_t> r[10];
  // 4 waves per SIMD: 1024-thread workgroups, one per CU (the pad keeps a second out: what k_eval's tiles allow); 8: two per CU (k_encrypt_mm, k_expand_mm)
  run<0, 4, 94240>(key, d_t0, d_gt, d_out, 1024, 1, &r[0]); run<1, 4, 94240>(key, d_t0, d_gt, d_out, 1024, 1, &r[1]); run<2, 4, 94240>(key, d_t0, d_gt, d_out, 1024, 1, &r[2]);
  run<3, 4, 94240>(key, d_t0, d_gt, d_out, 1024, 1, &r[3]); run<4, 4, 94240>(key, d_t0, d_gt, d_out, 1024, 1, &r[4]);
  run<0, 8, 0>(key, d_t0, d_gt, d_out, 1024, 2, &r[5]); run<1, 8, 0>(key, d_t0, d_gt, d_out, 1024, 2, &r[6]); run<2, 8, 0>(key, d_t0, d_gt, d_out, 1024, 2, &r[7]);
  run<3, 8, 0>(key, d_t0, d_gt, d_out, 1024, 2, &r[8]); run<4, 8, 0>(key, d_t0, d_gt, d_out, 1024, 2, &r[9]);
  bool same = true; for (int i = 1; i < 10; i++) same = same && r[i] == r[0];
  printf("same keystream digests in all variants: %s\n", same ? "yes" : "NO");
  return 0;
}
