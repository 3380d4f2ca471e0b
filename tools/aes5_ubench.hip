// aes5_ubench.hip -- can the vector-L1 (global load) path carry part of the T-table lookups in parallel with LDS?  (dev tool)
// NG = number of lookups per round served from a 4 KiB global table {T0,T1,T2,T3} instead of LDS (column 3, bytes 3,2,1,0 in that order).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "aes_dev.hpp"
using mf::AesKey;
using mf::AesLane;
template <int NG>
__device__ __forceinline__ void aes_hybrid(const uint8_t *tab, const AesLane &L, const uint32_t *__restrict__ gt, const AesKey &k, uint64_t ctr, uint32_t out[4]) {
  uint32_t s0 = k.nonce_lo ^ k.rk[0], s1 = k.nonce_hi ^ k.rk[1], s2 = (uint32_t)ctr ^ k.rk[2], s3 = (uint32_t)(ctr >> 32) ^ k.rk[3];
#pragma unroll
  for (int r = 1; r < 14; r++) {
    uint32_t t0 = mf::aes_col(tab, L, s0, s1, s2, s3, k.rk[4 * r]);
    uint32_t t1 = mf::aes_col(tab, L, s1, s2, s3, s0, k.rk[4 * r + 1]);
    uint32_t t2 = mf::aes_col(tab, L, s2, s3, s0, s1, k.rk[4 * r + 2]);
    uint32_t t3;
    if (NG == 0) {
      t3 = mf::aes_col(tab, L, s3, s0, s1, s2, k.rk[4 * r + 3]);
    } else {
      // column 3 = T0[b0(s3)] ^ T1[b1(s0)] ^ T2[b2(s1)] ^ T3[b3(s2)] ^ rk ; the last NG terms come from global tables (no rotation needed)
      uint32_t x0 = NG >= 4 ? gt[s3 & 255] : MF_LD(MF_A(s3, L.lo0, 0));
      uint32_t x1 = NG >= 3 ? gt[256 + ((s0 >> 8) & 255)] : __builtin_amdgcn_alignbit(MF_LD(MF_A(s0, L.lo0, 1)), MF_LD(MF_A(s0, L.lo0, 1)), 24);
      uint32_t x2 = NG >= 2 ? gt[512 + ((s1 >> 16) & 255)] : MF_LD(MF_A(s1, L.lo2, 2));
      uint32_t x3 = gt[768 + (s2 >> 24)];
      t3 = MF_XOR3(x0 ^ k.rk[4 * r + 3], x1, x2) ^ x3;
    }
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  out[0] = mf::aes_last(tab, L, s0, s1, s2, s3, k.rk[56]);
  out[1] = mf::aes_last(tab, L, s1, s2, s3, s0, k.rk[57]);
  out[2] = mf::aes_last(tab, L, s2, s3, s0, s1, k.rk[58]);
  out[3] = mf::aes_last(tab, L, s3, s0, s1, s2, k.rk[59]);
}
template <int NG>
__global__ __launch_bounds__(1024) void k_bench(AesKey key, const uint32_t *g_t0, const uint32_t *gt, uint32_t nb, uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const AesLane L = mf::aes_lane();
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < nb; i++) { uint32_t w[4]; aes_hybrid<NG>(smem, L, gt, key, base + i, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int NG>
static void run(const AesKey &key, const uint32_t *d_t0, const uint32_t *d_gt, uint32_t *d_out, std::vector<uint32_t> *res) {
  hipFuncSetAttribute((const void *)k_bench<NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 94240);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t nb = 256; float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bench<NG>, dim3(256), dim3(1024), 65536 + 94240, 0, key, d_t0, d_gt, nb, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
  }
  res->resize(1024); hipMemcpy(res->data(), d_out, 4096, hipMemcpyDeviceToHost);
  double blocks = 256.0 * 1024 * nb;
  printf("global lookups per round = %d: %7.3f ms %7.2f Gblk/s %6.2f clk/blk/CU@2.4GHz\n", NG, best, blocks / best / 1e6, 256.0 * 2.4e9 / (blocks / (best * 1e-3)));
}
int main() {
  uint8_t seed[40]; for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key; mf::expand_key(key, seed);
  uint32_t t0[256], gt[1024]; mf::make_t0_le(t0);
  for (int a = 0; a < 256; a++) { uint32_t v = t0[a]; gt[a] = v; gt[256 + a] = (v << 8) | (v >> 24); gt[512 + a] = (v << 16) | (v >> 16); gt[768 + a] = (v << 24) | (v >> 8); }
  uint32_t *d_t0, *d_gt, *d_out; hipMalloc(&d_t0, sizeof t0); hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice);
  hipMalloc(&d_gt, sizeof gt); hipMemcpy(d_gt, gt, sizeof gt, hipMemcpyHostToDevice); hipMalloc(&d_out, 256 * 1024 * 4);
  std::vector<uint32_t> a, b, c, d, e;
  run<0>(key, d_t0, d_gt, d_out, &a); run<1>(key, d_t0, d_gt, d_out, &b); run<2>(key, d_t0, d_gt, d_out, &c); run<3>(key, d_t0, d_gt, d_out, &d); run<4>(key, d_t0, d_gt, d_out, &e);
  printf("same results: %s\n", (a == b && a == c && a == d && a == e) ? "yes" : "NO");
  return 0;
}
