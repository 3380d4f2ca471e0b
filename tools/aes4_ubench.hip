// aes4_ubench.hip -- round issued as [16 addresses][16 ds_reads][combine] with sched_barrier pins vs the compiler's own order (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "aes_dev.hpp"
using mf::AesKey;
using mf::AesLane;
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int PIN>
__device__ __forceinline__ void round16(const uint8_t *tab, const AesLane &L, uint32_t s[4], const uint32_t *rk) {
  uint32_t a[16], x[16];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    a[4 * j + 0] = MF_A(s[j], L.lo0, 0);
    a[4 * j + 1] = MF_A(s[(j + 1) & 3], L.lo0, 1);
    a[4 * j + 2] = MF_A(s[(j + 2) & 3], L.lo2, 2);
    a[4 * j + 3] = MF_A(s[(j + 3) & 3], L.lo2, 3);
  }
  if (PIN) SB();
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = MF_LD(a[i]);
  if (PIN) SB();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t y = x[4 * j + 1] ^ x[4 * j + 3];
    s[j] = MF_XOR3(x[4 * j] ^ rk[j], x[4 * j + 2], __builtin_amdgcn_alignbit(y, y, 24));
  }
  if (PIN == 2) SB();
}
template <int PIN>
__global__ __launch_bounds__(1024) void k_bench(AesKey key, const uint32_t *g_t0, uint32_t nb, uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const AesLane L = mf::aes_lane();
  const uint8_t *tab = smem;
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < nb; i++) {
    const uint64_t ctr = base + i;
    uint32_t s[4] = {key.nonce_lo ^ key.rk[0], key.nonce_hi ^ key.rk[1], (uint32_t)ctr ^ key.rk[2], (uint32_t)(ctr >> 32) ^ key.rk[3]};
#pragma unroll
    for (int r = 1; r < 14; r++) round16<PIN>(tab, L, s, key.rk + 4 * r);
    uint32_t w0 = mf::aes_last(tab, L, s[0], s[1], s[2], s[3], key.rk[56]), w1 = mf::aes_last(tab, L, s[1], s[2], s[3], s[0], key.rk[57]);
    uint32_t w2 = mf::aes_last(tab, L, s[2], s[3], s[0], s[1], key.rk[58]), w3 = mf::aes_last(tab, L, s[3], s[0], s[1], s[2], key.rk[59]);
    acc ^= w0 ^ w1 ^ w2 ^ w3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int PIN>
static void run(const char *name, const AesKey &key, const uint32_t *d_t0, uint32_t *d_out, std::vector<uint32_t> *res) {
  hipFuncSetAttribute((const void *)k_bench<PIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 94240);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t nb = 256; float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bench<PIN>, dim3(256), dim3(1024), 65536 + 94240, 0, key, d_t0, nb, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
  }
  res->resize(1024); hipMemcpy(res->data(), d_out, 4096, hipMemcpyDeviceToHost);
  double blocks = 256.0 * 1024 * nb;
  printf("%-34s %7.3f ms %7.2f Gblk/s %6.2f clk/blk/CU@2.4GHz\n", name, best, blocks / best / 1e6, 256.0 * 2.4e9 / (blocks / (best * 1e-3)));
}
int main() {
  uint8_t seed[40]; for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key; mf::expand_key(key, seed);
  uint32_t t0[256]; mf::make_t0_le(t0);
  uint32_t *d_t0, *d_out; hipMalloc(&d_t0, sizeof t0); hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice); hipMalloc(&d_out, 256 * 1024 * 4);
  std::vector<uint32_t> a, b, c;
  run<0>("compiler order", key, d_t0, d_out, &a);
  run<1>("pinned: 16 addr | 16 reads | combine", key, d_t0, d_out, &b);
  run<2>("pinned + round boundary", key, d_t0, d_out, &c);
  printf("same results: %s\n", (a == b && a == c) ? "yes" : "NO");
  return 0;
}
