// aes3_ubench.hip -- what limits the T-table AES loop?  (dev tool)  Variants are separate translation-unit builds:
//   -DFAKE_LD : table lookups replaced by a 1-op VALU stand-in (VALU stream alone)
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifdef FAKE_LD
#define MF_LD(addr) ((addr) ^ 0x9e3779b9u)
#endif
#include "aes_dev.hpp"
using mf::AesKey;
template <int MINW>
__global__ __launch_bounds__(1024, MINW) void k_bench(AesKey key, const uint32_t *g_t0, uint32_t nb, uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const mf::AesLane L = mf::aes_lane();
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < nb; i++) { uint32_t w[4]; mf::aes256_ctr_block(smem, L, key, base + i, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MINW>
static void run(const char *name, const AesKey &key, const uint32_t *d_t0, uint32_t *d_out, int threads, int wgcu, size_t lds) {
  hipFuncSetAttribute((const void *)k_bench<MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t nb = 256; float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bench<MINW>, dim3(256 * wgcu), dim3(threads), lds, 0, key, d_t0, nb, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
  }
  double blocks = 256.0 * wgcu * threads * nb;
  printf("%-10s thr=%4d wg/cu=%d waves/SIMD=%2d: %7.3f ms %7.2f Gblk/s %6.2f clk/blk/CU@2.4GHz %s\n", name, threads, wgcu, threads / 64 * wgcu / 4, best,
         blocks / best / 1e6, 256.0 * 2.4e9 / (blocks / (best * 1e-3)), hipGetErrorString(hipGetLastError()));
}
int main() {
  uint8_t seed[40]; for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key; mf::expand_key(key, seed);
  uint32_t t0[256]; mf::make_t0_le(t0);
  uint32_t *d_t0, *d_out; hipMalloc(&d_t0, sizeof t0); hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice); hipMalloc(&d_out, 256 * 2 * 1024 * 4);
#ifdef FAKE_LD
  const char *n = "VALU-only";
#else
  const char *n = "real";
#endif
  run<1>(n, key, d_t0, d_out, 256, 1, 65536);
  run<2>(n, key, d_t0, d_out, 512, 1, 65536);
  run<4>(n, key, d_t0, d_out, 1024, 1, 65536 + 94240);
  run<4>(n, key, d_t0, d_out, 512, 2, 65536);
  run<8>(n, key, d_t0, d_out, 1024, 2, 65536);
  return 0;
}
