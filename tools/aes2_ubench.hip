// aes2_ubench.hip -- does a software-pipelined 2-block interleave beat the compiler's schedule? (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "aes_dev.hpp"
using mf::AesKey;
using mf::AesLane;

// half-round pieces on explicit arrays so the order of issue is ours
#define ADDR4(dst, s0, s1, s2, s3)                       \
  dst[0] = MF_A(s0, L.lo0, 0); dst[1] = MF_A(s1, L.lo0, 1); dst[2] = MF_A(s2, L.lo2, 2); dst[3] = MF_A(s3, L.lo2, 3);
#define LOAD4(x, a) x[0] = MF_LD(a[0]); x[1] = MF_LD(a[1]); x[2] = MF_LD(a[2]); x[3] = MF_LD(a[3]);
#define COMB(x, rk) ({ uint32_t y_ = x[1] ^ x[3]; MF_XOR3(x[0] ^ (rk), x[2], __builtin_amdgcn_alignbit(y_, y_, 24)); })

struct Blk { uint32_t s[4]; uint32_t x[16]; };

__device__ __forceinline__ void issue_round(const uint8_t *tab, const AesLane &L, Blk &b) {
  uint32_t a[16];
  ADDR4((a + 0), b.s[0], b.s[1], b.s[2], b.s[3]);
  ADDR4((a + 4), b.s[1], b.s[2], b.s[3], b.s[0]);
  ADDR4((a + 8), b.s[2], b.s[3], b.s[0], b.s[1]);
  ADDR4((a + 12), b.s[3], b.s[0], b.s[1], b.s[2]);
#pragma unroll
  for (int i = 0; i < 16; i++) b.x[i] = MF_LD(a[i]);
}
__device__ __forceinline__ void finish_round(Blk &b, const AesKey &k, int r) {
  uint32_t t0 = COMB((b.x + 0), k.rk[4 * r]), t1 = COMB((b.x + 4), k.rk[4 * r + 1]);
  uint32_t t2 = COMB((b.x + 8), k.rk[4 * r + 2]), t3 = COMB((b.x + 12), k.rk[4 * r + 3]);
  b.s[0] = t0; b.s[1] = t1; b.s[2] = t2; b.s[3] = t3;
}
__device__ __forceinline__ void last_round(const uint8_t *tab, const AesLane &L, Blk &b, const AesKey &k, uint32_t out[4]) {
  out[0] = mf::aes_last(tab, L, b.s[0], b.s[1], b.s[2], b.s[3], k.rk[56]);
  out[1] = mf::aes_last(tab, L, b.s[1], b.s[2], b.s[3], b.s[0], k.rk[57]);
  out[2] = mf::aes_last(tab, L, b.s[2], b.s[3], b.s[0], b.s[1], k.rk[58]);
  out[3] = mf::aes_last(tab, L, b.s[3], b.s[0], b.s[1], b.s[2], k.rk[59]);
}
__device__ __forceinline__ void init_blk(Blk &b, const AesKey &k, uint64_t ctr) {
  b.s[0] = k.nonce_lo ^ k.rk[0]; b.s[1] = k.nonce_hi ^ k.rk[1];
  b.s[2] = (uint32_t)ctr ^ k.rk[2]; b.s[3] = (uint32_t)(ctr >> 32) ^ k.rk[3];
}
// two blocks, B runs half a round behind A: A's reads are in flight while B combines, and vice versa
__device__ __forceinline__ void aes2_pipelined(const uint8_t *tab, const AesLane &L, const AesKey &k, uint64_t c0, uint64_t c1, uint32_t o0[4], uint32_t o1[4]) {
  Blk A, B;
  init_blk(A, k, c0);
  init_blk(B, k, c1);
  issue_round(tab, L, A);
#pragma unroll
  for (int r = 1; r < 14; r++) {
    issue_round(tab, L, B);       // B round r lookups go out
    finish_round(A, k, r);        // A round r combine (its reads were issued before B's)
    if (r < 13) issue_round(tab, L, A);  // A round r+1 lookups
    finish_round(B, k, r);
  }
  last_round(tab, L, A, k, o0);
  last_round(tab, L, B, k, o1);
}

template <int V>
__global__ __launch_bounds__(1024) void k_bench(AesKey key, const uint32_t *g_t0, uint32_t nb, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[65536 + 94240];  // static, table at LDS address 0 (see aes3_ubench.hip); 1 WG/CU
  if (nb == 0xffffffffu) smem[65536 + threadIdx.x] = 1;
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const AesLane L = mf::aes_lane();
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  if (V == 0) {
    for (uint32_t i = 0; i < nb; i++) { uint32_t w[4]; mf::aes256_ctr_block(smem, L, key, base + i, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
  } else {
    for (uint32_t i = 0; i < nb; i += 2) {
      uint32_t w[4], x[4];
      aes2_pipelined(smem, L, key, base + i, base + i + 1, w, x);
      acc ^= w[0] ^ w[1] ^ w[2] ^ w[3] ^ (x[0] + x[1] + x[2] + x[3]);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void k_ref(AesKey key, const uint32_t *g_t0, uint32_t nb, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[65536 + 94240];  // static, table at LDS address 0 (see aes3_ubench.hip); 1 WG/CU
  if (nb == 0xffffffffu) smem[65536 + threadIdx.x] = 1;
  mf::lds_fill_tab(reinterpret_cast<uint32_t *>(smem), g_t0);
  __syncthreads();
  const AesLane L = mf::aes_lane();
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nb;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < nb; i += 2) {
    uint32_t w[4], x[4];
    mf::aes256_ctr_block(smem, L, key, base + i, w);
    mf::aes256_ctr_block(smem, L, key, base + i + 1, x);
    acc ^= w[0] ^ w[1] ^ w[2] ^ w[3] ^ (x[0] + x[1] + x[2] + x[3]);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <typename F>
static void run(const char *name, F kern, const AesKey &key, const uint32_t *d_t0, uint32_t *d_out, int threads, size_t lds) {

  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t nb = 256; float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, key, d_t0, nb, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
  }
  double blocks = 256.0 * threads * nb;
  printf("%-28s thr=%4d: %7.3f ms %7.2f Gblk/s %6.2f clk/blk/CU@2.4GHz %s\n", name, threads, best, blocks / best / 1e6, 256.0 * 2.4e9 / (blocks / (best * 1e-3)), hipGetErrorString(hipGetLastError()));
}
int main() {
  uint8_t seed[40]; for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key; mf::expand_key(key, seed);
  uint32_t t0[256]; mf::make_t0_le(t0);
  uint32_t *d_t0, *d_out; hipMalloc(&d_t0, sizeof t0); hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice); hipMalloc(&d_out, 256 * 1024 * 4);
  std::vector<uint32_t> a(1024), b(1024);
  hipLaunchKernelGGL(k_ref, dim3(1), dim3(1024), 0, 0, key, d_t0, 4u, d_out); hipMemcpy(a.data(), d_out, 4096, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(1024), 0, 0, key, d_t0, 4u, d_out); hipMemcpy(b.data(), d_out, 4096, hipMemcpyDeviceToHost);
  printf("pipelined == reference: %s\n", memcmp(a.data(), b.data(), 4096) ? "NO" : "yes");
  run("single block (product)", k_bench<0>, key, d_t0, d_out, 1024, 65536 + 94240);
  run("2 blocks sequential", k_ref, key, d_t0, d_out, 1024, 65536 + 94240);
  run("2 blocks sw-pipelined", k_bench<1>, key, d_t0, d_out, 1024, 65536 + 94240);
  run("2 blocks sw-pipelined", k_bench<1>, key, d_t0, d_out, 512, 65536 + 94240);
  return 0;
}
