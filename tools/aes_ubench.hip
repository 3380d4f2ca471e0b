// aes_ubench.hip -- microbenchmark of AES-256-CTR block variants on gfx950 (development tool, not product).
// build: hipcc -O3 --offload-arch=gfx950 -I c-lwe-snarks_amd/csrc tools/aes_ubench.hip -o tools/aes_ubench
// Measures Gblock/s and cycles/block/CU for: V0 current (128-B stride table, bfe+lshl_or), V1 64-KiB T0|T2 table
// with v_perm addressing + bitop3, V2 = V1 two blocks interleaved, V3 = V1 with lookups replaced by VALU (VALU-only
// time), V4 = LDS-only chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "aes_dev.hpp"
using mf::AesKey;

#define XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)

// ---- V1 building blocks: table word index a*64 + r ; r<32: T0, r>=32: T2 = rotl16(T0) -----------------
struct Sel { uint32_t k0, k1, k2, k3; };
__device__ __forceinline__ uint32_t rotl8(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 24); }

template <bool FAKE>
__device__ __forceinline__ uint32_t LD(const uint8_t *lds, uint32_t addr) {
  if (FAKE) return addr * 0x9e3779b1u;  // keeps a data dependency, no LDS
  return *reinterpret_cast<const uint32_t *>(lds + addr);
}

template <bool FAKE>
__device__ __forceinline__ void aes_v1(const uint8_t *lds, uint32_t lo0, uint32_t lo2, const AesKey &k, uint64_t ctr, uint32_t out[4]) {
  const uint32_t K0 = 0x0c0c0400u, K1 = 0x0c0c0500u, K2 = 0x0c0c0600u, K3 = 0x0c0c0700u;
  uint32_t s0 = k.nonce_lo ^ k.rk[0], s1 = k.nonce_hi ^ k.rk[1];
  uint32_t s2 = (uint32_t)ctr ^ k.rk[2], s3 = (uint32_t)(ctr >> 32) ^ k.rk[3];
#define A0(s, K) __builtin_amdgcn_perm((s), lo0, (K))
#define A2(s, K) __builtin_amdgcn_perm((s), lo2, (K))
#define COL(a, b, c, d, rk) (XOR3(LD<FAKE>(lds, A0(a, K0)), LD<FAKE>(lds, A2(c, K2)), rotl8(LD<FAKE>(lds, A0(b, K1)) ^ LD<FAKE>(lds, A2(d, K3)))) ^ (rk))
#pragma unroll
  for (int r = 1; r < 14; r++) {
    uint32_t t0 = COL(s0, s1, s2, s3, k.rk[4 * r]);
    uint32_t t1 = COL(s1, s2, s3, s0, k.rk[4 * r + 1]);
    uint32_t t2 = COL(s2, s3, s0, s1, k.rk[4 * r + 2]);
    uint32_t t3 = COL(s3, s0, s1, s2, k.rk[4 * r + 3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  // last round: T2.byte0 = S, T0.byte1 = S, T0.byte2 = S, T2.byte3 = S
#define BFI(m, a, b) (((a) & (m)) | ((b) & ~(m)))
#define LAST(a, b, c, d, rk)                                                                           \
  (BFI(0x0000ffffu, BFI(0x000000ffu, LD<FAKE>(lds, A2(a, K0)), LD<FAKE>(lds, A0(b, K1))),              \
       BFI(0x00ff0000u, LD<FAKE>(lds, A0(c, K2)), LD<FAKE>(lds, A2(d, K3)))) ^ (rk))
  out[0] = LAST(s0, s1, s2, s3, k.rk[56]);
  out[1] = LAST(s1, s2, s3, s0, k.rk[57]);
  out[2] = LAST(s2, s3, s0, s1, k.rk[58]);
  out[3] = LAST(s3, s0, s1, s2, k.rk[59]);
#undef A0
#undef A2
}

__device__ __forceinline__ void fill_t02(uint32_t *lt, const uint32_t *g_t0) {
  for (int i = threadIdx.x; i < 256 * 64; i += blockDim.x) {
    uint32_t v = g_t0[i >> 6];
    lt[i] = (i & 32) ? ((v << 16) | (v >> 16)) : v;
  }
}

template <int V>
__global__ __launch_bounds__(1024) void k_bench(AesKey key, const uint32_t *g_t0, uint32_t nblk_per_thread, uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *lt = reinterpret_cast<uint32_t *>(smem);
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * nblk_per_thread;
  uint32_t acc = 0;
  if (V == 0) {
    mf::lds_fill_t0(lt, g_t0);
    __syncthreads();
    const uint32_t *tl = lt + (threadIdx.x & 31);
    for (uint32_t i = 0; i < nblk_per_thread; i++) {
      uint32_t w[4];
      mf::aes256_ctr_block(tl, key, base + i, w);
      acc ^= w[0] ^ w[1] ^ w[2] ^ w[3];
    }
  } else if (V == 1 || V == 3) {
    fill_t02(lt, g_t0);
    __syncthreads();
    const uint32_t lo0 = (threadIdx.x & 31) * 4, lo2 = lo0 + 128;
    for (uint32_t i = 0; i < nblk_per_thread; i++) {
      uint32_t w[4];
      aes_v1<V == 3>(smem, lo0, lo2, key, base + i, w);
      acc ^= w[0] ^ w[1] ^ w[2] ^ w[3];
    }
  } else if (V == 2) {
    fill_t02(lt, g_t0);
    __syncthreads();
    const uint32_t lo0 = (threadIdx.x & 31) * 4, lo2 = lo0 + 128;
    for (uint32_t i = 0; i < nblk_per_thread; i += 2) {
      uint32_t w[4], x[4];
      aes_v1<false>(smem, lo0, lo2, key, base + i, w);
      aes_v1<false>(smem, lo0, lo2, key, base + i + 1, x);
      acc ^= w[0] ^ w[1] ^ w[2] ^ w[3] ^ x[0] ^ x[1] ^ x[2] ^ x[3];
    }
  } else if (V == 4) {  // LDS only: 224 dependent-ish lookups per "block", 1 VALU op per lookup
    fill_t02(lt, g_t0);
    __syncthreads();
    const uint32_t lo0 = (threadIdx.x & 31) * 4;
    uint32_t s0 = (uint32_t)base, s1 = s0 * 3, s2 = s0 * 5, s3 = s0 * 7;
    for (uint32_t i = 0; i < nblk_per_thread; i++) {
#pragma unroll
      for (int r = 0; r < 14; r++) {
        uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          t0 += *reinterpret_cast<const uint32_t *>(smem + ((s0 & (0xff00u)) | lo0) + q * 0);
          t1 += *reinterpret_cast<const uint32_t *>(smem + ((s1 & (0xff00u)) | lo0) + q * 0);
          t2 += *reinterpret_cast<const uint32_t *>(smem + ((s2 & (0xff00u)) | lo0) + q * 0);
          t3 += *reinterpret_cast<const uint32_t *>(smem + ((s3 & (0xff00u)) | lo0) + q * 0);
          s0 = __builtin_amdgcn_alignbit(s0, s0, 8); s1 = __builtin_amdgcn_alignbit(s1, s1, 8);
          s2 = __builtin_amdgcn_alignbit(s2, s2, 8); s3 = __builtin_amdgcn_alignbit(s3, s3, 8);
        }
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
      }
      acc ^= s0 ^ s1 ^ s2 ^ s3;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int V>
static void run(const char *name, const AesKey &key, const uint32_t *d_t0, uint32_t *d_out, int threads, int wg_per_cu, size_t lds, uint32_t nb) {
  int grid = 256 * wg_per_cu;
  hipFuncSetAttribute((const void *)k_bench<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bench<V>, dim3(grid), dim3(threads), lds, 0, key, d_t0, nb, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  hipError_t e = hipGetLastError();
  double blocks = (double)grid * threads * nb;
  printf("%-34s thr=%4d wg/cu=%d lds=%6zu: %8.3f ms  %7.2f Gblk/s  %6.2f clk/blk/CU @2.4GHz %s\n", name, threads, wg_per_cu, lds, best,
         blocks / best / 1e6, 256.0 * 2.4e9 / (blocks / (best * 1e-3)), e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
  uint8_t seed[40];
  for (int i = 0; i < 40; i++) seed[i] = (uint8_t)i;
  AesKey key;
  mf::expand_key(key, seed);
  uint32_t t0[256];
  mf::make_t0_le(t0);
  uint32_t *d_t0, *d_out;
  hipMalloc(&d_t0, sizeof t0);
  hipMemcpy(d_t0, t0, sizeof t0, hipMemcpyHostToDevice);
  hipMalloc(&d_out, 256 * 8 * 1024 * 4);
  // correctness of V1 vs V0 on one block set
  {
    hipFuncSetAttribute((const void *)k_bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipFuncSetAttribute((const void *)k_bench<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_bench<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<uint32_t> a(512), b(512), c(512);
    hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(512), 32768, 0, key, d_t0, 4u, d_out);
    hipMemcpy(a.data(), d_out, 2048, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(512), 65536, 0, key, d_t0, 4u, d_out);
    hipMemcpy(b.data(), d_out, 2048, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(512), 65536, 0, key, d_t0, 4u, d_out);
    hipMemcpy(c.data(), d_out, 2048, hipMemcpyDeviceToHost);
    printf("V1==V0: %s   V2==V0: %s\n", memcmp(a.data(), b.data(), 2048) ? "NO" : "yes", memcmp(a.data(), c.data(), 2048) ? "NO" : "yes");
  }
  const uint32_t nb = 256;
  run<0>("V0 current (32K tbl)", key, d_t0, d_out, 256, 5, 32768, nb);
  run<0>("V0 current (32K tbl)", key, d_t0, d_out, 512, 2, 32768 + 47120, nb);
  run<0>("V0 current (32K tbl)", key, d_t0, d_out, 1024, 1, 32768, nb);
  run<1>("V1 perm+bitop3 (64K tbl)", key, d_t0, d_out, 256, 2, 65536, nb);
  run<1>("V1 perm+bitop3 (64K tbl)", key, d_t0, d_out, 512, 2, 65536, nb);
  run<1>("V1 perm+bitop3 (64K tbl)", key, d_t0, d_out, 512, 1, 65536 + 47120, nb);
  run<1>("V1 perm+bitop3 (64K tbl)", key, d_t0, d_out, 1024, 1, 65536 + 94224, nb);
  run<2>("V2 = V1 x2 interleaved", key, d_t0, d_out, 512, 2, 65536, nb);
  run<2>("V2 = V1 x2 interleaved", key, d_t0, d_out, 512, 1, 65536 + 47120, nb);
  run<2>("V2 = V1 x2 interleaved", key, d_t0, d_out, 1024, 1, 65536 + 94224, nb);
  run<3>("V3 = V1 VALU only", key, d_t0, d_out, 512, 2, 65536, nb);
  run<3>("V3 = V1 VALU only", key, d_t0, d_out, 1024, 1, 65536 + 94224, nb);
  run<4>("V4 LDS only (224 lookups)", key, d_t0, d_out, 512, 2, 65536, nb);
  run<4>("V4 LDS only (224 lookups)", key, d_t0, d_out, 1024, 1, 65536 + 94224, nb);
  return 0;
}
