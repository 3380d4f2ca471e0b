// rate3: ds_read_b32 + N VALU per read: how much VALU hides under LDS? (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int N, int SLOW>
__global__ __launch_bounds__(1024) void k(uint32_t iters, uint32_t *out, uint32_t sc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *l32 = (uint32_t *)smem;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) l32[i] = i * 2654435761u;
  __syncthreads();
  uint32_t addr = (threadIdx.x & 31) * 4 + ((threadIdx.x >> 5) & 7) * 256;
  uint32_t a = threadIdx.x + sc, b = a * 3, c = a ^ 5, d = b + 7;
  for (uint32_t it = 0; it < iters; it++) {
    uint32_t r[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "i"(i * 2048));
#pragma unroll
        for (int n = 0; n < N; n++) {
          if (SLOW) asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(a) : "v"(b));
          else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(b));
          uint32_t t = a; a = b; b = c; c = d; d = t;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("" ::"v"(r[i]));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
template <int N, int SLOW>
void run(uint32_t *d_out, int threads, int wgcu) {
  hipFuncSetAttribute((const void *)k<N, SLOW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t iters = 2000; float best = 1e30f;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<N, SLOW>), dim3(256 * wgcu), dim3(threads), 65536, 0, iters, d_out, 5u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double wps = (double)threads / 64 * wgcu / 4;
  double clk = best * 1e-3 * 2.4e9;
  printf("N=%d %s waves/SIMD=%.0f: %7.3f ms  %6.2f SIMD-cycles per read (LDS-only floor ~8.6)\n", N, SLOW ? "slow(4c)" : "fast(2c)", wps, best, clk / (wps * iters * 64));
}
int main() {
  uint32_t *d_out; hipMalloc(&d_out, 256 * 4 * 1024 * 4);
  for (int cfg = 0; cfg < 2; cfg++) {
    int thr = 1024, wg = cfg + 1;
    run<0, 0>(d_out, thr, wg); run<1, 0>(d_out, thr, wg); run<2, 0>(d_out, thr, wg); run<3, 0>(d_out, thr, wg); run<4, 0>(d_out, thr, wg); run<6, 0>(d_out, thr, wg); run<8, 0>(d_out, thr, wg);
    run<1, 1>(d_out, thr, wg); run<2, 1>(d_out, thr, wg); run<3, 1>(d_out, thr, wg); run<4, 1>(d_out, thr, wg);
  }
}
