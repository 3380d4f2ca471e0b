// rate4: does a ds_read_b32 wave-instruction that touches 64 distinct banks go faster than one that touches 32 (lanes l and
// l+32 on the same bank, different rows -- the product's 32-replica table layout)?  (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int PAT>
__global__ __launch_bounds__(1024) void k(uint32_t iters, uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *l32 = (uint32_t *)smem;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) l32[i] = i * 2654435761u;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  uint32_t addr;
  if (PAT == 0) addr = (lane & 31) * 4 + (lane >> 5) * 256 + ((threadIdx.x >> 6) & 3) * 512;  // 32 banks, 2 rows
  else if (PAT == 1) addr = lane * 4 + ((threadIdx.x >> 6) & 3) * 512;                          // 64 consecutive words
  else if (PAT == 2) addr = (lane & 31) * 4 + (lane >> 5) * 128 + ((threadIdx.x >> 6) & 3) * 512;  // = PAT 1 (64 words) written like the table
  else addr = (lane & 31) * 8 + (lane >> 5) * 4;                                                 // interleaved: even/odd words
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
    uint32_t r[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "i"(i * 2048));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; i++) acc ^= r[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int PAT>
void run(const char *name, uint32_t *d_out, int threads, int wgcu) {
  hipFuncSetAttribute((const void *)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t iters = 2000; float best = 1e30f;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<PAT>), dim3(256 * wgcu), dim3(threads), 65536, 0, iters, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double waves = (double)threads / 64 * wgcu;
  double clk = best * 1e-3 * 2.4e9;
  printf("%-34s waves/CU=%2.0f: %7.3f ms  %5.2f CU-cycles per ds_read_b32 wave-instr (@2.4GHz)\n", name, waves, best, clk / (waves * iters * 64));
}
int main() {
  uint32_t *d_out; hipMalloc(&d_out, 256 * 4 * 1024 * 4);
  for (int wg = 1; wg <= 2; wg++) {
    run<0>("32 banks x 2 rows (table layout)", d_out, 1024, wg);
    run<1>("64 consecutive words", d_out, 1024, wg);
    run<2>("64 words as 2 x 32 (+128 B)", d_out, 1024, wg);
    run<3>("64 words, lanes interleaved", d_out, 1024, wg);
  }
}
