// Sustained int8 MFMA rate on random operands, two waves per SIMD, every CU: v_mfma_i32_16x16x64_i8 against v_mfma_i32_32x32x32_i8 at
// the same 128 accumulator registers per wave (the chip lowers its clock under matrix load, and the clock it holds can depend on the
// shape: MI355X_MICROARCH.md, DVFS give-back item 7).  dev tool.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_i8_rate.hip -o tools/mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const v4i *__restrict__ in, int *__restrict__ out, int iters) {
  const uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
  v4i a[4], b[8];
  for (int i = 0; i < 4; i++) a[i] = in[(tid * 12 + i) & 0xffff];
  for (int i = 0; i < 8; i++) b[i] = in[(tid * 12 + 4 + i) & 0xffff];
  if (SHAPE == 16) {
    v4i acc[32];
    for (int i = 0; i < 32; i++) acc[i] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 32; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[i & 7], acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 32; i++) s += acc[i][0] + acc[i][3];
    out[tid] = s;
  } else {
    v16i acc[8];
    for (int i = 0; i < 8; i++)
      for (int e = 0; e < 16; e++) acc[i][e] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][15];
    out[tid] = s;
  }
}
int main() {
  const int nblk = 256, nthr = 512, iters = 20000;
  v4i *in; int *out;
  hipMalloc(&in, 65536 * 16); hipMalloc(&out, nblk * nthr * 4);
  uint32_t *h = (uint32_t *)malloc(65536 * 16); uint32_t x = 12345;
  for (int i = 0; i < 65536 * 4; i++) { x = x * 1664525u + 1013904223u; h[i] = x; }
  hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape : {16, 32, 16, 32}) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(nblk), dim3(nthr), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k<32>, dim3(nblk), dim3(nthr), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per wave and iteration: 16x16x64: 32 MFMAs x 32768 ops; 32x32x32: 16 MFMAs x 65536 ops -> 1 048 576 ops either way
      const double ops = (double)nblk * (nthr / 64) * iters * 1048576.0;
      if (rep == 2) printf("shape %dx: %.2f ms  %.0f TOPS int8\n", shape, ms, ops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
