// mul_rate_ubench.hip -- issue cost of the integer multiply instructions on gfx950 (inline asm, 8 independent chains per wave, 8 waves per
// SIMD on every CU): which of them are quarter rate.  Development tool.  hipcc -O3 --offload-arch=gfx950 tools/mul_rate_ubench.hip -o tools/mul_rate_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define OP2(name) asm volatile(name " %0, %0, %1" : "+v"(a[i]) : "v"(b))
#define OP3(name) asm volatile(name " %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c))
template <int V>
__global__ __launch_bounds__(512) void k(uint32_t iters, uint32_t *out, uint32_t seedv) {
  uint32_t a[8], b = threadIdx.x * 2654435761u + seedv, c = b ^ 0x9e3779b9u;
  for (int i = 0; i < 8; i++) a[i] = b * (i + 3);
  uint64_t w = 0;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (V == 0) OP2("v_xor_b32");
        if (V == 1) OP2("v_mul_lo_u32");
        if (V == 2) OP2("v_mul_hi_u32");
        if (V == 3) OP2("v_mul_u32_u24");
        if (V == 4) OP2("v_mul_hi_u32_u24");
        if (V == 5) OP3("v_mad_u32_u24");
        if (V == 6) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (V == 7) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (V == 8) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (V == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(a[i]), "v"(b) : "vcc");
        if (V == 10) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
  }
  uint32_t s = (uint32_t)w;
  for (int i = 0; i < 8; i++) s ^= a[i];
  if (s == 0x12345678) out[0] = s;
}
template <int V>
void run(const char *name) {
  uint32_t *out;
  hipMalloc(&out, 4);
  const uint32_t iters = 2000;
  hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(512), 0, 0, 10u, out, 1u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(512), 0, 0, iters, out, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // 256 CUs x 4 WGs x 8 waves = 8 waves per SIMD; wave-instructions per SIMD = 8 waves x iters x 128
  const double winst = 8.0 * iters * 128;
  printf("%-20s %8.3f ms  %6.2f ns per wave-instruction per SIMD (x 2.4 GHz = %5.1f clk)\n", name, ms, ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
  hipFree(out);
}
int main() {
  run<0>("v_xor_b32");
  run<1>("v_mul_lo_u32");
  run<2>("v_mul_hi_u32");
  run<3>("v_mul_u32_u24");
  run<4>("v_mul_hi_u32_u24");
  run<5>("v_mad_u32_u24");
  run<6>("v_dot2_u32_u16");
  run<7>("v_dot4_u32_u8");
  run<8>("v_pk_mul_lo_u16");
  run<9>("v_mad_u64_u32");
  run<10>("v_pk_mad_u16");
  return 0;
}
