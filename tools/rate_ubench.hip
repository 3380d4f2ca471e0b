// rate_ubench.hip -- raw issue rates on gfx950: VALU op kinds, ds_read widths, mixed. Development tool.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
template <int V>
__global__ __launch_bounds__(1024) void k(uint32_t iters, uint32_t *out, uint32_t seedv) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *l32 = (uint32_t *)smem;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) l32[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a = threadIdx.x * 7 + seedv, b = a * 3 + 1, c = a ^ 0x55, d = b + 77, e = a + 9, f = b ^ 3, g = c + 1, h = d ^ 9;
  uint32_t addr = (threadIdx.x & 31) * 4 + ((threadIdx.x >> 5) & 7) * 256;
  uint32_t addr8 = (threadIdx.x & 63) * 8, addr16 = (threadIdx.x & 63) * 16;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
    if (V == 0) {  // xor chain x8 independent
#pragma unroll
      for (int r = 0; r < REP / 8; r++) { a ^= b; b ^= c; c ^= d; d ^= e; e ^= f; f ^= g; g ^= h; h ^= a; }
    } else if (V == 1) {  // v_perm
#pragma unroll
      for (int r = 0; r < REP / 8; r++) {
        a = __builtin_amdgcn_perm(a, b, 0x0c0c0400); b = __builtin_amdgcn_perm(b, c, 0x0c0c0500); c = __builtin_amdgcn_perm(c, d, 0x07060100);
        d = __builtin_amdgcn_perm(d, e, 0x01000504); e = __builtin_amdgcn_perm(e, f, 0x0c0c0400); f = __builtin_amdgcn_perm(f, g, 0x0c0c0500);
        g = __builtin_amdgcn_perm(g, h, 0x07060100); h = __builtin_amdgcn_perm(h, a, 0x01000504);
      }
    } else if (V == 2) {  // bitop3
#pragma unroll
      for (int r = 0; r < REP / 8; r++) {
        a = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); b = __builtin_amdgcn_bitop3_b32(b, c, d, 0x96); c = __builtin_amdgcn_bitop3_b32(c, d, e, 0x96);
        d = __builtin_amdgcn_bitop3_b32(d, e, f, 0x96); e = __builtin_amdgcn_bitop3_b32(e, f, g, 0x96); f = __builtin_amdgcn_bitop3_b32(f, g, h, 0x96);
        g = __builtin_amdgcn_bitop3_b32(g, h, a, 0x96); h = __builtin_amdgcn_bitop3_b32(h, a, b, 0x96);
      }
    } else if (V == 3) {  // alignbit
#pragma unroll
      for (int r = 0; r < REP / 8; r++) {
        a = __builtin_amdgcn_alignbit(a, b, 24); b = __builtin_amdgcn_alignbit(b, c, 24); c = __builtin_amdgcn_alignbit(c, d, 24); d = __builtin_amdgcn_alignbit(d, e, 24);
        e = __builtin_amdgcn_alignbit(e, f, 24); f = __builtin_amdgcn_alignbit(f, g, 24); g = __builtin_amdgcn_alignbit(g, h, 24); h = __builtin_amdgcn_alignbit(h, a, 24);
      }
    } else if (V == 4) {  // mad_u64_u32
      uint64_t x = a, y = b, z = c, w = d;
#pragma unroll
      for (int r = 0; r < REP / 4; r++) { x = (uint64_t)(uint32_t)x * e + y; y = (uint64_t)(uint32_t)y * f + z; z = (uint64_t)(uint32_t)z * g + w; w = (uint64_t)(uint32_t)w * h + x; }
      a = (uint32_t)x; b = (uint32_t)y; c = (uint32_t)z; d = (uint32_t)w;
    } else if (V == 5) {  // ds_read_b32 conflict-free, 16 in flight
      uint32_t r[16];
#pragma unroll
      for (int q = 0; q < REP / 16; q++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "i"(i * 2048));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" ::"v"(r[i]));
      }
    } else if (V == 6) {  // ds_read_b64
      uint64_t r[16];
#pragma unroll
      for (int q = 0; q < REP / 16; q++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr8), "i"(i * 2048));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" ::"v"(r[i]));
      }
    } else if (V == 7) {  // ds_read_b32 + 2 independent VALU per read (mixed)
      uint32_t r[16];
#pragma unroll
      for (int q = 0; q < REP / 16; q++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "i"(i * 2048));
          a ^= b; b ^= c;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" ::"v"(r[i]));
      }
    } else if (V == 8) {  // ds_read_b32 with data-dependent random-ish addresses in replicated layout (stride 256B, 64 entries)
      uint32_t r[16];
#pragma unroll
      for (int q = 0; q < REP / 16; q++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
          uint32_t ad = __builtin_amdgcn_perm(a, addr, 0x0c0c0400 + ((i & 3) << 8));
          asm volatile("ds_read_b32 %0, %1" : "=v"(r[i]) : "v"(ad));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) a ^= r[i];
      }
    } else if (V == 9) {  // v_lshl_or / bfe (VOP3 with literal-free)
#pragma unroll
      for (int r = 0; r < REP / 8; r++) {
        a = (a << 7) | b; b = (b >> 8) & 0xff; c = (c << 7) | d; d = (d >> 16) & 0xff; e = (e << 7) | f; f = (f >> 8) & 0xff; g = (g << 7) | h; h = (h >> 16) & 0xff;
        b += a; d += c; f += e; h += g;
      }
    }
    acc ^= a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int V>
void run(const char *name, uint32_t *d_out, int threads, int wgcu, double ops_per_iter) {
  hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t iters = 2000; float best = 1e30f;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(256 * wgcu), dim3(threads), 65536, 0, iters, d_out, (uint32_t)it);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double waves_per_cu = (double)threads / 64 * wgcu;
  double winstr = waves_per_cu * iters * ops_per_iter;  // wave-instructions per CU
  double clk = best * 1e-3 * 2.4e9;
  printf("%-28s thr=%4d wg/cu=%d: %7.3f ms  %6.3f clk per wave-instr per CU  (%.1f lane-ops/clk/CU)\n", name, threads, wgcu, best, clk / winstr, 64.0 * winstr / clk);
}
int main() {
  uint32_t *d_out; hipMalloc(&d_out, 256 * 4 * 1024 * 4);
  for (int cfg = 0; cfg < 2; cfg++) {
    int thr = cfg ? 1024 : 256, wg = cfg ? 1 : 2;
    run<0>("v_xor_b32", d_out, thr, wg, REP);
    run<1>("v_perm_b32", d_out, thr, wg, REP);
    run<2>("v_bitop3_b32", d_out, thr, wg, REP);
    run<3>("v_alignbit_b32", d_out, thr, wg, REP);
    run<4>("v_mad_u64_u32", d_out, thr, wg, REP);
    run<9>("shift/and mix (12 per 8)", d_out, thr, wg, REP * 12 / 8);
    run<5>("ds_read_b32", d_out, thr, wg, REP);
    run<6>("ds_read_b64", d_out, thr, wg, REP);
    run<7>("ds_read_b32 + 2 xor (per read)", d_out, thr, wg, REP);
    run<8>("perm + ds_read_b32 dep (per rd)", d_out, thr, wg, REP);
  }
  return 0;
}
