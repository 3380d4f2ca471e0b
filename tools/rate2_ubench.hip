// rate2_ubench.hip -- per-instruction VALU issue cost on gfx950 via inline asm (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define A8(INS) \
  asm volatile(INS(%0, %1, %2) "\n" INS(%1, %2, %3) "\n" INS(%2, %3, %4) "\n" INS(%3, %4, %5) "\n" INS(%4, %5, %6) "\n" INS(%5, %6, %7) "\n" INS(%6, %7, %0) "\n" INS(%7, %0, %1) \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(sc), "v"(kk) )
#define I_XOR(x, y, z) "v_xor_b32 " #x ", " #y ", " #z
#define I_AND(x, y, z) "v_and_b32 " #x ", " #y ", " #z
#define I_ADD(x, y, z) "v_add_u32 " #x ", " #y ", " #z
#define I_LSHR(x, y, z) "v_lshrrev_b32 " #x ", 8, " #z
#define I_BFE(x, y, z) "v_bfe_u32 " #x ", " #y ", 8, 8"
#define I_LSHLOR(x, y, z) "v_lshl_or_b32 " #x ", " #y ", 7, " #z
#define I_ANDOR(x, y, z) "v_and_or_b32 " #x ", " #y ", %8, " #z
#define I_PERM(x, y, z) "v_perm_b32 " #x ", " #y ", " #z ", %8"
#define I_PERMV(x, y, z) "v_perm_b32 " #x ", " #y ", " #z ", %9"
#define I_BITOP3(x, y, z) "v_bitop3_b32 " #x ", " #x ", " #y ", " #z " bitop3:0x96"
#define I_BITOP3S(x, y, z) "v_bitop3_b32 " #x ", " #y ", " #z ", %8 bitop3:0x96"
#define I_ALIGN(x, y, z) "v_alignbit_b32 " #x ", " #y ", " #y ", 24"
#define I_ALIGN3(x, y, z) "v_alignbit_b32 " #x ", " #y ", " #z ", 24"
#define I_BFI(x, y, z) "v_bfi_b32 " #x ", %8, " #y ", " #z
#define I_MULLO(x, y, z) "v_mul_lo_u32 " #x ", " #y ", " #z
#define I_MULHI(x, y, z) "v_mul_hi_u32 " #x ", " #y ", " #z
#define I_MAD24(x, y, z) "v_mad_u32_u24 " #x ", " #y ", " #z ", " #x
#define I_SDWA(x, y, z) "v_lshlrev_b32_sdwa " #x ", %9, " #z " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
#define I_XORSDWA(x, y, z) "v_xor_b32_sdwa " #x ", " #y ", " #z " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
#define I_MOV(x, y, z) "v_mov_b32 " #x ", " #y
#define I_DPP(x, y, z) "v_mov_b32_dpp " #x ", " #y " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_XOR_E64(x, y, z) "v_xor_b32_e64 " #x ", " #y ", " #z
#define I_ADD3(x, y, z) "v_add3_u32 " #x ", " #x ", " #y ", " #z
#define I_LSHLADD(x, y, z) "v_lshl_add_u32 " #x ", " #y ", 7, " #z
template <int V>
__global__ __launch_bounds__(1024) void k(uint32_t iters, uint32_t *out, uint32_t sc) {
  uint32_t a = threadIdx.x * 7 + sc, b = a * 3 + 1, c = a ^ 0x55, d = b + 77, e = a + 9, f = b ^ 3, g = c + 1, h = d ^ 9, kk = 0x0c0c0400;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if (V == 0) A8(I_XOR); if (V == 1) A8(I_AND); if (V == 2) A8(I_ADD); if (V == 3) A8(I_LSHR); if (V == 4) A8(I_BFE);
      if (V == 5) A8(I_LSHLOR); if (V == 6) A8(I_ANDOR); if (V == 7) A8(I_PERM); if (V == 8) A8(I_BITOP3); if (V == 9) A8(I_ALIGN);
      if (V == 10) A8(I_BFI); if (V == 11) A8(I_MULLO); if (V == 12) A8(I_MULHI); if (V == 13) A8(I_MAD24); if (V == 14) A8(I_SDWA);
      if (V == 15) A8(I_XORSDWA); if (V == 16) A8(I_MOV); if (V == 17) A8(I_DPP); if (V == 18) A8(I_PERMV); if (V == 19) A8(I_BITOP3S);
      if (V == 20) A8(I_ALIGN3); if (V == 21) A8(I_XOR_E64); if (V == 22) A8(I_ADD3); if (V == 23) A8(I_LSHLADD);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int V>
void run(const char *name, uint32_t *d_out, int threads, int wgcu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t iters = 4000; float best = 1e30f;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(256 * wgcu), dim3(threads), 0, 0, iters, d_out, 0x0c0c0500u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double waves_per_simd = (double)threads / 64 * wgcu / 4;
  double clk = best * 1e-3 * 2.4e9;
  printf("%-26s waves/SIMD=%.0f: %7.3f ms  %5.2f cycles per wave-instr per SIMD (@2.4GHz)\n", name, waves_per_simd, best, clk / (waves_per_simd * iters * REP));
}
int main() {
  uint32_t *d_out; hipMalloc(&d_out, 256 * 4 * 1024 * 4);
  for (int cfg = 0; cfg < 2; cfg++) {
    int thr = cfg ? 1024 : 256, wg = 1;
    run<0>("v_xor_b32 (e32)", d_out, thr, wg); run<21>("v_xor_b32_e64", d_out, thr, wg); run<1>("v_and_b32", d_out, thr, wg); run<2>("v_add_u32", d_out, thr, wg);
    run<3>("v_lshrrev_b32", d_out, thr, wg); run<4>("v_bfe_u32", d_out, thr, wg); run<5>("v_lshl_or_b32", d_out, thr, wg); run<23>("v_lshl_add_u32", d_out, thr, wg);
    run<6>("v_and_or_b32 (sgpr mask)", d_out, thr, wg); run<7>("v_perm_b32 (sgpr sel)", d_out, thr, wg); run<18>("v_perm_b32 (vgpr sel)", d_out, thr, wg);
    run<8>("v_bitop3_b32 (3 vgpr)", d_out, thr, wg); run<19>("v_bitop3_b32 (2v+sgpr)", d_out, thr, wg); run<9>("v_alignbit (x,x,24)", d_out, thr, wg);
    run<20>("v_alignbit (x,y,24)", d_out, thr, wg); run<10>("v_bfi_b32 (sgpr mask)", d_out, thr, wg); run<22>("v_add3_u32", d_out, thr, wg);
    run<11>("v_mul_lo_u32", d_out, thr, wg); run<12>("v_mul_hi_u32", d_out, thr, wg); run<13>("v_mad_u32_u24", d_out, thr, wg);
    run<14>("v_lshlrev_b32_sdwa", d_out, thr, wg); run<15>("v_xor_b32_sdwa", d_out, thr, wg); run<16>("v_mov_b32", d_out, thr, wg); run<17>("v_mov_b32_dpp", d_out, thr, wg);
  }
  return 0;
}
