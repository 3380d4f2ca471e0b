// mfma_i8_probe: pins the lane -> element maps of v_mfma_i32_32x32x32_i8 on gfx950 with exact, asymmetric integer data (dev tool).
// Assumed: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 16h + j] and B[k = 16h + j][col r] in byte j = 0..15 of its 16-byte
// fragment; D register q (0..15) of lane l is D[row (q & 3) + 8 (q >> 2) + 4 h][col r].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t *A, const int8_t *B, int *D) {  // A[32][32] row-major (m,k), B[32][32] row-major (k,n)
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { v4i v; int8_t b[16]; } a, b;
  for (int j = 0; j < 16; j++) { a.b[j] = A[r * 32 + 16 * h + j]; b.b[j] = B[(16 * h + j) * 32 + r]; }
  v16i acc = {0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, acc, 0, 0, 0);
  for (int q = 0; q < 16; q++) D[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
}
int main() {
  int8_t A[1024], B[1024]; int D[1024], R[1024];
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { A[i * 32 + j] = (int8_t)((i * 7 + j * 3) % 251 - 125); B[i * 32 + j] = (int8_t)((i * 11 + j * 5 + 1) % 127); }
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += (int)A[i * 32 + k] * (int)B[k * 32 + j]; R[i * 32 + j] = s; }
  int8_t *dA, *dB; int *dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; i++) bad += D[i] != R[i];
  printf("v_mfma_i32_32x32x32_i8 with the assumed maps: %d of 1024 elements wrong (%s)\n", bad, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
