// mfma_i8_probe16: lane -> element maps of v_mfma_i32_16x16x64_i8 on gfx950, exact asymmetric integer data (dev tool).
// Assumed: lane l (c = l & 15, g = l >> 4) holds A[row c][k = 16 g + j] and B[k = 16 g + j][col c] in byte j = 0..15; D register
// q (0..3) of lane l is D[row 4 g + q][col c].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t *A, const int8_t *B, int *D) {  // A[16][64] (m,k), B[64][16] (k,n)
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  union { v4i v; int8_t b[16]; } a, b;
  for (int j = 0; j < 16; j++) { a.b[j] = A[c * 64 + 16 * g + j]; b.b[j] = B[(16 * g + j) * 16 + c]; }
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, acc, 0, 0, 0);
  for (int q = 0; q < 4; q++) D[(4 * g + q) * 16 + c] = acc[q];
}
int main() {
  int8_t A[1024], B[1024]; int D[256], R[256];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 64; j++) A[i * 64 + j] = (int8_t)((i * 7 + j * 3) % 251 - 125);
  for (int i = 0; i < 64; i++) for (int j = 0; j < 16; j++) B[i * 16 + j] = (int8_t)((i * 11 + j * 5 + 1) % 255 - 127);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += (int)A[i * 64 + k] * (int)B[k * 16 + j]; R[i * 16 + j] = s; }
  int8_t *dA, *dB; int *dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; i++) bad += D[i] != R[i];
  printf("v_mfma_i32_16x16x64_i8 with the assumed maps: %d of 256 elements wrong (%s)\n", bad, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
