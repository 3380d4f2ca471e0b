// ntt_lazy_ubench.hip -- would Harvey-style lazy butterflies pay in the polynomial step's LDS kernel?  (dev tool, round 6; csrc/poly.hip untouched)
//
// k_ntt_lds_mul8 (csrc/poly.hip) squares a 2048-point block in registers and LDS: 88 butterflies + 8 pointwise products per thread, each a Montgomery product (three
// quarter-rate 32-bit multiplies) with canonical add / sub around it (about ten full-rate instructions).  With primes below 2^30 the values can live in [0, 2p): the
// product needs no final subtraction, the difference no comparison.  This file holds the kernel's body twice -- canonical as in the product, and lazy -- over the same
// three 30-bit primes, checks that both give the same residues, and times them on the grid of a super-group's cyclic product (255 polynomials x 3 primes x 16 blocks).
//   hipcc -O3 --offload-arch=gfx950 tools/ntt_lazy_ubench.hip -o tools/ntt_lazy_ubench && tools/ntt_lazy_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

struct Prime { uint32_t p, ninv, r2; };
struct Primes3 { Prime q[3]; };

__host__ __device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t ninv) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * ninv;
  uint32_t u = (uint32_t)((t + (uint64_t)m * p) >> 32);
  return u >= p ? u - p : u;
}
__device__ __forceinline__ uint32_t add_mod(uint32_t a, uint32_t b, uint32_t p) { uint32_t s = a + b; return s >= p ? s - p : s; }
__device__ __forceinline__ uint32_t sub_mod(uint32_t a, uint32_t b, uint32_t p) { return a >= b ? a - b : a + p - b; }
// lazy forms: operands and results in [0, 2p), p < 2^30
__device__ __forceinline__ uint32_t lmont(uint32_t a, uint32_t b, uint32_t p, uint32_t ninv) {  // a b < 2^32 p  ->  result < 2p, no correction
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * ninv;
  return (uint32_t)((t + (uint64_t)m * p) >> 32);
}
__device__ __forceinline__ uint32_t ladd(uint32_t a, uint32_t b, uint32_t p2) { uint32_t s = a + b; return min(s, s - p2); }  // (s - 2p wraps when s < 2p)
__device__ __forceinline__ uint32_t lpad(uint32_t i) { return i + (i >> 5); }

template <int K, bool LAZY>
__device__ __forceinline__ void dif_regs(uint32_t *v, uint32_t j, uint32_t qd, uint32_t len, const uint32_t *t, const Prime q) {
  constexpr int R = 1 << K;
#pragma unroll
  for (int st = 0; st < K; st++) {
    const int h = R >> (st + 1);
    const uint32_t tstep = 2048 / (len >> st);
#pragma unroll
    for (int m = 0; m < R; m++)
      if ((m & h) == 0) {
        const uint32_t ti_ = (j + (uint32_t)(m & (h - 1)) * qd) * tstep, w = t[ti_ + (ti_ >> 5)];
        const uint32_t u = v[m], z = v[m + h];
        if (LAZY) {
          v[m] = ladd(u, z, 2 * q.p);
          v[m + h] = lmont(u - z + 2 * q.p, w, q.p, q.ninv);  // u - z + 2p < 4p < 2^32
        } else {
          v[m] = add_mod(u, z, q.p);
          v[m + h] = mont_mul(sub_mod(u, z, q.p), w, q.p, q.ninv);
        }
      }
  }
}
template <int K, bool LAZY>
__device__ __forceinline__ void dit_regs(uint32_t *v, uint32_t j, uint32_t qd, uint32_t len, const uint32_t *t, const Prime q) {
  constexpr int R = 1 << K;
#pragma unroll
  for (int st = 0; st < K; st++) {
    const int h = 1 << st;
    const uint32_t tstep = 2048 / (len << st);
#pragma unroll
    for (int m = 0; m < R; m++)
      if ((m & h) == 0) {
        const uint32_t ti_ = (j + (uint32_t)(m & (h - 1)) * qd) * tstep, w = t[ti_ + (ti_ >> 5)];
        if (LAZY) {
          const uint32_t u = v[m], z = lmont(v[m + h], w, q.p, q.ninv);
          v[m] = ladd(u, z, 2 * q.p);
          const uint32_t d = u - z + 2 * q.p;
          v[m + h] = min(d, d - 2 * q.p);
        } else {
          const uint32_t u = v[m], z = mont_mul(v[m + h], w, q.p, q.ninv);
          v[m] = add_mod(u, z, q.p);
          v[m + h] = sub_mod(u, z, q.p);
        }
      }
  }
}
// the body of k_ntt_lds_mul8 (squaring): forward low 11 stages, pointwise square, inverse low 11 stages of one 2048-point block
template <bool LAZY>
__global__ __launch_bounds__(256) void k_sq(uint32_t *__restrict__ a, uint32_t N, const uint32_t *__restrict__ tw, const uint32_t *__restrict__ twi, Primes3 P) {
  __shared__ uint32_t sm[2048 + 64];
  __shared__ uint32_t tws[2][1024 + 32];
  const Prime q = P.q[blockIdx.y % 3];
  const uint32_t tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.y * N + (size_t)blockIdx.x * 2048;
  {
    const uint32_t *tg = tw + (size_t)(blockIdx.y % 3) * 1024, *tig = twi + (size_t)(blockIdx.y % 3) * 1024;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      tws[0][lpad(tid + 256 * i)] = tg[tid + 256 * i];
      tws[1][lpad(tid + 256 * i)] = tig[tid + 256 * i];
    }
  }
  const uint32_t *t = tws[0], *ti = tws[1];
  uint32_t v[8];
  const uint32_t i1 = tid, j2 = tid & 31, i2 = (tid >> 5) * 256 + j2, j3 = tid >> 6, i3 = (tid & 63) * 32 + j3;
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = a[base + i1 + 256 * m];
  __syncthreads();
  dif_regs<3, LAZY>(v, tid, 256, 2048, t, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i1 + 256 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i2 + 32 * m)];
  dif_regs<3, LAZY>(v, j2, 32, 256, t, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i2 + 32 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i3 + 4 * m)];
  dif_regs<3, LAZY>(v, j3, 4, 32, t, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i3 + 4 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(8 * tid + m)];
  dif_regs<2, LAZY>(v, 0, 1, 4, t, q);
  dif_regs<2, LAZY>(v + 4, 0, 1, 4, t, q);
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = LAZY ? lmont(v[m], v[m], q.p, q.ninv) : mont_mul(v[m], v[m], q.p, q.ninv);  // (2p)^2 < 2^32 p
  dit_regs<2, LAZY>(v, 0, 1, 2, ti, q);
  dit_regs<2, LAZY>(v + 4, 0, 1, 2, ti, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(8 * tid + m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i3 + 4 * m)];
  dit_regs<3, LAZY>(v, j3, 4, 8, ti, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i3 + 4 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i2 + 32 * m)];
  dit_regs<3, LAZY>(v, j2, 32, 64, ti, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i2 + 32 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i1 + 256 * m)];
  dit_regs<3, LAZY>(v, tid, 256, 512, ti, q);
#pragma unroll
  for (int m = 0; m < 8; m++) a[base + i1 + 256 * m] = LAZY ? min(v[m], v[m] - q.p) : v[m];  // canonical on the way out (what the next kernel would take lazily)
}

static uint64_t powmod(uint64_t a, uint64_t e, uint64_t p) {
  uint64_t r = 1;
  a %= p;
  while (e) {
    if (e & 1) r = (unsigned __int128)r * a % p;
    a = (unsigned __int128)a * a % p;
    e >>= 1;
  }
  return r;
}

int main() {
  const uint32_t primes[3] = {998244353u, 754974721u, 1004535809u};  // 119 2^23 + 1, 45 2^24 + 1, 479 2^21 + 1: below 2^30, product 2^89.3
  const uint32_t N = 32768, NB = 255;
  Primes3 P;
  std::vector<uint32_t> tw(3 * 1024), twi(3 * 1024);
  for (int k = 0; k < 3; k++) {
    Prime &q = P.q[k];
    q.p = primes[k];
    uint32_t inv = 1;
    for (int it = 0; it < 6; it++) inv *= 2 - q.p * inv;
    q.ninv = 0u - inv;
    q.r2 = (uint32_t)(((unsigned __int128)1 << 64) % q.p);
    uint64_t g = 2;  // a generator of the 2048-th roots: g^((p-1)/2048) of order exactly 2048
    uint64_t w = 0;
    for (;; g++) {
      w = powmod(g, (q.p - 1) / 2048, q.p);
      if (powmod(w, 1024, q.p) == q.p - 1) break;
    }
    const uint64_t wi = powmod(w, q.p - 2, q.p);
    uint64_t a = 1, b = 1;
    for (int i = 0; i < 1024; i++) {
      tw[k * 1024 + i] = (uint32_t)(((unsigned __int128)a << 32) % q.p);
      twi[k * 1024 + i] = (uint32_t)(((unsigned __int128)b << 32) % q.p);
      a = a * w % q.p;
      b = b * wi % q.p;
    }
  }
  const size_t words = (size_t)NB * 3 * N;
  std::vector<uint32_t> h(words);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < words; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (uint32_t)(s % primes[(i / N) % 3]);
  }
  uint32_t *d_a, *d_b, *d_tw, *d_twi;
  HK(hipMalloc(&d_a, words * 4));
  HK(hipMalloc(&d_b, words * 4));
  HK(hipMalloc(&d_tw, tw.size() * 4));
  HK(hipMalloc(&d_twi, twi.size() * 4));
  HK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(d_twi, twi.data(), twi.size() * 4, hipMemcpyHostToDevice));
  const dim3 grid(N / 2048, 3 * NB);
  HK(hipMemcpy(d_a, h.data(), words * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(d_b, h.data(), words * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sq<false>, grid, dim3(256), 0, 0, d_a, N, d_tw, d_twi, P);
  hipLaunchKernelGGL(k_sq<true>, grid, dim3(256), 0, 0, d_b, N, d_tw, d_twi, P);
  HK(hipDeviceSynchronize());
  std::vector<uint32_t> ra(words), rb(words);
  HK(hipMemcpy(ra.data(), d_a, words * 4, hipMemcpyDeviceToHost));
  HK(hipMemcpy(rb.data(), d_b, words * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < words; i++) bad += ra[i] != rb[i];
  printf("canonical and lazy bodies agree on %zu of %zu residues\n", words - bad, words);
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0));
  HK(hipEventCreate(&e1));
  for (int rnd = 0; rnd < 3; rnd++)
    for (int lazy = 0; lazy < 2; lazy++) {
      HK(hipEventRecord(e0, 0));
      for (int it = 0; it < 20; it++) {
        if (lazy) hipLaunchKernelGGL(k_sq<true>, grid, dim3(256), 0, 0, d_b, N, d_tw, d_twi, P);
        else hipLaunchKernelGGL(k_sq<false>, grid, dim3(256), 0, 0, d_a, N, d_tw, d_twi, P);
      }
      HK(hipEventRecord(e1, 0));
      HK(hipEventSynchronize(e1));
      float ms = 0;
      HK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s butterflies: %7.1f us per launch (255 polynomials x 3 primes x 2^15 points)\n", lazy ? "lazy     " : "canonical", ms / 20 * 1e3);
    }
  return bad != 0;
}
