// poly.hip -- polynomial arithmetic over F_p, p = 2^32 - 5, on the GPU.
//
// The prover's only non-streaming step is  h = (v^2 - 1) / t  (reference src/snark.c:166-169, FLINT's
// nmod_poly_pow / nmod_poly_sub / nmod_poly_div).  p - 1 = 2 * 2147483645 has 2-adicity 1, so there is no NTT mod p;
// products are computed exactly over the integers with three NTT-friendly 30-bit primes + CRT (coefficients < 2^32,
// length <= 2^22: every convolution coefficient is < 2^86 < p1*p2*p3 ~ 2^89.3) and reduced mod p afterwards.  Because every
// result is the canonical representative in [0, p), it is bit-identical to FLINT's whatever algorithm FLINT picks.
// Division is a multiplication by the power-series inverse of rev(t), computed once per SSP by Newton iteration
// (mfh_poly_prepare_t): q = rev( rev(A)[:n] * rev(t)^-1 mod x^n ), n = deg A - deg t + 1.
//
// Round 6, the exact-division path of a BATCH of statements: a prover with a valid witness divides exactly (t | v^2 - 1, src/ssp.c:37-77), and an exact quotient
// of d coefficients is determined by the identity h t = v^2 - 1 taken modulo x^N - 1, N = the power of two >= d: h = (v^2 - 1 mod x^N - 1) * T' mod x^N - 1 with
// T' = t^-1 in F_p[x] / (x^N - 1) -- two CYCLIC products of length N instead of two linear ones of length 2N: half the transform work (2^15 points instead of 2^16 at the
// default size) and no reversals.  T' is computed once per SSP by the norm recursion a^-1 = a(-x) [a(x) a(-x)]^-1, the bracket being a polynomial in x^2, i.e. an
// element of the ring of half the length (log N levels down to a scalar).  Whether the division WAS exact is not assumed: the batch's results are checked on the device
// (h(r) t(r) = v(r)^2 - 1 at four points r: a wrong h of ANY origin survives with probability <= (2d / p)^4 < 2^-64), and when one statement fails the Euclidean path above
// recomputes the statements that failed -- its kernels are queued behind the check, sized for the whole batch, and each workgroup returns at once unless the list
// of failed statements reaches it: nothing waits for the host, and a batch with k such statements pays the Euclidean path for k.
#include <algorithm>

#include "ctx.hpp"

namespace {

constexpr uint32_t P32 = 0xfffffffbu;

// kernels of the Euclidean path take `need`: nullptr = every polynomial of the launch; else only the first *need ones -- the statements whose exact-division result
// failed the check, compacted: polynomial vs of the launch is statement need[2 + vs] of the batch (k_exact_check), which matters where a launch reads the
// batch's input or writes its output (`map` = need + 2 there); the transform and scratch buffers in between hold the compacted polynomials.  Nothing failed: every
// workgroup returns at once.
#define MF_NEEDED(need, vs) do { if ((need) && (vs) >= *(need)) return; } while (0)

struct NttPrime {
  uint32_t p, ninv, r2;  // modulus, -p^-1 mod 2^32, 2^64 mod p
};
struct Primes3 {
  NttPrime q[3];
};

__host__ __device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t ninv) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * ninv;
  uint32_t u = (uint32_t)((t + (uint64_t)m * p) >> 32);
  return u >= p ? u - p : u;
}
__host__ __device__ __forceinline__ uint32_t add_mod(uint32_t a, uint32_t b, uint32_t p) {
  uint32_t s = a + b;  // p < 2^31: no overflow
  return s >= p ? s - p : s;
}
__host__ __device__ __forceinline__ uint32_t sub_mod(uint32_t a, uint32_t b, uint32_t p) { return a >= b ? a - b : a + p - b; }
// Lazy forms for the stages a kernel runs in registers (round 6; the primes are below 2^30 for them): values in [0, 2p) between stages, so that the product needs no
// final subtraction and the difference no comparison -- the three quarter-rate multiplies of a butterfly stay, five of the ten instructions around them go
// (tools/ntt_lazy_ubench.hip: 15 - 20 % of k_ntt_lds_mul8).  Every kernel still reads and writes CANONICAL residues: lz_canon on the way out.
__device__ __forceinline__ uint32_t lz_mont(uint32_t a, uint32_t b, uint32_t p, uint32_t ninv) {  // a b < 2^32 p (a < 4p, b < p; or a, b < 2p)  ->  [0, 2p)
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * ninv;
  return (uint32_t)((t + (uint64_t)m * p) >> 32);
}
__device__ __forceinline__ uint32_t lz_add(uint32_t a, uint32_t b, uint32_t p2) {  // a, b in [0, 2p), p2 = 2p  ->  [0, 2p)   (s - 2p wraps when s < 2p)
  uint32_t s = a + b;
  return min(s, s - p2);
}
__device__ __forceinline__ uint32_t lz_canon(uint32_t a, uint32_t p) { return min(a, a - p); }  // [0, 2p) -> [0, p)
// one butterfly of a decimation-in-frequency / decimation-in-time stage on lazy values
__device__ __forceinline__ void lz_dif(uint32_t &x, uint32_t &y, uint32_t w, uint32_t p, uint32_t ninv) {
  const uint32_t u = x, z = y;
  x = lz_add(u, z, 2 * p);
  y = lz_mont(u - z + 2 * p, w, p, ninv);  // u - z + 2p < 4p < 2^32
}
__device__ __forceinline__ void lz_dit(uint32_t &x, uint32_t &y, uint32_t w, uint32_t p, uint32_t ninv) {
  const uint32_t u = x, z = lz_mont(y, w, p, ninv), d = u - z + 2 * p;
  x = lz_add(u, z, 2 * p);
  y = min(d, d - 2 * p);
}

// x mod (2^32 - 5) for x < 2^64:  2^32 = 5
__host__ __device__ __forceinline__ uint32_t red_p32(uint64_t x) {
  x = (x >> 32) * 5 + (uint32_t)x;  // < 5*2^32 + 2^32
  x = (x >> 32) * 5 + (uint32_t)x;  // < 30 + 2^32
  if (x >= P32) x -= P32;
  if (x >= P32) x -= P32;
  return (uint32_t)x;
}

uint64_t h_powmod(uint64_t a, uint64_t e, uint64_t p) {
  uint64_t r = 1;
  a %= p;
  while (e) {
    if (e & 1) r = (unsigned __int128)r * a % p;
    a = (unsigned __int128)a * a % p;
    e >>= 1;
  }
  return r;
}

// ---- kernels -------------------------------------------------------------------------------------------
// load `len` coefficients (mod p32) into [3][N] Montgomery residues, zero padded
// grid.y = 3 * batch everywhere below: blockIdx.y % 3 is the prime, blockIdx.y / 3 the polynomial of the batch; transform buffers are
// [batch][3][N], so blockIdx.y * N addresses them as before
__global__ void k_ntt_load(const uint32_t *__restrict__ in, uint32_t len, uint32_t N, Primes3 P, uint32_t *__restrict__ out, size_t in_stride,
                           const uint32_t *__restrict__ need = nullptr, const uint32_t *__restrict__ map = nullptr) {
  MF_NEEDED(need, blockIdx.y / 3);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const NttPrime q = P.q[blockIdx.y % 3];
  uint32_t x = i < len ? in[(size_t)(map ? map[blockIdx.y / 3] : blockIdx.y / 3) * in_stride + i] : 0u;
  out[(size_t)blockIdx.y * N + i] = mont_mul(x, q.r2, q.p, q.ninv);  // x * R mod p (x < 2^32, r2 < p: product < p * 2^32)
}

// decimation-in-frequency stage (natural -> bit-reversed order overall)
__global__ void k_ntt_dif(uint32_t *__restrict__ a, uint32_t N, uint32_t len, const uint32_t *__restrict__ tw, uint32_t tw_stride_n, Primes3 P) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N / 2) return;
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t half = len >> 1;
  const uint32_t j = i & (half - 1);
  const uint32_t s = (i - j) * 2;
  uint32_t *x = a + (size_t)blockIdx.y * N + s + j;
  const uint32_t w = tw[(size_t)(blockIdx.y % 3) * tw_stride_n + (size_t)j * (tw_stride_n * 2 / len)];
  uint32_t u = x[0], v = x[half];
  x[0] = add_mod(u, v, q.p);
  x[half] = mont_mul(sub_mod(u, v, q.p), w, q.p, q.ninv);
}
// decimation-in-time stage with inverse twiddles (bit-reversed -> natural)
__global__ void k_ntt_dit(uint32_t *__restrict__ a, uint32_t N, uint32_t len, const uint32_t *__restrict__ tw, uint32_t tw_stride_n, Primes3 P) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N / 2) return;
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t half = len >> 1;
  const uint32_t j = i & (half - 1);
  const uint32_t s = (i - j) * 2;
  uint32_t *x = a + (size_t)blockIdx.y * N + s + j;
  const uint32_t w = tw[(size_t)(blockIdx.y % 3) * tw_stride_n + (size_t)j * (tw_stride_n * 2 / len)];
  uint32_t u = x[0], v = mont_mul(x[half], w, q.p, q.ninv);
  x[0] = add_mod(u, v, q.p);
  x[half] = sub_mod(u, v, q.p);
}
// ---- fused stages ---------------------------------------------------------------------------------------------
// K consecutive DIF stages (block lengths len, len/2, ..., len>>(K-1)) in registers: a thread owns the 2^K elements
// {s + j + m*q}, q = len >> K.  grid = (N >> K) threads x 3 primes.
// in != nullptr (the first pass of a transform): the elements are taken from the coefficient array `in` (`in_len` coefficients mod p32 per
// polynomial, `in_stride` apart, zero padded) and converted to Montgomery residues on the way -- k_ntt_load's pass over the buffer saved.
template <int K>
__global__ __launch_bounds__(256) void k_ntt_dif_multi(uint32_t *__restrict__ a, uint32_t N, uint32_t len, const uint32_t *__restrict__ tw,
                                                       uint32_t half_max, Primes3 P, const uint32_t *__restrict__ in = nullptr, uint32_t in_len = 0,
                                                       size_t in_stride = 0, const uint32_t *__restrict__ need = nullptr,
                                                       const uint32_t *__restrict__ map = nullptr) {
  MF_NEEDED(need, blockIdx.y / 3);
  constexpr int R = 1 << K;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (N >> K)) return;
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t qd = len >> K;              // distance between a thread's elements
  const uint32_t j = i & (qd - 1);
  const uint32_t s = (i - j) << K;
  uint32_t *x = a + (size_t)blockIdx.y * N + s + j;
  const uint32_t *t = tw + (size_t)(blockIdx.y % 3) * half_max;
  uint32_t v[R];
  if (in) {
    const uint32_t *src = in + (size_t)(map ? map[blockIdx.y / 3] : blockIdx.y / 3) * in_stride;
#pragma unroll
    for (int m = 0; m < R; m++) {
      const uint32_t e = s + j + (uint32_t)m * qd;
      v[m] = e < in_len ? mont_mul(src[e], q.r2, q.p, q.ninv) : 0u;  // x * R mod p
    }
  } else {
#pragma unroll
    for (int m = 0; m < R; m++) v[m] = x[(size_t)m * qd];
  }
#pragma unroll
  for (int st = 0; st < K; st++) {
    const int h = R >> (st + 1);             // pair distance in register index
    const uint32_t L = len >> st;            // block length of this stage
    const uint32_t tstep = (half_max * 2) / L;
#pragma unroll
    for (int m = 0; m < R; m++) {
      if ((m & h) == 0) {
        const uint32_t pos = j + (uint32_t)(m & (h - 1)) * qd;  // position inside the half block
        lz_dif(v[m], v[m + h], t[(size_t)pos * tstep], q.p, q.ninv);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < R; m++) x[(size_t)m * qd] = lz_canon(v[m], q.p);
}
// K consecutive DIT stages with block lengths len, 2len, ..., len<<(K-1) (inverse twiddles)
template <int K>
__global__ __launch_bounds__(256) void k_ntt_dit_multi(uint32_t *__restrict__ a, uint32_t N, uint32_t len, const uint32_t *__restrict__ tw,
                                                       uint32_t half_max, Primes3 P, const uint32_t *__restrict__ need = nullptr) {
  MF_NEEDED(need, blockIdx.y / 3);
  constexpr int R = 1 << K;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (N >> K)) return;
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t qd = len >> 1;              // distance between a thread's elements = half of the first stage
  const uint32_t j = i & (qd - 1);
  const uint32_t s = (i - j) << K;
  uint32_t *x = a + (size_t)blockIdx.y * N + s + j;
  const uint32_t *t = tw + (size_t)(blockIdx.y % 3) * half_max;
  uint32_t v[R];
#pragma unroll
  for (int m = 0; m < R; m++) v[m] = x[(size_t)m * qd];
#pragma unroll
  for (int st = 0; st < K; st++) {
    const int h = 1 << st;
    const uint32_t L = len << st;
    const uint32_t tstep = (half_max * 2) / L;
#pragma unroll
    for (int m = 0; m < R; m++) {
      if ((m & h) == 0) {
        const uint32_t pos = j + (uint32_t)(m & (h - 1)) * qd;
        lz_dit(v[m], v[m + h], t[(size_t)pos * tstep], q.p, q.ninv);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < R; m++) x[(size_t)m * qd] = lz_canon(v[m], q.p);
}
// the lowest B stages (block lengths 2^B .. 2, or 2 .. 2^B for DIT) of every contiguous 2^B block, in LDS.  B <= 11.
template <bool DIT>
__global__ __launch_bounds__(256) void k_ntt_lds(uint32_t *__restrict__ a, uint32_t N, uint32_t B, const uint32_t *__restrict__ tw,
                                                 uint32_t half_max, Primes3 P) {
  __shared__ uint32_t sm[2048];
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t BL = 1u << B;
  uint32_t *x = a + (size_t)blockIdx.y * N + (size_t)blockIdx.x * BL;
  const uint32_t *t = tw + (size_t)(blockIdx.y % 3) * half_max;
  for (uint32_t i = threadIdx.x; i < BL; i += 256) sm[i] = x[i];
  __syncthreads();
  for (uint32_t st = 0; st < B; st++) {
    const uint32_t len = DIT ? (2u << st) : (BL >> st);
    const uint32_t half = len >> 1, tstep = (half_max * 2) / len;
    for (uint32_t i = threadIdx.x; i < BL / 2; i += 256) {
      const uint32_t j = i & (half - 1);
      const uint32_t p0 = ((i - j) << 1) + j;
      const uint32_t w = t[(size_t)j * tstep];
      const uint32_t u = sm[p0];
      if (DIT) {
        const uint32_t z = mont_mul(sm[p0 + half], w, q.p, q.ninv);
        sm[p0] = add_mod(u, z, q.p);
        sm[p0 + half] = sub_mod(u, z, q.p);
      } else {
        const uint32_t z = sm[p0 + half];
        sm[p0] = add_mod(u, z, q.p);
        sm[p0 + half] = mont_mul(sub_mod(u, z, q.p), w, q.p, q.ninv);
      }
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < BL; i += 256) x[i] = sm[i];
}

// forward low stages + pointwise product + inverse low stages of one 2^B block, all in LDS: a <- INTT_low( NTT_low(a) .* rhs ),
// rhs = NTT_low-transformed b (b == nullptr: a itself, i.e. squaring; b_is_hat: b is already fully transformed, e.g. the cached G^).
__global__ __launch_bounds__(256) void k_ntt_lds_mul(uint32_t *__restrict__ a, const uint32_t *__restrict__ b, int b_is_hat, uint32_t N, uint32_t B,
                                                     const uint32_t *__restrict__ tw, const uint32_t *__restrict__ twi, uint32_t half_max, Primes3 P,
                                                     const uint32_t *__restrict__ need = nullptr) {
  MF_NEEDED(need, blockIdx.y / 3);
  __shared__ uint32_t sa[2048];
  __shared__ uint32_t sb[2048];
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t BL = 1u << B;
  const size_t base = (size_t)blockIdx.y * N + (size_t)blockIdx.x * BL;
  const uint32_t *t = tw + (size_t)(blockIdx.y % 3) * half_max, *ti = twi + (size_t)(blockIdx.y % 3) * half_max;
  for (uint32_t i = threadIdx.x; i < BL; i += 256) {
    sa[i] = a[base + i];
    if (b) sb[i] = b_is_hat ? b[(size_t)(blockIdx.y % 3) * N + (size_t)blockIdx.x * BL + i] : b[base + i];  // a cached transform is shared by the batch
  }
  __syncthreads();
  const bool fwd_b = b && !b_is_hat;
  for (uint32_t st = 0; st < B; st++) {
    const uint32_t len = BL >> st, half = len >> 1, tstep = (half_max * 2) / len;
    for (uint32_t i = threadIdx.x; i < BL / 2; i += 256) {
      const uint32_t j = i & (half - 1), p0 = ((i - j) << 1) + j;
      const uint32_t w = t[(size_t)j * tstep];
      uint32_t u = sa[p0], z = sa[p0 + half];
      sa[p0] = add_mod(u, z, q.p);
      sa[p0 + half] = mont_mul(sub_mod(u, z, q.p), w, q.p, q.ninv);
      if (fwd_b) {
        u = sb[p0]; z = sb[p0 + half];
        sb[p0] = add_mod(u, z, q.p);
        sb[p0 + half] = mont_mul(sub_mod(u, z, q.p), w, q.p, q.ninv);
      }
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < BL; i += 256) sa[i] = mont_mul(sa[i], b ? sb[i] : sa[i], q.p, q.ninv);
  __syncthreads();
  for (uint32_t st = 0; st < B; st++) {
    const uint32_t len = 2u << st, half = len >> 1, tstep = (half_max * 2) / len;
    for (uint32_t i = threadIdx.x; i < BL / 2; i += 256) {
      const uint32_t j = i & (half - 1), p0 = ((i - j) << 1) + j;
      const uint32_t w = ti[(size_t)j * tstep];
      const uint32_t u = sa[p0], z = mont_mul(sa[p0 + half], w, q.p, q.ninv);
      sa[p0] = add_mod(u, z, q.p);
      sa[p0 + half] = sub_mod(u, z, q.p);
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < BL; i += 256) a[base + i] = sa[i];
}

// ---- the same for B = 11 (every transform of 2^11 points or more) with the stages in REGISTERS: a thread owns 8 of the block's 2048
// points; three radix-8 passes and one radix-4 pass per direction, the block crosses LDS only between passes (6 exchanges and barriers
// instead of 22 stage sweeps), the pointwise product and the first inverse pass stay in registers.  LDS index i lives at i + (i >> 5)
// (every pass's 64 lanes then spread over the 32 banks twice, the minimum).  rhs: nullptr = squaring, else a cached full transform
// (b_is_hat); two fresh operands keep the generic kernel above.
__device__ __forceinline__ uint32_t lpad(uint32_t i) { return i + (i >> 5); }
// K DIF stages (block lengths len, len/2, ...) on v[0 .. 2^K): element m sits at position j + m * qd of its block, qd = len >> K.  LAZY values: in [0, 2p) in and out
// PAD: the twiddle table lies in LDS at lpad() indices -- the stages read it with power-of-two strides (8, 16, ... entries between neighbouring lanes: 4 - 8 of
// the 64 banks, PMC round 3: SQ_LDS_BANK_CONFLICT was a quarter of k_ntt_lds_mul8's cycles); one skipped word per 32 spreads every such stride over the banks.
// (Measured, round 4: the chain of 255 statements takes the same 1.80 ms with and without the padding -- the kernel is bound by its Montgomery multiplies and
// its six barriers, not by these gathers; kept because it costs nothing.)
template <int K, bool PAD = false>
__device__ __forceinline__ void dif_regs(uint32_t *v, uint32_t j, uint32_t qd, uint32_t len, const uint32_t *__restrict__ t, uint32_t half_max,
                                         const NttPrime q) {
  constexpr int R = 1 << K;
#pragma unroll
  for (int st = 0; st < K; st++) {
    const int h = R >> (st + 1);
    const uint32_t tstep = (half_max * 2) / (len >> st);
#pragma unroll
    for (int m = 0; m < R; m++) {
      if ((m & h) == 0) {
        const uint32_t ti_ = (j + (uint32_t)(m & (h - 1)) * qd) * tstep;
        lz_dif(v[m], v[m + h], PAD ? t[ti_ + (ti_ >> 5)] : t[(size_t)ti_], q.p, q.ninv);
      }
    }
  }
}
// K DIT stages (block lengths len, 2 len, ...; inverse twiddles): element m at position j + m * qd, qd = len / 2
template <int K, bool PAD = false>
__device__ __forceinline__ void dit_regs(uint32_t *v, uint32_t j, uint32_t qd, uint32_t len, const uint32_t *__restrict__ t, uint32_t half_max,
                                         const NttPrime q) {
  constexpr int R = 1 << K;
#pragma unroll
  for (int st = 0; st < K; st++) {
    const int h = 1 << st;
    const uint32_t tstep = (half_max * 2) / (len << st);
#pragma unroll
    for (int m = 0; m < R; m++) {
      if ((m & h) == 0) {
        const uint32_t ti_ = (j + (uint32_t)(m & (h - 1)) * qd) * tstep;
        lz_dit(v[m], v[m + h], PAD ? t[ti_ + (ti_ >> 5)] : t[(size_t)ti_], q.p, q.ninv);
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_ntt_lds_mul8(uint32_t *__restrict__ a, const uint32_t *__restrict__ bhat, uint32_t N,
                                                      const uint32_t *__restrict__ tw, const uint32_t *__restrict__ twi, uint32_t half_max, Primes3 P,
                                                      const uint32_t *__restrict__ need = nullptr) {
  MF_NEEDED(need, blockIdx.y / 3);
  __shared__ uint32_t sm[2048 + 64];
  // the block's stages use every (half_max / 1024)-th entry of the twiddle tables: 2 x 1024 words, staged in LDS once per workgroup -- the
  // 88 twiddle reads per thread are then LDS gathers instead of global ones (64 different cache lines per wave-load)
  __shared__ uint32_t tws[2][1024 + 32];  // entry i at lpad(i)
  const NttPrime q = P.q[blockIdx.y % 3];
  const uint32_t tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.y * N + (size_t)blockIdx.x * 2048;
  {
    const uint32_t *tg = tw + (size_t)(blockIdx.y % 3) * half_max, *tig = twi + (size_t)(blockIdx.y % 3) * half_max;
    const uint32_t sc = half_max >> 10;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      tws[0][lpad(tid + 256 * i)] = tg[(size_t)(tid + 256 * i) * sc];
      tws[1][lpad(tid + 256 * i)] = tig[(size_t)(tid + 256 * i) * sc];
    }
  }
  const uint32_t *t = tws[0], *ti = tws[1];
  half_max = 1024;
  uint32_t v[8];
  // the three strided views of the block: element m of the thread at i1 + 256 m, i2 + 32 m, i3 + 4 m; and the contiguous one at 8 tid + m
  const uint32_t i1 = tid, j2 = tid & 31, i2 = (tid >> 5) * 256 + j2, j3 = tid >> 6, i3 = (tid & 63) * 32 + j3;
  // ---- forward: block lengths 2048 .. 2
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = a[base + i1 + 256 * m];
  __syncthreads();  // the staged twiddles
  dif_regs<3, true>(v, tid, 256, 2048, t, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i1 + 256 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i2 + 32 * m)];
  dif_regs<3, true>(v, j2, 32, 256, t, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i2 + 32 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i3 + 4 * m)];
  dif_regs<3, true>(v, j3, 4, 32, t, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i3 + 4 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(8 * tid + m)];
  dif_regs<2, true>(v, 0, 1, 4, t, half_max, q);
  dif_regs<2, true>(v + 4, 0, 1, 4, t, half_max, q);
  // ---- pointwise product with the right-hand side's transform (itself when squaring)
  if (bhat) {
    const uint4 *bp = reinterpret_cast<const uint4 *>(bhat + (size_t)(blockIdx.y % 3) * N + (size_t)blockIdx.x * 2048 + 8 * tid);
    const uint4 b0 = bp[0], b1 = bp[1];
    const uint32_t bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = lz_mont(v[m], bb[m], q.p, q.ninv);
  } else {
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = lz_mont(v[m], v[m], q.p, q.ninv);  // (2p)^2 < 2^32 p
  }
  // ---- inverse: block lengths 2 .. 2048
  dit_regs<2, true>(v, 0, 1, 2, ti, half_max, q);
  dit_regs<2, true>(v + 4, 0, 1, 2, ti, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(8 * tid + m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i3 + 4 * m)];
  dit_regs<3, true>(v, j3, 4, 8, ti, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i3 + 4 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i2 + 32 * m)];
  dit_regs<3, true>(v, j2, 32, 64, ti, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) sm[lpad(i2 + 32 * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = sm[lpad(i1 + 256 * m)];
  dit_regs<3, true>(v, tid, 256, 512, ti, half_max, q);
#pragma unroll
  for (int m = 0; m < 8; m++) a[base + i1 + 256 * m] = lz_canon(v[m], q.p);
}

__global__ void k_pointwise(uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t N, Primes3 P) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const NttPrime q = P.q[blockIdx.y % 3];
  size_t o = (size_t)blockIdx.y * N + i;
  a[o] = mont_mul(a[o], b[o], q.p, q.ninv);
}
struct Crt {
  uint32_t ninv_std[3];  // N^-1 mod p_i (standard form): mont_mul(xR, ninv_std) = x / N in standard form
  uint32_t inv_p1_p2, inv_p1_p3, inv_p2_p3;  // Montgomery form of p1^-1 mod p2, p1^-1 mod p3, p2^-1 mod p3
  uint32_t p1_mod, p1p2_mod;                 // p1 mod p32, p1*p2 mod p32
};
// the three residues of one coefficient (Montgomery form, as an unscaled inverse transform leaves them) -> the coefficient mod p32
__device__ __forceinline__ uint32_t crt_coeff(uint32_t a1, uint32_t a2, uint32_t a3, const Primes3 &P, const Crt &C) {
  const NttPrime q1 = P.q[0], q2 = P.q[1], q3 = P.q[2];
  uint32_t x1 = mont_mul(a1, C.ninv_std[0], q1.p, q1.ninv);
  uint32_t r2 = mont_mul(a2, C.ninv_std[1], q2.p, q2.ninv);
  uint32_t r3 = mont_mul(a3, C.ninv_std[2], q3.p, q3.ninv);
  // Garner: X = x1 + x2 p1 + x3 p1 p2
  uint32_t x1m2 = x1 >= q2.p ? x1 - q2.p : x1;  // x1 < p1 < 2 p2
  uint32_t x2 = mont_mul(sub_mod(r2, x1m2, q2.p), C.inv_p1_p2, q2.p, q2.ninv);
  uint32_t x1m3 = x1 >= q3.p ? x1 - q3.p : x1;
  uint32_t x2m3 = x2 >= q3.p ? x2 - q3.p : x2;
  uint32_t t3 = mont_mul(sub_mod(r3, x1m3, q3.p), C.inv_p1_p3, q3.p, q3.ninv);
  uint32_t x3 = mont_mul(sub_mod(t3, x2m3, q3.p), C.inv_p2_p3, q3.p, q3.ninv);
  uint64_t acc = (uint64_t)red_p32(x1) + red_p32((uint64_t)x2 * C.p1_mod) + red_p32((uint64_t)x3 * C.p1p2_mod);
  return red_p32(acc);
}
// residues (Montgomery, unscaled inverse transform) -> coefficient mod p32: out[i] = coefficient i, i < count -- or, rev_top >= 0, coefficient rev_top - i where that
// lies in [0, nsrc) and zero elsewhere (the reversals of the Euclidean path written by the kernel that produces the coefficients: rev(A)[:n], and the quotient
// turned back and padded); map: see MF_NEEDED
__global__ void k_crt(const uint32_t *__restrict__ r, uint32_t N, uint32_t count, Primes3 P, Crt C, uint32_t *__restrict__ out, size_t out_stride,
                      const uint32_t *__restrict__ need = nullptr, int64_t rev_top = -1, uint32_t nsrc = 0, const uint32_t *__restrict__ map = nullptr) {
  MF_NEEDED(need, blockIdx.y);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  r += (size_t)blockIdx.y * 3 * N;  // grid.y = batch
  out += (size_t)(map ? map[blockIdx.y] : blockIdx.y) * out_stride;
  if (rev_top < 0) {
    out[i] = crt_coeff(r[i], r[(size_t)N + i], r[(size_t)2 * N + i], P, C);
    return;
  }
  const int64_t e = rev_top - (int64_t)i;
  out[i] = e >= 0 && e < (int64_t)nsrc ? crt_coeff(r[e], r[(size_t)N + e], r[(size_t)2 * N + e], P, C) : 0u;
}
// ---- the exact-division path's seams, fused (2^12 <= N <= 2^16: one register pass of K = log N - 11 stages above the 2048-point blocks) ------------------------------
// Between the two cyclic products: the K top inverse stages of the square, the coefficient itself (crt_coeff), "- 1", back to residues, the K top forward stages of
// the second product -- k_ntt_dit_multi<K>, k_crt, k_sub_const0 and k_ntt_dif_multi<K> work on the SAME 2^K strided points {j + m N / 2^K} per thread, so one
// thread carries them through all four (three passes over the transform buffer saved).  LAST: the seam behind the second product -- top inverse stages, coefficient,
// out (the first `keep` of them): k_ntt_dit_multi<K> + k_crt.  grid = (N / 2^K / 256, batch); `a` = [batch][3][N], transformed in place.
template <int K, bool LAST>
__global__ __launch_bounds__(256) void k_exact_seam(uint32_t *__restrict__ a, uint32_t N, const uint32_t *__restrict__ tw, const uint32_t *__restrict__ twi,
                                                    uint32_t half_max, Primes3 P, Crt C, uint32_t *__restrict__ out, uint32_t keep, size_t out_stride) {
  constexpr int R = 1 << K;
  constexpr uint32_t qd = 2048;  // = N >> K: the launch is for N = 2^(11 + K) (one block of length N per polynomial: s = 0); a constant, so that the 3 R strided
                                 // addresses are immediates off one base instead of 3 R register pairs
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= qd || N != (qd << K)) return;
  uint32_t *x = a + (size_t)blockIdx.y * 3 * N + j;
  // prime by prime, Garner's digits as they become available (crt_coeff's arithmetic; x1 and x2 kept instead of the three residues).  K = 4 still takes 190 registers --
  // the 32 twiddles and 16 points of a prime in flight at once -- i.e. 2 waves per SIMD: 89 us per 255 polynomials against 93 for the three-residue form and 131 for the
  // four kernels it replaces; a 128-register build spills (444 bytes per lane) and was not kept
  const NttPrime q1 = P.q[0], q2 = P.q[1], q3 = P.q[2];
  uint32_t r[R], x1[R], x2[R], cf[R];
#pragma unroll
  for (int m = 0; m < R; m++) r[m] = x[(size_t)m * qd];
  dit_regs<K>(r, j, qd, 2 * qd, twi, half_max, q1);  // block lengths 2 qd .. N
#pragma unroll
  for (int m = 0; m < R; m++) x1[m] = mont_mul(r[m], C.ninv_std[0], q1.p, q1.ninv);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < R; m++) r[m] = x[(size_t)N + (size_t)m * qd];
  dit_regs<K>(r, j, qd, 2 * qd, twi + (size_t)half_max, half_max, q2);
#pragma unroll
  for (int m = 0; m < R; m++) {
    const uint32_t r2 = mont_mul(r[m], C.ninv_std[1], q2.p, q2.ninv), x1m2 = x1[m] >= q2.p ? x1[m] - q2.p : x1[m];  // x1 < p1 < 2 p2
    x2[m] = mont_mul(sub_mod(r2, x1m2, q2.p), C.inv_p1_p2, q2.p, q2.ninv);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < R; m++) r[m] = x[(size_t)2 * N + (size_t)m * qd];
  dit_regs<K>(r, j, qd, 2 * qd, twi + (size_t)2 * half_max, half_max, q3);
#pragma unroll
  for (int m = 0; m < R; m++) {
    const uint32_t r3 = mont_mul(r[m], C.ninv_std[2], q3.p, q3.ninv);
    const uint32_t x1m3 = x1[m] >= q3.p ? x1[m] - q3.p : x1[m], x2m3 = x2[m] >= q3.p ? x2[m] - q3.p : x2[m];
    const uint32_t t3 = mont_mul(sub_mod(r3, x1m3, q3.p), C.inv_p1_p3, q3.p, q3.ninv);
    const uint32_t x3 = mont_mul(sub_mod(t3, x2m3, q3.p), C.inv_p2_p3, q3.p, q3.ninv);
    cf[m] = red_p32((uint64_t)red_p32(x1[m]) + red_p32((uint64_t)x2[m] * C.p1_mod) + red_p32((uint64_t)x3 * C.p1p2_mod));
  }
  if (LAST) {
    out += (size_t)blockIdx.y * out_stride;
#pragma unroll
    for (int m = 0; m < R; m++)
      if (j + (uint32_t)m * qd < keep) out[j + (uint32_t)m * qd] = cf[m];
    return;
  }
  if (j == 0) cf[0] = cf[0] ? cf[0] - 1 : P32 - 1;  // v^2 - 1
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const NttPrime q = P.q[k];
#pragma unroll
    for (int m = 0; m < R; m++) r[m] = mont_mul(cf[m], q.r2, q.p, q.ninv);
    dif_regs<K>(r, j, qd, N, tw + (size_t)k * half_max, half_max, q);  // block lengths N .. 2 qd
#pragma unroll
    for (int m = 0; m < R; m++) x[(size_t)k * N + (size_t)m * qd] = lz_canon(r[m], q.p);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ void k_reverse(const uint32_t *__restrict__ in, int64_t top, uint32_t count, uint32_t *__restrict__ out, size_t in_stride = 0,
                          size_t out_stride = 0, const uint32_t *__restrict__ need = nullptr) {
  MF_NEEDED(need, blockIdx.y);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // out[i] = in[top - i] (0 if top - i < 0); grid.y = batch
  if (i >= count) return;
  in += (size_t)blockIdx.y * in_stride;
  out += (size_t)blockIdx.y * out_stride;
  int64_t s = top - (int64_t)i;
  out[i] = s >= 0 ? in[s] : 0u;
}
__global__ void k_two_minus(uint32_t *__restrict__ e, uint32_t count) {  // e <- 2 - e  (mod p32)
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t v = e[i];
  uint32_t neg = v ? P32 - v : 0u;
  e[i] = i == 0 ? red_p32((uint64_t)neg + 2) : neg;
}
__global__ void k_sub_const0(uint32_t *__restrict__ a, uint32_t c, size_t stride = 0, const uint32_t *__restrict__ need = nullptr) {  // a[0] -= c; grid.x = batch
  MF_NEEDED(need, blockIdx.x);
  if (threadIdx.x == 0) a[(size_t)blockIdx.x * stride] = red_p32((uint64_t)a[(size_t)blockIdx.x * stride] + P32 - c);
}
__global__ void k_add_vec(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t count, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = red_p32((uint64_t)a[i] + b[i]);
}
// out[i] = q[n-1-i] for i < min(n, d), zero above (quotient reversed back, first d coefficients)
__global__ void k_unreverse_pad(const uint32_t *__restrict__ qrev, uint32_t n, uint32_t d, uint32_t *__restrict__ out, size_t in_stride = 0,
                                size_t out_stride = 0, const uint32_t *__restrict__ need = nullptr, const uint32_t *__restrict__ map = nullptr) {
  MF_NEEDED(need, blockIdx.y);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  qrev += (size_t)blockIdx.y * in_stride;
  out += (size_t)(map ? map[blockIdx.y] : blockIdx.y) * out_stride;
  out[i] = i < n ? qrev[n - 1 - i] : 0u;
}

// ---- the ring F_p[x] / (x^n - 1), n a power of two (the exact-division path) --------------------------------------------------------
// out(x) = a(-x): an automorphism of the ring for even n
__global__ void k_conj(const uint32_t *__restrict__ a, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (i & 1) && a[i] ? P32 - a[i] : a[i];
}
// c: the 2n - 1 coefficients of a linear product of two ring elements.  out[j] = c[j] + c[j + n] (the product in the ring), j < n ...
__global__ void k_fold(const uint32_t *__restrict__ c, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = red_p32((uint64_t)c[j] + (j + n < 2 * n - 1 ? c[j + n] : 0u));
}
// ... and its even coefficients only, as an element of the ring of half the length in y = x^2 (the norm a(x) a(-x) has no odd ones): out[i] = c[2i] + c[2i + n], i < n / 2
__global__ void k_fold_even(const uint32_t *__restrict__ c, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n / 2) out[i] = red_p32((uint64_t)c[2 * i] + c[2 * i + n]);
}
// out(x) = b(x^2): n / 2 coefficients spread over n
__global__ void k_spread2(const uint32_t *__restrict__ b, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (i & 1) ? 0u : b[i >> 1];
}
// the check behind the exact-division path, one workgroup per statement: h(r) t(r) = v(r)^2 - 1 at the four points?  pw[j][i] = r_j^i, t_at[j] = t(r_j).  A statement that
// fails at one of them appends itself to the list the Euclidean kernels queued behind work through (need[0] = length, need[2 ..] = statements), counts in need[1] and
// leaves a mark in host memory (`seen`).  (The dot products are summed as two 32-bit halves per term -- 2^15 terms per lane at most -- and reduced mod p once per lane.)
__global__ __launch_bounds__(1024) void k_exact_check(const uint32_t *__restrict__ v, const uint32_t *__restrict__ h, uint32_t d, const uint32_t *__restrict__ pw,
                                                      uint32_t pw_stride, uint4 t_at, uint32_t *__restrict__ need, uint32_t *__restrict__ seen) {
  __shared__ uint32_t red[16][8];
  const uint32_t *vk = v + (size_t)blockIdx.x * d, *hk = h + (size_t)blockIdx.x * d;
  uint64_t lo[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // v at the four points, then h
  for (uint32_t i = threadIdx.x; i < d; i += 1024) {
    const uint32_t x = vk[i], y = hk[i];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t w = pw[(size_t)j * pw_stride + i];
      const uint64_t a = (uint64_t)x * w, b = (uint64_t)y * w;
      lo[j] += (uint32_t)a;
      hi[j] += a >> 32;
      lo[4 + j] += (uint32_t)b;
      hi[4 + j] += b >> 32;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint32_t a = red_p32((uint64_t)red_p32(hi[j]) * 5u + red_p32(lo[j]));  // 2^32 = 5 (mod p)
#pragma unroll
    for (int o = 32; o; o >>= 1) a = red_p32((uint64_t)a + __shfl_xor(a, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tj[4] = {t_at.x, t_at.y, t_at.z, t_at.w};
    uint64_t sum[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      sum[j] = 0;
      for (int w = 0; w < 16; w++) sum[j] = red_p32(sum[j] + red[w][j]);
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) bad |= red_p32(sum[4 + j] * tj[j]) != red_p32(sum[j] * sum[j] + (P32 - 1));
    if (bad) {
      need[2 + atomicAdd(need, 1u)] = blockIdx.x;
      atomicAdd(need + 1, 1u);
      *(volatile uint32_t *)seen = 1u;  // (host memory: mfh_poly_h_multi looks at it, some batch later, to decide whether the path pays for this caller)
    }
  }
}

inline dim3 g1(uint32_t n) { return dim3((n + 255) / 256); }

}  // namespace

struct PolyState {
  Primes3 P{};
  uint32_t root[3]{};  // generator of the 2^logmax subgroup is derived from these primitive roots
  uint32_t logmax = 0;
  uint32_t *d_tw = nullptr;   // [3][Nmax/2] forward twiddles (Montgomery)
  uint32_t *d_twi = nullptr;  // [3][Nmax/2] inverse twiddles
  uint32_t *d_bufA = nullptr, *d_bufB = nullptr;  // [3][Nmax] transform buffers
  uint32_t *d_tmp = nullptr;  // Nmax coefficients
  uint32_t *d_tmp2 = nullptr;
  uint32_t nb_cap = 1;  // polynomials the buffers above hold side by side (poly_batch_reserve)
  // per-SSP quotient precomputation
  bool have_t = false;
  uint32_t d = 0, dt = 0, n = 0, logN2 = 0;
  uint32_t *d_G = nullptr;     // n coefficients of rev(t)^-1
  uint32_t *d_Ghat = nullptr;  // [3][N2] forward transform of G
  uint32_t *d_f = nullptr;     // rev(t), dt+1 coefficients
  // ... and the exact-division path (when deg t = d - 1 and t is a unit modulo x^Nc - 1)
  bool cyc = false;
  uint32_t logNc = 0;            // Nc = 2^logNc >= d
  uint32_t *d_That = nullptr;    // [3][Nc] forward transform of t^-1 mod (x^Nc - 1)
  uint32_t *d_chk = nullptr;     // [4][Nc] powers of the four check points
  uint32_t chk_t[4]{};           // t at the check points
  uint32_t *d_need = nullptr;    // [0] statements of the batch that failed the check, [1] the same since the last reset, [2 ..] which ones (k_exact_check)
  uint32_t *h_seen = nullptr;    // pinned host word the check sets when a statement fails: read (never waited for) by later calls
  uint32_t rest = 0;             // batches left on the Euclidean path alone after such a mark (poly_exact == 1)
  ~PolyState() {
    for (uint32_t *p : {d_tw, d_twi, d_bufA, d_bufB, d_tmp, d_tmp2, d_G, d_Ghat, d_f, d_That, d_chk, d_need})
      if (p) hipFree(p);
    if (h_seen) hipHostFree(h_seen);
  }
};

namespace {

// below 2^30 (the lazy butterflies keep values in [0, 2p) and form u - z + 2p < 4p in 32 bits), 2-adicity 23, product 2^89.35; p1 < 2 p2, p1 < 2 p3, p2 < 2 p3 (k_crt)
const uint32_t kPrimes[3] = {998244353u, 897581057u, 880803841u};  // 119*2^23+1, 107*2^23+1, 105*2^23+1
const uint32_t kRoots[3] = {3u, 3u, 26u};                          // primitive roots (checked at init)
constexpr uint32_t kMaxBatch = 21845;  // polynomials side by side: 3 per polynomial on grid.y (<= 65535)

uint32_t to_mont(uint64_t x, const NttPrime &q) { return (uint32_t)(((unsigned __int128)(x % q.p) << 32) % q.p); }

int poly_init(mfh_ctx *c, uint32_t logmax) {
  if (c->poly && c->poly->logmax >= logmax) return MFH_OK;
  if (logmax > 23) {
    c->err = "polynomial too long for the NTT primes (max 2^23)";
    return MFH_EUNSUPPORTED;
  }
  delete c->poly;
  c->poly = nullptr;
  PolyState *S = new PolyState();
  S->logmax = logmax;
  const size_t Nmax = (size_t)1 << logmax;
  for (int k = 0; k < 3; k++) {
    NttPrime &q = S->P.q[k];
    q.p = kPrimes[k];
    uint32_t inv = 1;  // Newton for p^-1 mod 2^32
    for (int it = 0; it < 6; it++) inv *= 2 - q.p * inv;
    q.ninv = (uint32_t)(0u - inv);
    q.r2 = (uint32_t)(((unsigned __int128)1 << 64) % q.p);
    // primitive root check: g^((p-1)/f) != 1 for every prime factor f of p-1
    uint64_t pm1 = q.p - 1, rest = pm1;
    for (uint64_t f = 2; f * f <= rest; f++)
      if (rest % f == 0) {
        if (h_powmod(kRoots[k], pm1 / f, q.p) == 1) { c->err = "NTT root is not primitive"; delete S; return MFH_EINVAL; }
        while (rest % f == 0) rest /= f;
      }
    if (rest > 1 && h_powmod(kRoots[k], pm1 / rest, q.p) == 1) { c->err = "NTT root is not primitive"; delete S; return MFH_EINVAL; }
  }
  std::vector<uint32_t> tw(3 * Nmax / 2), twi(3 * Nmax / 2);
  for (int k = 0; k < 3; k++) {
    const NttPrime &q = S->P.q[k];
    uint64_t w = h_powmod(kRoots[k], (q.p - 1) >> logmax, q.p), wi = h_powmod(w, q.p - 2, q.p);
    uint64_t a = 1, b = 1;
    for (size_t i = 0; i < Nmax / 2; i++) {
      tw[k * (Nmax / 2) + i] = to_mont(a, q);
      twi[k * (Nmax / 2) + i] = to_mont(b, q);
      a = a * w % q.p;
      b = b * wi % q.p;
    }
  }
  bool ok = hipMalloc(&S->d_tw, tw.size() * 4) == hipSuccess && hipMalloc(&S->d_twi, twi.size() * 4) == hipSuccess &&
            hipMalloc(&S->d_bufA, 3 * Nmax * 4) == hipSuccess && hipMalloc(&S->d_bufB, 3 * Nmax * 4) == hipSuccess &&
            hipMalloc(&S->d_tmp, Nmax * 4) == hipSuccess && hipMalloc(&S->d_tmp2, Nmax * 4) == hipSuccess &&
            hipMemcpy(S->d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(S->d_twi, twi.data(), twi.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    delete S;
    c->err = "poly_init: device allocation failed";
    return MFH_ENOMEM;
  }
  c->poly = S;
  return MFH_OK;
}

// grow the transform / scratch buffers to hold nb polynomials side by side ([nb][3][Nmax] and [nb][Nmax])
int poly_batch_reserve(mfh_ctx *c, uint32_t nb) {
  PolyState *S = c->poly;
  if (nb <= S->nb_cap) return MFH_OK;
  const size_t Nmax = (size_t)1 << S->logmax;
  hipStreamSynchronize(c->stream);
  for (uint32_t **p : {&S->d_bufA, &S->d_bufB, &S->d_tmp, &S->d_tmp2}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  S->nb_cap = 1;
  bool ok = hipMalloc(&S->d_bufA, (size_t)nb * 3 * Nmax * 4) == hipSuccess && hipMalloc(&S->d_bufB, (size_t)nb * 3 * Nmax * 4) == hipSuccess &&
            hipMalloc(&S->d_tmp, (size_t)nb * Nmax * 4) == hipSuccess && hipMalloc(&S->d_tmp2, (size_t)nb * Nmax * 4) == hipSuccess;
  if (!ok) {
    c->err = "poly_batch_reserve: device allocation failed";
    return MFH_ENOMEM;
  }
  S->nb_cap = nb;
  return MFH_OK;
}

uint32_t ceil_log2(size_t x) {
  uint32_t l = 0;
  while (((size_t)1 << l) < x) l++;
  return l;
}

// the top (register) stages of a forward / inverse transform; the low B = min(logN, 11) stages run in LDS
// (up to five stages per pass: 32 elements per thread; 2^16 -> one pass of 5, 2^21 -> two.  `in`: the coefficients to load in the first
// pass instead of a separate k_ntt_load; returns false when there is no register pass to fuse the load into)
bool forward_top(mfh_ctx *c, uint32_t *buf, uint32_t logN, uint32_t nb = 1, const uint32_t *in = nullptr, uint32_t in_len = 0, size_t in_stride = 0,
                 const uint32_t *need = nullptr, const uint32_t *map = nullptr) {
  PolyState *S = c->poly;
  const uint32_t N = 1u << logN, half_max = 1u << (S->logmax - 1);
  uint32_t top = logN - std::min(logN, 11u), len = N;
  if (!top) return false;
  while (top) {
    const uint32_t k = top > 5 ? std::min(top - 3, 5u) : top;  // never leave a pass of fewer than three stages behind a full one
    dim3 g(((N >> k) + 255) / 256, 3 * nb);
    switch (k) {
      case 5: hipLaunchKernelGGL(k_ntt_dif_multi<5>, g, dim3(256), 0, c->stream, buf, N, len, S->d_tw, half_max, S->P, in, in_len, in_stride, need, map); break;
      case 4: hipLaunchKernelGGL(k_ntt_dif_multi<4>, g, dim3(256), 0, c->stream, buf, N, len, S->d_tw, half_max, S->P, in, in_len, in_stride, need, map); break;
      case 3: hipLaunchKernelGGL(k_ntt_dif_multi<3>, g, dim3(256), 0, c->stream, buf, N, len, S->d_tw, half_max, S->P, in, in_len, in_stride, need, map); break;
      case 2: hipLaunchKernelGGL(k_ntt_dif_multi<2>, g, dim3(256), 0, c->stream, buf, N, len, S->d_tw, half_max, S->P, in, in_len, in_stride, need, map); break;
      default: hipLaunchKernelGGL(k_ntt_dif_multi<1>, g, dim3(256), 0, c->stream, buf, N, len, S->d_tw, half_max, S->P, in, in_len, in_stride, need, map); break;
    }
    in = nullptr;
    map = nullptr;
    len >>= k;
    top -= k;
  }
  return true;
}
void inverse_top(mfh_ctx *c, uint32_t *buf, uint32_t logN, uint32_t nb = 1, const uint32_t *need = nullptr) {
  PolyState *S = c->poly;
  const uint32_t N = 1u << logN, half_max = 1u << (S->logmax - 1);
  const uint32_t B = std::min(logN, 11u);
  uint32_t top = logN - B, len = 2u << B;
  while (top) {
    const uint32_t k = top > 5 ? std::min(top - 3, 5u) : top;
    dim3 g(((N >> k) + 255) / 256, 3 * nb);
    switch (k) {
      case 5: hipLaunchKernelGGL(k_ntt_dit_multi<5>, g, dim3(256), 0, c->stream, buf, N, len, S->d_twi, half_max, S->P, need); break;
      case 4: hipLaunchKernelGGL(k_ntt_dit_multi<4>, g, dim3(256), 0, c->stream, buf, N, len, S->d_twi, half_max, S->P, need); break;
      case 3: hipLaunchKernelGGL(k_ntt_dit_multi<3>, g, dim3(256), 0, c->stream, buf, N, len, S->d_twi, half_max, S->P, need); break;
      case 2: hipLaunchKernelGGL(k_ntt_dit_multi<2>, g, dim3(256), 0, c->stream, buf, N, len, S->d_twi, half_max, S->P, need); break;
      default: hipLaunchKernelGGL(k_ntt_dit_multi<1>, g, dim3(256), 0, c->stream, buf, N, len, S->d_twi, half_max, S->P, need); break;
    }
    len <<= k;
    top -= k;
  }
}
void forward(mfh_ctx *c, uint32_t *buf, uint32_t logN) {  // complete forward transform (used for the cached G^)
  PolyState *S = c->poly;
  const uint32_t N = 1u << logN, half_max = 1u << (S->logmax - 1), B = std::min(logN, 11u);
  forward_top(c, buf, logN);
  hipLaunchKernelGGL(k_ntt_lds<false>, dim3(N >> B, 3), dim3(256), 0, c->stream, buf, N, B, S->d_tw, half_max, S->P);
}
Crt make_crt(const PolyState *S, uint32_t logN) {
  Crt C{};
  const NttPrime *q = S->P.q;
  for (int k = 0; k < 3; k++) C.ninv_std[k] = (uint32_t)h_powmod((uint64_t)1 << logN, q[k].p - 2, q[k].p);
  C.inv_p1_p2 = to_mont(h_powmod(q[0].p, q[1].p - 2, q[1].p), q[1]);
  C.inv_p1_p3 = to_mont(h_powmod(q[0].p, q[2].p - 2, q[2].p), q[2]);
  C.inv_p2_p3 = to_mont(h_powmod(q[1].p, q[2].p - 2, q[2].p), q[2]);
  C.p1_mod = (uint32_t)(q[0].p % P32);
  C.p1p2_mod = (uint32_t)(((uint64_t)q[0].p * q[1].p) % P32);
  return C;
}

// c[0..keep) = (a * b)[0..keep) mod p32.  bhat != null: use that precomputed forward transform (size 2^logN) instead of b.
// nb > 1: nb products side by side, polynomial k at a + k a_stride (and b + k a_stride), result at out + k out_stride.
// log_cyc != 0: the product in F_p[x] / (x^N - 1), N = 2^log_cyc >= la, lb (the transform of that length IS the cyclic product).  need, a_map (which polynomial at `a`
// product k reads): see MF_NEEDED.  rev_top, nsrc, out_map: the output written reversed / into the statements of the batch (k_crt).
int poly_mul(mfh_ctx *c, const uint32_t *a, uint32_t la, const uint32_t *b, uint32_t lb, const uint32_t *bhat, uint32_t logN_hat,
             uint32_t *out, uint32_t keep, uint32_t nb = 1, size_t a_stride = 0, size_t out_stride = 0, const uint32_t *need = nullptr, uint32_t log_cyc = 0,
             const uint32_t *a_map = nullptr, int64_t rev_top = -1, uint32_t nsrc = 0, const uint32_t *out_map = nullptr) {
  PolyState *S = c->poly;
  uint32_t logN = log_cyc ? log_cyc : bhat ? logN_hat : ceil_log2((size_t)la + lb - 1);
  if (logN < 1) logN = 1;
  if (logN > S->logmax || (log_cyc ? std::max(la, lb) : (size_t)la + lb - 1) > ((size_t)1 << logN) || (bhat && logN != logN_hat)) {
    c->err = "poly_mul: size exceeds the prepared NTT length";
    return MFH_EINVAL;
  }
  const uint32_t N = 1u << logN, half_max = 1u << (S->logmax - 1), B = std::min(logN, 11u);
  if (!forward_top(c, S->d_bufA, logN, nb, a, la, a_stride, need, a_map)) {  // (no register pass at this size: load on its own)
    hipLaunchKernelGGL(k_ntt_load, dim3((N + 255) / 256, 3 * nb), dim3(256), 0, c->stream, a, la, N, S->P, S->d_bufA, a_stride, need, a_map);
  }
  const uint32_t *rhs = bhat;  // already fully transformed
  int is_hat = 1;
  if (!bhat) {
    is_hat = 0;
    if (b == a && lb == la) {
      rhs = nullptr;  // square
    } else {
      if (!forward_top(c, S->d_bufB, logN, nb, b, lb, a_stride, need))
        hipLaunchKernelGGL(k_ntt_load, dim3((N + 255) / 256, 3 * nb), dim3(256), 0, c->stream, b, lb, N, S->P, S->d_bufB, a_stride, need, (const uint32_t *)nullptr);
      rhs = S->d_bufB;
    }
  }
  // low forward stages of both operands, pointwise product, low inverse stages: one kernel, the block never leaves LDS
  if (B == 11 && (rhs == nullptr || is_hat))
    hipLaunchKernelGGL(k_ntt_lds_mul8, dim3(N >> B, 3 * nb), dim3(256), 0, c->stream, S->d_bufA, rhs, N, S->d_tw, S->d_twi, half_max, S->P, need);
  else
    hipLaunchKernelGGL(k_ntt_lds_mul, dim3(N >> B, 3 * nb), dim3(256), 0, c->stream, S->d_bufA, rhs, is_hat, N, B, S->d_tw, S->d_twi, half_max, S->P, need);
  inverse_top(c, S->d_bufA, logN, nb, need);
  hipLaunchKernelGGL(k_crt, dim3((keep + 255) / 256, nb), dim3(256), 0, c->stream, S->d_bufA, N, keep, S->P, make_crt(S, logN), out, out_stride, need, rev_top, nsrc,
                     out_map);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

// T' = t^-1 in F_p[x] / (x^Nc - 1) by the norm recursion, its transform, and the check points: everything the exact-division path of mfh_poly_h_multi needs.
// Leaves S->cyc false (and the Euclidean path in charge) when deg t < d - 1 or t is not a unit of that ring.
int prepare_exact(mfh_ctx *c, const uint32_t *d_t, const std::vector<uint32_t> &t) {
  PolyState *S = c->poly;
  S->cyc = false;
  for (uint32_t **p : {&S->d_That, &S->d_chk})
    if (*p) { hipFree(*p); *p = nullptr; }
  const uint32_t d = S->d;
  if (S->dt != d - 1 || d < 2) return MFH_OK;
  const uint32_t logNc = ceil_log2(d), Nc = 1u << logNc;
  if (logNc + 1 > S->logmax) return MFH_OK;  // (the linear products below have 2 Nc - 1 coefficients)
  if (!S->h_seen) HIP_TRY(c, hipHostMalloc(&S->h_seen, 64, hipHostMallocDefault));
  if (!S->d_need) {
    HIP_TRY(c, hipMalloc(&S->d_need, (size_t)(2 + kMaxBatch) * 4));
    HIP_TRY(c, hipMemsetAsync(S->d_need, 0, 8, c->stream));
  }
  *S->h_seen = 0;
  S->rest = 0;
  // scratch: the ring elements a_0 = t, a_1, ... of lengths Nc, Nc / 2, ..., 1 (2 Nc words), a(-x), b(x^2) and two inverses (Nc each)
  uint32_t *scr = nullptr;
  HIP_TRY(c, hipMalloc(&scr, (size_t)6 * Nc * 4));
  struct Free { uint32_t *p; ~Free() { hipFree(p); } } guard{scr};
  uint32_t *lev = scr, *cj = scr + 2 * (size_t)Nc, *sp = cj + Nc, *inv[2] = {sp + Nc, sp + 2 * (size_t)Nc};
  HIP_TRY(c, hipMemsetAsync(lev, 0, (size_t)Nc * 4, c->stream));
  HIP_TRY(c, hipMemcpyAsync(lev, d_t, (size_t)d * 4, hipMemcpyDeviceToDevice, c->stream));
  std::vector<size_t> off{0};
  int rc;
  for (uint32_t n = Nc; n > 1; n >>= 1) {  // down: a_{k+1}(x^2) = a_k(x) a_k(-x)
    uint32_t *a = lev + off.back();
    hipLaunchKernelGGL(k_conj, g1(n), dim3(256), 0, c->stream, a, n, cj);
    if ((rc = poly_mul(c, a, n, cj, n, nullptr, 0, S->d_tmp, 2 * n - 1))) return rc;
    off.push_back(off.back() + n);
    hipLaunchKernelGGL(k_fold_even, g1(n / 2), dim3(256), 0, c->stream, S->d_tmp, n, lev + off.back());
  }
  uint32_t bottom = 0;  // the ring of length 1: a scalar
  HIP_TRY(c, hipMemcpyAsync(&bottom, lev + off.back(), 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!bottom) return MFH_OK;  // t shares a factor with x^Nc - 1 (probability about 2 / p for a random t)
  const uint32_t binv = (uint32_t)h_powmod(bottom, (uint64_t)P32 - 2, P32);
  HIP_TRY(c, hipMemcpyAsync(inv[0], &binv, 4, hipMemcpyHostToDevice, c->stream));
  int cur = 0;
  for (int k = (int)off.size() - 2; k >= 0; k--) {  // up: a_k^-1 = a_k(-x) a_{k+1}^-1(x^2)
    const uint32_t n = Nc >> k;
    hipLaunchKernelGGL(k_conj, g1(n), dim3(256), 0, c->stream, lev + off[k], n, cj);
    hipLaunchKernelGGL(k_spread2, g1(n), dim3(256), 0, c->stream, inv[cur], n, sp);
    if ((rc = poly_mul(c, cj, n, sp, n, nullptr, 0, S->d_tmp, 2 * n - 1))) return rc;
    hipLaunchKernelGGL(k_fold, g1(n), dim3(256), 0, c->stream, S->d_tmp, n, inv[cur ^ 1]);
    cur ^= 1;
  }
  // t T' = 1 in the ring?  (checked, not assumed: one more product and Nc words to the host, once per SSP)
  if ((rc = poly_mul(c, lev, Nc, inv[cur], Nc, nullptr, 0, S->d_tmp, 2 * Nc - 1))) return rc;
  hipLaunchKernelGGL(k_fold, g1(Nc), dim3(256), 0, c->stream, S->d_tmp, Nc, sp);
  std::vector<uint32_t> one(Nc);
  HIP_TRY(c, hipMemcpyAsync(one.data(), sp, (size_t)Nc * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (one[0] != 1 || std::any_of(one.begin() + 1, one.end(), [](uint32_t x) { return x != 0; })) return MFH_OK;
  HIP_TRY(c, hipMalloc(&S->d_That, (size_t)3 * Nc * 4));
  hipLaunchKernelGGL(k_ntt_load, dim3((Nc + 255) / 256, 3), dim3(256), 0, c->stream, inv[cur], Nc, Nc, S->P, S->d_That, (size_t)0, (const uint32_t *)nullptr,
                     (const uint32_t *)nullptr);
  forward(c, S->d_That, logNc);
  // the check points: fixed, odd, spread over F_p (splitmix64 of 1..4); their powers and t at them
  std::vector<uint32_t> pw((size_t)4 * Nc);
  for (int j = 0; j < 4; j++) {
    uint64_t z = 0x9e3779b97f4a7c15ull * (uint64_t)(j + 1);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    const uint64_t r = 2 + (z ^ (z >> 31)) % (P32 - 2);
    uint64_t x = 1, tv = 0;
    for (uint32_t i = 0; i < Nc; i++) {
      pw[(size_t)j * Nc + i] = (uint32_t)x;
      if (i < d) tv = (tv + x * t[i]) % P32;
      x = x * r % P32;
    }
    S->chk_t[j] = (uint32_t)tv;
  }
  HIP_TRY(c, hipMalloc(&S->d_chk, pw.size() * 4));
  HIP_TRY(c, hipMemcpyAsync(S->d_chk, pw.data(), pw.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  S->logNc = logNc;
  S->cyc = true;
  return MFH_OK;
}

}  // namespace

void mfh_poly_destroy(mfh_ctx *c) {
  delete c->poly;
  c->poly = nullptr;
}

extern "C" {

int mfh_poly_mul(mfh_ctx *c, const uint32_t *d_a, size_t la, const uint32_t *d_b, size_t lb, uint32_t *d_c) {
  if (!c || !d_a || !d_b || !d_c || !la || !lb) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = poly_init(c, std::max(ceil_log2(la + lb - 1), 1u));
  if (rc) return rc;
  return poly_mul(c, d_a, (uint32_t)la, d_b, (uint32_t)lb, nullptr, 0, d_c, (uint32_t)(la + lb - 1));
}

int mfh_poly_prepare_t(mfh_ctx *c, const uint32_t *d_t) {
  if (c) c->ssp_frag_src = nullptr;  // a (new) SSP is being prepared: derived images are stale
  if (!c || !d_t) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t d = c->P.d;
  std::vector<uint32_t> t(d);
  HIP_TRY(c, hipMemcpyAsync(t.data(), d_t, (size_t)d * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int64_t dt = -1;
  for (int64_t i = (int64_t)d - 1; i >= 0; i--)
    if (t[i]) { dt = i; break; }
  if (dt < 0) {
    c->err = "t(x) is the zero polynomial: (v^2-1)/t is undefined (nmod_poly_div would raise)";
    return MFH_EINVAL;
  }
  const uint32_t n = 2 * d - 1 - (uint32_t)dt;  // quotient length for the nominal degree 2d-2 numerator
  // products: v*v (2d-1 coeffs) and rev(A)[:n] * G[:n] (2n-1 coeffs); Newton steps never exceed 2n
  uint32_t logmax = std::max(ceil_log2((size_t)2 * d), ceil_log2((size_t)2 * n + dt + 2));
  int rc = poly_init(c, logmax);
  if (rc) return rc;
  PolyState *S = c->poly;
  for (uint32_t **p : {&S->d_G, &S->d_Ghat, &S->d_f})
    if (*p) { hipFree(*p); *p = nullptr; }
  S->have_t = false;
  S->d = d; S->dt = (uint32_t)dt; S->n = n;
  S->logN2 = std::max(ceil_log2((size_t)2 * n - 1), 1u);
  const uint32_t N2 = 1u << S->logN2;
  HIP_TRY(c, hipMalloc(&S->d_G, (size_t)std::max(n, 2u) * 4 * 2));
  HIP_TRY(c, hipMalloc(&S->d_Ghat, (size_t)3 * N2 * 4));
  HIP_TRY(c, hipMalloc(&S->d_f, (size_t)(dt + 1) * 4));
  hipLaunchKernelGGL(k_reverse, g1((uint32_t)dt + 1), dim3(256), 0, c->stream, d_t, dt, (uint32_t)dt + 1, S->d_f);  // f = rev(t)
  // Newton: g <- g (2 - f g) mod x^(2k)
  uint32_t g0 = (uint32_t)h_powmod(t[dt], (uint64_t)P32 - 2, P32);
  HIP_TRY(c, hipMemsetAsync(S->d_G, 0, (size_t)std::max(n, 2u) * 4 * 2, c->stream));
  HIP_TRY(c, hipMemcpyAsync(S->d_G, &g0, 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t k = 1; k < n;) {
    const uint32_t k2 = std::min(2 * k, n);
    const uint32_t lf = std::min(k2, (uint32_t)dt + 1);
    rc = poly_mul(c, S->d_f, lf, S->d_G, k, nullptr, 0, S->d_tmp, std::min(k2, lf + k - 1));  // e = f g mod x^k2
    if (rc) return rc;
    if (lf + k - 1 < k2) HIP_TRY(c, hipMemsetAsync(S->d_tmp + (lf + k - 1), 0, (size_t)(k2 - (lf + k - 1)) * 4, c->stream));
    hipLaunchKernelGGL(k_two_minus, g1(k2), dim3(256), 0, c->stream, S->d_tmp, k2);
    rc = poly_mul(c, S->d_G, k, S->d_tmp, k2, nullptr, 0, S->d_tmp2, k2);  // g (2 - e) mod x^k2
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(S->d_G, S->d_tmp2, (size_t)k2 * 4, hipMemcpyDeviceToDevice, c->stream));
    k = k2;
  }
  // forward transform of G at the size used per proof
  hipLaunchKernelGGL(k_ntt_load, dim3((N2 + 255) / 256, 3), dim3(256), 0, c->stream, S->d_G, n, N2, S->P, S->d_Ghat, (size_t)0, (const uint32_t *)nullptr,
                     (const uint32_t *)nullptr);
  forward(c, S->d_Ghat, S->logN2);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if ((rc = prepare_exact(c, d_t, t))) return rc;
  S->have_t = true;
  return MFH_OK;
}

// h_k = floor((v_k^2 - 1) / t) for nb polynomials side by side (v_k = d_v + k d, h_k = d_h + k d): the same launches as for one, nb times
// the work each
int mfh_poly_h_multi(mfh_ctx *c, const uint32_t *d_v, uint32_t *d_h, uint32_t nb) {
  if (!c || !d_v || !d_h || !nb || nb > kMaxBatch) return MFH_EINVAL;
  PolyState *S = c->poly;
  if (!S || !S->have_t) {
    c->err = "mfh_poly_prepare_t has not been called for this SSP";
    return MFH_EINVAL;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = poly_batch_reserve(c, nb);
  if (rc) return rc;
  const uint32_t d = S->d, n = S->n;
  const size_t Nmax = (size_t)1 << S->logmax;
  // the exact-division path (a batch: for one polynomial the launches below are latency, not work): two cyclic products of length Nc and the check
  const uint32_t *need = nullptr;
  bool exact = S->cyc && c->poly_exact && nb >= 4;
  if (exact && c->poly_exact == 1) {
    // a caller whose statements do NOT satisfy the SSP pays the cyclic products on top of the Euclidean division: after a batch in which the check has failed (seen
    // whenever the device gets there: this is a hint, nothing is waited for) the next 64 batches take the Euclidean path alone, then the exact path is tried again
    if (*(volatile uint32_t *)S->h_seen) {
      *(volatile uint32_t *)S->h_seen = 0;
      S->rest = 64;
    }
    if (S->rest) {
      S->rest--;
      exact = false;
    }
  }
  if (exact) {
    const uint32_t Nc = 1u << S->logNc;
    HIP_TRY(c, hipMemsetAsync(S->d_need, 0, 4, c->stream));
    if (S->logNc >= 12 && S->logNc <= 16) {  // five launches: the seams between and behind the two products are fused (k_exact_seam)
      const uint32_t half_max = 1u << (S->logmax - 1), K = S->logNc - 11;
      const Crt C = make_crt(S, S->logNc);
      const dim3 gs(((Nc >> K) + 255) / 256, nb), gl(Nc >> 11, 3 * nb);
      forward_top(c, S->d_bufA, S->logNc, nb, d_v, d, d);
      hipLaunchKernelGGL(k_ntt_lds_mul8, gl, dim3(256), 0, c->stream, S->d_bufA, (const uint32_t *)nullptr, Nc, S->d_tw, S->d_twi, half_max, S->P, (const uint32_t *)nullptr);
#define MF_SEAM(K_, LAST_, out_, keep_, stride_) \
  hipLaunchKernelGGL((k_exact_seam<K_, LAST_>), gs, dim3(256), 0, c->stream, S->d_bufA, Nc, S->d_tw, S->d_twi, half_max, S->P, C, out_, keep_, stride_)
#define MF_SEAM_K(LAST_, out_, keep_, stride_)                  \
  switch (K) {                                                  \
    case 1: MF_SEAM(1, LAST_, out_, keep_, stride_); break;     \
    case 2: MF_SEAM(2, LAST_, out_, keep_, stride_); break;     \
    case 3: MF_SEAM(3, LAST_, out_, keep_, stride_); break;     \
    case 4: MF_SEAM(4, LAST_, out_, keep_, stride_); break;     \
    default: MF_SEAM(5, LAST_, out_, keep_, stride_); break;    \
  }
      MF_SEAM_K(false, (uint32_t *)nullptr, 0u, (size_t)0)  // v^2 mod x^Nc - 1, minus 1, on its way into the second transform
      hipLaunchKernelGGL(k_ntt_lds_mul8, gl, dim3(256), 0, c->stream, S->d_bufA, (const uint32_t *)S->d_That, Nc, S->d_tw, S->d_twi, half_max, S->P, (const uint32_t *)nullptr);
      MF_SEAM_K(true, d_h, d, (size_t)d)  // times t^-1 in the ring: h when t | v^2 - 1
#undef MF_SEAM_K
#undef MF_SEAM
    } else {
      rc = poly_mul(c, d_v, d, d_v, d, nullptr, 0, S->d_tmp, Nc, nb, d, Nmax, nullptr, S->logNc);  // v^2 mod x^Nc - 1
      if (rc) return rc;
      hipLaunchKernelGGL(k_sub_const0, dim3(nb), dim3(64), 0, c->stream, S->d_tmp, 1u, Nmax, (const uint32_t *)nullptr);
      rc = poly_mul(c, S->d_tmp, Nc, nullptr, Nc, S->d_That, S->logNc, d_h, d, nb, Nmax, d, nullptr, S->logNc);  // times t^-1 in the ring: h when t | v^2 - 1
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_exact_check, dim3(nb), dim3(1024), 0, c->stream, d_v, d_h, d, S->d_chk, Nc, uint4{S->chk_t[0], S->chk_t[1], S->chk_t[2], S->chk_t[3]},
                       S->d_need, S->h_seen);
    need = S->d_need;  // the Euclidean kernels below work through the statements that failed the check: none, as a rule
  }
  const uint32_t *map = need ? need + 2 : nullptr;
  if (S->dt >= 1) {
    // rev(A)[:n], A = v^2 - 1 (2d - 1 coefficients, nominal degree 2d - 2): coefficients dt .. 2d - 2 of the square, written reversed by the kernel that produces them
    // (the "- 1" sits at coefficient 0, which the quotient does not see when deg t >= 1)
    rc = poly_mul(c, d_v, d, d_v, d, nullptr, 0, S->d_tmp2, n, nb, d, Nmax, need, 0, map, (int64_t)(2 * d - 2), 2 * d - 1);
    if (rc) return rc;
    // qrev = rev(A)[:n] * G mod x^n, turned back: h[i] = qrev[n - 1 - i] for i < min(n, d), zero above
    rc = poly_mul(c, S->d_tmp2, n, nullptr, n, S->d_Ghat, S->logN2, d_h, d, nb, Nmax, d, need, 0, nullptr, (int64_t)n - 1, n, map);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    return MFH_OK;
  }
  // deg t = 0 (division by a constant): the general sequence
  rc = poly_mul(c, d_v, d, d_v, d, nullptr, 0, S->d_tmp, 2 * d - 1, nb, d, Nmax, need, 0, map);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sub_const0, dim3(nb), dim3(64), 0, c->stream, S->d_tmp, 1u, Nmax, need);
  // rev(A)[:n]
  hipLaunchKernelGGL(k_reverse, dim3((n + 255) / 256, nb), dim3(256), 0, c->stream, S->d_tmp, (int64_t)(2 * d - 2), n, S->d_tmp2, Nmax, Nmax, need);
  // qrev = rev(A)[:n] * G mod x^n
  rc = poly_mul(c, S->d_tmp2, n, nullptr, n, S->d_Ghat, S->logN2, S->d_tmp, n, nb, Nmax, Nmax, need);
  if (rc) return rc;
  hipLaunchKernelGGL(k_unreverse_pad, dim3((d + 255) / 256, nb), dim3(256), 0, c->stream, S->d_tmp, n, d, d_h, Nmax, (size_t)d, need, map);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}
int mfh_set_poly_exact(mfh_ctx *c, int mode) {
  if (!c || mode < 0 || mode > 2) return MFH_EINVAL;
  c->poly_exact = mode;
  if (c->poly && c->poly->h_seen) {  // (what earlier batches have taught is forgotten)
    *c->poly->h_seen = 0;
    c->poly->rest = 0;
  }
  return MFH_OK;
}
// statements whose exact-division result failed the check (and were recomputed by Euclidean division) since the last call; waits for the stream.
// -1: no SSP prepared / no exact path for this t
long mfh_poly_exact_fallbacks(mfh_ctx *c) {
  if (!c || !c->poly || !c->poly->cyc) return -1;
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t w[2] = {0, 0};
  if (hipMemcpyAsync(w, c->poly->d_need, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
      hipMemsetAsync(c->poly->d_need + 1, 0, 4, c->stream) != hipSuccess)
    return -1;
  return (long)w[1];
}
int mfh_poly_h(mfh_ctx *c, const uint32_t *d_v, uint32_t *d_h) { return mfh_poly_h_multi(c, d_v, d_h, 1); }

int mfh_poly_add(mfh_ctx *c, const uint32_t *d_a, const uint32_t *d_b, size_t count, uint32_t *d_out) {
  if (!c || !d_a || !d_b || !d_out) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_add_vec, g1((uint32_t)count), dim3(256), 0, c->stream, d_a, d_b, (uint32_t)count, d_out);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

}  // extern "C"
