// snark.hip -- setup() and prover() of the reference (src/snark.c:57-190) as sequences of device launches.
//
// CRS device layout: ONE contiguous array of (2d + m) compressed ciphertexts (CT_BYTES each) in public-stream
// order -- s[0..d) | as[0..d) | t | v[0..m-1) -- which is exactly the order setup() encrypts them in and the order
// of the stream offsets CTR_S / CTR_AS / CTR_BT / CTR_BV (src/snark.h:8-12).  The reference keeps four mallocs
// (struct crs, src/snark.h:27-33); the host shim copies between the two.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr uint32_t P32 = 0xfffffffbu;

__device__ __forceinline__ uint32_t red_p32(uint64_t x) {
  x = (x >> 32) * 5 + (uint32_t)x;
  x = (x >> 32) * 5 + (uint32_t)x;
  if (x >= P32) x -= P32;
  if (x >= P32) x -= P32;
  return (uint32_t)x;
}
__device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b) { return red_p32((uint64_t)a * b); }

// coefficient k of SSP slot `slot` from either source (ssp_prg.hpp)
__device__ __forceinline__ uint32_t ssp_coef(const mf::SspSrc &src, uint32_t slot, uint32_t rowkey, uint32_t d, uint32_t k) {
  if (src.dense) return src.dense[(uint64_t)slot * d + k];
  return slot == 0 ? src.t[k] : mf::ssp_prg_coeff(rowkey, k);
}

// pw[k] = s^k mod p
__global__ void k_powers(uint32_t s, uint32_t d, uint32_t *__restrict__ pw) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  uint32_t r = 1, b = s, e = k;
  while (e) {
    if (e & 1) r = mulmod(r, b);
    b = mulmod(b, b);
    e >>= 1;
  }
  pw[k] = r;
}
// msg[i] = pw[i], msg[d+i] = alpha * pw[i]      (src/snark.c:73-91)
__global__ void k_msg_powers(const uint32_t *__restrict__ pw, uint32_t d, uint32_t alpha, uint32_t *__restrict__ msg) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  msg[k] = pw[k];
  msg[d + k] = mulmod(alpha, pw[k]);
}
// msg[2d + r] = beta * <slot(r), pw>, slot(0) = t (slot 0), slot(r) = v_r (slot r+1) for r = 1..m-1: Horner's value
// nmod_poly_evaluate_nmod(v_i, s) * beta (src/snark.c:97-98,105-106), computed as a dot product with the powers of s.
__global__ __launch_bounds__(256) void k_msg_evals(mf::SspSrc src, const uint32_t *__restrict__ pw, uint32_t d, uint32_t beta,
                                                   uint32_t *__restrict__ msg) {
  __shared__ uint64_t red[4];
  const uint32_t r = blockIdx.x;
  const uint32_t slot = r == 0 ? 0 : r + 1;
  uint64_t acc = 0;
  if (src.dense || slot == 0) {
    const uint32_t *row = src.dense ? src.dense + (uint64_t)slot * d : src.t;
    for (uint32_t k = threadIdx.x * 4; k < d; k += 256 * 4) {
      const uint4 v = *reinterpret_cast<const uint4 *>(row + k);
      const uint4 w = *reinterpret_cast<const uint4 *>(pw + k);
      uint64_t p0 = (uint64_t)v.x * w.x, p1 = (uint64_t)v.y * w.y, p2 = (uint64_t)v.z * w.z, p3 = (uint64_t)v.w * w.w;
      acc += (p0 >> 32) * 5 + (uint32_t)p0;  // each < 2^35: > 2^28 terms before overflow
      acc += (p1 >> 32) * 5 + (uint32_t)p1;
      acc += (p2 >> 32) * 5 + (uint32_t)p2;
      acc += (p3 >> 32) * 5 + (uint32_t)p3;
    }
  } else {
    const uint32_t rk = mf::ssp_prg_rowkey(src.seed, slot);
    for (uint32_t k = threadIdx.x; k < d; k += 256) {
      const uint64_t pr = (uint64_t)mf::ssp_prg_coeff(rk, k) * pw[k];
      acc += (pr >> 32) * 5 + (uint32_t)pr;
    }
  }
  acc = red_p32(acc);
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) msg[2 * d + r] = mulmod(red_p32(red[0] + red[1] + red[2] + red[3]), beta);
}

// out = a + (slot `slot` of the SSP) mod p
__global__ void k_add_slot(const uint32_t *__restrict__ a, mf::SspSrc src, uint32_t slot, uint32_t d, uint32_t *__restrict__ out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  out[k] = red_p32((uint64_t)a[k] + ssp_coef(src, slot, mf::ssp_prg_rowkey(src.seed, slot), d, k));
}

// the same for `gridDim.y` polynomials side by side (a, out strided by d)
__global__ void k_add_slot_multi(const uint32_t *__restrict__ a, mf::SspSrc src, uint32_t slot, uint32_t d, uint32_t *__restrict__ out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  const size_t o = (size_t)blockIdx.y * d + k;
  out[o] = red_p32((uint64_t)a[o] + ssp_coef(src, slot, mf::ssp_prg_rowkey(src.seed, slot), d, k));
}
// b_w of proof blockIdx.y += delta[blockIdx.y] * ct_t (ct_addmul_ui, src/lwe.c:141-149, for a batch: proofs are 5 ciphertexts apart)
__global__ void k_bw_add_delta_ct(uint64_t *__restrict__ proofs, const uint64_t *__restrict__ ct_t, const uint32_t *__restrict__ delta, uint32_t nvalues,
                                  uint32_t L, uint32_t KW) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvalues) return;
  const uint32_t x = delta[blockIdx.y];
  const uint32_t *aw = reinterpret_cast<const uint32_t *>(ct_t + (size_t)i * L);
  uint32_t *rw = reinterpret_cast<uint32_t *>(proofs + ((size_t)blockIdx.y * 5 + 4) * nvalues * L + (size_t)i * L);
  uint64_t carry = 0;
  for (uint32_t l = 0; l < KW; l++) {
    const uint64_t t = (uint64_t)aw[l] * x + carry + rw[l];
    rw[l] = (uint32_t)t;
    carry = t >> 32;
  }
}

// ciphertext values (K significant 64-bit limbs each) <-> NL = ceil(64 K / 56) uint64 "lanes" of 56 bits: lane j = bits [56 j, 56 j + 56) of the value.  Lanes
// of several partial ciphertexts can be added lane-wise (RCCL sum on uint64: 8 bits of headroom = 256 ranks) and the carries propagated once afterwards:
// sums mod 2^(64 K) do not depend on the order.  13 lanes per 704-bit value (27 at 1472) -- rounds 1-3 sent one 32-bit word per lane, 22 (46): the
// all-reduce per proof and the reduce-scatter per batch step carry 41 % fewer bytes.
__global__ void k_ct_to_lanes(const uint64_t *__restrict__ cts, uint64_t nvalues, uint32_t L, uint32_t K, uint32_t NL, uint64_t *__restrict__ lanes) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvalues * NL) return;
  const uint64_t v = i / NL;
  const uint32_t bit = 56u * (uint32_t)(i % NL), l = bit >> 6, sh = bit & 63;
  const uint64_t *x = cts + v * L;
  uint64_t r = x[l] >> sh;  // (l < K: 56 (NL - 1) < 64 K)
  if (sh > 8 && l + 1 < K) r |= x[l + 1] << (64 - sh);
  lanes[i] = r & ((1ull << 56) - 1);
}
__global__ void k_ct_from_lanes(const uint64_t *__restrict__ lanes, uint64_t nvalues, uint32_t L, uint32_t K, uint32_t NL, uint64_t *__restrict__ cts) {
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvalues) return;
  uint64_t *o = cts + v * L;
  // acc = (sum of the lanes added so far) >> 64 `done`: after lane j every bit below 56 (j + 1) is final, so whole limbs below that are emitted
  // at once; the shift 56 j - 64 done stays below 64, so a 64-bit lane sum never leaves the 128-bit accumulator
  unsigned __int128 acc = 0;
  uint32_t done = 0;
  for (uint32_t j = 0; j < NL; j++) {
    acc += (unsigned __int128)lanes[v * NL + j] << (56 * j - 64 * done);
    while (done < K && 56 * (j + 1) >= 64 * (done + 1)) {
      o[done++] = (uint64_t)acc;
      acc >>= 64;
    }
  }
  if (done < K) o[done++] = (uint64_t)acc;  // (the last lane is partial: 64 K is no multiple of 56)
  for (uint32_t l = K; l < L; l++) o[l] = 0;  // modq: what exceeds 2^(64 K) is dropped
}

// scal[r] = <slot r, pw> mod p for r = 0 (t) and 1 (v_0): nmod_poly_evaluate_nmod of src/snark.c:201,213
__global__ __launch_bounds__(256) void k_eval_slots01(mf::SspSrc src, const uint32_t *__restrict__ pw, uint32_t d, uint32_t *__restrict__ scal) {
  __shared__ uint64_t red[4];
  const uint32_t slot = blockIdx.x;
  const uint32_t rk = mf::ssp_prg_rowkey(src.seed, slot);
  uint64_t acc = 0;
  for (uint32_t k = threadIdx.x; k < d; k += 256) {
    uint64_t pr = (uint64_t)ssp_coef(src, slot, rk, d, k) * pw[k];
    acc += (pr >> 32) * 5 + (uint32_t)pr;
  }
  acc = red_p32(acc);
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) scal[blockIdx.x] = red_p32(red[0] + red[1] + red[2] + red[3]);
}
// the four checks of verifier() (src/snark.c:219-235) on decrypted values dec[5*i .. 5*i+5) = h, hat_h, hat_v, v_w, b_w
__global__ void k_verify(const uint32_t *__restrict__ dec, const uint32_t *__restrict__ scal, uint32_t alpha, uint32_t beta, uint32_t count,
                         uint8_t *__restrict__ ok) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t h_s = dec[5 * i], hath_s = dec[5 * i + 1], hatv_s = dec[5 * i + 2], w_s = dec[5 * i + 3], b_s = dec[5 * i + 4];
  const uint32_t t_s = scal[0];
  const uint32_t v_s = red_p32((uint64_t)scal[1] + w_s);
  bool good = mulmod(h_s, alpha) == hath_s;                                         // eq-pke
  good = good && mulmod(v_s, alpha) == hatv_s;
  good = good && red_p32((uint64_t)mulmod(v_s, v_s) + P32 - 1) == mulmod(h_s, t_s);  // eq-div
  good = good && mulmod(w_s, beta) == b_s;                                          // eq-lin
  ok[i] = good ? 1 : 0;  // the reference's "test-error" bound (src/snark.c:238-241) can never reject
}

inline dim3 g1(uint32_t n) { return dim3((n + 255) / 256); }

}  // namespace

int aux_reserve(mfh_ctx *c, size_t bytes) {
  if (bytes <= c->aux_bytes) return MFH_OK;
  if (c->aux) {
    hipStreamSynchronize(c->stream);
    hipFree(c->aux);
    c->aux = nullptr;
    c->aux_bytes = 0;
  }
  if (hipMalloc(&c->aux, bytes) != hipSuccess) {
    c->err = "hipMalloc(aux) failed";
    return MFH_ENOMEM;
  }
  c->aux_bytes = bytes;
  return MFH_OK;
}

extern "C" {

int mfh_ssp_prepare(mfh_ctx *c, const uint32_t *d_ssp) {
  if (!c) return MFH_EINVAL;
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  return mfh_poly_prepare_t(c, src.t);  // slot 0 = t
}

int mfh_setup_messages(mfh_ctx *c, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, uint32_t *d_msg) {
  if (!c || !d_msg) return MFH_EINVAL;
  if (alpha >= P32 || beta >= P32 || s >= P32) { c->err = "alpha, beta, s must be < p"; return MFH_EINVAL; }
  const uint32_t d = c->P.d, m = c->P.m;
  if (d % 4) { c->err = "d must be a multiple of 4"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  rc = aux_reserve(c, (size_t)d * 4);
  if (rc) return rc;
  uint32_t *pw = (uint32_t *)c->aux;
  hipLaunchKernelGGL(k_powers, g1(d), dim3(256), 0, c->stream, s, d, pw);
  hipLaunchKernelGGL(k_msg_powers, g1(d), dim3(256), 0, c->stream, pw, d, alpha, d_msg);
  hipLaunchKernelGGL(k_msg_evals, dim3(m), dim3(256), 0, c->stream, src, pw, d, beta, d_msg);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_setup(mfh_ctx *c, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, const uint64_t *d_sk, const uint64_t *d_err,
              uint8_t *d_crs_c8) {
  if (!c || !d_sk || !d_err || !d_crs_c8) return MFH_EINVAL;
  const size_t rows = (size_t)2 * c->P.d + c->P.m;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->d_msg || c->msg_rows < rows) {
    if (c->d_msg) { hipStreamSynchronize(c->stream); hipFree(c->d_msg); c->d_msg = nullptr; }
    HIP_TRY(c, hipMalloc(&c->d_msg, rows * 4));
    c->msg_rows = rows;
  }
  int rc = mfh_setup_messages(c, d_ssp, alpha, beta, s, c->d_msg);
  if (rc) return rc;
  // all 2d+m encryptions are consecutive rows of the stream starting at CTR_S = 0 (src/snark.c:75-110)
  return mfh_encrypt_rows(c, 0, rows, d_sk, c->d_msg, d_err, d_crs_c8);
}

int mfh_setup_image(mfh_ctx *c, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, const uint64_t *d_sk, const uint64_t *d_err,
                    uint8_t *d_crs_c8, void *d_rows_image) {
  int rc = mfh_setup(c, d_ssp, alpha, beta, s, d_sk, d_err, d_crs_c8);
  if (rc || !d_rows_image) return rc;
  return mfh_crs_expand(c, 0, (size_t)2 * c->P.d + c->P.m, d_crs_c8, d_rows_image);  // (same stream: behind the encryptions that write d_crs_c8)
}

int mfh_verify(mfh_ctx *c, const uint32_t *d_ssp, uint32_t alpha, uint32_t beta, uint32_t s, const uint64_t *d_sk, const uint64_t *d_proofs,
               size_t count, uint8_t *d_ok) {
  if (!c || !d_sk || (count && (!d_proofs || !d_ok))) return MFH_EINVAL;
  if (alpha >= P32 || beta >= P32 || s >= P32) { c->err = "alpha, beta, s must be < p"; return MFH_EINVAL; }
  if (!count) return MFH_OK;
  const uint32_t d = c->P.d;
  HIP_TRY(c, hipSetDevice(c->device));
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  rc = aux_reserve(c, (size_t)d * 4 + 16 + count * 5 * 4);
  if (rc) return rc;
  uint32_t *pw = (uint32_t *)c->aux, *scal = pw + d, *dec = scal + 4;
  hipLaunchKernelGGL(k_powers, g1(d), dim3(256), 0, c->stream, s, d, pw);
  hipLaunchKernelGGL(k_eval_slots01, dim3(2), dim3(256), 0, c->stream, src, pw, d, scal);
  HIP_TRY(c, hipGetLastError());
  rc = mfh_decrypt(c, d_sk, d_proofs, 5 * count, dec);
  if (rc) return rc;
  hipLaunchKernelGGL(k_verify, g1((uint32_t)count), dim3(256), 0, c->stream, dec, scal, alpha, beta, (uint32_t)count, d_ok);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

uint32_t mfh_lanes_per_value(const mfh_ctx *c) { return c ? (64 * (c->P.logq / 64) + 55) / 56 : 0; }
int mfh_ct_to_lanes(mfh_ctx *c, const uint64_t *d_cts, size_t count, uint64_t *d_lanes) {
  if (!c || !d_cts || !d_lanes) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t L = (c->P.logq + 63) / 64, K = c->P.logq / 64, NL = mfh_lanes_per_value(c);
  const uint64_t nvalues = (uint64_t)count * (c->P.n + 1);
  if (!nvalues) return MFH_OK;
  hipLaunchKernelGGL(k_ct_to_lanes, dim3((uint32_t)((nvalues * NL + 255) / 256)), dim3(256), 0, c->stream, d_cts, nvalues, L, K, NL, d_lanes);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_ct_from_lanes(mfh_ctx *c, const uint64_t *d_lanes, size_t count, uint64_t *d_cts) {
  if (!c || !d_cts || !d_lanes) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t L = (c->P.logq + 63) / 64, K = c->P.logq / 64, NL = mfh_lanes_per_value(c);
  const uint64_t nvalues = (uint64_t)count * (c->P.n + 1);
  if (!nvalues) return MFH_OK;
  hipLaunchKernelGGL(k_ct_from_lanes, dim3((uint32_t)((nvalues + 255) / 256)), dim3(256), 0, c->stream, d_lanes, nvalues, L, K, NL, d_cts);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

static int prove_partial_impl(mfh_ctx *c, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
                              uint32_t rank, uint32_t world, const uint64_t *d_wlanes, uint64_t *d_partial) {
  if (!c || !d_crs_c8 || !h_witness_bits || !d_partial || world == 0 || rank >= world) return MFH_EINVAL;
  if (delta >= P32) { c->err = "delta must be < p"; return MFH_EINVAL; }
  mf::SspSrc src;
  {
    int rc0 = ssp_src(c, d_ssp, src);
    if (rc0) return rc0;
  }
  const uint32_t d = c->P.d, m = c->P.m, n = c->P.n;
  const uint32_t L = (c->P.logq + 63) / 64, ctb = c->P.logq / 8;
  const size_t ctl = (size_t)(n + 1) * L;
  HIP_TRY(c, hipSetDevice(c->device));
  // prover scratch: w, v, h (d each), cw (m)
  if (!c->d_prover || c->prover_words < (size_t)3 * d + m) {
    if (c->d_prover) { hipStreamSynchronize(c->stream); hipFree(c->d_prover); c->d_prover = nullptr; }
    HIP_TRY(c, hipMalloc(&c->d_prover, ((size_t)3 * d + m) * 4));
    c->prover_words = (size_t)3 * d + m;
  }
  uint32_t *w = c->d_prover, *v = w + d, *h = v + d, *cw = h + d;
  uint64_t *pi_h = d_partial, *pi_hat_h = d_partial + ctl, *pi_hat_v = d_partial + 2 * ctl, *pi_v_w = d_partial + 3 * ctl,
           *pi_b_w = d_partial + 4 * ctl;
  // this rank's contiguous share of each region's rows
  auto share = [&](uint32_t rows, uint32_t &lo, uint32_t &cnt) {
    lo = (uint32_t)((uint64_t)rows * rank / world);
    cnt = (uint32_t)((uint64_t)rows * (rank + 1) / world) - lo;
  };

  // The witness pass (HBM-bound over the SSP) and the polynomial step (a chain of short launches) do not depend on b_w, and b_w's
  // rows need neither: the two run side by side, chain on the side stream, b_w's rows on the caller's stream, joined before the
  // S / AS regions (which need w, v and h).  The side stream waits for everything already queued on the caller's stream first, so
  // the previous proof's reads of w, v, h are over before they are rewritten.
  // Only for the single-GPU prover: with the witness lanes coming out of a collective the chain is short and the gain is within noise
  // (tools/rank_cost.py), and two processes sharing one GPU (the one-GPU rehearsal of the multi-rank path) serialise badly on
  // cross-queue event waits.
  const bool fork = c->overlap && world == 1;
  hipStream_t main_stream = c->stream;
  if (fork) {
    if (!c->side) {
      HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
      HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    HIP_TRY(c, hipEventRecord(c->ev_fork, main_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
  }
  struct StreamSwap {  // every helper launches on c->stream
    mfh_ctx *c;
    hipStream_t keep;
    StreamSwap(mfh_ctx *c_, hipStream_t s) : c(c_), keep(c_->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = keep; }
  };
  int rc;
  // b_w = delta * ct_t + sum_{bit} ct_{v_i}: rows BT, BV.. are m consecutive stream rows (src/snark.c:143-155)
  uint32_t *h_cw = (uint32_t *)pin_acquire(c, c->pin_cw, (size_t)m * 4);
  if (!h_cw) return MFH_ENOMEM;
  h_cw[0] = delta;
  for (uint32_t i = 1; i < m; i++) h_cw[i] = (h_witness_bits[(i - 1) >> 3] >> ((i - 1) & 7)) & 1;
  HIP_TRY(c, hipMemcpyAsync(cw, h_cw, (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
  pin_release(c, c->pin_cw);
  const uint64_t ctr_ct = (uint64_t)ctb * n;
  uint32_t lo, cnt;
  share(m, lo, cnt);
  const void *res = c->resident_rows;
  if (res && c->resident_sharded && (c->res_rank != rank || c->res_world != world)) {
    c->err = "the resident CRS image holds the shares of a different (rank, world)";
    return MFH_EINVAL;
  }
  // One region = `cnt` consecutive CRS rows starting at absolute stream row `abs_row` (compressed bytes at d_crs_c8 + abs_row*ctb).
  // Rows that are resident (whole image, a prefix of it, or this rank's share at image row `img_row`) are streamed from HBM, the
  // rest is regenerated from the seed and accumulated on top.
  auto eval_region = [&](size_t abs_row, size_t img_row, uint32_t cnt, const uint32_t *c0, const uint32_t *c1, uint64_t *r0, uint64_t *r1) -> int {
    size_t nres = 0;
    if (res) nres = c->resident_sharded ? cnt : (abs_row >= c->resident_nrows ? 0 : std::min<uint64_t>(cnt, c->resident_nrows - abs_row));
    int rc2 = MFH_OK;
    if (nres) rc2 = mfh_eval_rows_resident(c, res, c->resident_sharded ? img_row : abs_row, nres, c0, c1, r0, r1, 0);
    if (rc2 || nres == cnt) return rc2;
    return mfh_eval_rows(c, ctr_ct * (abs_row + nres), cnt - nres, d_crs_c8 + (abs_row + nres) * ctb, c0 + nres, c1 ? c1 + nres : nullptr, r0, r1,
                         nres ? 1 : 0);
  };
  uint32_t loS, cS;
  share(d, loS, cS);
  auto run_bw = [&]() -> int { return eval_region((size_t)2 * d + lo, (size_t)2 * cS, cnt, cw + lo, nullptr, pi_b_w, nullptr); };
  auto run_chain = [&]() -> int {
    StreamSwap on_side(c, fork ? c->side : main_stream);
    // w(x) = delta t + sum_{bit} v_i   (src/snark.c:141,147-155); every rank needs all of w for the polynomial step
    int r = d_wlanes ? mfh_witness_from_lanes(c, d_ssp, d_wlanes, delta, w) : mfh_witness_poly(c, d_ssp, h_witness_bits, delta, w);
    if (r) return r;
    // v = w + v_0 ; h = (v^2 - 1) / t   (src/snark.c:161-169)
    hipLaunchKernelGGL(k_add_slot, g1(d), dim3(256), 0, c->stream, w, src, 1u, d, v);
    HIP_TRY(c, hipGetLastError());
    r = mfh_poly_h(c, v, h);
    if (r) return r;
    if (fork) HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
    return MFH_OK;
  };
  // Host queueing order.  b_w first: the GPU already runs that long kernel while the host enqueues the chain's ~25 short launches --
  // pays when the b_w launch is long (it holds every CU's LDS, so the chain's LDS kernels mostly run after it anyway).  Chain first:
  // the chain runs at full speed alone and b_w's rows follow -- better when b_w's share is short (many ranks, resident CRS).
  const bool bw_first = c->overlap_mode == 2 || (c->overlap_mode == 1 && !res && (uint64_t)cnt * 2 >= m);
  if (bw_first) {
    rc = run_bw();
    if (!rc) rc = run_chain();
  } else {
    rc = run_chain();
    if (!rc) rc = run_bw();
  }
  if (rc) return rc;
  if (fork) HIP_TRY(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
  // S rows: (w, h) -> (v_w, h);  AS rows: (v, h) -> (hat_v, hat_h)   (src/snark.c:157-158,163-164,171-174, each row expanded once)
  // the coefficients of w, v, h are residues mod p of polynomial arithmetic: a zero is a 2^-32 event, so the row compaction would
  // only copy the identity; rows with a zero pair are still correct without it (they add zero)
  c->eval_dense = true;
  rc = eval_region(loS, 0, cS, w + loS, h + loS, pi_v_w, pi_h);
  if (!rc) rc = eval_region((size_t)d + loS, cS, cS, v + loS, h + loS, pi_hat_v, pi_hat_h);
  c->eval_dense = false;
  return rc;
}

int mfh_prove_partial(mfh_ctx *c, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
                      uint32_t rank, uint32_t world, uint64_t *d_partial) {
  return prove_partial_impl(c, d_crs_c8, d_ssp, h_witness_bits, delta, rank, world, nullptr, d_partial);
}

int mfh_prove_partial_w(mfh_ctx *c, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
                        uint32_t rank, uint32_t world, const uint64_t *d_wlanes, uint64_t *d_partial) {
  if (!d_wlanes) return MFH_EINVAL;
  return prove_partial_impl(c, d_crs_c8, d_ssp, h_witness_bits, delta, rank, world, d_wlanes, d_partial);
}

int mfh_prove_finish(mfh_ctx *c, uint64_t *d_proof, const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign) {
  if (!c || !d_proof || !h_smudge_mag || !h_smudge_sign) return MFH_EINVAL;
  const size_t ctl = (size_t)(c->P.n + 1) * ((c->P.logq + 63) / 64);
  // smudging: h, hat_h, hat_v, v_w, then v_w AGAIN; b_w is never smudged (src/snark.c:185-189)
  int rc = mfh_ct_smudge(c, d_proof, 4, h_smudge_mag, maglen, h_smudge_sign);
  if (rc) return rc;
  return mfh_ct_smudge(c, d_proof + 3 * ctl, 1, h_smudge_mag + 4 * maglen, maglen, h_smudge_sign + 4);
}

int mfh_prove(mfh_ctx *c, const uint8_t *d_crs_c8, const uint32_t *d_ssp, const uint8_t *h_witness_bits, uint32_t delta,
              const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign, uint64_t *d_proof) {
  if (!h_smudge_mag || !h_smudge_sign) return MFH_EINVAL;
  int rc = mfh_prove_partial(c, d_crs_c8, d_ssp, h_witness_bits, delta, 0, 1, d_proof);
  if (rc) return rc;
  return mfh_prove_finish(c, d_proof, h_smudge_mag, maglen, h_smudge_sign);
}

// ---- prover() for a batch of statements under one CRS and SSP ---------------------------------------------------------------------
// The S and AS regions are expanded ONCE per group of up to 31 proofs and the BT+BV region once per up to 255 (b_w's coefficients are
// witness bits: one byte-digit column per proof), the multiply-accumulate of all their coefficient vectors runs on the matrix cores
// (eval_rows_multi_io*, evalmm.hip); the witness pass reads the SSP once per 124 statements (a GEMM of the witness bits with the SSP
// bytes on the matrix cores); the polynomial step and the smudging stay per proof.  Proof b is bit-identical to mfh_prove with the
// same inputs.  Three building blocks, shared by the single-GPU call (mfh_prove_batch) and the row-sharded multi-GPU sequence
// (mfh_batch_chain -> exchange -> mfh_prove_batch_partial -> lane reduction -> mfh_prove_batch_finish):
//   batch_chain_launch      w = delta t + sum_bits v_i, v = w + v_0, h = (v^2 - 1) / t for a slab of statements   (src/snark.c:141-169)
//   batch_rows_supergroup   b_w's BT+BV rows and the S / AS rows of up to BSG statements, restricted to a rank's row shares   (:143-174)
//   batch_smudge            ct_smudge x 5 per proof in two launches                                               (:185-189)
}  // extern "C"

namespace {

constexpr uint32_t BG = 31;    // proofs per S / AS expansion when every group carries its own ones column: 62 coefficient vectors x 4 bytes + 1 = 249 of 256 digit columns
// proofs per super-group = per BT+BV expansion: one byte column each + the ones column = 256 digit columns.  The S / AS launches of the streaming regime
// serve the super-group's 2 x 255 coefficient vectors per region in two launches of 4 groups: 63 + 64 + 64 + 64 vectors, the first group carrying the ones
// column (sum_i A'[i][m] depends on the rows only) for the other three (MmIo::sa_from1) -- 255 proofs per two passes over a region's image instead of 248.
constexpr uint32_t BSG = 255;
// ... and when every group regenerates the keystream (no image: one group of 31 proofs = 62 vectors + its own ones column per expansion) a super-group is 8 x 31
// statements, so that no expansion runs for a handful of left-over vectors
constexpr uint32_t BSG_REGEN = 248;

struct OnStream {  // helpers launch on c->stream
  mfh_ctx *c;
  hipStream_t keep;
  OnStream(mfh_ctx *c_, hipStream_t s) : c(c_), keep(c_->stream) { c->stream = s; }
  ~OnStream() { c->stream = keep; }
};
struct OnSide {  // eval_rows_multi_io* launch on c->stream with the workspace c->mm_ws_sel selects
  mfh_ctx *c;
  hipStream_t keep;
  OnSide(mfh_ctx *c_, hipStream_t s) : c(c_), keep(c_->stream) { c->stream = s; c->mm_ws_sel = 1; }
  ~OnSide() { c->stream = keep; c->mm_ws_sel = 0; }
};

// scratch of a batch call: [whv x (w | h | v) of a super-group] | CW (packed witness bits + deltas of a super-group) | ONE | CT_T | the
// launches' column-sum slots (256 int64 each)
struct BatchScratch {
  uint32_t *WHV;
  uint8_t *CW;      // ncw areas of cw_stride bytes: the packed witness bits of a super-group, then its deltas
  size_t cw_stride, cw_delta_off;
  uint32_t *SMU;    // staged smudging terms u p of the call: [proof][5][KW] words, then the signs [proof][5]
  uint8_t *SMS;
  uint32_t *ONE;
  uint64_t *CT_T;
  int64_t *SCZ;
  size_t nslots;
};
int batch_scratch(mfh_ctx *c, uint32_t nproofs, uint32_t whv /* w | h | v areas of a super-group */, BatchScratch &B, uint32_t ncw = 1,
                  uint32_t nsmudge = 0 /* proofs whose smudging terms are staged */) {
  const uint32_t d = c->P.d, m = c->P.m, n = c->P.n;
  const size_t ctl = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
  const size_t nsg = ((size_t)nproofs + BSG_REGEN - 1) / BSG_REGEN;  // (the smaller super-group size: never fewer slots than a call uses)
  B.nslots = nsg * (2 + 2 * ((BSG + BG - 1) / BG));  // multi-vector launches of the call
  const size_t whv_b = (size_t)whv * 3 * BSG * d * 4;
  const size_t packed = ((size_t)BSG * ((m + 6) / 8) + 3) & ~(size_t)3;
  const size_t cw_b = packed + (size_t)BSG * 4;
  const size_t cw_pad = (cw_b + 255) & ~(size_t)255;
  const size_t KW = 2 * (c->P.logq / 64);
  const size_t sm_b = ((size_t)nsmudge * 5 * (KW * 4 + 1) + 255) & ~(size_t)255;
  const size_t need = whv_b + cw_pad * ncw + sm_b + 256 + ctl * 8 + B.nslots * 2048;
  if (c->batch_bytes < need) {
    if (c->d_batch) { hipDeviceSynchronize(); hipFree(c->d_batch); c->d_batch = nullptr; c->batch_bytes = 0; }
    HIP_TRY(c, hipMalloc(&c->d_batch, need));
    c->batch_bytes = need;
  }
  uint8_t *base = (uint8_t *)c->d_batch;
  B.WHV = (uint32_t *)base;
  B.CW = base + whv_b;
  B.cw_stride = cw_pad;
  B.cw_delta_off = packed;
  B.SMU = (uint32_t *)(base + whv_b + cw_pad * ncw);
  B.SMS = (uint8_t *)B.SMU + (size_t)nsmudge * 5 * KW * 4;
  B.ONE = (uint32_t *)(base + whv_b + cw_pad * ncw + sm_b);
  B.CT_T = (uint64_t *)(base + whv_b + cw_pad * ncw + sm_b + 256);
  B.SCZ = (int64_t *)(base + whv_b + cw_pad * ncw + sm_b + 256 + ctl * 8);
  return MFH_OK;
}
// Everything the host contributes to a call, staged in one pinned buffer and copied once, on c->stream: per super-group the packed
// witness bits (the m - 1 bits of a statement repacked densely) and the deltas, then the smudging terms u p of all five draws of every
// proof (schoolbook by 32-bit words: src/lwe.c:65-76) and their signs.  B was sized with ncw = super-groups, nsmudge = nproofs.
int batch_stage_host(mfh_ctx *c, const BatchScratch &B, uint32_t nproofs, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                     const uint8_t *h_mag, size_t maglen, const uint8_t *h_sign, uint32_t SG) {
  const uint32_t m = c->P.m, nsg = (nproofs + SG - 1) / SG;
  const uint32_t bstride = (m + 6) / 8, KW = 2 * (c->P.logq / 64);
  if (maglen + 4 > (size_t)KW * 4) { c->err = "smudge magnitude too wide"; return MFH_EINVAL; }
  const size_t total = (size_t)((uint8_t *)B.SMS - B.CW) + (size_t)nproofs * 5;
  uint8_t *st = (uint8_t *)pin_acquire(c, c->pin_cw, total);
  if (!st) return MFH_ENOMEM;
  for (uint32_t g = 0; g < nsg; g++) {
    const uint32_t s0 = g * SG, sg = std::min(SG, nproofs - s0);
    uint8_t *a = st + (size_t)g * B.cw_stride;
    for (uint32_t b = 0; b < sg; b++) memcpy(a + (size_t)b * bstride, h_bits + (size_t)(s0 + b) * bits_stride, bstride);
    memcpy(a + B.cw_delta_off, h_delta + s0, (size_t)sg * 4);
  }
  uint32_t *up = (uint32_t *)(st + ((uint8_t *)B.SMU - B.CW));
  for (size_t i = 0; i < (size_t)nproofs * 5; i++) {
    uint64_t carry = 0;
    for (uint32_t l = 0; l < KW; l++) {
      uint32_t w = 0;
      const size_t o = (size_t)l * 4;
      if (o < maglen) memcpy(&w, h_mag + i * maglen + o, std::min<size_t>(4, maglen - o));
      const uint64_t t = (uint64_t)w * P32 + carry;
      up[i * KW + l] = (uint32_t)t;
      carry = t >> 32;
    }
  }
  memcpy(st + ((uint8_t *)B.SMS - B.CW), h_sign, (size_t)nproofs * 5);
  HIP_TRY(c, hipMemcpyAsync(B.CW, st, total, hipMemcpyHostToDevice, c->stream));
  pin_release(c, c->pin_cw);
  return MFH_OK;
}
// b += +-(u p) on the b coordinate of ciphertext (i / per) * 5 + slot0 + i % per of a super-group's proof structs, with staged term
// (i / per) * 5 + uslot0 + i % per (ct_smudge, src/lwe.c:65-76): per = 4, slots 0.. and terms 0..: h, hat_h, hat_v, v_w; per = 1, slot 3,
// term 4: v_w AGAIN (src/snark.c:185-189).  b_w's ciphertext is never touched.
__global__ void k_smudge_batch(uint64_t *cts, uint32_t n, uint32_t L, uint32_t KW, const uint32_t *__restrict__ up, const uint8_t *__restrict__ sign,
                               uint32_t count, uint32_t per, uint32_t slot0, uint32_t uslot0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t pb = i / per, j = i % per;
  uint32_t *b = reinterpret_cast<uint32_t *>(cts + ((uint64_t)(pb * 5 + slot0 + j) * (n + 1) + n) * L);
  const uint32_t *u = up + (uint64_t)(pb * 5 + uslot0 + j) * KW;
  if (sign[pb * 5 + uslot0 + j] & 1) {
    uint32_t borrow = 0;
    for (uint32_t l = 0; l < KW; l++) {
      const uint64_t t = (uint64_t)b[l] - u[l] - borrow;
      b[l] = (uint32_t)t;
      borrow = (uint32_t)(t >> 63);
    }
  } else {
    uint32_t carry = 0;
    for (uint32_t l = 0; l < KW; l++) {
      const uint64_t t = (uint64_t)b[l] + u[l] + carry;
      b[l] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
  }
  for (uint32_t l = KW; l < 2 * L; l++) b[l] = 0;
}
int batch_smudge_staged(mfh_ctx *c, const BatchScratch &B, uint64_t *sproofs, uint32_t s0, uint32_t sg) {
  const uint32_t L = (c->P.logq + 63) / 64, KW = 2 * (c->P.logq / 64);
  const uint32_t *up = B.SMU + (size_t)s0 * 5 * KW;
  const uint8_t *sn = B.SMS + (size_t)s0 * 5;
  hipLaunchKernelGGL(k_smudge_batch, dim3((sg * 4 + 63) / 64), dim3(64), 0, c->stream, sproofs, c->P.n, L, KW, up, sn, sg * 4, 4u, 0u, 0u);
  hipLaunchKernelGGL(k_smudge_batch, dim3((sg + 63) / 64), dim3(64), 0, c->stream, sproofs, c->P.n, L, KW, up, sn, sg, 1u, 3u, 4u);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}
int batch_streams(mfh_ctx *c) {
  if (!c->side) {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  if (!c->ev_chain) {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_chain, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_chain_done, hipEventDisableTiming));
  }
  return MFH_OK;
}

// The chain of sg <= BSG statements on c->stream: W = delta t + sum_bits v_i (src/snark.c:141,147-155), V = W + v_0, H = (V^2 - 1) / t
// (src/snark.c:161-169); W, H, V are sg x d coefficients each.  The witness pass and the polynomial step have their own scratch (wws,
// the poly buffers).
int batch_chain_launch(mfh_ctx *c, const mf::SspSrc &src, const uint32_t *d_ssp, uint32_t sg, const uint8_t *h_bits, size_t bits_stride,
                       const uint32_t *h_delta, uint32_t *W, uint32_t *H, uint32_t *V, hipEvent_t ev_witness = nullptr) {
  const uint32_t d = c->P.d;
  int rc = MFH_OK;
  // d % 128 == 0: a GEMM on the matrix cores, one read (dense SSP) or one generation (generator-defined SSP) of the selected rows per
  // 124 statements; otherwise the VALU form, read or generated once per 12 statements
  if (d % 128 == 0) {
    // the whole super-group (up to 255 statements) in one read (dense SSP) or one generation (generator-defined SSP) of the rows: k_witness_mm8q / k_witness_mm8q_prg
    const uint32_t per = c->witness_per ? c->witness_per : 256u;  // (mfh_set_witness_per: A/B knob; a super-group is at most BSG = 255 statements)
    for (uint32_t b0 = 0; b0 < sg; b0 += per) {
      rc = mfh_witness_poly_mm(c, d_ssp, std::min(per, sg - b0), h_bits + (size_t)b0 * bits_stride, bits_stride, h_delta + b0, W + (size_t)b0 * d);
      if (rc) return rc;
    }
  } else {
    for (uint32_t b0 = 0; b0 < sg; b0 += 12) {
      rc = mfh_witness_poly_multi(c, d_ssp, std::min(12u, sg - b0), h_bits + (size_t)b0 * bits_stride, bits_stride, h_delta + b0, W + (size_t)b0 * d);
      if (rc) return rc;
    }
  }
  if (ev_witness) HIP_TRY(c, hipEventRecord(ev_witness, c->stream));  // the HBM-bound part of the chain is over
  hipLaunchKernelGGL(k_add_slot_multi, dim3((d + 255) / 256, sg), dim3(256), 0, c->stream, W, src, 1u, d, V);
  HIP_TRY(c, hipGetLastError());
  return mfh_poly_h_multi(c, V, H, sg);  // one set of launches, sg times the work each
}

// where the coefficient vectors of a super-group's statements are: statement b's w / h / v restricted to the S / AS rows of the share at
// w / h / v + b * stride (mfh_prove_batch: the whole polynomials where the chain left them, stride d)
struct BatchCoef {
  const uint32_t *w, *h, *v;
  uint64_t stride;
};

// The row work of one super-group of sg <= BSG statements over rank `rank`'s contiguous row shares (the whole regions when world == 1):
//   b_w = [delta ct_t +] sum_{bit} ct_{v_i} over the share of the m BT+BV rows (src/snark.c:143-155), the bits of all sg statements as
//   byte coefficients; S rows with (w, h) -> (v_w, h) on the caller's stream and AS rows with (h, v) -> (hat_h, hat_v) on the side
//   stream, every row expanded (or streamed from the image) once per group of 31 proofs (src/snark.c:157-174).
// sproofs: the sg proof structs (5 ciphertexts each, h | hat_h | hat_v | v_w | b_w).  h_delta == nullptr: no delta ct_t term (a partial
// proof: the term is added once, after the ranks' shares have been summed).  wait_ev: awaited before the S / AS launches (the chain).
// accumulate: add onto what sproofs holds (mod 2^(64K)) -- the row slabs of mfh_prove_batch when the image does not fit HBM.
// b_w of a super-group on c->stream from the staged area d_cw (packed bits, then -- has_delta -- the deltas at B.cw_delta_off)
int batch_bw(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, uint32_t sg, const uint8_t *d_cw, bool has_delta, uint64_t *sproofs,
             const BatchScratch &B, size_t &slot, int accumulate) {
  const uint32_t d = c->P.d, m = c->P.m, n = c->P.n;
  const uint32_t ctb = c->P.logq / 8;
  const size_t ctl = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
  const uint64_t ctr_ct = (uint64_t)ctb * n;
  const uint64_t pstride = 5 * ctl;
  const uint32_t lo = (uint32_t)((uint64_t)m * rank / world), cnt = (uint32_t)((uint64_t)m * (rank + 1) / world) - lo;
  const uint32_t bstride = (m + 6) / 8;
  MmIo io_bw = {{nullptr, nullptr}, sg, {sproofs + 4 * ctl, nullptr}, sg, pstride, d_cw, bstride, B.SCZ + 256 * slot++};
  io_bw.bits_row0 = lo;
  if (has_delta) {  // + delta_b ct_t (src/snark.c:143-145) in the launch's epilogue
    io_bw.add_ct = B.CT_T;
    io_bw.add_scale = (const uint32_t *)(d_cw + B.cw_delta_off);
  }
  return eval_rows_multi_io(c, ctr_ct * ((uint64_t)2 * d + lo), cnt, d_crs_c8 + ((size_t)2 * d + lo) * ctb, io_bw, sg, 1, accumulate);
}
// d_cw: the super-group's staged area (batch_stage_host), or nullptr: staged here from h_bits / h_delta.
int batch_rows_supergroup(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, uint32_t sg, const uint8_t *h_bits,
                          size_t bits_stride, const BatchCoef &co, uint64_t *sproofs, const BatchScratch &B, size_t &slot,
                          const uint32_t *h_delta, hipEvent_t wait_ev, int accumulate = 0, const uint8_t *d_cw = nullptr, bool bw_done = false,
                          uint32_t ws_slot = 0 /* early mode: which half of ws3 */, bool *tail_on_side = nullptr /* early mode: the epilogues were queued on c->side, not joined */) {
  const uint32_t d = c->P.d, m = c->P.m, n = c->P.n;
  const uint32_t ctb = c->P.logq / 8;
  const size_t ctl = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
  const uint64_t ctr_ct = (uint64_t)ctb * n;
  const uint64_t pstride = 5 * ctl;  // component `slot` of consecutive proofs (struct order h | hat_h | hat_v | v_w | b_w)
  hipStream_t const main_stream = c->stream, side_stream = c->side;
  const uint32_t loS = (uint32_t)((uint64_t)d * rank / world), cS = (uint32_t)((uint64_t)d * (rank + 1) / world) - loS;
  int rc = MFH_OK;
  if (!bw_done) {
    if (!d_cw) {
      // ---- b_w: the packed bits travel as they are (2.7 KB per statement at the default size; the digit kernel unpacks them), the deltas
      // behind them, through one pinned staging buffer and one copy
      const uint32_t bstride = (m + 6) / 8;  // the m - 1 bits of a statement, repacked densely
      const size_t staged = B.cw_delta_off + (h_delta ? (size_t)sg * 4 : 0);
      uint8_t *h_cw = (uint8_t *)pin_acquire(c, c->pin_cw, staged);
      if (!h_cw) return MFH_ENOMEM;
      for (uint32_t b = 0; b < sg; b++) memcpy(h_cw + (size_t)b * bstride, h_bits + (size_t)b * bits_stride, bstride);
      if (h_delta) memcpy(h_cw + B.cw_delta_off, h_delta, (size_t)sg * 4);
      HIP_TRY(c, hipMemcpyAsync(B.CW, h_cw, staged, hipMemcpyHostToDevice, c->stream));
      pin_release(c, c->pin_cw);
      d_cw = B.CW;
    }
    rc = batch_bw(c, d_crs_c8, rank, world, sg, d_cw, h_delta != nullptr, sproofs, B, slot, accumulate);
    if (rc) return rc;
  }
  if (wait_ev) HIP_TRY(c, hipStreamWaitEvent(main_stream, wait_ev, 0));
  HIP_TRY(c, hipEventRecord(c->ev_fork, main_stream));
  HIP_TRY(c, hipStreamWaitEvent(side_stream, c->ev_fork, 0));
  // with the image registered one streaming launch serves NGL = 4 groups: the image is then read from HBM once per 124 proofs (2 and 8
  // groups per launch measured 2 % and 1 % slower, 1 group 10 % slower)
  constexpr uint32_t NGLMAX = 8;
  const uint32_t NGL = c->mm_image ? c->batch_ngl : 1u;
  const MmRegion regs[2] = {{ctr_ct * loS, d_crs_c8 + (size_t)loS * ctb}, {ctr_ct * ((uint64_t)d + loS), d_crs_c8 + ((size_t)d + loS) * ctb}};
  // image registered for both regions: the S and the AS groups of a round share ONE streaming launch (no two launches of the kernel overlap:
  // one region's fragments at a time in the L2s); otherwise two streams, so that one stream's small kernels run under the other's row kernel
  const bool merged = mm_image_covers(c, regs[0].off, cS) && mm_image_covers(c, regs[1].off, cS) && c->batch_merge;
  if (merged) {
    // Streaming regime, everything on the caller's stream: all rounds' operands (digit fragments, column sums: 0.02 ms per group) first,
    // then per round the streaming launch and its epilogues (partial products -> ciphertext words in the proof structs: 0.05 ms per group).
    // Every round has its own digit / partial-product area in ws3.
    constexpr uint32_t RMAX = (2 * BSG + 62) / 63;  // rounds of a super-group with one group per launch and region
    MmIo io[RMAX][2 * NGLMAX];
    uint32_t nv[RMAX][2 * NGLMAX];
    MmsPlan plan[RMAX];
    uint32_t R = 0;
    size_t ws_need = 0;
    // A region's coefficient vectors of the super-group as one list [X_0 .. X_(sg-1), Y_0 .. Y_(sg-1)] (S: X = w -> v_w, Y = h -> h; AS: X = h -> hat_h,
    // Y = v -> hat_v), cut into consecutive runs: per launch the first group of a region takes 63 vectors and carries the ones column, the others 64 each and
    // borrow it.  A run crosses the X | Y boundary at most once, which is what MmIo's two operand / result arrays express.
    const uint32_t nvtot = 2 * sg;
    auto run_io = [&](int region, uint32_t p, uint32_t cnt, uint32_t lender1) -> MmIo {
      const uint32_t x0 = std::min(p, sg), x1 = std::min(p + cnt, sg), y0 = std::max(p, sg) - sg;
      const uint32_t *X = region == 0 ? co.w : co.h, *Y = region == 0 ? co.h : co.v;
      uint64_t *OX = sproofs + (region == 0 ? 3 : 1) * ctl, *OY = sproofs + (region == 0 ? 0 : 2) * ctl;
      MmIo m = MmIo{{X + (uint64_t)x0 * co.stride, Y + (uint64_t)y0 * co.stride}, x1 - x0, {OX + (uint64_t)x0 * pstride, OY + (uint64_t)y0 * pstride}, x1 - x0, pstride,
                    nullptr, 0, B.SCZ + 256 * slot++, co.stride, 0};
      m.sa_from1 = lender1;
      return m;
    };
    for (uint32_t p0 = 0; p0 < nvtot; R++) {
      uint32_t cnt[NGLMAX], pos[NGLMAX], ng = 0, p = p0;
      for (; ng < NGL && p < nvtot; ng++) {
        pos[ng] = p;
        cnt[ng] = std::min(ng == 0 ? 63u : 64u, nvtot - p);
        p += cnt[ng];
      }
      for (uint32_t k = 0; k < ng; k++) {
        io[R][k] = run_io(0, pos[k], cnt[k], k ? 1 : 0);            // S groups: lender = group 0 of the launch
        io[R][ng + k] = run_io(1, pos[k], cnt[k], k ? ng + 1 : 0);  // AS groups: lender = group ng
        nv[R][k] = nv[R][ng + k] = cnt[k];
      }
      p0 = p;
      if (!mms_plan(c, regs, 2, cS, io[R], nv[R], ng, 4, plan[R])) { c->err = "mfh_prove_batch: the registered image does not serve the S / AS regions"; return MFH_EINVAL; }
      ws_need += (mms_ws_bytes(plan[R]) + 255) & ~(size_t)255;
    }
    // early mode (mfh_set_mm_width): the epilogues of this super-group run on the side stream beside the NEXT super-group's digit and streaming kernels, which
    // therefore get the other half of ws3
    const bool early = c->batch_early_chain && tail_on_side;
    // The two halves are ONE stride apart for the whole call: the stride of the call's first super-group, which is its largest (a ragged last one has fewer
    // groups and a smaller ws_need: placed at ITS ws_need, an odd-indexed short super-group would land inside the half its predecessor's epilogues still read).
    if (early && (ws_slot == 0 || c->early_ws_half < ws_need)) {
      if (ws_slot != 0) { c->err = "early chain: a later super-group needs more scratch than the call's first"; return MFH_EINVAL; }
      c->early_ws_half = ws_need;
    }
    rc = buf_reserve(c, c->ws3, c->ws3_bytes, early ? 2 * c->early_ws_half : ws_need);
    if (rc) return rc;
    size_t wo = early ? (size_t)(ws_slot & 1) * c->early_ws_half : 0;
    for (uint32_t r = 0; r < R; r++) {
      mms_bind(plan[r], (uint8_t *)c->ws3 + wo);
      wo += (mms_ws_bytes(plan[r]) + 255) & ~(size_t)255;
      rc = mms_digits(c, plan[r], io[r], nv[r]);
      if (rc) return rc;
    }
    for (uint32_t r = 0; r < R; r++) {
      rc = mms_stream(c, plan[r]);
      if (rc) return rc;
      if (early) {  // epilogue on the side stream, behind this round's launch only
        while (c->ev_round.size() <= r) {
          hipEvent_t e;
          HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
          c->ev_round.push_back(e);
        }
        HIP_TRY(c, hipEventRecord(c->ev_round[r], main_stream));
        HIP_TRY(c, hipStreamWaitEvent(side_stream, c->ev_round[r], 0));
        OnStream side(c, side_stream);
        rc = mms_finish(c, plan[r], io[r], nv[r], accumulate);
        if (rc) return rc;
        continue;
      }
      rc = mms_finish(c, plan[r], io[r], nv[r], accumulate);  // (on a side stream beside the next launch the epilogues starve -- 0.5 ms each instead of 0.05 -- and slow that launch by more than they take alone)
      if (rc) return rc;
    }
    if (early) {  // the caller continues this super-group (smudging, completion event) on the side stream and joins once, at the end of the call
      *tail_on_side = true;
      return MFH_OK;
    }
  } else {
    // Regenerating regime (or one group per launch): two streams, so that one stream's small kernels run under the other's row kernel
    for (uint32_t g0 = 0; g0 < sg; g0 += NGL * BG) {
      MmIo io[2 * NGLMAX];
      uint32_t nv[2 * NGLMAX], ng = 0;
      for (uint32_t k = 0; k < NGL && g0 + k * BG < sg; k++) ng++;
      for (uint32_t k = 0; k < ng; k++) {
        const uint32_t gg = g0 + k * BG, g = std::min(BG, sg - gg);
        uint64_t *proofs = sproofs + (size_t)gg * 5 * ctl;
        const uint64_t o = (uint64_t)gg * co.stride;
        io[k] = MmIo{{co.w + o, co.h + o}, g, {proofs + 3 * ctl, proofs}, g, pstride, nullptr, 0, B.SCZ + 256 * slot++, co.stride, 0};                 // S: (w, h) -> (v_w, h)
        io[ng + k] = MmIo{{co.h + o, co.v + o}, g, {proofs + ctl, proofs + 2 * ctl}, g, pstride, nullptr, 0, B.SCZ + 256 * slot++, co.stride, 0};  // AS: (h, v) -> (hat_h, hat_v)
        nv[k] = nv[ng + k] = 2 * g;
      }
      rc = eval_rows_multi_io_regions(c, regs, 1, cS, io, nv, ng, 4, accumulate);
      if (rc) return rc;
      {
        OnSide side(c, side_stream);
        rc = eval_rows_multi_io_regions(c, regs + 1, 1, cS, io + ng, nv + ng, ng, 4, accumulate);
        if (rc) return rc;
      }
    }
  }
  HIP_TRY(c, hipEventRecord(c->ev_join, side_stream));
  HIP_TRY(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
  return MFH_OK;
}

// smudging of sg proofs in two launches: h, hat_h, hat_v, v_w with draws 0..3, then v_w AGAIN with draw 4; b_w never
// (src/snark.c:185-189).  A zero magnitude leaves a ciphertext unchanged.
int batch_smudge(mfh_ctx *c, uint64_t *sproofs, uint32_t sg, const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign) {
  std::vector<uint8_t> mags((size_t)sg * 5 * maglen), signs((size_t)sg * 5);
  for (int pass = 0; pass < 2; pass++) {
    std::fill(mags.begin(), mags.end(), 0);
    std::fill(signs.begin(), signs.end(), 0);
    for (uint32_t b = 0; b < sg; b++) {
      const uint8_t *pm = h_smudge_mag + (size_t)b * 5 * maglen, *ps = h_smudge_sign + (size_t)b * 5;
      if (pass == 0) {
        memcpy(&mags[(size_t)b * 5 * maglen], pm, 4 * maglen);
        memcpy(&signs[(size_t)b * 5], ps, 4);
      } else {
        memcpy(&mags[((size_t)b * 5 + 3) * maglen], pm + 4 * maglen, maglen);
        signs[(size_t)b * 5 + 3] = ps[4];
      }
    }
    int rc = mfh_ct_smudge(c, sproofs, (size_t)sg * 5, mags.data(), maglen, signs.data());
    if (rc) return rc;
  }
  return MFH_OK;
}

// ct_t = the BT row as a ciphertext (eval_poly of one row with coefficient 1): b_w's delta * ct_t term (src/snark.c:143-145)
int batch_ct_t(mfh_ctx *c, const uint8_t *d_crs_c8, const BatchScratch &B) {
  const uint32_t d = c->P.d, ctb = c->P.logq / 8;
  HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)B.ONE, 1, 1, c->stream));
  return mfh_eval_rows(c, (uint64_t)ctb * c->P.n * 2 * d, 1, d_crs_c8 + (size_t)2 * d * ctb, B.ONE, nullptr, B.CT_T, nullptr, 0);
}

// More than one group and no image registered: expand this rank's row shares of the CRS once for the whole call into a transient image in
// MFMA A-fragment order (scratch kept by the context) and stream it for every group, instead of running AES per group.
struct ImageGuard {
  mfh_ctx *c;
  bool on;
  ~ImageGuard() { if (on) mfh_crs_set_resident_mm(c, nullptr); }
};
int batch_transient_image(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t nproofs, uint32_t rank, uint32_t world, ImageGuard &guard) {
  const uint32_t ctb = c->P.logq / 8;
  if (c->mm_image || !c->batch_image || nproofs <= BG || (((uint64_t)c->P.n * ctb) & 7)) return MFH_OK;
  const size_t ib = mfh_crs_mm_share_bytes(c, rank, world);
  if (c->batch_img_bytes < ib) {
    if (c->batch_img) { hipDeviceSynchronize(); hipFree(c->batch_img); c->batch_img = nullptr; c->batch_img_bytes = 0; }
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0;
    // no room (the rest of the call and the caller need memory too): the groups regenerate the keystream
    if (ib <= mem_free / 4 * 3 && hipMalloc(&c->batch_img, ib) == hipSuccess) c->batch_img_bytes = ib;
    else { c->batch_img = nullptr; (void)hipGetLastError(); }
  }
  if (!c->batch_img) return MFH_OK;
  int rc = mfh_crs_expand_mm_share(c, d_crs_c8, rank, world, (uint8_t *)c->batch_img);
  if (rc) return rc;
  mfh_crs_set_resident_mm_share(c, (const uint8_t *)c->batch_img, rank, world);
  guard.on = true;
  return MFH_OK;
}

}  // namespace

extern "C" {

int mfh_prove_batch(mfh_ctx *c, const uint8_t *d_crs_c8, const uint32_t *d_ssp, uint32_t nproofs, const uint8_t *h_witness_bits,
                    size_t bits_stride, const uint32_t *h_delta, const uint8_t *h_smudge_mag, size_t maglen, const uint8_t *h_smudge_sign,
                    uint64_t *d_proofs) {
  if (!c) return MFH_EINVAL;
  if (!nproofs) return MFH_OK;
  if (!d_crs_c8 || !h_witness_bits || !h_delta || !h_smudge_mag || !h_smudge_sign || !d_proofs) return MFH_EINVAL;
  if (c->resident_rows) { c->err = "mfh_prove_batch regenerates the keystream: clear the resident CRS image first"; return MFH_EUNSUPPORTED; }
  if (c->mm_image && c->mm_world != 1) { c->err = "mfh_prove_batch: the registered matrix-core image holds one rank's row shares (use mfh_prove_batch_partial)"; return MFH_EINVAL; }
  mf::SspSrc src;
  {
    int rc0 = ssp_src(c, d_ssp, src);
    if (rc0) return rc0;
  }
  const uint32_t d = c->P.d, m = c->P.m, n = c->P.n;
  if (bits_stride < (m + 6) / 8) { c->err = "bits_stride shorter than the m - 1 witness bits"; return MFH_EINVAL; }
  for (uint32_t b = 0; b < nproofs; b++)
    if (h_delta[b] >= P32) { c->err = "delta must be < p"; return MFH_EINVAL; }
  // (checked before any GPU work in every regime: the row-slab path smudges only after all its launches)
  if (maglen + 4 > (size_t)(c->P.logq / 64) * 8) { c->err = "smudge magnitude too wide"; return MFH_EINVAL; }
  const size_t ctl = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
  HIP_TRY(c, hipSetDevice(c->device));
  // The chain (witness pass + polynomial step) of a super-group runs on its own stream: the first one beside the CRS expansion, the chain
  // of super-group k + 1 after the row work of super-group k (beside its smudging), into the other of two w | h | v areas.  Queued further
  // ahead (one area per super-group, chains beside the previous super-groups' streaming launches) the call takes the same time --
  // measured with 2, 3, 4, 8 areas: 88.2, 88.7, 89.4, 89.9 ms per 992 statements -- because every kernel of the call fills the CUs it
  // gets (one k_mmstream workgroup owns a CU's registers and LDS, k_expand_mm runs 8 waves per SIMD): concurrent streams time-share
  // the GPU and the work is conserved.  In turn keeps the streaming launches' durations clean.
  // super-group size: 255 statements when the row work streams an image (the caller's, or the call's transient one: groups of 63 / 64 coefficient vectors
  // with a shared ones column), 248 = 8 x 31 when every group regenerates the keystream
  const bool will_stream = c->mm_image || (c->batch_image && nproofs > BG && (((uint64_t)n * (c->P.logq / 8)) & 7) == 0);
  const uint32_t SG = will_stream ? BSG : BSG_REGEN;
  const uint32_t nsg = (nproofs + SG - 1) / SG;
  // Row slabs.  When the call would stream an image that does not fit HBM (363 GB at 2^20 constraints), the CRS rows are cut into
  // `nsl` slabs -- exactly the row shares of the multi-GPU prover, one after the other on this GPU: a slab's image is expanded once and
  // streamed for EVERY group of the call (results accumulated mod 2^(64K)), so the keystream is generated once per call instead of
  // once per group of 31 proofs.  All chains run first (one w | h | v area per super-group).
  uint32_t nsl = 1;
  const uint32_t ctb0 = c->P.logq / 8;
  const bool slabs_ok = !c->mm_image && c->batch_image && nproofs > BG && (((uint64_t)n * ctb0) & 7) == 0;
  if (slabs_ok && c->batch_slabs) nsl = std::min(c->batch_slabs, std::max(1u, std::min(d, m)));  // forced (tests, tuning)
  else if (slabs_ok) {
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) {
      const size_t ib = mfh_crs_mm_image_bytes(c), avail = mem_free + c->batch_img_bytes + c->batch_bytes;  // (what the context holds is re-used)
      if (ib > avail / 4 * 3) {
        const size_t budget = std::max<size_t>(avail / 3, (size_t)1 << 30);
        nsl = (uint32_t)std::min<size_t>(256, (ib + budget - 1) / budget);
        const size_t area = (size_t)3 * BSG * d * 4, maxsg = std::max<size_t>(1, avail / 4 / area);
        if (nsg > maxsg) {  // the w | h | v areas of the whole call would not fit: sub-calls of maxsg super-groups
          const size_t ctl0 = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
          for (uint32_t s0 = 0; s0 < nproofs; s0 += (uint32_t)maxsg * SG) {
            const uint32_t cnt = std::min<uint32_t>((uint32_t)maxsg * SG, nproofs - s0);
            int r = mfh_prove_batch(c, d_crs_c8, d_ssp, cnt, h_witness_bits + (size_t)s0 * bits_stride, bits_stride, h_delta + s0,
                                    h_smudge_mag + (size_t)s0 * 5 * maglen, maglen, h_smudge_sign + (size_t)s0 * 5, d_proofs + (size_t)s0 * 5 * ctl0);
            if (r) return r;
          }
          c->last_batch_sg = 0;  // (several sub-calls: no per-super-group completion to hand out; a caller drains after mfh_sync)
          return MFH_OK;
        }
      }
    }
  }
  const uint32_t nbuf = nsl > 1 ? nsg : 2;
  // one completion event per super-group (mfh_prove_batch_stream_wait: a caller's copy stream drains super-group k while k + 1 runs)
  while (c->ev_sgdone.size() < nsg) {
    hipEvent_t e;
    HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_sgdone.push_back(e);
  }
  c->last_batch_sg = 0;  // (set when the whole call has been queued)
  BatchScratch B;
  int rc = batch_scratch(c, nproofs, nbuf, B, nsl > 1 ? 1 : nsg, nsl > 1 ? 0 : nproofs);
  if (rc) return rc;
  size_t slot = 0;
  HIP_TRY(c, hipMemsetAsync(B.SCZ, 0, B.nslots * 2048, c->stream));
  rc = batch_streams(c);
  if (rc) return rc;
  while (c->ev_cdone.size() < nbuf) {
    hipEvent_t e0, e1, e2;
    HIP_TRY(c, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    c->ev_cdone.push_back(e0);
    c->ev_rdone.push_back(e1);
    c->ev_wdone.push_back(e2);
  }
  hipStream_t const main_stream = c->stream, chain_stream = c->side2;
  auto whv_of = [&](uint32_t sgi, uint32_t *&W, uint32_t *&H, uint32_t *&V) {
    W = B.WHV + (size_t)(sgi % nbuf) * 3 * BSG * d;
    H = W + (size_t)BSG * d;
    V = H + (size_t)BSG * d;
  };
  auto launch_chain = [&](uint32_t sgi) -> int {  // the chain stream has been told what to wait for
    const uint32_t s0 = sgi * SG, sg = std::min(SG, nproofs - s0);
    uint32_t *WALL, *HALL, *VALL;
    whv_of(sgi, WALL, HALL, VALL);
    OnStream chain(c, chain_stream);
    int r = batch_chain_launch(c, src, d_ssp, sg, h_witness_bits + (size_t)s0 * bits_stride, bits_stride, h_delta + s0, WALL, HALL, VALL, c->ev_wdone[sgi % nbuf]);
    if (r) return r;
    HIP_TRY(c, hipEventRecord(c->ev_cdone[sgi % nbuf], chain_stream));
    return MFH_OK;
  };
  HIP_TRY(c, hipEventRecord(c->ev_chain, main_stream));  // what the caller's stream has been given so far no longer reads the scratch
  HIP_TRY(c, hipStreamWaitEvent(chain_stream, c->ev_chain, 0));
  rc = launch_chain(0);
  if (rc) return rc;
  if (nsl > 1) {
    for (uint32_t k = 1; k < nsg; k++) {
      rc = launch_chain(k);
      if (rc) return rc;
    }
    // The slab count was sized from the free memory BEFORE this call's own scratch (w | h | v areas of every super-group, digit and partial-product areas) was
    // reserved; when the slab image then does not fit, halve the slabs (twice as many, up to 256) instead of giving up: the result does not depend on the count.
    for (;;) {
      size_t ib = 0;
      for (uint32_t r = 0; r < nsl; r++) ib = std::max(ib, mfh_crs_mm_share_bytes(c, r, nsl));
      if (c->batch_img_bytes >= ib) break;
      if (c->batch_img) { hipDeviceSynchronize(); hipFree(c->batch_img); c->batch_img = nullptr; c->batch_img_bytes = 0; }
      size_t mem_free = 0, mem_total = 0;
      const bool fits = hipMemGetInfo(&mem_free, &mem_total) != hipSuccess || ib + ((size_t)8 << 30) <= mem_free;  // (leave room for the launches' workspaces)
      if (fits && hipMalloc(&c->batch_img, ib) == hipSuccess) { c->batch_img_bytes = ib; break; }
      c->batch_img = nullptr;
      (void)hipGetLastError();
      const uint32_t cap = std::min<uint32_t>(256u, std::max(1u, std::min(d, m)));
      if (nsl >= cap || c->batch_slabs) { c->err = "mfh_prove_batch: no room for a row slab of the CRS image"; return MFH_ENOMEM; }
      nsl = std::min(cap, nsl * 2);
    }
    rc = batch_ct_t(c, d_crs_c8, B);
    if (rc) return rc;
    ImageGuard slab{c, true};
    for (uint32_t r = 0; r < nsl; r++) {
      rc = mfh_crs_expand_mm_share(c, d_crs_c8, r, nsl, (uint8_t *)c->batch_img);  // (queued behind the previous slab's launches on the caller's stream)
      if (rc) return rc;
      mfh_crs_set_resident_mm_share(c, (const uint8_t *)c->batch_img, r, nsl);
      HIP_TRY(c, hipMemsetAsync(B.SCZ, 0, B.nslots * 2048, c->stream));
      slot = 0;
      const uint32_t loS = (uint32_t)((uint64_t)d * r / nsl);
      for (uint32_t s0 = 0, sgi = 0; s0 < nproofs; s0 += SG, sgi++) {
        const uint32_t sg = std::min(SG, nproofs - s0);
        uint32_t *WALL, *HALL, *VALL;
        whv_of(sgi, WALL, HALL, VALL);
        const BatchCoef co = {WALL + loS, HALL + loS, VALL + loS, d};
        rc = batch_rows_supergroup(c, d_crs_c8, r, nsl, sg, h_witness_bits + (size_t)s0 * bits_stride, bits_stride, co, d_proofs + (size_t)s0 * 5 * ctl, B, slot,
                                   r == 0 ? h_delta + s0 : nullptr, r == 0 ? c->ev_cdone[sgi % nbuf] : nullptr, r > 0);
        if (rc) return rc;
      }
    }
    for (uint32_t s0 = 0; s0 < nproofs; s0 += SG) {
      rc = batch_smudge(c, d_proofs + (size_t)s0 * 5 * ctl, std::min(SG, nproofs - s0), h_smudge_mag + (size_t)s0 * 5 * maglen, maglen, h_smudge_sign + (size_t)s0 * 5);
      if (rc) return rc;
      HIP_TRY(c, hipEventRecord(c->ev_sgdone[s0 / SG], c->stream));  // (every slab touches every proof: all super-groups complete together, at the end)
    }
    c->last_batch_sg = SG;
    c->last_batch_n = nproofs;
    return MFH_OK;
  }
  ImageGuard transient{c, false};
  rc = batch_transient_image(c, d_crs_c8, nproofs, 0, 1, transient);
  if (rc) return rc;
  // everything the host contributes, in one copy (no host-side wait between the super-groups), staged while the GPU expands the CRS
  rc = batch_stage_host(c, B, nproofs, h_witness_bits, bits_stride, h_delta, h_smudge_mag, maglen, h_smudge_sign, SG);
  if (rc) return rc;
  rc = batch_ct_t(c, d_crs_c8, B);
  if (rc) return rc;
  // (b_w needs nothing of the chain and nothing needs b_w before the call ends, so all b_w launches of a call -- HBM-bound passes over the
  // BT+BV image -- could run in the background under the matrix-core bound S / AS launches.  Built and measured: on an unrestricted side
  // stream the call takes the same 81.5 ms, on a stream masked to 64 / 32 / 16 CUs 88.5 / 98.1 / 116.3 ms -- a k_mmstream workgroup pulls
  // ~18 GB/s whether 16 or 256 CUs stream, so fewer CUs only stretch the pass.  b_w stays in line.)
  // b_w of ALL super-groups of the call in one streaming launch per (up to) 8 of them: b_w needs the witness bits only (staged above), so it does not have to wait
  // for its super-group's turn, and 4 - 8 groups over the BT+BV image share its fragments in the XCDs' L2s like the S / AS groups of a super-group do (one group
  // per launch is HBM-bound at 3.4 TB/s: 0.9 ms per super-group).  mfh_set_batch_bw(ctx, 0) restores one launch per super-group.
  bool bw_done = false;
  if (c->batch_bw_merged && nsg > 1 && mm_image_covers(c, (uint64_t)ctb0 * n * 2 * d, m)) {
    const uint64_t ctr_ct = (uint64_t)ctb0 * n, pstride = 5 * ctl;
    const uint32_t bstride = (m + 6) / 8;
    const MmRegion reg = {ctr_ct * 2 * d, d_crs_c8 + (size_t)2 * d * ctb0};
    for (uint32_t g0 = 0; g0 < nsg; g0 += 8) {
      MmIo ios[8];
      uint32_t nvs[8];
      const uint32_t ng = std::min(8u, nsg - g0);
      for (uint32_t k = 0; k < ng; k++) {
        const uint32_t sgi = g0 + k, s0 = sgi * SG, sg = std::min(SG, nproofs - s0);
        const uint8_t *d_cw = B.CW + (size_t)sgi * B.cw_stride;
        ios[k] = MmIo{{nullptr, nullptr}, sg, {d_proofs + (size_t)s0 * 5 * ctl + 4 * ctl, nullptr}, sg, pstride, d_cw, bstride, B.SCZ + 256 * slot++};
        ios[k].add_ct = B.CT_T;  // + delta_b ct_t (src/snark.c:143-145) in the epilogue
        ios[k].add_scale = (const uint32_t *)(d_cw + B.cw_delta_off);
        nvs[k] = sg;
      }
      rc = ng > 1 ? eval_rows_multi_io_regions(c, &reg, 1, m, ios, nvs, ng, 1, 0) : eval_rows_multi_io(c, reg.off, m, reg.c8, ios[0], nvs[0], 1, 0);
      if (rc) return rc;
    }
    bw_done = true;
  }
  // early mode (mfh_set_mm_width(ctx, w < 32, 1): the persistent streaming launch leaves CUs of every XCD free): the chain of super-group k + 1 is queued BESIDE the
  // row work of super-group k (its w | h | v area was last read by k - 1) and the epilogues + smudging of k beside the row work of k + 1, instead of between them
  const bool early = c->batch_early_chain && bw_done;
  bool any_on_side = false;
  for (uint32_t s0 = 0, sgi = 0; s0 < nproofs; s0 += SG, sgi++) {
    const uint32_t sg = std::min(SG, nproofs - s0);
    uint64_t *sproofs = d_proofs + (size_t)s0 * 5 * ctl;
    uint32_t *WALL, *HALL, *VALL;
    whv_of(sgi, WALL, HALL, VALL);
    const BatchCoef co = {WALL, HALL, VALL, d};
    if (early && sgi + 1 < nsg) {
      if (sgi >= 1) HIP_TRY(c, hipStreamWaitEvent(chain_stream, c->ev_rdone[(sgi - 1) % nbuf], 0));
      rc = launch_chain(sgi + 1);
      if (rc) return rc;
    }
    // early mode: this super-group's digit kernels and streaming launches write the half of the scratch that super-group sgi - 2 used; its epilogues read that half
    // on the side stream and its completion event is recorded behind them
    if (early && sgi >= 2) HIP_TRY(c, hipStreamWaitEvent(main_stream, c->ev_sgdone[sgi - 2], 0));
    // b_w's pass over the BT+BV image is HBM-bound like the chain's witness pass, and the chain is what the S / AS launches wait for: b_w
    // starts when the witness pass is over and runs beside the polynomial step (NTT: VALU / LDS)
    HIP_TRY(c, hipStreamWaitEvent(main_stream, c->ev_wdone[sgi % nbuf], 0));
    // the multi-vector launches read their coefficient vectors where the polynomial step left them and write the proof structs in place (MmIo)
    bool on_side = false;
    rc = batch_rows_supergroup(c, d_crs_c8, 0, 1, sg, h_witness_bits + (size_t)s0 * bits_stride, bits_stride, co, sproofs, B, slot, h_delta + s0,
                               c->ev_cdone[sgi % nbuf], 0, B.CW + (size_t)sgi * B.cw_stride, bw_done, sgi, early ? &on_side : nullptr);
    if (rc) return rc;
    if (early) HIP_TRY(c, hipEventRecord(c->ev_rdone[sgi % nbuf], main_stream));  // (the digit kernels were the last readers of the area)
    else if (sgi + 1 < nsg) {  // the next super-group's chain, into the other area (last read by super-group sgi - 1)
      HIP_TRY(c, hipEventRecord(c->ev_rdone[sgi % nbuf], main_stream));
      HIP_TRY(c, hipStreamWaitEvent(chain_stream, c->ev_rdone[sgi % nbuf], 0));
      rc = launch_chain(sgi + 1);
      if (rc) return rc;
    }
    {
      OnStream tail(c, on_side ? c->side : main_stream);
      rc = batch_smudge_staged(c, B, sproofs, s0, sg);
      if (rc) return rc;
      HIP_TRY(c, hipEventRecord(c->ev_sgdone[sgi], c->stream));  // this super-group's proofs are final (b_w was written before the loop or by batch_rows_supergroup)
    }
    any_on_side = any_on_side || on_side;
  }
  if (any_on_side) {  // the caller's stream ends behind everything the call queued
    HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
    HIP_TRY(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
  }
  c->last_batch_sg = SG;
  c->last_batch_n = nproofs;
  return MFH_OK;
}

uint32_t mfh_prove_batch_supergroup(const mfh_ctx *c) { return c ? c->last_batch_sg : 0; }

int mfh_prove_batch_stream_wait(mfh_ctx *c, uint32_t upto, void *hip_stream) {
  if (!c) return MFH_EINVAL;
  if (!c->last_batch_sg || !upto || upto > c->last_batch_n) { c->err = "mfh_prove_batch_stream_wait: no such statements in the last mfh_prove_batch call"; return MFH_EINVAL; }
  HIP_TRY(c, hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_sgdone[(upto - 1) / c->last_batch_sg], 0));
  return MFH_OK;
}

// ---- the row-sharded batch prover (BASELINE configs 3/4): see include/mfhip.h ------------------------------------------
int mfh_batch_chain(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_witness_bits, size_t bits_stride, const uint32_t *h_delta,
                    uint32_t *d_w, uint32_t *d_h, uint32_t *d_v) {
  if (!c) return MFH_EINVAL;
  if (!nstmt) return MFH_OK;
  if (!h_witness_bits || !h_delta || !d_w || !d_h || !d_v) return MFH_EINVAL;
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  const uint32_t d = c->P.d, m = c->P.m;
  if (bits_stride < (m + 6) / 8) { c->err = "bits_stride shorter than the m - 1 witness bits"; return MFH_EINVAL; }
  for (uint32_t b = 0; b < nstmt; b++)
    if (h_delta[b] >= P32) { c->err = "delta must be < p"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  for (uint32_t s0 = 0; s0 < nstmt; s0 += BSG) {
    const size_t o = (size_t)s0 * d;
    rc = batch_chain_launch(c, src, d_ssp, std::min(BSG, nstmt - s0), h_witness_bits + (size_t)s0 * bits_stride, bits_stride, h_delta + s0, d_w + o, d_h + o,
                            d_v + o);
    if (rc) return rc;
  }
  return MFH_OK;
}

// The chain cut in two for the row-sharded prover of a generator-defined (or very large) SSP, where the witness pass IS the chain's cost
// and shards by COEFFICIENT RANGE without any reduction: every rank computes the coefficients [col0, col0 + ncols) of w of ALL statements
// (1 / world of the rows' generation or read), the slices are exchanged (all-to-all), and the owner of a statement finishes its chain.
int mfh_batch_witness_cols(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_witness_bits, size_t bits_stride, const uint32_t *h_delta,
                           uint32_t col0, uint32_t ncols, uint32_t *d_w, size_t w_stride) {
  if (!c) return MFH_EINVAL;
  if (!nstmt || !ncols) return MFH_OK;
  if (!h_witness_bits || !h_delta || !d_w) return MFH_EINVAL;
  if (bits_stride < (c->P.m + 6) / 8) { c->err = "bits_stride shorter than the m - 1 witness bits"; return MFH_EINVAL; }
  for (uint32_t s0 = 0; s0 < nstmt; s0 += BSG) {
    int rc = mfh_witness_poly_mm_cols(c, d_ssp, std::min(BSG, nstmt - s0), h_witness_bits + (size_t)s0 * bits_stride, bits_stride, h_delta + s0, col0, ncols,
                                      d_w + (size_t)s0 * w_stride, w_stride);
    if (rc) return rc;
  }
  return MFH_OK;
}
// d_v = d_w + v_0, d_h = (d_v^2 - 1) / t for nstmt statements whose whole w polynomials are in d_w (src/snark.c:161-169)
int mfh_batch_chain_from_w(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint32_t *d_w, uint32_t *d_h, uint32_t *d_v) {
  if (!c) return MFH_EINVAL;
  if (!nstmt) return MFH_OK;
  if (!d_w || !d_h || !d_v) return MFH_EINVAL;
  mf::SspSrc src;
  int rc = ssp_src(c, d_ssp, src);
  if (rc) return rc;
  const uint32_t d = c->P.d;
  HIP_TRY(c, hipSetDevice(c->device));
  for (uint32_t s0 = 0; s0 < nstmt; s0 += BSG) {
    const uint32_t sg = std::min(BSG, nstmt - s0);
    const size_t o = (size_t)s0 * d;
    hipLaunchKernelGGL(k_add_slot_multi, dim3((d + 255) / 256, sg), dim3(256), 0, c->stream, d_w + o, src, 1u, d, d_v + o);
    HIP_TRY(c, hipGetLastError());
    rc = mfh_poly_h_multi(c, d_v + o, d_h + o, sg);
    if (rc) return rc;
  }
  return MFH_OK;
}

int mfh_prove_batch_partial(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, uint32_t nstmt, const uint8_t *h_witness_bits,
                            size_t bits_stride, const uint32_t *d_w, const uint32_t *d_h, const uint32_t *d_v, size_t coef_stride, uint64_t *d_partial) {
  if (!c || !world || rank >= world) return MFH_EINVAL;
  if (!nstmt) return MFH_OK;
  if (!d_crs_c8 || !h_witness_bits || !d_w || !d_h || !d_v || !d_partial) return MFH_EINVAL;
  if (c->resident_rows) { c->err = "mfh_prove_batch_partial: clear the single-proof resident CRS image first"; return MFH_EUNSUPPORTED; }
  if (c->mm_image && (c->mm_rank != rank || c->mm_world != world)) {
    c->err = "the registered matrix-core image holds the row shares of a different (rank, world)";
    return MFH_EINVAL;
  }
  const uint32_t d = c->P.d, m = c->P.m, n = c->P.n;
  const uint32_t cS = (uint32_t)((uint64_t)d * (rank + 1) / world) - (uint32_t)((uint64_t)d * rank / world);
  if (bits_stride < (m + 6) / 8) { c->err = "bits_stride shorter than the m - 1 witness bits"; return MFH_EINVAL; }
  if (coef_stride < cS) { c->err = "coef_stride shorter than the rank's share of the S / AS rows"; return MFH_EINVAL; }
  const size_t ctl = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
  HIP_TRY(c, hipSetDevice(c->device));
  BatchScratch B;
  int rc = batch_scratch(c, nstmt, 0, B);
  if (rc) return rc;
  size_t slot = 0;
  HIP_TRY(c, hipMemsetAsync(B.SCZ, 0, B.nslots * 2048, c->stream));
  rc = batch_streams(c);
  if (rc) return rc;
  ImageGuard transient{c, false};
  rc = batch_transient_image(c, d_crs_c8, nstmt, rank, world, transient);
  if (rc) return rc;
  for (uint32_t s0 = 0; s0 < nstmt; s0 += BSG) {
    const uint32_t sg = std::min(BSG, nstmt - s0);
    const uint64_t o = (uint64_t)s0 * coef_stride;
    const BatchCoef co = {d_w + o, d_h + o, d_v + o, coef_stride};
    rc = batch_rows_supergroup(c, d_crs_c8, rank, world, sg, h_witness_bits + (size_t)s0 * bits_stride, bits_stride, co, d_partial + (size_t)s0 * 5 * ctl, B,
                               slot, nullptr, nullptr);
    if (rc) return rc;
  }
  return MFH_OK;
}

int mfh_prove_batch_finish(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t nstmt, const uint32_t *h_delta, const uint8_t *h_smudge_mag, size_t maglen,
                           const uint8_t *h_smudge_sign, uint64_t *d_proofs) {
  if (!c) return MFH_EINVAL;
  if (!nstmt) return MFH_OK;
  if (!d_crs_c8 || !h_delta || !h_smudge_mag || !h_smudge_sign || !d_proofs) return MFH_EINVAL;
  for (uint32_t b = 0; b < nstmt; b++)
    if (h_delta[b] >= P32) { c->err = "delta must be < p"; return MFH_EINVAL; }
  if (maglen + 4 > (size_t)(c->P.logq / 64) * 8) { c->err = "smudge magnitude too wide"; return MFH_EINVAL; }  // before delta ct_t is added
  const uint32_t n = c->P.n, KW = 2 * (c->P.logq / 64);
  const size_t ctl = (size_t)(n + 1) * ((c->P.logq + 63) / 64);
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t cap = std::min(nstmt, BSG);
  BatchScratch B;
  int rc = batch_scratch(c, cap, 0, B, 1, cap);
  if (rc) return rc;
  rc = batch_ct_t(c, d_crs_c8, B);
  if (rc) return rc;
  // Everything the host contributes to the call -- per chunk of up to BSG statements the deltas, the smudging terms u p of all five draws (schoolbook by 32-bit words,
  // src/lwe.c:65-76) and their signs -- is staged in ONE pinned buffer taken once, so the call queues its copies and kernels and returns: the host does not wait for
  // the GPU to reach this call between two smudging passes (it did until round 6, which held a pipelined caller -- mfuoco_prover_batch_sharded -- to the GPU's pace).
  // (pin_smudge, not the pin_cw the row work stages its witness bits in: a caller that queues finish(k) behind rows(k + 1) would otherwise wait in rows(k + 2) for finish(k)'s copy)
  const size_t per = 4 + (size_t)5 * KW * 4 + 5, total = (size_t)nstmt * per + 4 * (((size_t)nstmt + BSG - 1) / BSG);
  uint8_t *st = (uint8_t *)pin_acquire(c, c->pin_smudge, total);
  if (!st) return MFH_ENOMEM;
  size_t at = 0;
  for (uint32_t s0 = 0; s0 < nstmt; s0 += BSG) {
    const uint32_t sg = std::min(BSG, nstmt - s0);
    uint64_t *sproofs = d_proofs + (size_t)s0 * 5 * ctl;
    uint32_t *h_dl = (uint32_t *)(st + at), *up = h_dl + sg;
    uint8_t *sn = (uint8_t *)(up + (size_t)sg * 5 * KW);
    memcpy(h_dl, h_delta + s0, (size_t)sg * 4);
    for (size_t i = 0; i < (size_t)sg * 5; i++) {
      const uint8_t *mag = h_smudge_mag + ((size_t)s0 * 5 + i) * maglen;
      uint64_t carry = 0;
      for (uint32_t l = 0; l < KW; l++) {
        uint32_t w = 0;
        const size_t o = (size_t)l * 4;
        if (o < maglen) memcpy(&w, mag + o, std::min<size_t>(4, maglen - o));
        const uint64_t t = (uint64_t)w * P32 + carry;
        up[i * KW + l] = (uint32_t)t;
        carry = t >> 32;
      }
    }
    memcpy(sn, h_smudge_sign + (size_t)s0 * 5, (size_t)sg * 5);
    at += ((size_t)sg * per + 3) & ~(size_t)3;
    HIP_TRY(c, hipMemcpyAsync(B.CW, h_dl, (size_t)sg * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(B.SMU, up, (size_t)sg * 5 * KW * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(B.SMS, sn, (size_t)sg * 5, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_bw_add_delta_ct, dim3((n + 1 + 255) / 256, sg), dim3(256), 0, c->stream, sproofs, B.CT_T, (const uint32_t *)B.CW, n + 1,
                       (c->P.logq + 63) / 64, 2 * (c->P.logq / 64));
    HIP_TRY(c, hipGetLastError());
    rc = batch_smudge_staged(c, B, sproofs, 0, sg);
    if (rc) break;
  }
  pin_release(c, c->pin_smudge);
  return rc;
}

}  // extern "C"
