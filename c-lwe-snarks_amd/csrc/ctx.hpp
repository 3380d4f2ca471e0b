// ctx.hpp -- internal: the context object behind include/mfhip.h and helpers shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "aes_dev.hpp"
#include "mfhip.h"
#include "ssp_prg.hpp"

#define HIP_TRY(ctx, expr)                                                        \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);             \
      return MFH_EDEVICE;                                                         \
    }                                                                             \
  } while (0)

struct PolyState;

// pinned host staging buffer whose previous async copy is awaited only when the buffer is reused
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;
  size_t used = 0;  // bytes the last user asked for: what pin_scrub zeroes
};

struct mfh_ctx {
  mfh_params P{};
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  mf::AesKey key{};
  bool have_seed = false;
  uint32_t *d_t0 = nullptr;  // 256 words
  void *ws = nullptr;        // scratch (partials etc.)
  size_t ws_bytes = 0;
  void *ws2 = nullptr;       // second scratch of mfh_eval_rows_multi: the launch in flight on the side stream (mfh_prove_batch)
  size_t ws2_bytes = 0;
  int mm_ws_sel = 0;         // 0: ws, 1: ws2
  void *ws3 = nullptr;       // mfh_prove_batch, streaming regime: digit fragments and partial products of all rounds of a super-group
  size_t ws3_bytes = 0;
  std::vector<hipEvent_t> ev_cdone, ev_rdone, ev_wdone;  // mfh_prove_batch: chain of super-group k done / its w | h | v area read / its witness pass done (per area)
  std::vector<hipEvent_t> ev_round;  // one per round: its streaming launch has finished
  std::vector<hipEvent_t> ev_sgdone;  // mfh_prove_batch: super-group k's proofs are final in d_proofs (mfh_prove_batch_stream_wait)
  uint32_t last_batch_sg = 0, last_batch_n = 0;  // super-group size and statement count of the last mfh_prove_batch call
  void *wws = nullptr;       // scratch of the witness pass (its own buffer: the pass may run beside an eval launch that owns `ws`)
  size_t wws_bytes = 0;
  // lazy-carry image (one u64 per accumulator word and coordinate) + active-row counter of the eval launches.  Invariant: all
  // zero between launches (k_eval_reduce_carry clears what it reads), so no memset is queued per evaluation.
  uint64_t *lazy = nullptr;
  uint32_t *lazy_cnt = nullptr;
  size_t lazy_bytes = 0;
  bool eval_dense = false;  // caller's hint: the coefficient vectors have (practically) no zero entry, skip the row compaction
  // prover overlap: witness pass + polynomial step on `side` while b_w's rows are evaluated on `stream` (snark.hip)
  bool overlap = true;
  int overlap_mode = 1;  // 1 = pick the queueing order by the size of b_w's share, 2 = b_w first, 3 = chain first (mfh_set_overlap)
  hipStream_t side = nullptr, side2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_chain = nullptr, ev_chain_done = nullptr;  // mfh_prove_batch: witness pass + polynomial step on `side`
  std::string err;
  // kernel timing (bench.py's roofline leg): HIP events recorded on the launch stream, resolved lazily
  bool timing = false;
  struct Timed {
    hipEvent_t e0, e1;
    int kind;        // 0 keystream, 1/2 eval (1/2 coeff vectors), 3 encrypt, 4 expand, 5/6 resident MAC (1/2 vectors)
    uint64_t rows;   // rows handed to the launch
    uint64_t work;   // rows x evaluations the launch serves (k_mmstream with several groups; = rows elsewhere)
    int sub;         // 1: the launch ran the persistent one-workgroup-per-CU grid (k_mmstream_p / _pb / _w); 0 otherwise
  };
  std::vector<Timed> timed;
  double last_busy_ms = 0;  // union of the spans of the launches the last mfh_timing_drain matched
  uint64_t last_work_rows = 0;
  std::vector<hipEvent_t> ev_pool;
  PolyState *poly = nullptr;  // NTT tables and per-SSP precomputation (poly.hip)
  void *aux = nullptr;        // small scratch that must survive an eval/encrypt launch (snark.hip)
  size_t aux_bytes = 0;
  uint32_t *d_msg = nullptr;  // setup messages
  size_t msg_rows = 0;
  uint32_t *d_prover = nullptr;  // prover polynomials w, v, h and the b_w coefficient vector
  size_t prover_words = 0;
  std::vector<uint32_t> h_cw;
  const uint8_t *mm_image = nullptr;  // CRS expanded for the matrix-core path (mfh_crs_expand_mm): S | AS | BT+BV regions
  uint32_t mm_rank = 0, mm_world = 1;  // whose row shares the image holds (mfh_crs_set_resident_mm_share)
  uint64_t mm_off[3] = {0, 0, 0}, mm_rows[3] = {0, 0, 0};
  size_t mm_base[3] = {0, 0, 0};
  void *ssp_frag = nullptr;  // the dense SSP in MFMA B-fragment order (evalmm.hip: witness pass of the batch prover); built lazily
  size_t ssp_frag_bytes = 0;
  const uint32_t *ssp_frag_src = nullptr;  // the d_ssp it was built from; mfh_ssp_prepare / mfh_ssp_upload reset it
  void *d_batch = nullptr;  // mfh_prove_batch group scratch: W | H | V | CW | ONE | CT_T
  size_t batch_bytes = 0;
  // mfh_prove_batch, more than one group of proofs and no image registered: the CRS is expanded ONCE PER CALL into this scratch in
  // MFMA A-fragment order and streamed for every group (k_mmstream) instead of running AES again per group (mfh_set_batch_image)
  void *batch_img = nullptr;
  size_t batch_img_bytes = 0;
  int batch_image = 1;
  // k_mmstream launches with several groups (mfh_set_mm_stream): slot -> (group, tile group) map, persistent grid, rendezvous of the sharers
  int mm_map = 1;
  bool mm_persist = true;
  bool mm_wave1 = false;  // persistent grid with one wave per SIMD and 256 accumulators in AccVGPRs (k_mmstream_w)
  uint32_t mm_sync_mode = 0, mm_spin = 64;
  uint32_t mm_width = 32;  // workgroups per XCD of the persistent S / AS launch (mfh_set_mm_width): 32 = every CU
  size_t early_ws_half = 0;        // early chain: distance of the two halves of ws3 = the scratch of the call's first (largest) super-group
  bool batch_early_chain = false;  // mfh_prove_batch: chain of super-group k + 1 and epilogue of k queued BESIDE the streaming launch of k / k + 1 (for the CUs a narrower grid leaves free)
  uint32_t *mm_sync = nullptr;  // 8 x 32 counters of the persistent grid's rendezvous
  int poly_exact = 1;  // batches of h = (v^2 - 1) / t try the exact-division path first: 0 never, 1 unless recent batches failed its check, 2 always (poly.hip; mfh_set_poly_exact)
  bool mm_pack = true;  // the streaming kernels hand their partial products to the epilogue recombined (evalmm.hip: mms_store_packed); mfh_set_mm_pack
  uint32_t mm_chunk_rows = 131071;  // rows per row chunk of the matrix-core launches (int32 accumulators: |A'C'| <= 2^14 per row)
  int expand_path = 0;     // mfh_crs_expand_mm*: 0 = k_expand_mm (lane = row, MFMA transposition, no LDS tile), 1 = k_evalmm16<MODE 1> (LDS tile + byte gathers)
  int enc_path = 0;        // mfh_encrypt_rows: 0 = pick by batch size, 1 = VALU kernel (k_encrypt), 2 = matrix-core kernel (k_encrypt_mm)
  uint32_t ncu = 256;        // compute units of the device (launch sizing)
  int dec_path = 0;          // mfh_decrypt: 0 = by batch size, 1 = VALU kernel (k_decrypt), 2 = matrix-core kernel (k_decrypt_mm)
  int eval_path = 0;         // mfh_eval_rows / mfh_prove: 0 = tile kernel (k_eval), 1 = wave-autonomous kernel (k_eval_w, logq 736 only; measured 4 % slower)
  uint32_t enc_chunks = 0;   // k_encrypt_mm: 0 = column chunks per row picked from the batch size, n = forced (mfh_set_encrypt_chunks; tuning)
  uint32_t witness_per = 0;  // batch chain: statements per witness GEMM pass, 0 = one pass per super-group (mfh_set_witness_per; A/B knob)
  uint32_t batch_slabs = 0;  // mfh_prove_batch: 0 = row slabs only when the image does not fit HBM (count picked from free memory), n = always n slabs
  int batch_bw_merged = 1;  // b_w of all super-groups of a call in one streaming launch per 8 of them (mfh_set_batch_bw)
  int batch_merge = 1;     // the S and AS groups of a round in one streaming launch (0: two launches on two streams)
  uint32_t batch_ngl = 8;  // groups of 63 / 64 coefficient vectors per streaming launch and region (1..8; 8 = a super-group's S and AS regions in ONE launch)
  PinBuf pin_rows, pin_cw, pin_smudge;
  // the batch chain's witness staging (one per super-group of a call, evalmm.hip): a ring, so that queueing super-group k + 1 does not wait on the host for
  // super-group k's copy to have RUN (with one buffer mfh_prove_batch blocked its caller for half of the call's GPU time)
  PinBuf pin_wring[8];
  uint32_t pin_wnext = 0;
  hipEvent_t ev_sample = nullptr;  // ... recorded behind the last reader of sample_tmp: the next call (on whatever stream) waits for it before it overwrites the buffer
  void *sample_tmp = nullptr;  // mfh_sample_rows: the rows' raw stream bytes (small requests; kept so that the call neither allocates nor waits)
  size_t sample_bytes = 0;
  void *uploader = nullptr;  // mfh_ssp_upload: per-thread pinned / device staging pairs and streams (mfhip.hip), made on the first large upload
  // generator-defined SSP (ssp_prg.hpp): used by every entry point that is handed d_ssp == NULL
  bool prg_on = false;
  uint64_t prg_seed = 0;
  const uint32_t *prg_t = nullptr;
  const uint8_t *resident_rows = nullptr;  // expanded CRS (mfh_crs_expand layout) or null: regenerate the keystream
  uint64_t resident_nrows = ~0ull;         // full image: rows [0, resident_nrows) in stream order are resident, the rest is regenerated
  bool resident_sharded = false;           // image holds only rank res_rank's shares: S share | AS share | BT+BV share
  uint32_t res_rank = 0, res_world = 1;
};

struct Timer {  // brackets one launch with events when timing is on; never synchronises
  mfh_ctx *c;
  mfh_ctx::Timed t{};
  bool on;
  Timer(mfh_ctx *c_, int kind, uint64_t rows, uint64_t work = 0, int sub = 0) : c(c_), on(c_->timing) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
      else hipEventCreate(&e);
      return e;
    };
    t.e0 = get(); t.e1 = get(); t.kind = kind; t.rows = rows; t.work = work ? work : rows; t.sub = sub;
    hipEventRecord(t.e0, c->stream);
  }
  ~Timer() {
    if (!on) return;
    hipEventRecord(t.e1, c->stream);
    c->timed.push_back(t);
  }
};

inline void *pin_acquire(mfh_ctx *c, PinBuf &b, size_t bytes) {
  if (b.pending) { hipEventSynchronize(b.ev); b.pending = false; }
  if (bytes > b.cap) {
    if (b.p) hipHostFree(b.p);
    b.cap = (bytes + 4095) & ~(size_t)4095;
    if (hipHostMalloc(&b.p, b.cap, hipHostMallocDefault) != hipSuccess) { b.p = nullptr; b.cap = 0; c->err = "hipHostMalloc failed"; return nullptr; }
  }
  if (!b.ev) hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
  b.used = bytes;
  return b.p;
}
inline void pin_release(mfh_ctx *c, PinBuf &b) {
  hipEventRecord(b.ev, c->stream);
  b.pending = true;
}
// zero what the last user staged, once its copy has run (mfh_scrub_staging: witness bits, deltas and smudging terms do not outlive their call in pinned memory)
inline void pin_scrub(PinBuf &b) {
  if (b.pending) { hipEventSynchronize(b.ev); b.pending = false; }
  if (b.p && b.used) {
    volatile unsigned char *q = (volatile unsigned char *)b.p;  // (volatile: not elided as a dead store)
    memset((void *)q, 0, b.used);
    __asm__ __volatile__("" : : "r"(b.p) : "memory");
  }
  b.used = 0;
}
inline void pin_free(PinBuf &b) {
  if (b.pending) hipEventSynchronize(b.ev);
  if (b.p) hipHostFree(b.p);
  if (b.ev) hipEventDestroy(b.ev);
  b = PinBuf();
}

// resolves the d_ssp argument of an entry point: a dense image, or (NULL) the registered generator-defined SSP
inline int ssp_src(mfh_ctx *c, const uint32_t *d_ssp, mf::SspSrc &src) {
  if (d_ssp) { src = mf::SspSrc{d_ssp, d_ssp, 0}; return MFH_OK; }
  if (!c->prg_on || !c->prg_t) { c->err = "d_ssp is NULL and no generator-defined SSP is registered (mfh_ssp_set_prg)"; return MFH_EINVAL; }
  src = mf::SspSrc{nullptr, c->prg_t, c->prg_seed};
  return MFH_OK;
}

void mfh_poly_destroy(mfh_ctx *c);
extern "C" int mfh_witness_poly_mm(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                                   uint32_t *d_w);
extern "C" int mfh_witness_poly_mm_cols(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                                        uint32_t col0, uint32_t ncols, uint32_t *d_w, size_t w_stride);
extern "C" int mfh_poly_h_multi(mfh_ctx *c, const uint32_t *d_v, uint32_t *d_h, uint32_t nb);
int aux_reserve(mfh_ctx *c, size_t bytes);
// expandmm.hip: one region of the matrix-core CRS image (rows at stream offset off, their compressed ciphertexts c8), barrier-free writer
int expand_mm_region(mfh_ctx *c, uint64_t off, uint32_t nrows, const uint8_t *c8, uint8_t *image);
// encmm.hip: mfh_encrypt_rows with <sk, a> on the matrix cores (off and the row length multiples of 8)
int encrypt_rows_mm(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint32_t *msg, const uint64_t *err, uint8_t *c8);
// encmm.hip: regev_decrypt batches with <a, sk> on the matrix cores: full ciphertexts in HBM / seed-compressed ciphertexts (a regenerated)
int decrypt_mm(mfh_ctx *c, const uint64_t *sk, const uint64_t *cts, size_t count, uint32_t *out);
int decrypt_rows_mm(mfh_ctx *c, uint64_t off, size_t nrows, const uint64_t *sk, const uint8_t *c8, uint32_t *out);

// Operands of one multi-vector launch (evalmm.hip): coefficient vector v is coef[0] + v * nrows for v < csplit, else
// coef[1] + (v - csplit) * nrows; its result goes to out[0] + v * ostride for v < osplit, else out[1] + (v - osplit) * ostride
// (uint64 words).  mfh_prove_batch reads w | h | v where the polynomial step left them and writes the proof structs in place.
struct MmIo {
  const uint32_t *coef[2];
  uint32_t csplit;
  uint64_t *out[2];
  uint32_t osplit;
  uint64_t ostride;
  // bits != nullptr replaces coef: vector v is the packed witness bits of statement v (bits + v * bits_stride) as b_w's coefficients
  // over the BT+BV rows -- row 0 (BT) -> 0, row i -> bit i - 1 (src/snark.c:143-155); coeff_bytes = 1
  const uint8_t *bits;
  uint32_t bits_stride;
  // optional: 256 int64 column sums the CALLER has zeroed (mfh_prove_batch clears the slots of all its launches with one memset);
  // nullptr: a slot of the workspace, cleared by a memset in front of the digit kernel
  int64_t *sc_zeroed;
  // words between consecutive coefficient vectors of coef[0] / coef[1]; 0 = nrows (vectors back to back).  The row-sharded batch prover
  // hands over slices [statement][w | h | v][rows of the share] as they come out of the all-to-all.
  uint64_t cstride;
  // bits: the launch covers rows [bits_row0, bits_row0 + nrows) of the BT+BV region (a rank's share): local row i <-> bit bits_row0 + i - 1
  uint32_t bits_row0;
  // optional: out_v += scale[v] * ct (ct_addmul_ui, src/lwe.c:141-149) in the epilogue, before the carries are resolved -- b_w's
  // delta_b * ct_t term (src/snark.c:143-145); add_ct = one ciphertext (n + 1 values of L limbs), add_scale = one uint32 < p per vector
  const uint64_t *add_ct;
  const uint32_t *add_scale;
  // streaming launches of several groups (mms_*): 0 = the group carries its own ones column (sum_i A'[i][m], one of its 256 digit columns: at most 63
  // four-byte vectors); g + 1 = it borrows the ones column of group g of the same launch and region (the same rows: the same sums) and may hold 64 vectors
  uint32_t sa_from1;
};
int eval_rows_multi_io(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, const MmIo &io, uint32_t nvec, uint32_t coeff_bytes, int accumulate);
// ng of them over the same region in one streaming launch when the matrix-core image is registered (io.sc_zeroed required); else one by one
int eval_rows_multi_io_set(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, const MmIo *ios, const uint32_t *nvecs, uint32_t ng,
                           uint32_t coeff_bytes);
// the same over nreg <= 2 regions of equal row count in ONE streaming launch: group r * ng + k evaluates region r with ios[r * ng + k]
struct MmRegion { uint64_t off; const uint8_t *c8; };
int eval_rows_multi_io_regions(mfh_ctx *c, const MmRegion *regs, uint32_t nreg, size_t nrows, const MmIo *ios, const uint32_t *nvecs, uint32_t ng,
                               uint32_t coeff_bytes, int accumulate);
bool mm_image_covers(const mfh_ctx *c, uint64_t off, size_t nrows);
// the phases of such a launch (evalmm.hip), for callers that queue them on different streams
struct MmsPlan {
  uint32_t ng, ngt, ND, nrows, nchunks, rpc, rpad, ntiles, mtiles;
  size_t cd_bytes, part_bytes;
  const uint8_t *img[2];
  int8_t *cd;
  int *part;
};
bool mms_plan(mfh_ctx *c, const MmRegion *regs, uint32_t nreg, size_t nrows, const MmIo *ios, const uint32_t *nvecs, uint32_t ng, uint32_t coeff_bytes,
              MmsPlan &P);                       // false: the registered image does not serve this shape
size_t mms_ws_bytes(const MmsPlan &P);
void mms_bind(MmsPlan &P, void *ws);              // the launch's digit fragments and partial products live in ws (mms_ws_bytes)
int mms_digits(mfh_ctx *c, const MmsPlan &P, const MmIo *ios, const uint32_t *nvecs);
int mms_stream(mfh_ctx *c, const MmsPlan &P);
int mms_finish(mfh_ctx *c, const MmsPlan &P, const MmIo *ios, const uint32_t *nvecs, int accumulate);

inline int buf_reserve(mfh_ctx *c, void *&buf, size_t &have, size_t bytes) {
  if (bytes <= have) return MFH_OK;
  if (buf) {
    hipStreamSynchronize(c->stream);
    hipFree(buf);
    buf = nullptr;
    have = 0;
  }
  bytes = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
  if (hipMalloc(&buf, bytes) != hipSuccess) {
    c->err = "hipMalloc(workspace) failed";
    return MFH_ENOMEM;
  }
  have = bytes;
  return MFH_OK;
}
inline int lazy_reserve(mfh_ctx *c, size_t bytes) {
  if (bytes <= c->lazy_bytes) return MFH_OK;
  if (c->lazy) {
    hipStreamSynchronize(c->stream);
    hipFree(c->lazy);
    c->lazy = nullptr;
    c->lazy_bytes = 0;
  }
  bytes = (bytes + 255) & ~(size_t)255;
  if (hipMalloc((void **)&c->lazy, bytes + 256) != hipSuccess) {
    c->err = "hipMalloc(lazy image) failed";
    return MFH_ENOMEM;
  }
  if (hipMemsetAsync(c->lazy, 0, bytes + 256, c->stream) != hipSuccess) return MFH_EDEVICE;
  c->lazy_cnt = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(c->lazy) + bytes);
  c->lazy_bytes = bytes;
  return MFH_OK;
}
inline int ws_reserve(mfh_ctx *c, size_t bytes) { return buf_reserve(c, c->ws, c->ws_bytes, bytes); }
inline int wws_reserve(mfh_ctx *c, size_t bytes) { return buf_reserve(c, c->wws, c->wws_bytes, bytes); }
inline int ws2_reserve(mfh_ctx *c, size_t bytes) { return buf_reserve(c, c->ws2, c->ws2_bytes, bytes); }

