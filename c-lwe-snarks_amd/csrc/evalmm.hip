// evalmm.hip -- eval_poly (reference src/lwe.c:160-178) for MANY coefficient vectors over the same CRS rows: one expansion of the
// rows, the multiply-accumulate on the matrix cores.
//
// With one or two coefficient vectors (one proof) the MAC is GEMV-shaped and k_eval (mfhip.hip) does it on the VALU.  With V
// vectors -- the S / AS / BV regions of a batch of proofs under one CRS -- it is a GEMM over the rows:
//
//     out_v[j] = sum_i c_v[i] * a_ij   (mod 2^704)
//   a_ij = sum_u A[i][(j,u)] 256^u   (88 significant bytes of the 92-byte stream value)
//   c_v[i] = sum_w C[i][(v,w)] 256^w (four bytes)
//   G[(j,u)][(v,w)] = sum_i A[i][(j,u)] * C[i][(v,w)]           M = 1471*88, N = 4V + 1, K = rows
//   out_v[j] = sum_{u,w} G[(j,u)][(v,w)] 256^(u + w)
//
// v_mfma_i32_32x32x32_i8 is signed: both operands go in offset by 128.  The keystream bytes are produced as A' = A - 128 (0x80808080
// folded into the last AES round key, so the offset costs nothing), the digit matrix holds C' = C - 128, and
//     sum_i A C = sum_i A'C' + 128 sum_i A' + 128 sum_i C' + 16384 rows:
// sum_i A'[m] comes out of the same MFMA through one extra digit column of ones, sum_i C'[n] from a small kernel.  |A'C'| <= 16384: an
// int32 accumulator holds 131 071 rows; launches split the rows accordingly.  128 digit columns = 31 vectors + the ones column.
//
// Kernels in this file:
//   k_evalmm<NT>     128 digit columns (v_mfma_i32_32x32x32_i8, 4-coordinate column tiles): described next
//   k_evalmm16<0>    256 digit columns (v_mfma_i32_16x16x64_i8, 2-coordinate column tiles): 31 proofs' vector pairs per expansion
//   k_evalmm16<1>    expansion only: the rows written to HBM in MFMA A-fragment order (the resident image of the batch prover)
//   k_mmstream       the same GEMM streamed from that image: no AES, HBM / matrix-core bound
//   k_witness_mm     the witness pass of up to 128 statements as a GEMM of witness bits x SSP bytes (one read of the SSP)
//   k_mm_digits / k_evalmm_finish, k_ssp_frag / k_witness_bits / k_witness_mm_finish: operand preparation and epilogues
//
// k_evalmm: workgroup = 1024 threads, one column tile of CT = 4 coordinates (352 byte positions = 11 MFMA row tiles) x one row chunk.
// Per unit of RT = 128 rows: (1) all 16 waves expand the 128 x 368-byte row segments into a row-major LDS tile (23-24 AES blocks
// per row, the product AES of aes_dev.hpp); (2) waves 0..10 gather their A fragments from the tile (byte position m of 16
// consecutive rows: the row<->byte transposition every MFMA formulation of this product needs, done on the read side), load the
// B fragments (coefficient digits, 16 consecutive rows per lane, contiguous in the digit matrix) and issue NT MFMAs per 32 rows,
// while waves 11..15 compute the next unit's counter-mode span constants.  LDS: 64 KiB table (first, at address 0) + 48 KiB tile
// + 8 KiB span constants.
#include <algorithm>
#include <cstring>

#include "ctx.hpp"

namespace {

using mf::AesKey;

constexpr int CT = 4;             // coordinates per column tile
constexpr int SB = 88;            // significant bytes per value at logq = 736 (K = 11 limbs)
constexpr int VB = 92;            // stream bytes per value (CT_BYTES)
constexpr int MB = CT * SB;       // 352 byte positions per column tile
constexpr int MT = MB / 32;       // 11 MFMA row tiles
constexpr int RT = 128;           // rows per unit
constexpr int TSTRIDE = 384;      // tile row stride: 24 AES blocks
constexpr int BLK_PER_ROW = 24;
static_assert(MB % 32 == 0 && CT * VB + 15 <= TSTRIDE, "tile geometry");

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Digit matrix in MFMA B-fragment order: for k-step K = i / 32 (32 rows), column tile q and lane (r = n & 31, h): 16 bytes = digit
// n = 32 q + r of rows 32 K + 16 h + e, e = 0..15, at cd[((K * NT + q) * 64 + 32 h + r) * 16 + e].  A unit's (RT rows) fragments are one
// contiguous RT * N bytes.  Column n = ND v + w holds byte w of c_v[i] minus 128; column ND nvec is the ones column; every other
// column, and every row >= nrows, is zero (such rows and columns then add nothing to G').
__device__ __forceinline__ uint32_t coef_at(const MmIo &io, uint32_t v, uint32_t nrows, uint32_t i) {
  if (io.bits) {
    const uint32_t a = io.bits_row0 + i;
    return a ? (io.bits[(uint64_t)v * io.bits_stride + ((a - 1) >> 3)] >> ((a - 1) & 7)) & 1u : 0u;
  }
  const uint64_t cs = io.cstride ? io.cstride : nrows;
  return (v < io.csplit ? io.coef[0] + (uint64_t)v * cs : io.coef[1] + (uint64_t)(v - io.csplit) * cs)[i];
}
// block = the N digit columns (thread n) x RG consecutive groups of 16 rows: a thread builds whole 16-byte fragment elements (digit n of
// 16 consecutive rows: its vector's 16 coefficients are one 64-byte line, shared by the ND byte columns of that vector) and adds its
// column sum sc[n] = sum_i C'[i][n] (signed digits; sc zeroed by the caller) with one atomic per block.
constexpr int DG_RG = 8;   // row groups per block: 128 rows
__global__ void k_mm_digits(MmIo io, uint32_t nvec, uint32_t ND, uint32_t nrows, uint32_t rpad, uint32_t NT, int wide, int8_t *__restrict__ cd,
                            int64_t *__restrict__ sc) {
  const uint32_t n = threadIdx.x;  // blockDim.x = N
  const uint32_t v = n / ND, w = n % ND;
  const bool ones = n == ND * nvec;
  int colsum = 0;
  for (uint32_t rg = blockIdx.x * DG_RG; rg < (blockIdx.x + 1) * DG_RG && rg * 16 < rpad; rg++) {
    const uint32_t i0 = rg * 16;
    uint32_t pk[4] = {0, 0, 0, 0};
    if (io.bits && v < nvec) {
      // packed witness bits: rows i0 .. i0 + 15 are bits i0 - 1 .. i0 + 14 (row 0 -> 0): one 24-bit window instead of 16 byte loads
      const uint8_t *bp = io.bits + (uint64_t)v * io.bits_stride;
      const uint32_t a0 = io.bits_row0 + i0;  // row of the BT+BV region (the launch may cover a rank's share of it)
      const uint32_t first = a0 ? a0 - 1 : 0, b0 = first >> 3, nb = io.bits_stride;  // a statement's bit string is bits_stride bytes
      uint32_t win = 0;
#pragma unroll
      for (int k = 0; k < 3; k++)
        if (b0 + k < nb) win |= (uint32_t)bp[b0 + k] << (8 * k);
      win >>= first & 7;
      if (!a0) win <<= 1;  // row 0 is the BT row: coefficient 0
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int dgt = i0 + e < nrows ? (int)((win >> e) & 1u) - 128 : 0;
        colsum += dgt;
        pk[e >> 2] |= (uint32_t)(dgt & 255) << (8 * (e & 3));
      }
    } else if (v < nvec || ones) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int dgt = 0;
        if (i0 + e < nrows) dgt = ones ? 1 : (int)((coef_at(io, v, nrows, i0 + e) >> (8 * w)) & 255u) - 128;
        colsum += dgt;
        pk[e >> 2] |= (uint32_t)(dgt & 255) << (8 * (e & 3));
      }
    }
    uint64_t at;
    if (!wide) {  // v_mfma_i32_32x32x32_i8: k-step = 32 rows, lane = 32 h + r
      const uint32_t K = i0 >> 5, h = (i0 >> 4) & 1, q = n >> 5, r = n & 31;
      at = ((uint64_t)K * NT + q) * 64 + 32 * h + r;
    } else {      // v_mfma_i32_16x16x64_i8: k-step = 64 rows, lane = 16 g + c, NT column tiles of 16
      const uint32_t K = i0 >> 6, g = (i0 >> 4) & 3, q = n >> 4, c = n & 15;
      at = ((uint64_t)K * NT + q) * 64 + 16 * g + c;
    }
    reinterpret_cast<uint4 *>(cd)[at] = uint4{pk[0], pk[1], pk[2], pk[3]};
  }
  if (v < nvec && colsum) atomicAdd(reinterpret_cast<unsigned long long *>(sc + n), (unsigned long long)(long long)colsum);
}

// The hot shape of the batch prover -- 256 digit columns, four-byte coefficients (62 vectors + the ones column) -- with the 16
// coefficients of a fragment element fetched as four 16-byte loads, DG4 row groups in flight per thread and the byte column picked with
// v_perm: thread = (vector n >> 2, byte n & 3).
constexpr int DG4 = 4;
struct MmGroupArgs {  // the operands of up to 16 groups of one round (kernel argument: 1.6 KB)
  MmIo io[16];
  uint32_t nvec[16];
};
__device__ __forceinline__ void mm_digits4w_body(const MmIo &io, uint32_t nvec, uint32_t nrows, uint32_t rpad, int8_t *__restrict__ cd, int64_t *__restrict__ sc) {
  const uint32_t n = threadIdx.x, v = n >> 2, w = n & 3;
  const bool ones = n == 4 * nvec;
  const uint64_t cs = io.cstride ? io.cstride : nrows;
  const uint32_t *cv = v < nvec ? (v < io.csplit ? io.coef[0] + (uint64_t)v * cs : io.coef[1] + (uint64_t)(v - io.csplit) * cs) : nullptr;
  const uint32_t rg0 = blockIdx.x * DG4;
  uint32_t x[DG4][16];
#pragma unroll
  for (int r = 0; r < DG4; r++) {
    const uint32_t i0 = (rg0 + r) * 16;
#pragma unroll
    for (int e = 0; e < 16; e++) x[r][e] = 0;
    if (!cv || i0 >= nrows) continue;
    if (i0 + 16 <= nrows && ((reinterpret_cast<uintptr_t>(cv + i0) & 15) == 0)) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 t = reinterpret_cast<const uint4 *>(cv + i0)[k];
        x[r][4 * k] = t.x; x[r][4 * k + 1] = t.y; x[r][4 * k + 2] = t.z; x[r][4 * k + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; e++)
        if (i0 + e < nrows) x[r][e] = cv[i0 + e];
    }
  }
  const uint32_t sel = 0x0c0c0400u + 0x00000101u * w;  // {a.byte w, b.byte w, 0, 0}
  uint32_t bytesum = 0, nvalid = 0;
#pragma unroll
  for (int r = 0; r < DG4; r++) {
    const uint32_t i0 = (rg0 + r) * 16;
    if (i0 >= rpad) continue;
    uint32_t pk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t lo = __builtin_amdgcn_perm(x[r][4 * k + 1], x[r][4 * k], sel);
      const uint32_t hi = __builtin_amdgcn_perm(x[r][4 * k + 3], x[r][4 * k + 2], sel);
      uint32_t b4 = lo | (hi << 16);  // byte w of coefficients 4k .. 4k+3
      // rows at or past nrows: digit 0 (not -128); valid rows: digit - 128 = byte ^ 0x80
      uint32_t m = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) m |= (i0 + 4 * k + e < nrows ? 0xffu : 0u) << (8 * e);
      bytesum = __builtin_amdgcn_sad_u8(b4 & m, 0u, bytesum);
      nvalid += __builtin_popcount(m) >> 3;
      b4 = (b4 ^ 0x80808080u) & m;
      pk[k] = cv ? b4 : (ones ? 0x01010101u & m : 0u);
    }
    const uint32_t K = i0 >> 6, g = (i0 >> 4) & 3, q = n >> 4, c = n & 15;
    reinterpret_cast<uint4 *>(cd)[((uint64_t)K * 16 /* NQ2 column tiles */ + q) * 64 + 16 * g + c] = uint4{pk[0], pk[1], pk[2], pk[3]};
  }
  if (cv) {
    const long long colsum = (long long)bytesum - 128ll * nvalid;
    if (colsum) atomicAdd(reinterpret_cast<unsigned long long *>(sc + n), (unsigned long long)colsum);
  }
}
__global__ __launch_bounds__(256) void k_mm_digits4w(MmIo io, uint32_t nvec, uint32_t nrows, uint32_t rpad, int8_t *__restrict__ cd, int64_t *__restrict__ sc) {
  mm_digits4w_body(io, nvec, nrows, rpad, cd, sc);
}
// all groups of a round in one launch: blockIdx.y = group, its fragments at cd + group * cd_stride, its column sums in io.sc_zeroed
__global__ __launch_bounds__(256) void k_mm_digits4w_groups(MmGroupArgs A, uint32_t nrows, uint32_t rpad, int8_t *__restrict__ cd, uint64_t cd_stride) {
  const uint32_t g = blockIdx.y;
  mm_digits4w_body(A.io[g], A.nvec[g], nrows, rpad, cd + g * cd_stride, A.io[g].sc_zeroed);
}

struct RowGeom {
  uint64_t cb0;   // first AES block of the row's segment
  uint32_t head;  // byte offset of the segment inside that block
  uint32_t nblk;
};
__device__ __forceinline__ RowGeom row_geom(uint64_t off, uint64_t row, uint32_t n, uint32_t j0, uint32_t nks /* keystream-backed coords */,
                                            uint32_t vb = VB) {
  const uint64_t B0 = off + row * ((uint64_t)n * vb) + (uint64_t)j0 * vb;
  RowGeom g;
  g.cb0 = B0 >> 4;
  g.head = (uint32_t)(B0 & 15);
  g.nblk = nks ? (g.head + nks * vb + 15) >> 4 : 0;
  return g;
}

// grid = (column tiles, row chunks); NT = 32-column tiles of the digit matrix (N = 32 NT)
template <int NT>
__global__ __launch_bounds__(1024) void k_evalmm(AesKey key /* rk[56..59] ^ 0x80808080 */, const uint32_t *__restrict__ g_t0, uint64_t off,
                                                 uint32_t n, uint32_t nrows, uint32_t rows_per_chunk, const uint8_t *__restrict__ c8,
                                                 const int8_t *__restrict__ cd, uint32_t rpad, int *__restrict__ part) {
  struct __attribute__((aligned(16))) Lds {
    uint32_t lt[mf::kTabBytes / 4];      // first: LDS address 0 (aes_dev.hpp)
    uint8_t tile[RT * TSTRIDE];
    uint32_t spanc[RT][2][8];            // per row: the rounds-1-2 constants of the (at most two) 256-counter spans its segment touches
    v4i bfrag[RT / 32][NT][64];          // the unit's coefficient-digit fragments, in the order the MFMA lanes read them
  };
  __shared__ Lds lds;
  mf::lds_fill_tab(lds.lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lds.lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t j0 = blockIdx.x * CT;
  const uint32_t nks = j0 >= n ? 0u : min((uint32_t)CT, n - j0);  // keystream-backed coordinates of this tile (the last tile also holds b)
  const bool has_b = j0 + CT > n && j0 <= n;                      // coordinate n = the row's b, read from the compressed CRS
  const uint32_t r0 = blockIdx.y * rows_per_chunk;
  const uint32_t r1 = min(nrows, r0 + rows_per_chunk);

  v16i acc[NT];
#pragma unroll
  for (int q = 0; q < NT; q++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[q][e] = 0;

  // per-lane column of this wave's MFMA row tile inside a row segment (without the row's head offset)
  const uint32_t r32 = lane & 31, h = lane >> 5;
  const uint32_t m = wave * 32 + r32;            // byte position, < MB for waves < MT
  const uint32_t mcol = (m / SB) * VB + (m % SB);

  auto span_task = [&](uint32_t u0, uint32_t task) {  // task = 2 * local row + which span
    const uint32_t lr = task >> 1, which = task & 1;
    const uint64_t row = (uint64_t)u0 + lr;
    if (row >= r1 || !nks) return;
    const RowGeom g = row_geom(off, row, n, j0, nks);
    const uint64_t sp0 = g.cb0 >> 8, sp1 = (g.cb0 + g.nblk - 1) >> 8;
    if (which && sp1 == sp0) return;
    uint32_t sc[5];
    mf::aes_span_consts(tab, L, key, sp0 + which, sc);
#pragma unroll
    for (int i = 0; i < 5; i++) lds.spanc[lr][which][i] = sc[i];
  };

  __syncthreads();
  if (tid < 2 * RT) span_task(r0, tid);
  __syncthreads();
  for (uint32_t u0 = r0; u0 < r1; u0 += RT) {
    // the unit's digit fragments: RT * N contiguous bytes, one 16-byte load per thread, in flight under the expansion
    v4i bstage = {0, 0, 0, 0};
    const bool bload = tid < (RT / 32) * NT * 64;
    if (bload) bstage = *reinterpret_cast<const v4i *>(cd + ((uint64_t)(u0 >> 5) * NT * 64 + tid) * 16);
    // ---- (1) expansion: block slot s -> (local row s / 24, block s % 24)
    for (uint32_t s = tid; s < RT * BLK_PER_ROW; s += 1024) {
      const uint32_t lr = s / BLK_PER_ROW, k = s % BLK_PER_ROW;
      const uint64_t row = (uint64_t)u0 + lr;
      if (row >= r1) continue;
      const RowGeom g = row_geom(off, row, n, j0, nks);
      if (k >= g.nblk) continue;
      const uint64_t ctr = g.cb0 + k;
      const uint32_t *scp = lds.spanc[lr][(uint32_t)((ctr >> 8) - (g.cb0 >> 8))];
      uint32_t sc[5] = {scp[0], scp[1], scp[2], scp[3], scp[4]};
      uint32_t w[4];
      mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
      *reinterpret_cast<uint4 *>(&lds.tile[lr * TSTRIDE + 16 * k]) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (bload) (&lds.bfrag[0][0][0])[tid] = bstage;
    __syncthreads();
    if (has_b) {  // b of each row: CT_BYTES from the compressed CRS, offset by 128 like the keystream, behind the keystream coordinates
      for (uint32_t s = tid; s < RT * VB; s += 1024) {
        const uint32_t lr = s / VB, k = s % VB;
        const uint64_t row = (uint64_t)u0 + lr;
        if (row >= r1) continue;
        const RowGeom g = row_geom(off, row, n, j0, nks);
        lds.tile[lr * TSTRIDE + g.head + nks * VB + k] = (uint8_t)(c8[row * VB + k] ^ 0x80);
      }
      __syncthreads();
    }
    // ---- (2) waves 0..MT-1: MFMA over the unit's rows; waves MT..15: span constants of the next unit
#ifndef MM_SKIP_MFMA
    if (wave < MT) {
      const uint32_t head0 = row_geom(off, u0, n, j0, 1).head, hstep = (n * VB) & 15;  // head of local row lr = (head0 + hstep lr) & 15
#pragma unroll 1
      for (int ks = 0; ks < RT / 32; ks++) {
        if ((uint64_t)u0 + ks * 32 >= r1) break;  // whole k-step beyond the chunk (wave-uniform)
        const uint32_t lrb = ks * 32 + 16 * h;
        // A fragment: byte position m of 16 consecutive rows.  Rows at or beyond nrows hold stale bytes; their digits are zero.
        uint32_t aw[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) {
          uint32_t x = 0;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const uint32_t lr = lrb + 4 * e4 + e;
            x |= (uint32_t)lds.tile[lr * TSTRIDE + ((head0 + hstep * lr) & 15) + mcol] << (8 * e);
          }
          aw[e4] = x;
        }
        const v4i a = {(int)aw[0], (int)aw[1], (int)aw[2], (int)aw[3]};
#pragma unroll
        for (int q = 0; q < NT; q++) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, lds.bfrag[ks][q][lane], acc[q], 0, 0, 0);
      }
    } else
#endif
    if (wave >= MT) {
      const uint32_t t2 = tid - MT * 64;  // 0..319 >= 2 RT
      if (t2 < 2 * RT) span_task(u0 + RT, t2);
    }
    __syncthreads();
  }
  // ---- partial G of this workgroup: part[((chunk * ntiles + tile) * MB + m) * N + n]
  if (wave < MT) {
    const uint32_t N = 32 * NT;
    int *p = part + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * MB * N;
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const uint32_t mm = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        p[(uint64_t)mm * N + 32 * q + r32] = acc[q][e];
      }
  }
}

// ---- the wide variant: 256 digit columns (63 four-byte vectors = 31 proofs' pairs per expansion) ---------------------------------
// Same scheme with v_mfma_i32_16x16x64_i8: the column tile shrinks to 2 coordinates (176 byte positions = 11 row tiles of 16, one per
// wave, 16 column tiles x 4 accumulator registers each = the same 64 accumulator VGPRs), a row segment is 184 bytes = exactly 12 AES
// blocks when the stream offset is a multiple of 8 (it is for every CRS region), a unit is 256 rows (3072 blocks = 3 full rounds of
// the 16 waves).  The unit's digit fragments (64 KiB) do not fit LDS beside the table and the tile: they are staged per 64-row
// k-step (16 KiB), double buffered, by the waves that have no row tile (11..15; they prefetch the fragments into their idle accumulator
// registers under the expansion), one barrier per k-step.
// logq = 1472 (values of 184 bytes, all significant) has the same 184-byte row segment with ONE coordinate per column tile: 184 byte
// positions = 11.5 row tiles, padded to 12 (the last 8 positions are never read back).
constexpr int RT2 = 256, TS2 = 192, BPR2 = 12, NQ2 = 16, N2 = 16 * NQ2;
template <int LOGQ> struct W16;
template <> struct W16<736> { static constexpr int CT = 2, SBY = 88, VBY = 92, MT = 11, MBP = 176, LL = 12; };
template <> struct W16<1472> { static constexpr int CT = 1, SBY = 184, VBY = 184, MT = 12, MBP = 192, LL = 23; };
static_assert(RT2 * 4 == 1024 && BPR2 % 4 == 0 && RT2 * BPR2 == 3 * 1024 && W16<736>::CT * W16<736>::VBY + 8 <= TS2 && W16<1472>::CT * W16<1472>::VBY + 8 <= TS2, "wide tile geometry");

// MODE 0: regenerate the keystream (AES) and multiply-accumulate.  MODE 1: regenerate and WRITE the rows to `image` in A-FRAGMENT order
// (offset-by-128 bytes, b coordinate included) -- the CRS expanded once for the matrix-core path, streamed by k_mmstream below.
// Fragment of (row tile mt = 11 * column tile + wave, 64-row k-step s): 64 lanes x 16 bytes at image + ((mt * KS + s) * 64 + lane) * 16,
// KS = 4 * ceil(rows / 256) k-steps per region.
template <int MODE, int LOGQ>
__global__ __launch_bounds__(1024) void k_evalmm16(AesKey key /* rk[56..59] ^ 0x80808080 */, const uint32_t *__restrict__ g_t0, uint64_t off,
                                                   uint32_t n, uint32_t nrows, uint32_t rows_per_chunk, const uint8_t *__restrict__ c8,
                                                   const int8_t *__restrict__ cd, int *__restrict__ part, uint8_t *__restrict__ image) {
  using G = W16<LOGQ>;
  constexpr int CT2 = G::CT, MT2 = G::MT, MB2 = G::MBP, NSW = (16 - G::MT) * 64;  // NSW threads stage fragments / span constants
  constexpr uint32_t SBq = G::SBY, VBq = G::VBY;
  struct __attribute__((aligned(16))) Lds {
    uint32_t lt[mf::kTabBytes / 4];  // first: LDS address 0 (aes_dev.hpp)
    uint8_t tile[RT2 * TS2];
    v4i bfrag[2][NQ2][64];           // the current and the next k-step's coefficient-digit fragments
    uint32_t spanc[RT2][2][5];
  };
  __shared__ Lds lds;
  mf::lds_fill_tab(lds.lt, g_t0);
  const uint8_t *tab = reinterpret_cast<const uint8_t *>(lds.lt);
  const mf::AesLane L = mf::aes_lane();
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t j0 = blockIdx.x * CT2;
  const uint32_t nks = j0 >= n ? 0u : min((uint32_t)CT2, n - j0);
  const bool has_b = j0 + CT2 > n && j0 <= n;
  const uint32_t r0 = blockIdx.y * rows_per_chunk;
  const uint32_t r1 = min(nrows, r0 + rows_per_chunk);

  v4i acc[NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; q++) acc[q] = v4i{0, 0, 0, 0};

  const uint32_t c16 = lane & 15, g4 = lane >> 4;
  const uint32_t m = wave * 16 + c16;  // byte position, < MB2 for waves < MT2
  const uint32_t mcol = (m / SBq) * VBq + (m % SBq);
  const uint32_t hstep = (n * VBq) & 15;

  auto span_task = [&](uint32_t u0, uint32_t task) {  // task = 2 * local row + which span
    const uint32_t lr = task >> 1, which = task & 1;
    const uint64_t row = (uint64_t)u0 + lr;
    if (row >= r1 || !nks) return;
    const RowGeom g = row_geom(off, row, n, j0, nks, VBq);
    const uint64_t sp0 = g.cb0 >> 8, sp1 = (g.cb0 + g.nblk - 1) >> 8;
    if (which && sp1 == sp0) return;
    uint32_t sc[5];
    mf::aes_span_consts(tab, L, key, sp0 + which, sc);
#pragma unroll
    for (int i = 0; i < 5; i++) lds.spanc[lr][which][i] = sc[i];
  };
  const v4i *cdv = reinterpret_cast<const v4i *>(cd);
  const uint32_t KS = (nrows + RT2 - 1) / RT2 * (RT2 / 64);  // 64-row k-steps of the region (MODE 1)

  __syncthreads();
  if (tid < 2 * RT2) span_task(r0, tid);
  __syncthreads();
  for (uint32_t u0 = r0; u0 < r1; u0 += RT2) {
    if (MODE == 1) {  // expansion only: tile -> image
      for (uint32_t s2 = tid; s2 < RT2 * BPR2; s2 += 1024) {
        const uint32_t lr = s2 / BPR2, k = s2 % BPR2;
        const uint64_t row = (uint64_t)u0 + lr;
        uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // rows past the region: A' = -128, i.e. A = 0
        if (row < r1) {
          const RowGeom g = row_geom(off, row, n, j0, nks, VBq);
          if (k < g.nblk) {
            const uint64_t ctr = g.cb0 + k;
            const uint32_t *scp = lds.spanc[lr][(uint32_t)((ctr >> 8) - (g.cb0 >> 8))];
            uint32_t sc[5] = {scp[0], scp[1], scp[2], scp[3], scp[4]};
            mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
          }
        }
        *reinterpret_cast<uint4 *>(&lds.tile[lr * TS2 + 16 * k]) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      __syncthreads();
      if (has_b) {
        for (uint32_t s2 = tid; s2 < RT2 * VBq; s2 += 1024) {
          const uint32_t lr = s2 / VBq, k = s2 % VBq;
          const uint64_t row = (uint64_t)u0 + lr;
          if (row >= r1) continue;
          const RowGeom g = row_geom(off, row, n, j0, nks, VBq);
          lds.tile[lr * TS2 + g.head + nks * VBq + k] = (uint8_t)(c8[row * VBq + k] ^ 0x80);
        }
        __syncthreads();
      }
      if (wave < MT2) {  // this wave's row tile, four k-steps: the byte gather of the MFMA phase, stored instead of multiplied
        const uint32_t head0 = row_geom(off, u0, n, j0, 1, VBq).head;
        v4i *dst = reinterpret_cast<v4i *>(image) + ((uint64_t)(MT2 * blockIdx.x + wave) * KS + (u0 >> 6)) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < RT2 / 64; ks++) {
          const uint32_t lrb = ks * 64 + 16 * g4;
          uint32_t aw[4];
#pragma unroll
          for (int e4 = 0; e4 < 4; e4++) {
            uint32_t x = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const uint32_t lr = lrb + 4 * e4 + e;
              x |= (uint32_t)lds.tile[lr * TS2 + ((head0 + hstep * lr) & 15) + mcol] << (8 * e);
            }
            aw[e4] = x;
          }
          dst[(uint64_t)ks * 64] = v4i{(int)aw[0], (int)aw[1], (int)aw[2], (int)aw[3]};
        }
      } else {
        const uint32_t t2 = tid - MT2 * 64;
        for (uint32_t task = t2; task < 2 * RT2; task += NSW) span_task(u0 + RT2, task);
      }
      __syncthreads();
      continue;
    }
    // first k-step's digit fragments: 16 KiB contiguous, one 16-byte load per thread, in flight under the expansion
    const v4i bstage = cdv[(uint64_t)(u0 >> 6) * NQ2 * 64 + tid];
    // the other k-steps' fragments: the waves without a row tile fetch them NOW, into the (for them unused) accumulator registers, so
    // that the loads land under the expansion and staging them later costs no global latency
    if (wave >= MT2) {
      const uint32_t t2 = tid - MT2 * 64;
#pragma unroll
      for (int k2 = 1; k2 < RT2 / 64; k2++)
#pragma unroll
        for (int i2 = 0; i2 < 4; i2++) {
          const uint32_t idx = t2 + NSW * i2;
          if (idx < NQ2 * 64) acc[(k2 - 1) * 4 + i2] = cdv[((uint64_t)(u0 >> 6) + k2) * NQ2 * 64 + idx];
        }
    }
    // ---- (1) expansion: four lanes per row, each takes blocks (tid & 3) + 4 it, it = 0..2: exactly three blocks per thread, one row
    //          geometry per thread and unit
    {
      const uint32_t lr = tid >> 2;
      const uint64_t row = (uint64_t)u0 + lr;
      if (row < r1) {
        const RowGeom g = row_geom(off, row, n, j0, nks, VBq);
#pragma unroll 1
        for (int it = 0; it < BPR2 / 4; it++) {
          const uint32_t k = (tid & 3) + 4 * it;
          if (k >= g.nblk) continue;
          const uint64_t ctr = g.cb0 + k;
          const uint32_t *scp = lds.spanc[lr][(uint32_t)((ctr >> 8) - (g.cb0 >> 8))];
          uint32_t sc[5] = {scp[0], scp[1], scp[2], scp[3], scp[4]};
          uint32_t w[4];
          mf::aes256_ctr_block_sc(tab, L, key, ctr, sc, w);
          *reinterpret_cast<uint4 *>(&lds.tile[lr * TS2 + 16 * k]) = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
    (&lds.bfrag[0][0][0])[tid] = bstage;
    __syncthreads();
    if (has_b) {
      for (uint32_t s2 = tid; s2 < RT2 * VBq; s2 += 1024) {
        const uint32_t lr = s2 / VBq, k = s2 % VBq;
        const uint64_t row = (uint64_t)u0 + lr;
        if (row >= r1) continue;
        const RowGeom g = row_geom(off, row, n, j0, nks, VBq);
        lds.tile[lr * TS2 + g.head + nks * VBq + k] = (uint8_t)(c8[row * VBq + k] ^ 0x80);
      }
      __syncthreads();
    }
    // ---- (2) per 64-row k-step: waves 0..10 MFMA from bfrag[ks & 1]; waves 11..15 stage the next k-step's fragments (and, once,
    //          the next unit's span constants)
    const uint32_t head0 = row_geom(off, u0, n, j0, 1, VBq).head;
#pragma unroll
    for (int ks = 0; ks < RT2 / 64; ks++) {
      if (wave < MT2) {
        if ((uint64_t)u0 + ks * 64 < r1) {  // else: whole k-step beyond the chunk (its digits are zero anyway)
          const uint32_t lrb = ks * 64 + 16 * g4;
          uint32_t aw[4];
#pragma unroll
          for (int e4 = 0; e4 < 4; e4++) {
            uint32_t x = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const uint32_t lr = lrb + 4 * e4 + e;
              x |= (uint32_t)lds.tile[lr * TS2 + ((head0 + hstep * lr) & 15) + mcol] << (8 * e);
            }
            aw[e4] = x;
          }
          const v4i a = {(int)aw[0], (int)aw[1], (int)aw[2], (int)aw[3]};
#ifndef MM_SKIP_MFMA
#pragma unroll
          for (int q = 0; q < NQ2; q++) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, lds.bfrag[ks & 1][q][lane], acc[q], 0, 0, 0);
#else
          acc[0][0] += a[0] + a[1] + a[2] + a[3];
#endif
        }
      } else {
        const uint32_t t2 = tid - MT2 * 64;  // 0..319
        if (ks + 1 < RT2 / 64) {
          v4i *dst = &lds.bfrag[(ks + 1) & 1][0][0];
#pragma unroll
          for (int i2 = 0; i2 < 4; i2++) {
            const uint32_t idx = t2 + NSW * i2;
            if (idx < NQ2 * 64) dst[idx] = acc[ks * 4 + i2];
          }
        }
        if (ks == 0)
          for (uint32_t task = t2; task < 2 * RT2; task += NSW) span_task(u0 + RT2, task);
      }
      __syncthreads();
    }
  }
  if (MODE == 0 && wave < MT2) {  // (MODE 1 has no partial products: part is null)
    int *p = part + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * MB2 * N2;
#pragma unroll
    for (int q = 0; q < NQ2; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) p[(uint64_t)(wave * 16 + 4 * g4 + e) * N2 + 16 * q + c16] = acc[q][e];
  }
}


// ---- k_mmstream: the resident regime of the batch prover -- the same GEMM streamed from the A-fragment image ---------------------
// No AES, no tile, no gather: a wave owns two row tiles (2 x 16 byte positions x 256 digit columns = 128 accumulator VGPRs) and reads
// their fragments straight from HBM (1 KiB per wave, row tile and 64-row k-step, coalesced, one 256-row stage ahead in registers);
// the stage's digit fragments (64 KiB) are shared by the 8 waves of the workgroup through LDS, double buffered, one barrier per 256
// rows.  HBM-bound by construction: per stage and CU 64 KiB of A against 128 x 8 MFMAs (1.7 us of matrix core) and 512 KiB of LDS
// fragment reads.  grid = (row-tile pairs / 8, row chunks); part[((chunk * Mtot) + m) * 256 + n] as k_evalmm16 writes it.
#ifndef MMS_SW
#define MMS_SW 8
#endif
constexpr int SW = MMS_SW;   // waves per workgroup: RG row groups x CG column groups (MMS_SW=4 MMS_RQ=4: one wave per SIMD, 256 accumulators)
constexpr int TPW = 16;      // row tiles per workgroup
#ifndef MMS_RQ
#define MMS_RQ 2
#endif
constexpr int RQ = MMS_RQ;        // row tiles per wave
constexpr int RG = 16 / RQ;       // row groups: the workgroup covers 16 row tiles
constexpr int CG = SW / RG;       // column groups
constexpr int CH = NQ2 / CG;      // column tiles per wave
// One launch serves ng groups of coefficient vectors (ng digit matrices cdv + g cd_stride, ng partial-product arrays part + g
// part_stride): workgroups b and b + 8 -- dispatched back to back onto the same XCD (blocks go round-robin over the 8 XCDs) -- take the
// same 16 row tiles for consecutive groups, so the second one's A fragments come out of that XCD's L2 (or the Infinity Cache) instead
// of HBM: the image is read from HBM once per ng groups.  grid.x = 8 ng ceil(tile groups / 8).
struct MmsImages { const v4i *image[16]; };  // the region image each group of a launch streams (S and AS groups share one launch)
// Which (group, tile group) the slot-th workgroup of an XCD takes.  `ngt` groups in the launch, `ngr` of them per region (ngt / ngr regions).
//   map 0: slot -> (slot % ngt, slot / ngt): the 32 workgroups an XCD runs at a time are 32 / ngt tile groups x all groups of BOTH regions -- an image's
//          fragments are shared by ngr workgroups, a group's digit fragments by 32 / ngt of them;
//   map 1: 32 consecutive slots are 32 / ngr tile groups x the ngr groups of ONE region (regions alternate per 32 slots): the fragments are shared by ngr
//          workgroups as before, a group's digit fragments by 32 / ngr -- the digit fragments are as many bytes per workgroup as the image's, so with two
//          regions this halves their share of the L2 misses.  Needs ngr | 32; the host picks map 0 otherwise.
__device__ __forceinline__ void mms_item(uint32_t xcd, uint32_t slot, uint32_t ngt, uint32_t ngr, uint32_t map, uint32_t &grp, uint32_t &tg) {
  if (map == 0) {
    grp = slot % ngt;
    tg = (slot / ngt) * 8 + xcd;
  } else {
    const uint32_t blk = slot >> 5, i = slot & 31, nreg = ngt / ngr, tpb = 32 / ngr;
    grp = (blk % nreg) * ngr + i % ngr;
    tg = ((blk / nreg) * tpb + i / ngr) * 8 + xcd;
  }
}
// ---- packed partial products (round 6) ------------------------------------------------------------------------------------------------------------------
// A lane of the streaming kernels ends an item with, per row tile and column tile, the int32 sums of FOUR consecutive byte positions (rows 4 g4 + e of the tile) of ONE
// digit column (16 q + c16).  Written as they are that is 1 KiB per byte position and group -- 2.1 GB per super-group launch, written by the kernel and read back by the
// epilogue.  The epilogue only ever needs them recombined, so the kernel recombines what it holds (exact integer arithmetic, the corrections stay in the epilogue):
//   pk = 1 (one-byte coefficient columns: b_w)   T = sum_e acc[e] 2^(8e), |T| < 2^56, one int64 per (byte-position quad, column): half the bytes;
//   pk = 4 (four-byte coefficient vectors)       additionally over the vector's four digit columns w = c16 & 3, which are the four lanes of a DPP quad:
//                                                U = sum_w T_w 2^(8w), |U| < 2^80, one 16-byte record {lo, mid, hi, 0} (two's complement, 96 bits) per
//                                                (byte-position quad, vector), written by the quad's lane 0: a quarter of the bytes.
// The ones column (sum_i A'[i][m], column 4 nvec) is the w = 0 lane of a quad whose other three columns are zero digits (k_mm_digits*: "every other column is zero"), so
// its record is T_0 itself.  pk = 0: the int32 layout of the regenerating kernels (k_evalmm16<0>).  mfh_set_mm_pack(ctx, 0) keeps pk = 0 everywhere (A/B, tests).
__device__ __forceinline__ long long mms_pack_e(const v4i &a) {
  return (long long)a[0] + ((long long)a[1] << 8) + ((long long)a[2] << 16) + ((long long)a[3] << 24);
}
template <int CTRL>
__device__ __forceinline__ long long mms_quad_bcast(long long t) {  // lane CTRL & 3 of every quad, to the whole quad
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(unsigned long long)t, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)((unsigned long long)t >> 32), CTRL, 0xf, 0xf, false);
  return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void mms_store_packed(int *__restrict__ part, uint32_t pk, uint64_t m4 /* byte-position quad, chunks included */, uint32_t ctile, uint32_t c16,
                                                 const v4i &acc) {
  const long long T = mms_pack_e(acc);
  if (pk == 1) {
    reinterpret_cast<long long *>(part)[m4 * N2 + 16 * ctile + c16] = T;
    return;
  }
  const long long T1 = mms_quad_bcast<0x55>(T), T2 = mms_quad_bcast<0xAA>(T), T3 = mms_quad_bcast<0xFF>(T);
  if (c16 & 3) return;
  const __int128 U = (__int128)T + (__int128)T1 * 256 + (__int128)T2 * 65536 + (__int128)T3 * 16777216;
  const unsigned __int128 u = (unsigned __int128)U;
  reinterpret_cast<uint4 *>(part)[m4 * (N2 / 4) + 4 * ctile + (c16 >> 2)] = uint4{(uint32_t)u, (uint32_t)(u >> 32), (uint32_t)(u >> 64), 0u};
}
__device__ __forceinline__ void mmstream_body(const MmsImages &imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                              const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t grp, uint32_t tg /* tile group = 16 row tiles */,
                                              uint32_t chunk, uint64_t cd_stride /* v4i */, uint64_t part_stride /* int */, uint32_t pk /* packed partial products */) {
  __shared__ v4i bfrag[2][RT2 / 64][NQ2][64];  // 2 x 64 KiB
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: the fragment addresses are a scalar base + the lane's 16 bytes
  const uint32_t c16 = lane & 15, g4 = lane >> 4;
  const uint32_t rq = wave % RG, ch = wave / RG;
  if (tg * 16 >= mtiles) return;  // (uniform)
  const v4i *__restrict__ image = imgs.image[grp];
  cdv += grp * cd_stride;
  part += grp * part_stride;
  const uint32_t mt0 = (tg * RG + rq) * RQ;  // this wave's row tiles (with CG > 1 the waves rq and rq + RG read the same A fragments)
  const uint32_t r0 = chunk * rows_per_chunk, r1 = min(nrows, r0 + rows_per_chunk);
  v4i acc[RQ][CH];
#pragma unroll
  for (int t = 0; t < RQ; t++)
#pragma unroll
    for (int q = 0; q < CH; q++) acc[t][q] = v4i{0, 0, 0, 0};
  constexpr int KSN = RT2 / 64;                          // k-steps per stage: 4
  constexpr int BPK = NQ2 * 64 / (SW * 64);              // digit fragments per thread and k-step: 2
#ifndef MMS_DEPTH
#define MMS_DEPTH 6
#endif
  constexpr int DEPTH = MMS_DEPTH;                       // digit fragments in flight from LDS ahead of the MFMAs
  auto a_load = [&](int t, uint32_t u0, int ks) -> v4i {
#ifdef MMS_SKIP_A  // timing-only build: no HBM stream (wrong results)
    return v4i{(int)u0, ks, t, (int)lane};
#endif
    // (a row tile past the end is clamped to the last one and its products are dropped at the end: the load is UNCONDITIONAL.  A branch
    // around it makes the number of younger loads unknown to the compiler's s_waitcnt vmcnt accounting, which then waits for far too
    // many: vmcnt(1) in the first k-step of a stage, i.e. the previous stage's last loads awaited one k-step after their issue)
    const v4i *ap = image + ((uint64_t)min(mt0 + t, mtiles - 1) * KS + (u0 >> 6) + ks) * 64;  // scalar
    return ap[lane];
  };
  v4i a[RQ][KSN];
  // Vector-memory operations complete in issue order (s_waitcnt vmcnt), so every load of the loop is used exactly one stage after it was
  // issued: the A fragments of stage s+1 and the digit fragments of stage s+2 (kept in registers for a stage, then written to the LDS
  // buffer stage s has just finished with) are issued in k-step ks of stage s and waited for in k-step ks of stage s+1.
  v4i bnr[KSN][BPK];
  auto b_load = [&](uint32_t u0, int ks, int i) -> v4i {
    const v4i *bp = cdv + (uint64_t)(u0 >> 6) * NQ2 * 64 + (uint64_t)ks * NQ2 * 64 + SW * 64 * i;  // scalar
    return bp[tid];
  };
  if (r0 >= r1) return;  // (uniform; the host never launches an empty chunk)
  const uint32_t ulast = r0 + (r1 - r0 - 1) / RT2 * RT2;  // first row of the chunk's last stage: prefetches past it re-read it (no branches in the loop)
#pragma unroll
  for (int i = 0; i < KSN * BPK; i++) (&bfrag[0][0][0][0])[tid + SW * 64 * i] = cdv[(uint64_t)(r0 >> 6) * NQ2 * 64 + tid + SW * 64 * i];
  // The prologue issues its loads in EXACTLY the order the loop does (pinned: the scheduler must not move them).  s_waitcnt vmcnt(n) is a
  // static count of younger loads, and the compiler takes the minimum over the paths into the loop: with the prologue's loads in another
  // order it emitted vmcnt(8) at the head of a stage and vmcnt(1) in its first k-step -- the loads of the previous stage's LAST k-step
  // awaited one k-step after their issue instead of one stage after it.
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KSN; ks++) {
#pragma unroll
    for (int t = 0; t < RQ; t++) a[t][ks] = a_load(t, r0, ks);
#pragma unroll
    for (int i = 0; i < BPK; i++) bnr[ks][i] = b_load(min(r0 + RT2, ulast), ks, i);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
#if defined(MMS_PRIO_YOUNG)  /* timing variants (MI355X_MICROARCH.md, two waves per SIMD, item 4): static priority for one half of the workgroup's waves */
  if (wave >= SW / 2) __builtin_amdgcn_s_setprio(1);
#elif defined(MMS_PRIO_OLD)
  if (wave < SW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  uint32_t buf = 0;
  for (uint32_t u0 = r0; u0 < r1; u0 += RT2) {
    const uint32_t un1 = min(u0 + RT2, ulast), un2 = min(u0 + 2 * RT2, ulast);
    const v4i *bcur = &bfrag[buf][0][ch * CH][lane];  // this wave's fragment (ks, q) at bcur[(ks * NQ2 + q) * 64]
    v4i *bnext = &bfrag[buf ^ 1][0][0][0];
    // a ring of DEPTH digit fragments keeps the LDS reads ahead of the MFMAs that use them
    v4i ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; i++) ring[i] = bcur[((i / CH) * NQ2 + i % CH) * 64];
#pragma unroll
    for (int ks = 0; ks < KSN; ks++) {
#ifdef MMS_LOADS_ONLY  // timing-only build (wrong results): the HBM stream and the staging of the digit fragments without the matrix core
#pragma unroll
      for (int t = 0; t < RQ; t++) acc[t][ks][0] += a[t][ks][0] ^ a[t][ks][3];
#else
#pragma unroll
      for (int q = 0; q < CH; q++) {
        const int it = ks * CH + q;
        const v4i b = ring[it % DEPTH];
        if (it + DEPTH < KSN * CH) ring[it % DEPTH] = bcur[(((it + DEPTH) / CH) * NQ2 + (it + DEPTH) % CH) * 64];
#pragma unroll
        for (int t = 0; t < RQ; t++) {
#ifdef MMS_ACC_ASM  // accumulators pinned to AccVGPRs (one wave per SIMD, 256 of them: MMS_SW=4 MMS_RQ=4): the compiler's own placement shuttles them through arch VGPRs and spills
          asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[t][q]) : "v"(a[t][ks]), "v"(b));
#else
          acc[t][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][ks], b, acc[t][q], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);  // keep [refill one ring slot, RQ MFMAs] as written: the reads stay DEPTH fragments ahead
      }
#endif
      // this k-step's share of the next stage's digit fragments (loaded a stage ago) -> the other LDS buffer; the A fragments of this
      // k-step are spent: their registers take the next stage's
#pragma unroll
      for (int i = 0; i < BPK; i++) bnext[ks * NQ2 * 64 + tid + SW * 64 * i] = bnr[ks][i];
#pragma unroll
      for (int t = 0; t < RQ; t++) a[t][ks] = a_load(t, un1, ks);
#pragma unroll
      for (int i = 0; i < BPK; i++) bnr[ks][i] = b_load(un2, ks, i);
    }
#ifndef MMS_NOSYNC  // (timing-only build without it: wrong results -- what the stage barrier costs)
    __syncthreads();  // the other buffer is complete; everyone is done with this one
#endif
    buf ^= 1;
  }
  const uint64_t Mtot = (uint64_t)mtiles * 16;
#ifdef MMS_ACC_ASM
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the hazard recogniser does not see inside the asm MFMAs: their last results must have landed before they are read)
#endif
  if (pk) {  // (uniform)
#pragma unroll
    for (int t = 0; t < RQ; t++) {
      if (mt0 + t >= mtiles) continue;
      const uint64_t m4 = (uint64_t)chunk * (Mtot / 4) + (uint64_t)(mt0 + t) * 4 + g4;
#pragma unroll
      for (int q = 0; q < CH; q++) mms_store_packed(part, pk, m4, ch * CH + q, c16, acc[t][q]);
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < RQ; t++) {
    if (mt0 + t >= mtiles) continue;
    int *p = part + ((uint64_t)chunk * Mtot + (uint64_t)(mt0 + t) * 16) * N2;
#pragma unroll
    for (int q = 0; q < CH; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) p[(uint64_t)(4 * g4 + e) * N2 + 16 * (ch * CH + q) + c16] = acc[t][q][e];
  }
}

// Two entry points over the one body so that profiles tell the two launch shapes apart: k_mmstream = several groups per launch (the S / AS
// rounds of the batch prover: matrix-core bound, the kernel bench.py's roofline describes), k_mmstream1 = one group per launch (b_w's pass
// over the BT+BV image, mfh_eval_rows_multi from a registered image: HBM-bound).
__global__ __launch_bounds__(SW * 64) void k_mmstream(MmsImages imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                      const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t ngt, uint32_t ngr, uint32_t map,
                                                      uint64_t cd_stride, uint64_t part_stride, uint32_t pk) {
  uint32_t grp, tg;
  mms_item(blockIdx.x & 7, blockIdx.x >> 3, ngt, ngr, map, grp, tg);
  mmstream_body(imgs, mtiles, KS, nrows, rows_per_chunk, cdv, part, grp, tg, blockIdx.y, cd_stride, part_stride, pk);
}
__global__ __launch_bounds__(SW * 64) void k_mmstream1(MmsImages imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                       const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t pk) {
  mmstream_body(imgs, mtiles, KS, nrows, rows_per_chunk, cdv, part, 0u, blockIdx.x, blockIdx.y, 0, 0, pk);
}
// The same launch as a PERSISTENT grid: one workgroup per CU (the 128 KiB of LDS allow no second one anyway), workgroup b = CU slot b >> 3 of the XCD
// b & 7 (blocks are dealt round-robin over the XCDs), looping over the slots cu, cu + 32, cu + 64, ... of its XCD and over the row chunks.  The ngr
// workgroups that stream the same fragments -- and, with sync_mode 2, all 32 workgroups of the XCD -- therefore stay on the same CUs for the whole launch
// and begin every item together: before an item lane 0 adds to the set's counter and polls it (sc1 loads, s_sleep between polls) until every member has
// arrived OR spin_max polls have passed.  The rendezvous is for speed only (the members then find each other's fragments in the XCD's L2): nothing is
// handed over, a member that is not resident (fewer CUs than workgroups, another kernel on the GPU) only costs the others spin_max polls per item, and
// every wave reaches the end of the grid whatever the counters hold.  sync: 8 x 32 counters zeroed before the launch.
// (`width`: workgroups per XCD = grid / 8, 32 by default -- every CU --; fewer leave CUs of every XCD to other streams' kernels, and the workgroup then strides
// over its XCD's slots by `width` (mfh_set_mm_width: the round-5 experiment on the power finding, EXPERIMENTS.md); the rendezvous needs width == 32)
__device__ __forceinline__ void mmstream_persistent(const MmsImages &imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                    const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t ngt, uint32_t ngr, uint32_t map,
                                                    uint64_t cd_stride, uint64_t part_stride, uint32_t nblk /* 32-slot blocks per XCD */, uint32_t nchunks,
                                                    uint32_t *__restrict__ sync, uint32_t sync_mode, uint32_t spin_max, uint32_t width, uint32_t pk) {
  const uint32_t xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const uint32_t members = sync_mode == 2 ? 32u : (map ? ngr : ngt);
  uint32_t *ctr = sync + xcd * 32 + (sync_mode == 2 ? 0u : cu / members);
  uint32_t k = 0;
  for (uint32_t chunk = 0; chunk < nchunks; chunk++)
    for (uint32_t slot = cu; slot < nblk * 32; slot += width, k++) {
      if (sync_mode && members > 1) {
        if (threadIdx.x == 0) {
          __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t target = (k + 1) * members;
          for (uint32_t spin = 0; spin < spin_max && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spin++) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
      }
      uint32_t grp, tg;
      mms_item(xcd, slot, ngt, ngr, map, grp, tg);
      mmstream_body(imgs, mtiles, KS, nrows, rows_per_chunk, cdv, part, grp, tg, chunk, cd_stride, part_stride, pk);
    }
}
__global__ __launch_bounds__(SW * 64) void k_mmstream_p(MmsImages imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                        const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t ngt, uint32_t ngr, uint32_t map,
                                                        uint64_t cd_stride, uint64_t part_stride, uint32_t nblk, uint32_t nchunks,
                                                        uint32_t *__restrict__ sync, uint32_t sync_mode, uint32_t spin_max, uint32_t width, uint32_t pk) {
  mmstream_persistent(imgs, mtiles, KS, nrows, rows_per_chunk, cdv, part, ngt, ngr, map, cd_stride, part_stride, nblk, nchunks, sync, sync_mode, spin_max, width, pk);
}
// the same grid under another name for the launch that serves b_w of several super-groups (one-byte coefficient columns over the BT+BV image): profiles then show
// the two launch shapes -- 16 groups x 32768 rows, matrix-core bound; up to 8 groups x 21845 rows -- as two rows
__global__ __launch_bounds__(SW * 64) void k_mmstream_pb(MmsImages imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                         const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t ngt, uint32_t ngr, uint32_t map,
                                                         uint64_t cd_stride, uint64_t part_stride, uint32_t nblk, uint32_t nchunks,
                                                         uint32_t *__restrict__ sync, uint32_t sync_mode, uint32_t spin_max, uint32_t width, uint32_t pk) {
  mmstream_persistent(imgs, mtiles, KS, nrows, rows_per_chunk, cdv, part, ngt, ngr, map, cd_stride, part_stride, nblk, nchunks, sync, sync_mode, spin_max, width, pk);
}

// ---- k_mmstream_w: the same GEMM with ONE wave per SIMD (round 4) -----------------------------------------------------------------------------------
// k_mmstream is power-limited: what raises its clock is fewer bytes moved per MFMA.  Here a workgroup is 4 waves and a wave owns FOUR row tiles
// (64 byte positions x 256 digit columns = 256 accumulator registers, all in AccVGPRs: the MFMAs are inline asm with "+a" operands -- left to itself the compiler
// spreads 256 accumulators over both register files, shuttles them with v_accvgpr moves and spills), so every digit fragment read from LDS feeds 4 MFMAs instead
// of 2: 256 KiB of LDS fragment reads per stage and CU instead of 512, the same L2 traffic.  With one wave per SIMD nothing hides a wave's own waits, so:
//   * the digit fragments come from LDS in groups of four, two groups ahead (a ring of 16 fragments), one ds_read per fragment in use, and a group's LAST fragment
//     is used first: LDS returns in order, so one s_waitcnt covers the group (the same trick for the four staging words of a k-step's ds_writes);
//   * NOTHING is issued in a burst: the 4 ds_writes, 8 global loads and 16 ds_reads of a k-step are dealt one per fragment (4 MFMAs);
//   * the ring runs across the stage boundary, so the stage's ONE barrier sits at the end of group 9 (its position and why one suffices: at the barrier), and
//     it waits for the wave's ds_writes only (s_waitcnt lgkmcnt(4): the four fragment reads issued after them stay in flight).
// Same operands, same partial products, same epilogues as k_mmstream; persistent grid only (mfh_set_mm_stream).
// WHY IT SHIPS although it is never the default: it takes the same time as the two-wave body with half the LDS reads and a quarter of the parked cycles -- the reproducible half of
// the finding that k_mmstream_p is power-limited; tools/mm_variant_clock.py A/Bs the two bodies on any box, and tests keep it bit-identical.
constexpr int WSW = 4, WRQ = 4;
__device__ __forceinline__ void mfma_acc(v4i &acc, const v4i &a, const v4i &b) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mmstream_body_w(const MmsImages &imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t grp, uint32_t tg, uint32_t chunk, uint64_t cd_stride,
                                                uint64_t part_stride) {
  __shared__ v4i bfrag[2][RT2 / 64][NQ2][64];  // 2 x 64 KiB
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t c16 = lane & 15, g4 = lane >> 4;
  if (tg * 16 >= mtiles) return;  // (uniform)
  const v4i *__restrict__ image = imgs.image[grp];
  cdv += grp * cd_stride;
  part += grp * part_stride;
  const uint32_t mt0 = tg * 16 + wave * WRQ;
  const uint32_t r0 = chunk * rows_per_chunk, r1 = min(nrows, r0 + rows_per_chunk);
  if (r0 >= r1) return;  // (uniform)
  constexpr int KSN = RT2 / 64;               // 4 k-steps per stage
  constexpr int BPK = NQ2 * 64 / (WSW * 64);  // 4 staging words per thread and k-step
  v4i acc[WRQ][NQ2];
#pragma unroll
  for (int t = 0; t < WRQ; t++)
#pragma unroll
    for (int q = 0; q < NQ2; q++) acc[t][q] = v4i{0, 0, 0, 0};
  auto a_load = [&](int t, uint32_t u0, int ks) -> v4i {
    const v4i *ap = image + ((uint64_t)min(mt0 + t, mtiles - 1) * KS + (u0 >> 6) + ks) * 64;  // scalar (clamped: the load is unconditional, see k_mmstream)
    return ap[lane];
  };
  auto b_load = [&](uint32_t u0, int ks, int i) -> v4i {
    const v4i *bp = cdv + (uint64_t)(u0 >> 6) * NQ2 * 64 + (uint64_t)ks * NQ2 * 64 + WSW * 64 * i;  // scalar
    return bp[tid];
  };
  const uint32_t ulast = r0 + (r1 - r0 - 1) / RT2 * RT2;
  v4i a[WRQ][KSN], bnr[KSN][BPK];
#pragma unroll
  for (int i = 0; i < KSN * BPK; i++) (&bfrag[0][0][0][0])[tid + WSW * 64 * i] = cdv[(uint64_t)(r0 >> 6) * NQ2 * 64 + tid + WSW * 64 * i];
  __builtin_amdgcn_sched_barrier(0);
  // The prologue issues its loads in the order they are outstanding at the head of a steady-state stage (vector-memory operations complete in issue order and
  // s_waitcnt vmcnt(n) is a static count over all paths into the loop): B[0], then (B[ks], A[ks - 1]) for ks = 1 .. 3; A[3] is loaded by the stage's k-step 0.
  const uint32_t ub1 = min(r0 + RT2, ulast);
#pragma unroll
  for (int i = 0; i < BPK; i++) bnr[0][i] = b_load(ub1, 0, i);
#pragma unroll
  for (int ks = 1; ks < KSN; ks++) {
#pragma unroll
    for (int i = 0; i < BPK; i++) bnr[ks][i] = b_load(ub1, ks, i);
#pragma unroll
    for (int t = 0; t < WRQ; t++) a[t][ks - 1] = a_load(t, r0, ks - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  // The ring: 4 slots of 4 fragments; group G of a stage (16 per stage: 4 per k-step) lives in rg[G & 3] and is fetched two groups ahead, ONE ds_read per
  // fragment of the group in use -- nothing of the loop is issued in a burst: with one wave per SIMD every instruction that is not an MFMA has to fit the gap
  // the previous MFMA leaves (a burst of 8 loads + 4 ds_writes at the end of every k-step was a fifth of the stage).
  v4i rg[4][4];
  uint32_t buf = 0;
  auto frag = [&](uint32_t b, int f) -> v4i { return bfrag[b][f >> 4][f & 15][lane]; };  // fragment f = 16 ks + q of buffer b
#pragma unroll
  for (int G = 0; G < 2; G++)
#pragma unroll
    for (int j = 0; j < 4; j++) rg[G][j] = frag(0, 4 * G + j);
  for (uint32_t u0 = r0; u0 < r1; u0 += RT2) {
    const uint32_t un1 = min(u0 + RT2, ulast), un2 = min(u0 + 2 * RT2, ulast);
    v4i *bnext = &bfrag[buf ^ 1][0][0][0];
#pragma unroll
    for (int ks = 0; ks < KSN; ks++) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int G = ks * 4 + g, Gn = G + 2;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = 3 - u;  // the group's last fragment first: LDS returns in order, one wait covers the group
#pragma unroll
          for (int t = WRQ - 1; t >= 0; t--) mfma_acc(acc[t][4 * g + j], a[t][ks], rg[G & 3][j]);
          // one fragment of the group two ahead (groups 14, 15 reach into the next stage's buffer: its k-step 0, written before this stage's barrier)
          rg[Gn & 3][u] = Gn < 16 ? frag(buf, 4 * Gn + u) : frag(buf ^ 1, 4 * (Gn - 16) + u);
          // one memory operation of the k-step's twelve
          if (g == 0) {  // this k-step's share of the next stage's digit fragments (loaded a stage ago) -> the other buffer; youngest staging word first
            bnext[ks * NQ2 * 64 + tid + WSW * 64 * (BPK - 1 - u)] = bnr[ks][BPK - 1 - u];
          } else if (g == 1) {  // ... and the staging words of the stage after
            bnr[ks][u] = b_load(un2, ks, u);
          } else if (g == 2) {  // the A fragments the previous k-step has finished with: k-step 3's take THIS stage's rows (they are used in its own k-step 3)
            if (ks == 0) a[u][KSN - 1] = a_load(u, u0, KSN - 1);
            else a[u][ks - 1] = a_load(u, un1, ks - 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (G == 9) {
          // The stage's ONE barrier.  Buffer X[ks] is written in group 4 ks of stage s (for stage s + 1), fetched for stage s + 1 in groups 4 ks - 2 .. 4 ks + 1
          // (k-step 0: from group 14 of stage s) and was last fetched for stage s - 1 two groups before its use: every write / first-fetch and last-fetch /
          // write pair has the end of group 9 of one of the stages between them.  Only the ds_writes have to have landed: the four fragment reads issued after
          // the last of them (group 9's) may still be in flight.
          asm volatile("s_waitcnt lgkmcnt(4)\n\ts_barrier" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    buf ^= 1;
  }
  __syncthreads();  // (every wave is out of the item's last stage before the next item's prologue rewrites buffer 0)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the hazard recogniser does not see inside the asm MFMAs: their last results must have landed
  const uint64_t Mtot = (uint64_t)mtiles * 16;
  // (int32 partial products only: the 256 accumulators live in AccVGPRs, and recombining them in the kernel -- mms_store_packed -- pulls them through the arch VGPRs all at once)
#pragma unroll
  for (int t = 0; t < WRQ; t++) {
    if (mt0 + t >= mtiles) continue;
    int *p = part + ((uint64_t)chunk * Mtot + (uint64_t)(mt0 + t) * 16) * N2;
#pragma unroll
    for (int q = 0; q < NQ2; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) p[(uint64_t)(4 * g4 + e) * N2 + 16 * q + c16] = acc[t][q][e];
  }
}
__global__ __launch_bounds__(WSW * 64) void k_mmstream_w(MmsImages imgs, uint32_t mtiles, uint32_t KS, uint32_t nrows, uint32_t rows_per_chunk,
                                                         const v4i *__restrict__ cdv, int *__restrict__ part, uint32_t ngt, uint32_t ngr, uint32_t map,
                                                         uint64_t cd_stride, uint64_t part_stride, uint32_t nblk, uint32_t nchunks) {
  const uint32_t xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
  for (uint32_t k = 0; k < nblk * nchunks; k++) {
    uint32_t grp, tg;
    mms_item(xcd, (k % nblk) * 32 + cu, ngt, ngr, map, grp, tg);
    mmstream_body_w(imgs, mtiles, KS, nrows, rows_per_chunk, cdv, part, grp, tg, k / nblk, cd_stride, part_stride);
  }
}

// out_v[j] = sum_{u,w} G[(j,u)][(v,w)] 256^(u + w) mod 2^704 with G = G' + 128 SA[(j,u)] + 128 sc[(v,w)] + 16384 nrows, G' and
// SA = G'[.][ones column] summed over the row chunks.  Block = one coordinate j x 64 vectors x 4 word groups.  Thread (v, lq)
// forms val_l = sum_k t_k 2^(8k) < 2^80 for its words l = lq, lq + 4, ... (no carries between them) and parks the three 32-bit pieces
// in LDS; the lq = 0 threads then add, per output word, the pieces of val_l, val_(l-1), val_(l-2) with the running carry (22 | 46 adds
// on LDS data) and write the element in 16-byte (LL even: 96-byte elements) or 8-byte pieces: the elements of neighbouring threads lie a
// ciphertext apart, so every store is its own memory transaction.
template <int ND, int KWM /* 32-bit words of a value that survive modq, at most: sizes the LDS exchange (22 at logq 736, 46 at 1472) */,
          bool PK /* packed partial products (the streaming kernels): instantiated apart, so that the int32 form keeps its registers and occupancy */>
__device__ __forceinline__ void evalmm_finish_body(const int *__restrict__ part, const int64_t *__restrict__ sc, uint32_t nchunks, uint32_t ntiles, uint32_t N,
                                                   uint32_t nvec, uint32_t n, uint32_t nrows, uint32_t ct, uint32_t MBv, uint32_t sby, uint32_t LL,
                                                   const MmIo &io, int accumulate, const int *__restrict__ sa_part, uint32_t sa_col) {
  __shared__ uint32_t sv[KWM][3][64];  // (17 KiB at logq 736: six workgroups per CU instead of the four that 35 KiB allow)
  const uint32_t vl = threadIdx.x & 63, lq = threadIdx.x >> 6;
  const uint32_t v = blockIdx.y * 64 + vl, j = blockIdx.x;
  const uint32_t tile = j / ct, jj = j % ct;
  const uint32_t KWv = sby / 4;  // 22 | 46 words survive modq
  if (v < nvec) {
    int64_t corr[ND];
#pragma unroll
    for (int w = 0; w < ND; w++) corr[w] = 128 * sc[ND * v + w] + 16384ll * nrows;
    // packed partial products (the streaming kernels, mms_store_packed): word l of the value is byte-position quad (tile MBv + jj sby) / 4 + l -- one record per
    // (quad, vector) [ND = 4: 96-bit U = sum_{k,w} G'[4 l + k][w] 2^(8 (k + w))] or per (quad, column) [ND = 1: T = sum_k G'[4 l + k] 2^(8 k)], and the same of the
    // ones column; the corrections 128 SA + 128 sc + 16384 rows are linear, so they are applied to the recombined sums
    if (PK) {
      const uint64_t M4 = (uint64_t)ntiles * MBv / 4;
      unsigned __int128 corrsum = 0;
#pragma unroll
      for (int w = 0; w < ND; w++) corrsum += (unsigned __int128)((uint64_t)corr[w] * 0x01010101ull) << (8 * w);  // (corr >= 0: sc >= -128 rows)
      for (uint32_t l = lq; l < KWv; l += 4) {
        const uint64_t m4 = ((uint64_t)tile * MBv + jj * sby) / 4 + l;
        __int128 U = 0, SA = 0;
        for (uint32_t ch = 0; ch < nchunks; ch++) {
          const uint64_t at = (uint64_t)ch * M4 + m4;
          if (ND == 4) {
            const uint4 r = reinterpret_cast<const uint4 *>(part)[at * (N / 4) + v], q = reinterpret_cast<const uint4 *>(sa_part)[at * (N / 4) + sa_col / 4];
            U += (__int128)(((unsigned __int128)(uint64_t)(int64_t)(int32_t)r.z << 64) | ((uint64_t)r.y << 32) | r.x);
            SA += (__int128)(((unsigned __int128)(uint64_t)(int64_t)(int32_t)q.z << 64) | ((uint64_t)q.y << 32) | q.x);
          } else {
            U += reinterpret_cast<const long long *>(part)[at * N + v];
            SA += reinterpret_cast<const long long *>(sa_part)[at * N + sa_col];
          }
        }
        // 128 SA[4 l + k] is added once per digit column w of the vector: sum_w 2^(8 w) times the recombined ones-column sum
        const unsigned __int128 val = (unsigned __int128)(U + SA * (__int128)(ND == 4 ? 128ll * 0x01010101ll : 128ll)) + corrsum;
        sv[l][0][vl] = (uint32_t)val;
        sv[l][1][vl] = (uint32_t)(val >> 32);
        sv[l][2][vl] = (uint32_t)(val >> 64);
      }
    } else
    for (uint32_t l = lq; l < KWv; l += 4) {
      unsigned __int128 val = 0;
      // the first chunk's partial products of the word's four byte positions are loaded TOGETHER (eight loads in flight per thread; read
      // position by position, each pair of loads was awaited before the next was issued), further chunks -- rare -- one by one
      int g0[4][ND], sa0[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int *row = part + ((uint64_t)tile * MBv + (jj * sby + 4 * l + k)) * N;
#pragma unroll
        for (int w = 0; w < ND; w++) g0[k][w] = row[ND * v + w];
        sa0[k] = (sa_part + ((uint64_t)tile * MBv + (jj * sby + 4 * l + k)) * N)[sa_col];  // the ones column: this group's, or the lender's (same rows)
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t mm = jj * sby + 4 * l + k;
        int64_t g[ND], sa = sa0[k];
#pragma unroll
        for (int w = 0; w < ND; w++) g[w] = g0[k][w];
        for (uint32_t ch = 1; ch < nchunks; ch++) {
          const int *row = part + (((uint64_t)ch * ntiles + tile) * MBv + mm) * N;
#pragma unroll
          for (int w = 0; w < ND; w++) g[w] += row[ND * v + w];
          sa += (sa_part + (((uint64_t)ch * ntiles + tile) * MBv + mm) * N)[sa_col];
        }
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < ND; w++) t += (uint64_t)(g[w] + 128 * sa + corr[w]) << (8 * w);  // each term is a true byte-product sum: >= 0, < 2^31
        val += (unsigned __int128)t << (8 * k);
      }
      sv[l][0][vl] = (uint32_t)val;
      sv[l][1][vl] = (uint32_t)(val >> 32);
      sv[l][2][vl] = (uint32_t)(val >> 64);
    }
  }
  __syncthreads();
  if (lq != 0 || v >= nvec) return;
  uint32_t *out = reinterpret_cast<uint32_t *>((v < io.osplit ? io.out[0] + (uint64_t)v * io.ostride : io.out[1] + (uint64_t)(v - io.osplit) * io.ostride) +
                                               (uint64_t)j * LL);
  const bool wide4 = (LL & 1) == 0 && (io.ostride & 1) == 0 && ((reinterpret_cast<uintptr_t>(io.out[0]) | reinterpret_cast<uintptr_t>(io.out[1])) & 15) == 0;
  uint32_t pend[4] = {0, 0, 0, 0};
  uint64_t carry = 0;
  const uint32_t *actw = io.add_ct ? reinterpret_cast<const uint32_t *>(io.add_ct + (uint64_t)j * LL) : nullptr;
  const uint64_t ax = io.add_ct ? io.add_scale[v] : 0;
  uint64_t acarry = 0;
  for (uint32_t l = 0; l < KWv; l++) {
    uint64_t word = carry + sv[l][0][vl];
    if (l >= 1) word += sv[l - 1][1][vl];
    if (l >= 2) word += sv[l - 2][2][vl];
    if (accumulate) word += out[l];
    if (actw) {  // + scale * ct: one 32 x 32 product per word, its high half carried into the next word
      const uint64_t t = (uint64_t)actw[l] * ax + acarry;
      word += (uint32_t)t;
      acarry = t >> 32;
    }
    carry = word >> 32;
    pend[l & 3] = (uint32_t)word;
    if (wide4 ? (l & 3) == 3 : (l & 1) == 1) {
      if (wide4) *reinterpret_cast<uint4 *>(out + (l & ~3u)) = uint4{pend[0], pend[1], pend[2], pend[3]};
      else *reinterpret_cast<uint2 *>(out + (l & ~1u)) = uint2{pend[(l & 2)], pend[(l & 2) + 1]};
    }
  }
  // modq: limbs >= K dropped (src/lwe.h:107-118)
  if (wide4) {
    if (KWv & 3) *reinterpret_cast<uint4 *>(out + (KWv & ~3u)) = uint4{pend[0], (KWv & 3) > 1 ? pend[1] : 0u, (KWv & 3) > 2 ? pend[2] : 0u, 0u};
    for (uint32_t l = (KWv + 3) & ~3u; l < 2 * LL; l += 4) *reinterpret_cast<uint4 *>(out + l) = uint4{0u, 0u, 0u, 0u};
  } else {
    if (KWv & 1) *reinterpret_cast<uint2 *>(out + (KWv & ~1u)) = uint2{pend[(KWv & 2)], 0u};
    for (uint32_t l = (KWv + 1) & ~1u; l < 2 * LL; l += 2) *reinterpret_cast<uint2 *>(out + l) = uint2{0u, 0u};
  }
}

template <int ND, int KWM, bool PK>
__global__ __launch_bounds__(256) void k_evalmm_finish(const int *__restrict__ part, const int64_t *__restrict__ sc, uint32_t nchunks, uint32_t ntiles,
                                                        uint32_t N, uint32_t nvec, uint32_t n, uint32_t nrows, uint32_t ct, uint32_t MBv, uint32_t sby,
                                                        uint32_t LL, MmIo io, int accumulate) {
  evalmm_finish_body<ND, KWM, PK>(part, sc, nchunks, ntiles, N, nvec, n, nrows, ct, MBv, sby, LL, io, accumulate, part, ND * nvec);
}
// all groups of a round in one launch: blockIdx.z = group, its partial products at part + group * part_stride
template <int ND, int KWM, bool PK>
__global__ __launch_bounds__(256) void k_evalmm_finish_groups(const int *__restrict__ part, uint64_t part_stride, uint32_t nchunks, uint32_t ntiles, uint32_t N,
                                                               uint32_t n, uint32_t nrows, uint32_t ct, uint32_t MBv, uint32_t sby, uint32_t LL, MmGroupArgs A,
                                                               int accumulate) {
  const uint32_t g = blockIdx.z, lender = A.io[g].sa_from1 ? A.io[g].sa_from1 - 1 : g;
  evalmm_finish_body<ND, KWM, PK>(part + g * part_stride, A.io[g].sc_zeroed, nchunks, ntiles, N, A.nvec[g], n, nrows, ct, MBv, sby, LL, A.io[g], accumulate,
                                  part + lender * part_stride, ND * A.nvec[lender]);
}

// ---- the witness pass of up to 32 statements as a GEMM over the SSP rows (one read of the SSP) ---------------------------------------
//   sum_b[k] = sum_i bit_b[i] * v_i[k]:   A = the statements' witness bits (0/1), B = the bytes of v_i[k] (offset by 128), K = rows.
// The SSP is row-major (v_i[k], k fastest) but the MFMA wants 16 consecutive ROWS per lane, so a second image of the SSP in B-fragment
// order is built once per SSP (k_ssp_frag, same size as the uint32 SSP): for row step K (32 rows), coefficient tile kt (32
// coefficients), byte w and lane (k = 32 kt + (l & 31), h = l >> 5): the 16 bytes [byte w of v_{32K+16h+e+1}[k]] ^ 0x80, e = 0..15, at
// frag[(((kt * KS + K) * 4 + w) * 64 + l) * 16 + e] (KS row steps: a coefficient tile's fragments are contiguous, so a wave reads one
// sequential stream -- with the row step outermost, 8 KiB pieces 4 MiB apart, the pass ran at 3.75 TB/s).  The pass is then a pure stream: four 16-byte loads and four 32x32x32 MFMAs
// (M = 32 statements) per wave and row step.
__global__ void k_ssp_frag(const uint32_t *__restrict__ ssp, uint32_t nrowsel, uint32_t d, uint32_t *__restrict__ frag) {
  // one thread = 4 rows x 1 coefficient -> one dword of each of the four byte columns
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t KT = d / 32;
  const uint32_t lane = gid & 63, eg = (gid >> 6) & 3;
  const uint64_t tile = gid >> 8;  // K * KT + kt
  const uint32_t kt = (uint32_t)(tile % KT), K = (uint32_t)(tile / KT);
  const uint32_t KSt = (nrowsel + 31) / 32;
  if (K >= KSt) return;
  const uint32_t k = kt * 32 + (lane & 31), rb = K * 32 + 16 * (lane >> 5) + 4 * eg;
  uint32_t x[4];
#pragma unroll
  for (int e = 0; e < 4; e++) x[e] = rb + e < nrowsel ? ssp[(uint64_t)(rb + e + 2) * d + k] ^ 0x80808080u : 0u;  // row r = v_{r+1} = slot r + 2
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t lo = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0400u + 0x00000101u * w);  // {x0.bw, x1.bw, 0, 0}
    const uint32_t hi = __builtin_amdgcn_perm(x[3], x[2], 0x04000c0cu + 0x01010000u * w);  // {0, 0, x2.bw, x3.bw}
    frag[(((((uint64_t)kt * KSt + K) * 4 + w) * 64 + lane) << 2) + eg] = lo | hi;
  }
}
// The witness kernels work on a RANGE of coefficients [col0, col0 + d) of the polynomials (the whole polynomial: col0 = 0, d = the SSP's
// d; a rank of the row-sharded batch prover computes its slice of every statement's w: mfh_witness_poly_mm_cols): `d` below is the width
// of the range (and of the partial arrays), WCols carries where it starts.
struct WCols {
  uint32_t kt0;      // col0 / 32: first 32-coefficient tile
  uint32_t KS;       // row steps of the whole SSP (the fragment image's tile stride)
  uint64_t wstride;  // coefficients between consecutive statements of the output
};
// grid = (d / 128, row chunks); block = 4 waves, one 32-coefficient tile each; MT = 1, 2 or 4 tiles of 32 statements (the SSP is read
// once per 32 MT statements).  part[((chunk * 4 + w) * 32 MT + stmt) * d + k].
template <int MT>
__global__ __launch_bounds__(256) void k_witness_mm(const v4i *__restrict__ sspfrag, const v4i *__restrict__ bitfrag, uint32_t nrowsel /* m - 1 */,
                                                    uint32_t ksteps_per_chunk, uint32_t d, int *__restrict__ part, WCols wc) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t r32 = lane & 31, h = lane >> 5;
  const uint32_t ktl = blockIdx.x * 4 + wave, kt = wc.kt0 + ktl;
  const uint32_t k = ktl * 32 + r32;  // (within the range)
  const uint32_t K0 = blockIdx.y * ksteps_per_chunk, K1 = min((nrowsel + 31) / 32, K0 + ksteps_per_chunk);
  v16i acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][w][e] = 0;
  if (K0 >= K1) return;  // (uniform)
  // PF row steps of fragments in flight per wave (MT = 4 holds 256 accumulator registers: one wave per SIMD, so the stream has to be
  // kept ahead by hand); loads past the chunk re-read its last step
  constexpr int PF = 3;
  v4i bq[PF][4], aq[PF][MT];
  auto fetch = [&](int slot, uint32_t K) {
    K = min(K, K1 - 1);
    const v4i *src = sspfrag + (((uint64_t)kt * wc.KS + K) * 4) * 64 + lane;
#pragma unroll
    for (int w = 0; w < 4; w++) bq[slot][w] = src[64 * w];
#pragma unroll
    for (int t = 0; t < MT; t++) aq[slot][t] = bitfrag[((uint64_t)K * MT + t) * 64 + lane];
  };
#pragma unroll
  for (int i = 0; i < PF; i++) fetch(i, K0 + i);
  for (uint32_t K = K0; K < K1; K += PF) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      if (K + i < K1) {
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
          for (int w = 0; w < 4; w++) acc[t][w] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[i][t], bq[i][w], acc[t][w], 0, 0, 0);
        fetch(i, K + i + PF);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const uint32_t stmt = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
        part[(((uint64_t)blockIdx.y * 4 + w) * (32 * MT) + stmt) * d + k] = acc[t][w][e];
      }
}
// The same pass over a GENERATOR-DEFINED SSP (csrc/ssp_prg.hpp; BASELINE configs 3/4, where the dense SSP would be 5.9 TB): the B
// fragments are not loaded but generated -- lane (coefficient k, row half h) hashes its 16 (row, k) pairs (9 integer operations each; the
// un-reduced 32-bit hash: sums of raw values and sums of coefficients agree mod p) and picks the four byte planes with v_perm -- so that a
// selected row is generated once per 32 MT statements instead of once per 12 (the VALU form, k_witness_partial_multi_prg): at 2^20
// constraints the witness pass of a statement drops from 19 ms to about 2.  rowkeys[r] = ssp_prg_rowkey(seed, slot r + 2), padded to a
// multiple of 32 rows.
__global__ void k_prg_rowkeys(uint64_t seed, uint32_t nrows_pad, uint32_t *__restrict__ rk) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrows_pad) rk[r] = mf::ssp_prg_rowkey(seed, r + 2);
}
template <int MT>
__global__ __launch_bounds__(256) void k_witness_mm_prg(const uint32_t *__restrict__ rowkeys, const v4i *__restrict__ bitfrag, uint32_t nrowsel /* m - 1 */,
                                                        uint32_t ksteps_per_chunk, uint32_t d, int *__restrict__ part, WCols wc) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t r32 = lane & 31, h = lane >> 5;
  const uint32_t kt = blockIdx.x * 4 + wave;
  const uint32_t k = kt * 32 + r32;  // (within the range)
  const uint32_t K0 = blockIdx.y * ksteps_per_chunk, K1 = min((nrowsel + 31) / 32, K0 + ksteps_per_chunk);
  v16i acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][w][e] = 0;
  if (K0 >= K1) return;  // (uniform)
  const uint32_t kc = wc.kt0 * 32 + k + 0x632BE5ABu;
  // the bit fragments and the row keys of a step are loaded two steps ahead (consumed in the step that issues them, the loads cost their
  // whole latency every step: 0.85 us per step against 0.35 of arithmetic); loads past the chunk re-read its last step
  constexpr int PF = 2;
  v4i aqr[PF][MT];
  uint4 rkr[PF][4];
  auto fetch = [&](uint32_t K, int slot) {
    K = min(K, K1 - 1);
#pragma unroll
    for (int t = 0; t < MT; t++) aqr[slot][t] = bitfrag[((uint64_t)K * MT + t) * 64 + lane];
    const uint4 *rk4 = reinterpret_cast<const uint4 *>(rowkeys + 32 * (uint64_t)K + 16 * h);
#pragma unroll
    for (int q = 0; q < 4; q++) rkr[slot][q] = rk4[q];
  };
  fetch(K0, 0);
  fetch(K0 + 1, 1);
  auto step = [&](uint32_t K, int slot) {
    v4i aq[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) aq[t] = aqr[slot][t];
    uint32_t x[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint4 r = rkr[slot][q];
      x[4 * q] = r.x; x[4 * q + 1] = r.y; x[4 * q + 2] = r.z; x[4 * q + 3] = r.w;
    }
    fetch(K + PF, slot);
#pragma unroll
    for (int e = 0; e < 16; e++) {  // mf::ssp_prg_raw(rowkey, k)
      uint32_t y = kc * x[e];
      y ^= y >> 16;
      y *= 0x7FEB352Du;
      y ^= y >> 15;
      y *= 0x846CA68Bu;
      y ^= y >> 16;
      x[e] = y;
    }
    v4i bq[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t lo = __builtin_amdgcn_perm(x[4 * j + 1], x[4 * j], 0x0c0c0400u + 0x00000101u * w);  // {x0.bw, x1.bw, 0, 0}
        const uint32_t hi = __builtin_amdgcn_perm(x[4 * j + 3], x[4 * j + 2], 0x04000c0cu + 0x01010000u * w);  // {0, 0, x2.bw, x3.bw}
        bq[w][j] = (int)((lo | hi) ^ 0x80808080u);
      }
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int w = 0; w < 4; w++) acc[t][w] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[t], bq[w], acc[t][w], 0, 0, 0);
  };
  uint32_t K = K0;
  for (; K + 2 <= K1; K += 2) {
    step(K, 0);
    step(K + 1, 1);
  }
  if (K < K1) step(K, 0);
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const uint32_t stmt = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
        part[(((uint64_t)blockIdx.y * 4 + w) * (32 * MT) + stmt) * d + k] = acc[t][w][e];
      }
}
// 256 statements per generation with TWO waves per SIMD.  The 8 statement tiles x 4 byte planes of a 32-coefficient tile (512 accumulator registers) go to
// four waves, 4 statement tiles x 2 planes each (128 registers; round 4 -- rounds 2-3 gave a wave all 8 statement tiles of ONE plane: 8 KiB of bit fragments
// + 1 KiB of coefficient bytes read from LDS per wave and 32-row step, 72 KiB per CU = 576 clk of the LDS pipe against 512 clk of MFMAs per SIMD: the pass was
// LDS-bound, a build without the MFMAs ran no faster; 4 + 2 KiB per wave are 384 clk).  The four waves SHARE the hashes four ways -- wave j hashes rows
// 4 j .. 4 j + 3 of a lane's 16 and publishes dword j of all four planes' fragments through LDS (a three-slot ring: the hashes of step K + 2 are issued between
// the MFMAs of step K, the fragments of step K + 1 are read during step K; the row keys are loaded four steps ahead), wave (sh, pp) reads the fragments of
// planes 2 pp and 2 pp + 1 with two 16-byte loads and the bit fragments of statement tiles 4 sh .. 4 sh + 3.  4 hashes and 8 MFMAs per wave and step, and
// with two waves per SIMD one wave's hashes run under the other's MFMAs.
// Chunk partials only (k_witness_mm_finish).  grid = (d / 64, row chunks), block = 8 waves = 2 coefficient tiles x (2 statement halves x 2 plane pairs).
__global__ __launch_bounds__(512) void k_witness_mm8q_prg(const uint32_t *__restrict__ rowkeys, const v4i *__restrict__ bitfrag, uint32_t nrowsel /* m - 1 */,
                                                          uint32_t ksteps_per_chunk, uint32_t d, int *__restrict__ part, WCols wc) {
  constexpr int MT = 8, RING = 4;
  __shared__ v4i bits[RING][MT][64];
  __shared__ uint32_t xch[3][2][4][64][4];  // [step % 3][tile][plane][lane][hashing wave]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t r32 = lane & 31, h = lane >> 5;
  const uint32_t tile = wave >> 2, pl = wave & 3;
  const uint32_t sh = pl >> 1, pp = pl & 1;        // the wave's statement half (tiles 4 sh ..) and plane pair (planes 2 pp, 2 pp + 1)
  // which 4 of a lane's 16 rows this wave hashes = which dword of the lane's fragments it publishes: rotated by the lane's 16-lane group, so that the 64
  // lanes of a ds_write_b32 into the [lane][4 dwords] slots hit 64 different banks (with pos = pl for every lane the addresses are 16 bytes apart: 16
  // distinct banks, every publish a 4-way conflict -- PMC: SQ_LDS_BANK_CONFLICT was 37 % of the LDS cycles of the pass)
  const uint32_t pos = (pl + (lane >> 4)) & 3;
  const uint32_t ktl = blockIdx.x * 2 + tile, kt = wc.kt0 + ktl;
  const uint32_t kc = kt * 32 + r32 + 0x632BE5ABu;
  const uint32_t K0 = blockIdx.y * ksteps_per_chunk, K1 = min((nrowsel + 31) / 32, K0 + ksteps_per_chunk);
  v16i acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][q][e] = 0;
  if (K0 >= K1) return;  // (uniform)
  auto bits_load = [&](uint32_t K) -> v4i { return bitfrag[(uint64_t)min(K, K1 - 1) * MT * 64 + tid]; };  // 512 elements per step: one per thread
  auto bits_store = [&](uint32_t K, v4i st) { (&bits[K % RING][0][0])[tid] = st; };
  auto rk_load = [&](uint32_t K) -> uint4 { return *reinterpret_cast<const uint4 *>(rowkeys + 32 * (uint64_t)min(K, K1 - 1) + 16 * h + 4 * pos); };
  auto hash1 = [&](uint32_t rowkey) -> uint32_t {  // mf::ssp_prg_raw(rowkey, k)
#ifdef WPRG_NOHASH  // timing-only build (wrong results): what the kernel costs without the generator's arithmetic
    return kc ^ rowkey;
#endif
    uint32_t y = kc * rowkey;
    y ^= y >> 16;
    y *= 0x7FEB352Du;
    y ^= y >> 15;
    y *= 0x846CA68Bu;
    y ^= y >> 16;
    return y;
  };
  auto publish = [&](uint32_t K, const uint32_t (&x)[4]) {  // dword `pos` of the four planes' fragments of step K
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t lo = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0400u + 0x00000101u * w);  // {x0.bw, x1.bw, 0, 0}
      const uint32_t hi = __builtin_amdgcn_perm(x[3], x[2], 0x04000c0cu + 0x01010000u * w);  // {0, 0, x2.bw, x3.bw}
      xch[K % 3][tile][w][lane][pos] = (lo | hi) ^ 0x80808080u;
    }
  };
  auto fragment = [&](uint32_t K, uint32_t q) -> v4i { return *reinterpret_cast<const v4i *>(&xch[K % 3][tile][2 * pp + q][lane][0]); };
  uint4 rkr[4];  // the row keys of steps K + 2 .. K + 5
  v4i sta, stb;  // the bit fragments of steps K + 1 / K + 2 on their way to the ring
  sta = bits_load(K0);
  bits_store(K0, sta);
  sta = bits_load(K0 + 1);
  stb = bits_load(K0 + 2);
  {
    const uint4 r0 = rk_load(K0), r1 = rk_load(K0 + 1);
    const uint32_t x0[4] = {hash1(r0.x), hash1(r0.y), hash1(r0.z), hash1(r0.w)};
    const uint32_t x1[4] = {hash1(r1.x), hash1(r1.y), hash1(r1.z), hash1(r1.w)};
    publish(K0, x0);
    publish(K0 + 1, x1);
  }
#pragma unroll
  for (int i = 2; i <= 5; i++) rkr[i & 3] = rk_load(K0 + i);
  __syncthreads();
  v4i bq0 = fragment(K0, 0), bq1 = fragment(K0, 1);
  uint32_t K = K0;
  auto step = [&](int slot, v4i &st) {  // st: the bit fragment of step K + 1 (loaded two steps ago); refilled with that of step K + 3
    const v4i bn0 = fragment(K + 1, 0), bn1 = fragment(K + 1, 1);  // (published a step ago, before the barrier)
    const uint4 rk = rkr[(slot + 2) & 3];  // step K + 2
    const uint32_t hr[4] = {rk.x, rk.y, rk.z, rk.w};
    uint32_t hx[4];
    bits_store(K + 1, st);
    const v4i *aq = &bits[K % RING][4 * sh][lane];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const v4i a = aq[t * 64];
#ifdef WPRG_NOMFMA  // timing-only build (wrong results): the generation, its LDS exchange and the barriers without the matrix cores
      acc[t][0][0] += a[0] ^ bq0[t];
      acc[t][1][0] += a[1] ^ bq1[t];
#else
      acc[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq0, acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq1, acc[t][1], 0, 0, 0);
#endif
      hx[t] = hash1(hr[t]);
    }
    publish(K + 2, hx);  // (slot last read during step K - 2, two barriers ago)
    rkr[(slot + 2) & 3] = rk_load(K + 6);
    bq0 = bn0;
    bq1 = bn1;
    st = bits_load(K + 3);
#ifndef WPRG_NOSYNC  // (timing-only build without it: wrong results -- what the step barrier costs)
    __syncthreads();
#endif
    K++;
  };
  while (K + 4 <= K1) {  // (K advances inside step)
    step(0, sta);
    step(1, stb);
    step(2, sta);
    step(3, stb);
  }
  if (K < K1) step(0, sta);
  if (K < K1) step(1, stb);
  if (K < K1) step(2, sta);
  uint32_t dd = d;
  asm volatile("" : "+s"(dd));  // (keeps the store addresses from being computed ahead of the loop)
#pragma unroll
  for (int q = 0; q < 2; q++) {
    int *dst = part + (((uint64_t)blockIdx.y * 4 + 2 * pp + q) * (32 * MT) + 128 * sh) * dd + ktl * 32 + r32 + (uint64_t)(4 * h) * dd;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int e = 0; e < 16; e++) dst[(uint64_t)((e & 3) + 8 * (e >> 2)) * dd] = acc[t][q][e];
      dst += (uint64_t)32 * dd;
    }
  }
}
// 256 statements (a whole super-group of 248) in ONE read of the dense SSP.  8 statement tiles x 4 byte planes are 512 accumulator
// registers per 32-coefficient tile: the four planes go to four waves (128 registers each, two waves per SIMD; a workgroup = 2
// coefficient tiles x 4 planes), the eight bit fragments of a row step -- 8 KiB, the same for all eight waves -- go through a four-slot
// LDS ring (fetched straight from L2 by every wave they made a first version L1-bound: 2.26 ms against 2 x 0.59 for two 124-statement
// passes) and are read from it a step ahead, under the previous step's MFMAs.  Vector-memory operations complete in issue order, so
// EVERY load of the loop is consumed exactly PF steps after its issue (the plane's fragment of step K + PF, the bit fragment of step
// K + 2 + PF, staged in registers and stored to the ring two steps ahead of its use), and the prologue issues its loads in the order
// the loop does, pinned: s_waitcnt vmcnt(n) is a static count of younger loads and the compiler takes the minimum over the paths into
// the loop (with the bit fragments staged two steps ahead, or all of them loaded first, it emitted vmcnt(4..9) where the steady state
// allows 12: the stream was awaited one or two steps after its issue whatever PF).  Timing-only builds split the pass: the stream
// alone 0.53 ms per 248 statements (5.4 TB/s), MFMAs + LDS alone 0.54, together 0.77 -- with one wave per SIMD (wave pairs, two
// planes each: the first version) as with two; the chip does not hold its clock under both.
// part == nullptr (one row chunk, m < 2^16): the four waves exchange their plane sums through LDS, one statement tile per round, and the
// tile's owner writes w_b[k] = delta_b t[k] + the byte sum mod p (what k_witness_mm_finish does from chunk partials: 0.5 GB written and
// read back per 248 statements otherwise).  grid = (d / 64, row chunks), block = 8 waves.
__global__ __launch_bounds__(512) void k_witness_mm8q(const v4i *__restrict__ sspfrag, const v4i *__restrict__ bitfrag, uint32_t nrowsel /* m - 1 */,
                                                      uint32_t ksteps_per_chunk, uint32_t d, int *__restrict__ part, const uint32_t *__restrict__ tpoly /* + col0 */,
                                                      const uint32_t *__restrict__ cnt_delta, uint32_t nstmt, uint32_t *__restrict__ w_out, WCols wc) {
  constexpr int MT = 8, RING = 4, PF = 4;
  __shared__ v4i bits[RING][MT][64];      // 32 KiB
  __shared__ uint32_t xch[2][3][16][64];  // the epilogue's exchange: [coefficient tile][sending wave (owner skipped)][e][lane], 24 KiB
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t r32 = lane & 31, h = lane >> 5;
  const uint32_t tile = wave >> 2, pl = wave & 3;
  const uint32_t ktl = blockIdx.x * 2 + tile, kt = wc.kt0 + ktl;
  const uint32_t K0 = blockIdx.y * ksteps_per_chunk, K1 = min((nrowsel + 31) / 32, K0 + ksteps_per_chunk);
  v16i acc[MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[t][e] = 0;
  if (K0 >= K1) return;  // (uniform)
  auto bits_load = [&](uint32_t K) -> v4i { return bitfrag[(uint64_t)min(K, K1 - 1) * MT * 64 + tid]; };  // 512 elements per step: one per thread
  auto bits_store = [&](uint32_t K, v4i st) { (&bits[K % RING][0][0])[tid] = st; };
  auto ssp_load = [&](uint32_t K) -> v4i { return sspfrag[(((uint64_t)kt * wc.KS + min(K, K1 - 1)) * 4 + pl) * 64 + lane]; };
  v4i stg[PF], bq[PF];
  {
    const v4i s0 = bits_load(K0), s1 = bits_load(K0 + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PF; i++) {  // (in the order the loop issues them)
      bq[i] = ssp_load(K0 + i);
      stg[i] = bits_load(K0 + 2 + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    bits_store(K0, s0);
    bits_store(K0 + 1, s1);
  }
  __syncthreads();
  v4i acur[MT];
#pragma unroll
  for (int t = 0; t < MT; t++) acur[t] = bits[K0 % RING][t][lane];
  uint32_t K = K0;
  auto step = [&](int slot) {
    bits_store(K + 2, stg[slot]);
    v4i anext[MT];
    const v4i *aq = &bits[(K + 1) % RING][0][lane];
#pragma unroll
    for (int t = 0; t < MT; t++) anext[t] = aq[t * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(acur[t], bq[slot], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bq[slot] = ssp_load(K + PF);
    stg[slot] = bits_load(K + 2 + PF);
#pragma unroll
    for (int t = 0; t < MT; t++) acur[t] = anext[t];
    __syncthreads();
    K++;
  };
  while (K + PF <= K1) {  // (K advances inside step)
#pragma unroll
    for (int i = 0; i < PF; i++) step(i);
  }
#pragma unroll
  for (int i = 0; i + 1 < PF; i++)
    if (K < K1) step(i);
  uint32_t dd = d;
  asm volatile("" : "+s"(dd));  // (keeps the store addresses from being computed ahead of the loop)
  if (part) {
    int *dst = part + ((uint64_t)blockIdx.y * 4 + pl) * (32 * MT) * dd + ktl * 32 + r32 + (uint64_t)(4 * h) * dd;
#pragma unroll
    for (int t = 0; t < MT; t++) {
#pragma unroll
      for (int e = 0; e < 16; e++) dst[(uint64_t)((e & 3) + 8 * (e >> 2)) * dd] = acc[t][e];
      dst += (uint64_t)32 * dd;
    }
    return;
  }
  // statement tile t is finished by wave t >> 1 of the coefficient tile: the other three hand over their plane sums (acc + 128 cnt_b:
  // the true byte sum, < 2^24 for m < 2^16), one statement tile per round
  asm volatile("" : "+s"(cnt_delta), "+s"(tpoly));
  const uint32_t k = ktl * 32 + r32;
  const uint64_t tk = tpoly[k], ws = wc.wstride, P = MFH_P;
#pragma unroll
  for (int t = 0; t < MT; t++) {
    const uint32_t owner = t >> 1;
    uint32_t mine[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const uint32_t b = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
      mine[e] = (uint32_t)acc[t][e] + (b < nstmt ? 128u * cnt_delta[2 * b] : 0u);
    }
    if (pl != owner) {  // (wave-uniform)
      const uint32_t sidx = pl - (pl > owner);
#pragma unroll
      for (int e = 0; e < 16; e++) xch[tile][sidx][e][lane] = mine[e];
    }
    __syncthreads();
    if (pl == owner) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const uint32_t b = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (b < nstmt) {
          uint64_t val = (uint64_t)mine[e] << (8 * owner);
#pragma unroll
          for (int o = 0; o < 4; o++)
            if (o != (int)owner) val += (uint64_t)xch[tile][o - (o > (int)owner)][e][lane] << (8 * o);
          w_out[(uint64_t)b * ws + k] = (uint32_t)((val % P + tk * cnt_delta[2 * b + 1] % P) % P);
        }
      }
    }
    __syncthreads();
  }
}
// bits of nstmt statements (packed, bits_stride bytes apart) -> A fragments: bitfrag[K][t][lane (stmt = 32 t + (l & 31), h)][e] = bit
// (32 K + 16 h + e) of that statement
// (one thread per lane's 16 bytes: two bytes of the statement's bit string in, one 16-byte store out)
__global__ void k_witness_bits(const uint8_t *__restrict__ bits, size_t bits_stride, uint32_t nstmt, uint32_t nrowsel, uint32_t ksteps, uint32_t MT,
                               int8_t *__restrict__ bitfrag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // one lane of one fragment: 16 output bytes
  if (i >= ksteps * MT * 64) return;
  const uint32_t lane = i & 63, t = (i >> 6) % MT, K = (i >> 6) / MT, stmt = 32 * t + (lane & 31), h = lane >> 5;
  const uint32_t r0 = K * 32 + 16 * h;  // rows r0 .. r0 + 15: bits of two consecutive bytes (r0 is a multiple of 16)
  uint32_t w = 0;
  if (stmt < nstmt && r0 < nrowsel) {
    const uint8_t *b = bits + (size_t)stmt * bits_stride + (r0 >> 3);
    w = b[0];
    if (r0 + 8 < nrowsel) w |= (uint32_t)b[1] << 8;
    if (nrowsel - r0 < 16) w &= (1u << (nrowsel - r0)) - 1;  // rows beyond the last selected one contribute nothing
  }
  uint32_t o[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t n4 = (w >> (4 * q)) & 15;  // four bits -> four bytes of 0 / 1
    o[q] = (n4 & 1) | ((n4 & 2) << 7) | ((n4 & 4) << 14) | ((n4 & 8) << 21);
  }
  reinterpret_cast<uint4 *>(bitfrag)[i] = uint4{o[0], o[1], o[2], o[3]};
}
// w_b[k] = delta_b t[k] + sum_i bit_b[i] v_i[k] mod p from the chunk partials: sum_w 256^w (G'_w + 128 cnt_b)
__global__ void k_witness_mm_finish(const int *__restrict__ part, uint32_t nchunks, const uint32_t *__restrict__ t, const uint32_t *__restrict__ cnt_delta,
                                    uint32_t nstmt, uint32_t mrows /* 32 MT */, uint32_t d, uint32_t *__restrict__ w_out, uint64_t wstride) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (k >= d || b >= nstmt) return;
  const uint64_t corr = 128ull * cnt_delta[2 * b];
  const uint32_t delta = cnt_delta[2 * b + 1];
  uint64_t val = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    int64_t g = 0;
    for (uint32_t ch = 0; ch < nchunks; ch++) g += part[(((uint64_t)ch * 4 + w) * mrows + b) * d + k];
    val += (uint64_t)(g + (int64_t)corr) << (8 * w);  // the true byte sum: >= 0
  }
  const uint64_t P = MFH_P;
  w_out[(uint64_t)b * wstride + k] = (uint32_t)((val % P + (uint64_t)t[k] * delta % P) % P);
}

}  // namespace

extern "C" {

// geometry of the 256-column kernels for the context's modulus
struct WideGeom { uint32_t ct, sby, vby, mt, mbp, LL; };
static WideGeom wide_geom(const mfh_ctx *c) {
  if (c->P.logq == 736) return {W16<736>::CT, W16<736>::SBY, W16<736>::VBY, W16<736>::MT, W16<736>::MBP, W16<736>::LL};
  return {W16<1472>::CT, W16<1472>::SBY, W16<1472>::VBY, W16<1472>::MT, W16<1472>::MBP, W16<1472>::LL};
}

// the epilogue kernels are instantiated per modulus: the LDS exchange holds 22 (logq 736) or 46 (1472) words per value
#define FINISH_LAUNCH(kern, ND_, PK_, grid, ...)                                                                         \
  do {                                                                                                                   \
    if ((PK_) && wg.sby / 4 <= 22) hipLaunchKernelGGL((kern<ND_, 22, true>), grid, dim3(256), 0, c->stream, __VA_ARGS__);  \
    else if (PK_) hipLaunchKernelGGL((kern<ND_, 46, true>), grid, dim3(256), 0, c->stream, __VA_ARGS__);                 \
    else if (wg.sby / 4 <= 22) hipLaunchKernelGGL((kern<ND_, 22, false>), grid, dim3(256), 0, c->stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL((kern<ND_, 46, false>), grid, dim3(256), 0, c->stream, __VA_ARGS__);                         \
  } while (0)

int mfh_eval_rows_multi(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, const uint32_t *d_coeffs, uint32_t nvec, uint32_t coeff_bytes,
                        uint64_t *d_rops, int accumulate) {
  if (!c || !d_rops || !nvec || (nrows && (!d_c8 || !d_coeffs)) || (coeff_bytes != 1 && coeff_bytes != 4)) return MFH_EINVAL;
  const MmIo io = {{d_coeffs, nullptr}, nvec, {d_rops, nullptr}, nvec, (uint64_t)(c->P.n + 1) * wide_geom(c).LL, nullptr, 0, nullptr};
  return eval_rows_multi_io(c, off, nrows, d_c8, io, nvec, coeff_bytes, accumulate);
}

}  // extern "C"

int eval_rows_multi_io(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, const MmIo &io, uint32_t nvec, uint32_t coeff_bytes, int accumulate) {
  if (!c || !nvec || !io.out[0] || (io.osplit < nvec && !io.out[1]) || (coeff_bytes != 1 && coeff_bytes != 4)) return MFH_EINVAL;
  if (nrows && (!d_c8 || (!io.bits && (!io.coef[0] || (io.csplit < nvec && !io.coef[1]))) || (io.bits && coeff_bytes != 1))) return MFH_EINVAL;
  const uint32_t ND = coeff_bytes;
  const uint32_t n = c->P.n;
  const bool q736 = c->P.logq == 736;
  const WideGeom wg = wide_geom(c);
  // 128 digit columns: k_evalmm (32x32x32 MFMA, 4-coordinate tiles, logq = 736 only); up to 256: k_evalmm16 (16x16x64, 2- or
  // 1-coordinate tiles), which needs every row segment to start at byte 0 or 8 of an AES block: stream offset and row length multiples of 8.
  // A registered matrix-core CRS image (mfh_crs_set_resident_mm) serves the region it was expanded from: always the 256-column layout.
  const uint8_t *img_region = nullptr;
  for (int r = 0; r < 3 && c->mm_image; r++)
    if (c->mm_off[r] == off && c->mm_rows[r] == nrows && nrows) img_region = c->mm_image + c->mm_base[r];
  const bool wide = img_region || !q736 || nvec * ND + 1 > 128;
  if (nvec * ND + 1 > 256) { c->err = "mfh_eval_rows_multi: at most 63 four-byte (255 one-byte) coefficient vectors per call (256 digit columns)"; return MFH_EINVAL; }
  if (wide && ((off & 7) || (((uint64_t)n * wg.vby) & 7))) {
    c->err = "mfh_eval_rows_multi: the 256-column kernel needs off and the row length to be multiples of 8";
    return MFH_EINVAL;
  }
  if (!c->have_seed) { c->err = "mfh_set_seed has not been called"; return MFH_EINVAL; }
  if (nrows > 0xffffffffu - 256) return MFH_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t ctl = (size_t)(n + 1) * wg.LL;
  if (io.ostride < ctl) return MFH_EINVAL;
  if (nrows == 0) {
    const uint32_t n0 = std::min(nvec, io.osplit);
    if (!accumulate) {
      HIP_TRY(c, hipMemset2DAsync(io.out[0], io.ostride * 8, 0, ctl * 8, n0, c->stream));
      if (n0 < nvec) HIP_TRY(c, hipMemset2DAsync(io.out[1], io.ostride * 8, 0, ctl * 8, nvec - n0, c->stream));
    }
    return MFH_OK;
  }
  const uint32_t ct = wide ? wg.ct : CT, mb = wide ? wg.mbp : CT * SB, rt = wide ? RT2 : RT;
  const uint32_t NT = wide ? NQ2 : (nvec * ND + 1 <= 64 ? 2 : 4), N = wide ? N2 : 32 * NT;
  const uint32_t ntiles = (n + 1 + ct - 1) / ct;
  // row chunks: 368 column tiles x 2 chunks = 736 workgroups = 2.9 rounds of the 256 CUs (one workgroup per CU at a time); an int32
  // accumulator holds 131 071 rows
  // (the 256-column kernels have 736 (1471) column tiles / 506 workgroups of row-tile pairs: one chunk already fills the CUs as evenly)
  uint32_t nchunks = (!wide && nrows >= 8 * rt) ? 2 : 1;
  // (the limit rounded DOWN to whole units first: rounding a chunk of <= 131 071 rows up to a unit afterwards could reach 131 072)
  const uint32_t lim = std::max(rt, c->mm_chunk_rows / rt * rt);
  nchunks = std::max<uint32_t>(nchunks, ((uint32_t)nrows + lim - 1) / lim);
  uint32_t rpc = ((uint32_t)nrows + nchunks - 1) / nchunks;
  rpc = (rpc + rt - 1) / rt * rt;  // <= lim
  nchunks = ((uint32_t)nrows + rpc - 1) / rpc;
  const uint32_t rpad = nchunks * rpc;  // a multiple of the unit: digit rows past nrows are zero
  const size_t cd_bytes = ((size_t)N * rpad + 255) & ~(size_t)255;
  const size_t sc_bytes = 256 * 8;
  const size_t part_bytes = (size_t)nchunks * ntiles * mb * N * 4;
  int rc = c->mm_ws_sel ? ws2_reserve(c, cd_bytes + sc_bytes + part_bytes) : ws_reserve(c, cd_bytes + sc_bytes + part_bytes);
  if (rc) return rc;
  uint8_t *wsp = (uint8_t *)(c->mm_ws_sel ? c->ws2 : c->ws);  // the launch on the side stream of mfh_prove_batch has its own scratch
  int8_t *cd = (int8_t *)wsp;
  int64_t *sc = io.sc_zeroed ? io.sc_zeroed : (int64_t *)(wsp + cd_bytes);
  int *part = (int *)(wsp + cd_bytes + sc_bytes);
  if (!io.sc_zeroed) HIP_TRY(c, hipMemsetAsync(sc, 0, sc_bytes, c->stream));
  if (wide && ND == 4 && !io.bits)
    hipLaunchKernelGGL(k_mm_digits4w, dim3((rpad / 16 + DG4 - 1) / DG4), dim3(256), 0, c->stream, io, nvec, (uint32_t)nrows, rpad, cd, sc);
  else
    hipLaunchKernelGGL(k_mm_digits, dim3((rpad / 16 + DG_RG - 1) / DG_RG), dim3(N), 0, c->stream, io, nvec, ND, (uint32_t)nrows, rpad, NT, wide ? 1 : 0, cd, sc);
  AesKey keyx = c->key;
  for (int i = 56; i < 60; i++) keyx.rk[i] ^= 0x80808080u;  // the kernel's keystream bytes come out as A - 128
  const uint32_t pk = img_region && c->mm_pack ? ND : 0u;  // the streaming kernel hands its partial products over recombined (mms_store_packed)
  {
    Timer t(c, img_region ? 8 : 7, nrows);
    if (img_region) {
      const uint32_t mtiles = ntiles * wg.mt, KS = ((uint32_t)nrows + RT2 - 1) / RT2 * (RT2 / 64);
      MmsImages imgs{};
      imgs.image[0] = (const v4i *)img_region;
      hipLaunchKernelGGL(k_mmstream1, dim3(((mtiles + TPW - 1) / TPW + 7) / 8 * 8, nchunks), dim3(SW * 64), 0, c->stream, imgs,
                         mtiles, KS, (uint32_t)nrows, rpc, (const v4i *)cd, part, pk);
    } else if (wide && q736)
      hipLaunchKernelGGL((k_evalmm16<0, 736>), dim3(ntiles, nchunks), dim3(1024), 0, c->stream, keyx, c->d_t0, off, n, (uint32_t)nrows, rpc, d_c8, cd, part,
                         (uint8_t *)nullptr);
    else if (wide)
      hipLaunchKernelGGL((k_evalmm16<0, 1472>), dim3(ntiles, nchunks), dim3(1024), 0, c->stream, keyx, c->d_t0, off, n, (uint32_t)nrows, rpc, d_c8, cd, part,
                         (uint8_t *)nullptr);
    else if (NT == 2)
      hipLaunchKernelGGL(k_evalmm<2>, dim3(ntiles, nchunks), dim3(1024), 0, c->stream, keyx, c->d_t0, off, n, (uint32_t)nrows, rpc, d_c8, cd, rpad,
                         part);
    else
      hipLaunchKernelGGL(k_evalmm<4>, dim3(ntiles, nchunks), dim3(1024), 0, c->stream, keyx, c->d_t0, off, n, (uint32_t)nrows, rpc, d_c8, cd, rpad,
                         part);
  }
  HIP_TRY(c, hipGetLastError());
  const uint32_t sby = wide ? wg.sby : SB;
  const dim3 fgrid(n + 1, (nvec + 63) / 64);
  if (ND == 4)
    FINISH_LAUNCH(k_evalmm_finish, 4, pk, fgrid, part, sc, nchunks, ntiles, N, nvec, n, (uint32_t)nrows, ct, mb, sby, wg.LL, io, accumulate);
  else
    FINISH_LAUNCH(k_evalmm_finish, 1, pk, fgrid, part, sc, nchunks, ntiles, N, nvec, n, (uint32_t)nrows, ct, mb, sby, wg.LL, io, accumulate);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

// ng evaluations over each of nreg regions (same row count) of the registered image in ONE k_mmstream launch: group r * ng + k streams
// region r with operands ios[r * ng + k] (the image is then read from HBM once for the ng groups of a region, see k_mmstream; the S and
// AS groups of the batch prover share a launch so that no two launches of the kernel overlap).  Three phases -- operand preparation
// (digit fragments + column sums), the streaming launch, the epilogues -- so that a caller can queue them on different streams
// (mfh_prove_batch: the epilogues of round r run beside the launch of round r + 1).
bool mms_plan(mfh_ctx *c, const MmRegion *regs, uint32_t nreg, size_t nrows, const MmIo *ios, const uint32_t *nvecs, uint32_t ng, uint32_t coeff_bytes,
              MmsPlan &P) {
  if (!c || !regs || !ios || !nvecs || !ng || !nreg || nreg > 2 || nreg * ng > 16) return false;
  bool ok = nreg * ng > 1 && c->have_seed && nrows && nrows <= 0xffffffffu - 256;
  for (uint32_t q = 0; q < nreg && ok; q++) {
    P.img[q] = nullptr;
    for (int r = 0; r < 3 && c->mm_image; r++)
      if (c->mm_off[r] == regs[q].off && c->mm_rows[r] == nrows) P.img[q] = c->mm_image + c->mm_base[r];
    ok = P.img[q] != nullptr;
  }
  P.ng = ng;
  P.ngt = nreg * ng;
  for (uint32_t g = 0; g < P.ngt && ok; g++) {
    ok = nvecs[g] && nvecs[g] * coeff_bytes + (ios[g].sa_from1 ? 0 : 1) <= 256 && ios[g].out[0] && ios[g].sc_zeroed && (ios[g].bits || ios[g].coef[0]);
    if (ok && ios[g].sa_from1) {  // borrowed ones column: a group of the same launch and region that carries its own; only the grouped digit / epilogue kernels know it
      const uint32_t l = ios[g].sa_from1 - 1;
      ok = l < P.ngt && l / ng == g / ng && !ios[l].sa_from1 && coeff_bytes == 4;
      for (uint32_t x = 0; x < P.ngt && ok; x++) ok = !ios[x].bits;  // (the per-group fallback kernels do not know about lenders)
    }
  }
  if (!ok) return false;
  const WideGeom wg = wide_geom(c);
  P.ND = coeff_bytes;
  P.nrows = (uint32_t)nrows;
  P.ntiles = (c->P.n + 1 + wg.ct - 1) / wg.ct;
  P.mtiles = P.ntiles * wg.mt;
  // row chunks: an int32 accumulator holds 131 071 rows (a rank's share of a 2^20-row region is 131 072)
  // (the limit rounded down to whole stages first, so that the rounded-up chunk cannot exceed it: 2 x 131 071 rows are three chunks)
  const uint32_t lim = std::max<uint32_t>(RT2, c->mm_chunk_rows / RT2 * RT2);
  uint32_t nchunks = ((uint32_t)nrows + lim - 1) / lim;
  P.rpc = (((uint32_t)nrows + nchunks - 1) / nchunks + RT2 - 1) / RT2 * RT2;  // <= lim
  P.nchunks = ((uint32_t)nrows + P.rpc - 1) / P.rpc;
  P.rpad = P.nchunks * P.rpc;
  P.cd_bytes = ((size_t)N2 * P.rpad + 255) & ~(size_t)255;
  P.part_bytes = ((size_t)P.nchunks * P.ntiles * wg.mbp * N2 * 4 + 255) & ~(size_t)255;
  P.cd = nullptr;
  P.part = nullptr;
  return true;
}
size_t mms_ws_bytes(const MmsPlan &P) { return (size_t)P.ngt * (P.cd_bytes + P.part_bytes); }
void mms_bind(MmsPlan &P, void *ws) {
  P.cd = (int8_t *)ws;
  P.part = (int *)((uint8_t *)ws + (size_t)P.ngt * P.cd_bytes);
}
static bool mms_group_args(const MmsPlan &P, const MmIo *ios, const uint32_t *nvecs, MmGroupArgs &A) {
  if (P.ND != 4) return false;
  for (uint32_t g = 0; g < P.ngt; g++) {
    if (ios[g].bits) return false;
    A.io[g] = ios[g];
    A.nvec[g] = nvecs[g];
  }
  return true;
}
int mms_digits(mfh_ctx *c, const MmsPlan &P, const MmIo *ios, const uint32_t *nvecs) {
  MmGroupArgs A;
  if (mms_group_args(P, ios, nvecs, A)) {  // one launch for the round's groups (8 launches of 21 us -> one of ~60)
    hipLaunchKernelGGL(k_mm_digits4w_groups, dim3((P.rpad / 16 + DG4 - 1) / DG4, P.ngt), dim3(256), 0, c->stream, A, P.nrows, P.rpad, P.cd, (uint64_t)P.cd_bytes);
    HIP_TRY(c, hipGetLastError());
    return MFH_OK;
  }
  for (uint32_t g = 0; g < P.ngt; g++) {
    if (P.ND == 4 && !ios[g].bits)
      hipLaunchKernelGGL(k_mm_digits4w, dim3((P.rpad / 16 + DG4 - 1) / DG4), dim3(256), 0, c->stream, ios[g], nvecs[g], P.nrows, P.rpad, P.cd + g * P.cd_bytes,
                         ios[g].sc_zeroed);
    else
      hipLaunchKernelGGL(k_mm_digits, dim3((P.rpad / 16 + DG_RG - 1) / DG_RG), dim3(N2), 0, c->stream, ios[g], nvecs[g], P.ND, P.nrows, P.rpad, (uint32_t)NQ2, 1,
                         P.cd + g * P.cd_bytes, ios[g].sc_zeroed);
  }
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}
// persistent grid or one workgroup per item (mfh_set_mm_stream), and in which form the launch hands its partial products to mms_finish
static bool mms_persistent(const mfh_ctx *c, const MmsPlan &P) {
  return P.ngt > 1 && c->mm_persist && c->ncu == 256 && ((c->mm_map && 32 % P.ng == 0) || 32 % P.ngt == 0);
}
static uint32_t mms_pk(const mfh_ctx *c, const MmsPlan &P) {
  if (!c->mm_pack || (mms_persistent(c, P) && c->mm_wave1)) return 0;  // (k_mmstream_w writes int32 only)
  return P.ND;
}
int mms_stream(mfh_ctx *c, const MmsPlan &P) {
  const uint32_t KS = (P.nrows + RT2 - 1) / RT2 * (RT2 / 64), tgs = (P.mtiles + TPW - 1) / TPW;
  const bool will_persist = mms_persistent(c, P);  // (the choice made below)
  Timer t(c, P.ngt > 1 ? (P.ND == 1 ? 14 : 10) : 8, P.nrows, (uint64_t)P.nrows * P.ngt, will_persist ? 1 : 0);  // kind 10 ("mmstream_rounds"): several groups per launch; 14 ("mmstream_bw"): b_w of several super-groups
  MmsImages imgs{};
  for (uint32_t g = 0; g < P.ngt; g++) imgs.image[g] = (const v4i *)P.img[g / P.ng];
  const uint32_t pk = mms_pk(c, P);  // (mms_finish reads what this launch writes: the same choice)
  if (P.ngt > 1) {
    // slot -> (group, tile group) map and grid shape (mfh_set_mm_stream): see mms_item / k_mmstream_p
    const uint32_t tgx = (tgs + 7) / 8;  // tile groups per XCD
    const uint32_t map = c->mm_map && 32 % P.ng == 0 ? 1u : 0u;
    const uint32_t slots = map ? (tgx + 32 / P.ng - 1) / (32 / P.ng) * (P.ngt / P.ng) * 32 : tgx * P.ngt;  // per XCD
    const bool persistent = will_persist;
    if (persistent && c->mm_wave1) {
      hipLaunchKernelGGL(k_mmstream_w, dim3(256), dim3(WSW * 64), 0, c->stream, imgs, P.mtiles, KS, P.nrows, P.rpc, (const v4i *)P.cd, P.part, P.ngt, P.ng, map,
                         (uint64_t)(P.cd_bytes / 16), (uint64_t)(P.part_bytes / 4), (slots + 31) / 32, P.nchunks);
    } else if (persistent) {
      if (!c->mm_sync) HIP_TRY(c, hipMalloc(&c->mm_sync, 8 * 32 * sizeof(uint32_t)));
      if (c->mm_sync_mode) HIP_TRY(c, hipMemsetAsync(c->mm_sync, 0, 8 * 32 * sizeof(uint32_t), c->stream));  // (the rendezvous counters; unused by default)
      const uint32_t width = P.ND == 1 ? 32u : c->mm_width;  // (b_w's HBM-bound launch keeps every CU)
      hipLaunchKernelGGL(P.ND == 1 ? k_mmstream_pb : k_mmstream_p, dim3(8 * width), dim3(SW * 64), 0, c->stream, imgs, P.mtiles, KS, P.nrows, P.rpc, (const v4i *)P.cd, P.part, P.ngt, P.ng, map,
                         (uint64_t)(P.cd_bytes / 16), (uint64_t)(P.part_bytes / 4), (slots + 31) / 32, P.nchunks, c->mm_sync, width == 32 ? c->mm_sync_mode : 0u, c->mm_spin, width, pk);
    } else {
      hipLaunchKernelGGL(k_mmstream, dim3(slots * 8, P.nchunks), dim3(SW * 64), 0, c->stream, imgs, P.mtiles, KS, P.nrows, P.rpc, (const v4i *)P.cd, P.part, P.ngt, P.ng,
                         map, (uint64_t)(P.cd_bytes / 16), (uint64_t)(P.part_bytes / 4), pk);
    }
  } else
    hipLaunchKernelGGL(k_mmstream1, dim3((tgs + 7) / 8 * 8, P.nchunks), dim3(SW * 64), 0, c->stream, imgs, P.mtiles, KS, P.nrows, P.rpc, (const v4i *)P.cd, P.part, pk);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}
int mms_finish(mfh_ctx *c, const MmsPlan &P, const MmIo *ios, const uint32_t *nvecs, int accumulate) {
  const WideGeom wg = wide_geom(c);
  const uint32_t n = c->P.n;
  const uint32_t pk = mms_pk(c, P);
  MmGroupArgs A;
  if (mms_group_args(P, ios, nvecs, A)) {
    uint32_t nvmax = 0;
    for (uint32_t g = 0; g < P.ngt; g++) nvmax = std::max(nvmax, nvecs[g]);
    FINISH_LAUNCH(k_evalmm_finish_groups, 4, pk, dim3(n + 1, (nvmax + 63) / 64, P.ngt), P.part, (uint64_t)(P.part_bytes / 4), P.nchunks, P.ntiles, (uint32_t)N2, n, P.nrows,
                  wg.ct, wg.mbp, wg.sby, wg.LL, A, accumulate);
    HIP_TRY(c, hipGetLastError());
    return MFH_OK;
  }
  for (uint32_t g = 0; g < P.ngt; g++) {
    const dim3 fgrid(n + 1, (nvecs[g] + 63) / 64);
    int *pg = P.part + g * (P.part_bytes / 4);
    if (P.ND == 4)
      FINISH_LAUNCH(k_evalmm_finish, 4, pk, fgrid, pg, ios[g].sc_zeroed, P.nchunks, P.ntiles, (uint32_t)N2, nvecs[g], n, P.nrows, wg.ct, wg.mbp, wg.sby, wg.LL, ios[g],
                    accumulate);
    else
      FINISH_LAUNCH(k_evalmm_finish, 1, pk, fgrid, pg, ios[g].sc_zeroed, P.nchunks, P.ntiles, (uint32_t)N2, nvecs[g], n, P.nrows, wg.ct, wg.mbp, wg.sby, wg.LL, ios[g],
                    accumulate);
  }
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}
int eval_rows_multi_io_regions(mfh_ctx *c, const MmRegion *regs, uint32_t nreg, size_t nrows, const MmIo *ios, const uint32_t *nvecs, uint32_t ng,
                               uint32_t coeff_bytes, int accumulate) {
  if (!c || !regs || !ios || !nvecs || !ng || !nreg || nreg * ng > 16) return MFH_EINVAL;
  MmsPlan P;
  if (!mms_plan(c, regs, nreg, nrows, ios, nvecs, ng, coeff_bytes, P)) {
    // anything else -- no image, a single evaluation, 128-column shapes -- : separate evaluations, region by region
    for (uint32_t g = 0; g < nreg * ng; g++) {
      int rc = eval_rows_multi_io(c, regs[g / ng].off, nrows, regs[g / ng].c8, ios[g], nvecs[g], coeff_bytes, accumulate);
      if (rc) return rc;
    }
    return MFH_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = c->mm_ws_sel ? ws2_reserve(c, mms_ws_bytes(P)) : ws_reserve(c, mms_ws_bytes(P));
  if (rc) return rc;
  mms_bind(P, c->mm_ws_sel ? c->ws2 : c->ws);
  rc = mms_digits(c, P, ios, nvecs);
  if (!rc) rc = mms_stream(c, P);
  if (!rc) rc = mms_finish(c, P, ios, nvecs, accumulate);
  return rc;
}
int eval_rows_multi_io_set(mfh_ctx *c, uint64_t off, size_t nrows, const uint8_t *d_c8, const MmIo *ios, const uint32_t *nvecs, uint32_t ng,
                           uint32_t coeff_bytes) {
  const MmRegion reg = {off, d_c8};
  return eval_rows_multi_io_regions(c, &reg, 1, nrows, ios, nvecs, ng, coeff_bytes, 0);
}
// true when mfh_eval_rows_multi over nrows rows at stream offset off would stream the registered image
bool mm_image_covers(const mfh_ctx *c, uint64_t off, size_t nrows) {
  for (int r = 0; r < 3 && c->mm_image; r++)
    if (c->mm_off[r] == off && c->mm_rows[r] == nrows && nrows) return true;
  return false;
}

extern "C" {

// ---- the CRS expanded once for the matrix-core path (second regime of SURVEY 8(d) for the batch prover) -------------------------
static size_t mm_region_bytes(const mfh_ctx *c, uint64_t rows) {  // row tiles x 64-row k-steps x 1 KiB fragments
  const WideGeom wg = wide_geom(c);
  const uint64_t mtiles = (uint64_t)((c->P.n + 1 + wg.ct - 1) / wg.ct) * wg.mt, ksteps = (rows + RT2 - 1) / RT2 * (RT2 / 64);
  return (size_t)(mtiles * ksteps * 1024);
}
// the three row ranges (absolute stream rows) of rank `rank`'s shares: S share | AS share | BT+BV share (the whole regions when world == 1)
struct MmShare { uint64_t row0[3], rows[3]; };
static MmShare mm_share(const mfh_ctx *c, uint32_t rank, uint32_t world) {
  const uint64_t d = c->P.d, m = c->P.m;
  const uint64_t loS = d * rank / world, cS = d * (rank + 1) / world - loS, lo = m * rank / world, cnt = m * (rank + 1) / world - lo;
  return MmShare{{loS, d + loS, 2 * d + lo}, {cS, cS, cnt}};
}
size_t mfh_crs_mm_share_bytes(const mfh_ctx *c, uint32_t rank, uint32_t world) {
  if (!c || !world || rank >= world) return 0;
  const MmShare sh = mm_share(c, rank, world);
  return mm_region_bytes(c, sh.rows[0]) + mm_region_bytes(c, sh.rows[1]) + mm_region_bytes(c, sh.rows[2]);
}
size_t mfh_crs_mm_image_bytes(const mfh_ctx *c) { return mfh_crs_mm_share_bytes(c, 0, 1); }
// expands rank `rank`'s shares of the S, AS and BT+BV regions of the compressed CRS into d_image (mfh_crs_mm_share_bytes bytes), in MFMA
// A-fragment order: every share is a region of its own (stream offset and row count), k-steps counted from its first row
int mfh_crs_expand_mm_share(mfh_ctx *c, const uint8_t *d_crs_c8, uint32_t rank, uint32_t world, uint8_t *d_image) {
  if (!c || !d_crs_c8 || !d_image || !world || rank >= world) return MFH_EINVAL;
  if (!c->have_seed) { c->err = "mfh_set_seed has not been called"; return MFH_EINVAL; }
  const WideGeom wg = wide_geom(c);
  const uint32_t n = c->P.n;
  if (((uint64_t)n * wg.vby) & 7) { c->err = "mfh_crs_expand_mm: the row length must be a multiple of 8"; return MFH_EUNSUPPORTED; }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t ctr_ct = (uint64_t)wg.vby * n;
  const MmShare sh = mm_share(c, rank, world);
  AesKey keyx = c->key;
  for (int i = 56; i < 60; i++) keyx.rk[i] ^= 0x80808080u;
  const uint32_t ntiles = (n + 1 + wg.ct - 1) / wg.ct;
  size_t base = 0;
  for (int r = 0; r < 3; r++) {
    if (!sh.rows[r]) continue;
    const uint32_t units = (uint32_t)((sh.rows[r] + RT2 - 1) / RT2);
    const uint32_t nchunks = std::max(1u, std::min(units, 4u));
    const uint32_t rpc = (units + nchunks - 1) / nchunks * RT2;
    Timer t(c, 9, sh.rows[r]);
    const dim3 grid(ntiles, (uint32_t)((sh.rows[r] + rpc - 1) / rpc));
    const uint8_t *c8 = d_crs_c8 + (size_t)sh.row0[r] * wg.vby;
    if (c->expand_path == 0 && ((ctr_ct * sh.row0[r]) & 7) == 0) {  // the barrier-free writer (expandmm.hip)
      int rc = expand_mm_region(c, ctr_ct * sh.row0[r], (uint32_t)sh.rows[r], c8, d_image + base);
      if (rc) return rc;
      base += mm_region_bytes(c, sh.rows[r]);
      continue;
    }
    if (c->P.logq == 736)
      hipLaunchKernelGGL((k_evalmm16<1, 736>), grid, dim3(1024), 0, c->stream, keyx, c->d_t0, ctr_ct * sh.row0[r], n, (uint32_t)sh.rows[r], rpc, c8,
                         (const int8_t *)nullptr, (int *)nullptr, d_image + base);
    else
      hipLaunchKernelGGL((k_evalmm16<1, 1472>), grid, dim3(1024), 0, c->stream, keyx, c->d_t0, ctr_ct * sh.row0[r], n, (uint32_t)sh.rows[r], rpc, c8,
                         (const int8_t *)nullptr, (int *)nullptr, d_image + base);
    base += mm_region_bytes(c, sh.rows[r]);
  }
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}
int mfh_crs_expand_mm(mfh_ctx *c, const uint8_t *d_crs_c8, uint8_t *d_image) { return mfh_crs_expand_mm_share(c, d_crs_c8, 0, 1, d_image); }
// rows per row chunk of the matrix-core launches: an int32 accumulator holds at most 131 071 rows (the default); smaller values split
// a region into more chunks (tuning / tests).  0 restores the default.
int mfh_set_mm_stream(mfh_ctx *c, int map, int persistent, int sync_mode, uint32_t spin_max) {
  if (!c || map < 0 || map > 1 || sync_mode < 0 || sync_mode > 2 || persistent < 0 || persistent > 2) return MFH_EINVAL;
  c->mm_map = map;
  c->mm_persist = persistent != 0;
  c->mm_wave1 = persistent == 2;  // the one-wave-per-SIMD body (k_mmstream_w)
  c->mm_sync_mode = (uint32_t)sync_mode;
  c->mm_spin = spin_max;
  return MFH_OK;
}
int mfh_set_mm_width(mfh_ctx *c, uint32_t per_xcd, int early_chain) {
  if (!c || per_xcd < 1 || per_xcd > 32) return MFH_EINVAL;
  c->mm_width = per_xcd;
  c->batch_early_chain = early_chain != 0;
  return MFH_OK;
}
int mfh_set_mm_pack(mfh_ctx *c, int on) {
  if (!c) return MFH_EINVAL;
  c->mm_pack = on != 0;
  return MFH_OK;
}
int mfh_set_mm_chunk_rows(mfh_ctx *c, uint32_t rows) {
  if (!c || rows > 131071) return MFH_EINVAL;
  c->mm_chunk_rows = rows ? rows : 131071;
  return MFH_OK;
}
// registers (or, with NULL, clears) the image: mfh_eval_rows_multi / mfh_prove_batch* then stream the three row ranges it holds from it
int mfh_crs_set_resident_mm_share(mfh_ctx *c, const uint8_t *d_image, uint32_t rank, uint32_t world) {
  if (!c || !world || rank >= world) return MFH_EINVAL;
  c->mm_image = d_image;
  c->mm_rank = rank;
  c->mm_world = world;
  const uint64_t ctr_ct = (uint64_t)wide_geom(c).vby * c->P.n;
  const MmShare sh = mm_share(c, rank, world);
  size_t base = 0;
  for (int r = 0; r < 3; r++) {
    c->mm_off[r] = ctr_ct * sh.row0[r];
    c->mm_rows[r] = sh.rows[r];
    c->mm_base[r] = base;
    base += mm_region_bytes(c, sh.rows[r]);
  }
  return MFH_OK;
}
int mfh_crs_set_resident_mm(mfh_ctx *c, const uint8_t *d_image) { return mfh_crs_set_resident_mm_share(c, d_image, 0, 1); }

// mfh_witness_poly for up to 256 statements in ONE read (dense SSP) or one generation (generator-defined SSP) of the selected rows, on the
// matrix cores, restricted to the coefficients [col0, col0 + ncols): d_w[b * w_stride + (k - col0)]
int mfh_witness_poly_mm_cols(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                             uint32_t col0, uint32_t ncols, uint32_t *d_w, size_t w_stride) {
  if (!c || !h_bits || !h_delta || !d_w || nstmt == 0 || nstmt > 256) return MFH_EINVAL;
  mf::SspSrc src;  // d_ssp == NULL: the registered generator-defined SSP (B fragments generated in the kernel)
  {
    int rc0 = ssp_src(c, d_ssp, src);
    if (rc0) return rc0;
  }
  const uint32_t MT = nstmt > 128 ? 8 : nstmt > 64 ? 4 : nstmt > 32 ? 2 : 1;
  const uint32_t d = c->P.d, m = c->P.m;
  if (d % 128 || m < 2) { c->err = "mfh_witness_poly_mm: d must be a multiple of 128"; return MFH_EUNSUPPORTED; }
  if ((uint64_t)col0 + ncols > d || w_stride < ncols) return MFH_EINVAL;
  if (ncols == 0) return MFH_OK;
  if (col0 % 128 || ncols % 128) { c->err = "mfh_witness_poly_mm_cols: the coefficient range must start and end at multiples of 128"; return MFH_EUNSUPPORTED; }
  const uint32_t nc = ncols;
  const WCols wc = {col0 / 32, (m - 1 + 31) / 32, (uint64_t)w_stride};
  const uint32_t *tpoly = src.t + col0;
  for (uint32_t b = 0; b < nstmt; b++)
    if (h_delta[b] >= MFH_P) { c->err = "delta must be < p"; return MFH_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t nrowsel = m - 1, ksteps = (nrowsel + 31) / 32;
  // (the 256-statement pass has d / 64 workgroups of one wave per SIMD: one row chunk fills the chip at d >= 2^14, and with one chunk --
  // and byte sums that fit 32 bits per plane pair -- it finishes in the kernel)
  const bool fused = MT == 8 && src.dense && m < 65536 && (nc >= 16384 || ksteps <= 64);  // (small instances: nothing to fill either way)
  const uint32_t nchunks = fused ? 1u : std::min(ksteps, 4u), kpc = (ksteps + nchunks - 1) / nchunks;
  // the SSP in B-fragment order: built on first use per SSP (mfh_ssp_prepare invalidates it), kept beside the uint32 image
  const size_t sfrag_b = (size_t)ksteps * 32 * d * 4;
  if (src.dense && (c->ssp_frag_src != d_ssp || c->ssp_frag_bytes < sfrag_b)) {
    if (c->ssp_frag_bytes < sfrag_b) {
      if (c->ssp_frag) { hipStreamSynchronize(c->stream); hipFree(c->ssp_frag); c->ssp_frag = nullptr; c->ssp_frag_bytes = 0; }
      HIP_TRY(c, hipMalloc(&c->ssp_frag, sfrag_b));
      c->ssp_frag_bytes = sfrag_b;
    }
    const uint64_t nthreads = (uint64_t)ksteps * (d / 32) * 256;
    hipLaunchKernelGGL(k_ssp_frag, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, c->stream, d_ssp, nrowsel, d, (uint32_t *)c->ssp_frag);
    HIP_TRY(c, hipGetLastError());
    c->ssp_frag_src = d_ssp;
  }
  const size_t packed = (size_t)nstmt * bits_stride, head_b = ((packed + 8 + 256 * 8 + 255) & ~(size_t)255);
  const size_t frag_b = (size_t)ksteps * MT * 1024, part_b = (size_t)nchunks * 4 * 32 * MT * nc * 4;
  const size_t rk_b = src.dense ? 0 : (((size_t)ksteps * 32 * 4 + 255) & ~(size_t)255);
  int rc = wws_reserve(c, head_b + frag_b + part_b + rk_b);
  if (rc) return rc;
  // staged: packed bits, then (count of selected rows, delta) per statement
  PinBuf &wpin = c->pin_wring[c->pin_wnext++ % 8];
  uint8_t *stage = (uint8_t *)pin_acquire(c, wpin, head_b);
  if (!stage) return MFH_ENOMEM;
  memcpy(stage, h_bits, packed);
  uint32_t *cd = (uint32_t *)(stage + packed + ((8 - packed % 8) % 8));
  for (uint32_t b = 0; b < nstmt; b++) {
    uint32_t cnt = 0;
    const uint8_t *hb = h_bits + (size_t)b * bits_stride;
    for (uint32_t r = 0; r + 8 <= nrowsel; r += 8) cnt += (uint32_t)__builtin_popcount(hb[r >> 3]);
    for (uint32_t r = nrowsel & ~7u; r < nrowsel; r++) cnt += (hb[r >> 3] >> (r & 7)) & 1;
    cd[2 * b] = cnt;
    cd[2 * b + 1] = h_delta[b];
  }
  uint8_t *dev = (uint8_t *)c->wws;
  HIP_TRY(c, hipMemcpyAsync(dev, stage, head_b, hipMemcpyHostToDevice, c->stream));
  pin_release(c, wpin);
  const uint32_t *d_cd = (const uint32_t *)(dev + packed + ((8 - packed % 8) % 8));
  int8_t *d_frag = (int8_t *)(dev + head_b);
  int *d_part = (int *)(dev + head_b + frag_b);
  hipLaunchKernelGGL(k_witness_bits, dim3((uint32_t)((frag_b / 16 + 255) / 256)), dim3(256), 0, c->stream, dev, bits_stride, nstmt, nrowsel, ksteps, MT, d_frag);
  if (!src.dense) {
    uint32_t *d_rk = (uint32_t *)(dev + head_b + frag_b + part_b);
    const dim3 grid(nc / 128, (ksteps + kpc - 1) / kpc);
    hipLaunchKernelGGL(k_prg_rowkeys, dim3((ksteps * 32 + 255) / 256), dim3(256), 0, c->stream, src.seed, ksteps * 32, d_rk);
    if (MT == 8)
      hipLaunchKernelGGL(k_witness_mm8q_prg, dim3(nc / 64, (ksteps + kpc - 1) / kpc), dim3(512), 0, c->stream, d_rk, (const v4i *)d_frag, nrowsel, kpc, nc, d_part, wc);
    else if (MT == 1) hipLaunchKernelGGL(k_witness_mm_prg<1>, grid, dim3(256), 0, c->stream, d_rk, (const v4i *)d_frag, nrowsel, kpc, nc, d_part, wc);
    else if (MT == 2) hipLaunchKernelGGL(k_witness_mm_prg<2>, grid, dim3(256), 0, c->stream, d_rk, (const v4i *)d_frag, nrowsel, kpc, nc, d_part, wc);
    else hipLaunchKernelGGL(k_witness_mm_prg<4>, grid, dim3(256), 0, c->stream, d_rk, (const v4i *)d_frag, nrowsel, kpc, nc, d_part, wc);
  } else if (MT == 8) {
    hipLaunchKernelGGL(k_witness_mm8q, dim3(nc / 64, (ksteps + kpc - 1) / kpc), dim3(512), 0, c->stream, (const v4i *)c->ssp_frag, (const v4i *)d_frag, nrowsel, kpc, nc,
                       fused ? (int *)nullptr : d_part, tpoly, d_cd, nstmt, d_w, wc);
    if (fused) {
      HIP_TRY(c, hipGetLastError());
      return MFH_OK;
    }
  } else if (MT == 1)
    hipLaunchKernelGGL(k_witness_mm<1>, dim3(nc / 128, (ksteps + kpc - 1) / kpc), dim3(256), 0, c->stream, (const v4i *)c->ssp_frag, (const v4i *)d_frag,
                       nrowsel, kpc, nc, d_part, wc);
  else if (MT == 2)
    hipLaunchKernelGGL(k_witness_mm<2>, dim3(nc / 128, (ksteps + kpc - 1) / kpc), dim3(256), 0, c->stream, (const v4i *)c->ssp_frag, (const v4i *)d_frag,
                       nrowsel, kpc, nc, d_part, wc);
  else
    hipLaunchKernelGGL(k_witness_mm<4>, dim3(nc / 128, (ksteps + kpc - 1) / kpc), dim3(256), 0, c->stream, (const v4i *)c->ssp_frag, (const v4i *)d_frag,
                       nrowsel, kpc, nc, d_part, wc);
  hipLaunchKernelGGL(k_witness_mm_finish, dim3((nc + 255) / 256, nstmt), dim3(256), 0, c->stream, d_part, (ksteps + kpc - 1) / kpc, tpoly, d_cd, nstmt, 32 * MT,
                     nc, d_w, (uint64_t)w_stride);
  HIP_TRY(c, hipGetLastError());
  return MFH_OK;
}

int mfh_witness_poly_mm(mfh_ctx *c, const uint32_t *d_ssp, uint32_t nstmt, const uint8_t *h_bits, size_t bits_stride, const uint32_t *h_delta,
                        uint32_t *d_w) {
  if (!c) return MFH_EINVAL;
  return mfh_witness_poly_mm_cols(c, d_ssp, nstmt, h_bits, bits_stride, h_delta, 0, c->P.d, d_w, c->P.d);
}

}  // extern "C"
