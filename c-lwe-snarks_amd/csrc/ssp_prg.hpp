// ssp_prg.hpp -- a Square Span Program whose wire polynomials are DEFINED by a counter-based generator instead of stored.
//
// The reference's SSP is a dense buffer of (M+3) x D uint64 coefficients (src/ssp.h:6-9): 5.7 GB at the default instance
// and 5.9 TB at the 2^20-constraint instance of BASELINE configs 4/5, which therefore cannot exist in memory.  SURVEY 8(d)
// prescribes "SSP coefficients defined by a counter-based PRG (not stored)" for those configs.  Here:
//     v_i[k] = coefficient(seed, slot = i + 1, k)   for slots >= 1   (same slot numbering as the dense layout)
//     t      = stored (slot 0: D uint32 coefficients), t = v_0 + sum_{w_i = 1} v_i - 1 as random_ssp builds it (src/ssp.c:59-71)
// The generator is a 32-bit integer hash (two multiplies, three xor-shifts) of (seed, slot, k), reduced into [0, p).  It only has
// to be cheap, deterministic and well mixed: it stands in for the reference's getrandom() coefficients (src/ssp.c:56,62).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MF_HD __host__ __device__ __forceinline__
#else
#define MF_HD inline
#endif

namespace mf {

MF_HD uint32_t ssp_prg_rowkey(uint64_t seed, uint32_t slot) {  // per-row part, uniform across a row
  uint32_t r = (uint32_t)seed + slot * 0x9E3779B1u;
  r ^= r >> 15;
  r *= 0x2C1B3C6Du;
  r ^= (uint32_t)(seed >> 32);
  return r | 1u;
}
// the un-reduced 32-bit hash; coefficient = raw mod p.  Sums of coefficients may be formed from the raw values and reduced once:
// raw and coefficient differ by a multiple of p (0 or p).
MF_HD uint32_t ssp_prg_raw(uint32_t rowkey, uint32_t k) {
  uint32_t x = (k + 0x632BE5ABu) * rowkey;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
MF_HD uint32_t ssp_prg_coeff(uint32_t rowkey, uint32_t k) {
  const uint32_t x = ssp_prg_raw(rowkey, k);
  return x >= 0xfffffffbu ? x - 0xfffffffbu : x;
}

// where a kernel takes its SSP coefficients from: a dense uint32 [(m+3)][d] image, or the generator (+ stored slot 0)
struct SspSrc {
  const uint32_t *dense;  // non-null: dense image
  const uint32_t *t;      // generator mode: slot 0
  uint64_t seed;
};

}  // namespace mf
