// Multi-word bit-parallel LCS for strings of 65..256 code units (K = 2 or 4 words of 64 bits).
// Shared by the RAW and the levels-mode Indel grids.  Real NAPKON `Term` strings (header + question
// + parameter, reference: napkon_string_matching/types/questionnaire.py:59-68) routinely exceed 64
// code units, so this is the path the reference's default configuration (compare_column: Term,
// score_func: fuzzy_match; config.yml:13-14) takes; the one-word kernels are the fast case.
//
// Layout per wavefront in LDS:
//   pm    [pm_stride][K] u64   match masks of the wave-uniform pattern, K words per symbol
//   text  [16 K][64]     u32   the lanes' texts, 4 code units per dword, column = lane (conflict-free)
// Hyyro's update with a carry chain across the words:
//   U_k = V_k & PM[c][k];  (V + U) over K words with carry;  V_k = sum_k | (V_k ^ U_k)
#pragma once
#include "nsm_common.hpp"

namespace nsm {

constexpr uint16_t kNeverWide = 0xffff;

// Build the wave's match-mask table for the pattern row `codes` (la code units).
template <int K>
__device__ __forceinline__ void wide_build_pm(unsigned long long* pm, int pm_stride, const uint8_t* __restrict__ codes,
                                              int la, int lane) {
  for (int c = lane; c < pm_stride * K; c += kWave) pm[c] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int pos = lane + kWave * k;
    if (pos < la) {
      const unsigned c = codes[pos];
      atomicOr(&pm[c * K + k], 1ull << lane);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Copy the lane's text row (64 K code units at `codes`) into the wave's LDS text image.
template <int K>
__device__ __forceinline__ void wide_store_text(uint32_t* text, const uint8_t* __restrict__ codes, int lane) {
  const uint4* tp = reinterpret_cast<const uint4*>(codes);
#pragma unroll
  for (int q = 0; q < 4 * K; ++q) {
    const uint4 v = tp[q];
    text[(4 * q + 0) * kWave + lane] = v.x;
    text[(4 * q + 1) * kWave + lane] = v.y;
    text[(4 * q + 2) * kWave + lane] = v.z;
    text[(4 * q + 3) * kWave + lane] = v.w;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// LCS length of the wave's pattern (masks in pm) and the lane's text (first nchars code units, the
// wave's longest text; shorter texts are padded with the all-zero-mask symbol).
template <int K>
__device__ __forceinline__ int wide_lcs(const unsigned long long* pm, const uint32_t* text, int nchars, int lane) {
  unsigned long long v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = ~0ull;
  for (int w = 0; 4 * w < nchars; ++w) {
    const uint32_t word = text[w * kWave + lane];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const unsigned c = (word >> (8 * b)) & 0xffu;
      const unsigned long long* e = pm + c * K;
      unsigned long long carry = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const unsigned long long u = v[k] & e[k];
        const unsigned long long t = __builtin_addcll(v[k], u, carry, &carry);
        v[k] = t | (v[k] ^ u);
      }
    }
  }
  int ones = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) ones += __popcll(v[k]);
  return kWave * K - ones;
}

}  // namespace nsm
