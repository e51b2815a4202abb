// Multi-word bit-parallel LCS for strings of 65..512 code units (K = 2, 4 or 8 words of 64 bits).
// Shared by the RAW and the levels-mode Indel grids.  Real NAPKON `Term` strings (header + question
// + parameter, reference: napkon_string_matching/types/questionnaire.py:59-68) routinely exceed 64
// code units, so this is the path the reference's default configuration (compare_column: Term,
// score_func: fuzzy_match; config.yml:13-14) takes; the one-word kernels are the fast case.
//
// Layout per wavefront in LDS:
//   pm    [pm_stride][K + 1] u64   match masks of the wave-uniform pattern, K words per symbol + 1 of padding
//   text  [16 K][64]     u32   the lanes' texts, 4 code units per dword, column = lane (conflict-free)
// Hyyro's update with a carry chain across the words:
//   U_k = V_k & PM[c][k];  (V + U) over K words with carry;  V_k = sum_k | (V_k ^ U_k)
#pragma once
#include "nsm_common.hpp"

namespace nsm {

constexpr uint16_t kNeverWide = 0xffff;

// One 64-bit word of a mask table in LDS, as a VOLATILE load: the compiler then keeps it a single ds_read_b64 (2 LDS
// cycles for 64 lanes) instead of fusing two into a ds_read2_b64, which the LDS serves at half the rate
// (MI355X_MICROARCH.md, LDS table; round 3: the shared-tile levels kernel went from 51.1 to 42.0 ms with this alone).
__device__ __forceinline__ unsigned long long lds_word(const unsigned long long* p) {
  return *(const volatile __attribute__((address_space(3))) unsigned long long*)(p);
}

// 64-bit words per mask-table entry.  K > 1: one word of padding, so that the entries of different symbols
// start in different LDS banks (entry stride (K + 1) * 2 dwords: 32 symbols without a conflict; at stride
// 2 K dwords symbols c and c + 32 / K collide -- measured on Term-like strings: 62 % of the LDS cycles of the
// levels kernel were bank conflicts of these reads)
template <int K>
constexpr int kPmWords = K == 1 ? 1 : K + 1;

// Build the wave's match-mask table for the pattern row `codes` (la code units).
template <int K>
__device__ __forceinline__ void wide_build_pm(unsigned long long* pm, int pm_stride, const uint8_t* __restrict__ codes,
                                              int la, int lane) {
  for (int c = lane; c < pm_stride * kPmWords<K>; c += kWave) pm[c] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int pos = lane + kWave * k;
    if (pos < la) {
      const unsigned c = codes[pos];
      atomicOr(&pm[c * kPmWords<K> + k], 1ull << lane);
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

// t = v + u over W 64-bit words as one v_add_co / v_addc_co chain (the compiler's own expansion is a
// 64-bit add, a 64-bit compare and a select per word)
template <int W>
__device__ __forceinline__ void add_chain(const unsigned long long (&v)[W], const unsigned long long (&u)[W],
                                          unsigned long long (&t)[W]) {
  uint32_t a[2 * W], b[2 * W], d[2 * W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    a[2 * k] = static_cast<uint32_t>(v[k]); a[2 * k + 1] = static_cast<uint32_t>(v[k] >> 32);
    b[2 * k] = static_cast<uint32_t>(u[k]); b[2 * k + 1] = static_cast<uint32_t>(u[k] >> 32);
  }
  if constexpr (W == 1) {
    asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
        : "=&v"(d[0]), "=&v"(d[1]) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]) : "vcc");
  } else if constexpr (W == 2) {
    asm("v_add_co_u32 %0, vcc, %4, %8\n\tv_addc_co_u32 %1, vcc, %5, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %6, %10, vcc\n\tv_addc_co_u32 %3, vcc, %7, %11, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
  } else if constexpr (W == 3) {
    asm("v_add_co_u32 %0, vcc, %6, %12\n\tv_addc_co_u32 %1, vcc, %7, %13, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %8, %14, vcc\n\tv_addc_co_u32 %3, vcc, %9, %15, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %10, %16, vcc\n\tv_addc_co_u32 %5, vcc, %11, %17, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]),
          "v"(b[3]), "v"(b[4]), "v"(b[5]) : "vcc");
  } else if constexpr (W > 4) {
    // 5..8 words (strings beyond 256 code units, rare): the compiler's carry chain
    unsigned long long carry = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) t[k] = __builtin_addcll(v[k], u[k], carry, &carry);
    return;
  } else {
    static_assert(W == 4, "1..8 words");
    asm("v_add_co_u32 %0, vcc, %8, %16\n\tv_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\tv_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\tv_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\tv_addc_co_u32 %7, vcc, %15, %23, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]),
          "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
  }
#pragma unroll
  for (int k = 0; k < W; ++k) t[k] = (static_cast<unsigned long long>(d[2 * k + 1]) << 32) | d[2 * k];
}

// LCS length of the wave's pattern (masks in pm, K words per symbol of which the first W are live:
// W = ceil(la / 64), wave-uniform) and the lane's text (first nchars code units, the wave's longest
// text; shorter texts are padded with the all-zero-mask symbol).
//
// EARLY: `lb` is the lane's text length and `need` the smallest LCS that still matters to it.  Every
// 8 code units the wave checks LCS-so-far + code units still to come >= need; when no lane can reach
// its `need` the scan stops and a value below every lane's `need` is returned (exact: one text code
// unit adds at most one to the LCS).
template <int K, int W, bool EARLY>
__device__ __forceinline__ int wide_lcs_words(const unsigned long long* pm, const uint32_t* text, int nchars,
                                              int lane, int lb, int need) {
  // Software-pipelined: the text dword of iteration w + 1 (W <= 2: w + 2) and, for W <= 2, the masks of
  // iteration w + 1 are requested before the recurrence of iteration w runs, so the two dependent LDS
  // latencies of a step (text dword -> mask words) overlap the arithmetic instead of stalling it
  // (measured on the levels kernels: one LDS round trip per dependent step left the waves waiting ~100
  // cycles per 2-4 code units).  Reads past the last text dword are clamped to it (results unused).
  constexpr bool kDeep = W <= 2;
  unsigned long long v[W];
#pragma unroll
  for (int k = 0; k < W; ++k) v[k] = ~0ull;
  const int nw = (nchars + 3) >> 2;
  if (nw == 0) return 0;
  auto load_masks = [&](uint32_t word, unsigned long long (&m)[4][W]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const unsigned long long* e = pm + ((word >> (8 * b)) & 0xffu) * kPmWords<K>;
#pragma unroll
      for (int k = 0; k < W; ++k) m[b][k] = lds_word(e + k);
    }
  };
  uint32_t w_next = text[lane];                            // dword of iteration 0
  uint32_t w_next2 = text[min(1, nw - 1) * kWave + lane];  // dword of iteration 1
  unsigned long long m_cur[4][W];
  if (kDeep) load_masks(w_next, m_cur);
  for (int w = 0; w < nw; ++w) {
    const uint32_t word = w_next;
    w_next = w_next2;
    w_next2 = text[min(w + 2, nw - 1) * kWave + lane];
    unsigned long long m_nx[4][W];
    if (kDeep) load_masks(w_next, m_nx);
    else load_masks(word, m_cur);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      unsigned long long u[W], t[W];
#pragma unroll
      for (int k = 0; k < W; ++k) u[k] = v[k] & m_cur[b][k];
      add_chain<W>(v, u, t);
#pragma unroll
      for (int k = 0; k < W; ++k) v[k] = t[k] | (v[k] ^ u[k]);
    }
    if (kDeep) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < W; ++k) m_cur[b][k] = m_nx[b][k];
    }
    if (EARLY && (w & 1)) {
      int ones = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) ones += __popcll(v[k]);
      const int reach = kWave * W - ones + max(0, lb - 4 * (w + 1));
      if (!__any(reach >= need)) return -1;
    }
  }
  int ones = 0;
#pragma unroll
  for (int k = 0; k < W; ++k) ones += __popcll(v[k]);
  return kWave * W - ones;
}

template <int K, bool EARLY = false>
__device__ __forceinline__ int wide_lcs(const unsigned long long* pm, const uint32_t* text, int nchars, int lane,
                                        int la, int lb = 0, int need = 0) {
  const int words = (la + kWave - 1) >> 6;  // wave-uniform
  if (words <= 1) return wide_lcs_words<K, 1, EARLY>(pm, text, nchars, lane, lb, need);
  if (K >= 2 && words == 2) return wide_lcs_words<K, (K >= 2 ? 2 : 1), EARLY>(pm, text, nchars, lane, lb, need);
  if (K >= 4 && words == 3) return wide_lcs_words<K, (K >= 4 ? 3 : 1), EARLY>(pm, text, nchars, lane, lb, need);
  if (K >= 4 && words == 4) return wide_lcs_words<K, (K >= 4 ? 4 : 1), EARLY>(pm, text, nchars, lane, lb, need);
  if (K >= 8 && words <= 6) return wide_lcs_words<K, (K >= 8 ? 6 : 1), EARLY>(pm, text, nchars, lane, lb, need);
  return wide_lcs_words<K, K, EARLY>(pm, text, nchars, lane, lb, need);
}

// Two patterns (mask tables pmA, pmB of the same layout) against the lane's text in ONE pass: the text dword,
// the symbol extraction and the table offset are shared, and the two carry chains are independent -- at the one
// to two waves per SIMD the 16 KB text images leave a CU, instruction-level parallelism is what hides the LDS
// and carry latencies.  W = live words of the LONGER pattern (the shorter one's upper words have all-zero masks).
template <int K, int W>
__device__ __forceinline__ void wide_lcs2_words(const unsigned long long* pmA, const unsigned long long* pmB,
                                                const uint32_t* text, int nchars, int lane, int& lcsA, int& lcsB) {
  unsigned long long va[W], vb[W];
#pragma unroll
  for (int k = 0; k < W; ++k) va[k] = vb[k] = ~0ull;
  const int nw = (nchars + 3) >> 2;
  uint32_t w_next = nw > 0 ? text[lane] : 0u;
  for (int w = 0; w < nw; ++w) {
    const uint32_t word = w_next;
    w_next = text[min(w + 1, nw - 1) * kWave + lane];
    unsigned long long ma[4][W], mb[4][W];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int off = static_cast<int>((word >> (8 * b)) & 0xffu) * kPmWords<K>;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        ma[b][k] = lds_word(pmA + off + k);
        mb[b][k] = lds_word(pmB + off + k);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      unsigned long long ua[W], ub[W], ta[W], tb[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        ua[k] = va[k] & ma[b][k];
        ub[k] = vb[k] & mb[b][k];
      }
      add_chain<W>(va, ua, ta);
      add_chain<W>(vb, ub, tb);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        va[k] = ta[k] | (va[k] ^ ua[k]);
        vb[k] = tb[k] | (vb[k] ^ ub[k]);
      }
    }
  }
  int oa = 0, ob = 0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    oa += __popcll(va[k]);
    ob += __popcll(vb[k]);
  }
  lcsA = kWave * W - oa;
  lcsB = kWave * W - ob;
}

// la_max = the longer of the two patterns; supported for up to 2 live words (longer pairs: two single passes)
template <int K>
__device__ __forceinline__ void wide_lcs2(const unsigned long long* pmA, const unsigned long long* pmB, const uint32_t* text,
                                          int nchars, int lane, int la_max, int& lcsA, int& lcsB) {
  if (la_max <= kWave) wide_lcs2_words<K, 1>(pmA, pmB, text, nchars, lane, lcsA, lcsB);
  else wide_lcs2_words<K, (K >= 2 ? 2 : 1)>(pmA, pmB, text, nchars, lane, lcsA, lcsB);
}

}  // namespace nsm
