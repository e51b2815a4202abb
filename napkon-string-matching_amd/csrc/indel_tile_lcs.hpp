// Bit-parallel LCS loops of the shared-tile kernel (indel_levels_tile.hpp), in a header of their own so that
// tools/lcs_loop_bench.hip can time them in isolation.
#pragma once
#include "nsm_common.hpp"

namespace nsm {

// left rows staged together = mask tables per wave (the scan uses two -- two rows per pass --, the dense pass one per batch
// row).  The lane-per-pair dense passes pool the parked pairs of ONE batch and run at 16..20 of 64 lanes; for one-word
// strings a table is only 0.6 KB, so 8 rows per batch fit (NSM_TILE_BATCH_K1=8) -- measured on 3 x 100k^2 word-like grids
// at 0.5 / 0.45: 37.8 / 58.4 ms against 38.6 / 58.6 with 4 (the passes halve, the mask tables to build per row do not, and
// the staging registers spill): kept at 4.
#ifndef NSM_TILE_BATCH_K1
#define NSM_TILE_BATCH_K1 4
#endif
constexpr int tile_batch(int K) { return K == 1 ? NSM_TILE_BATCH_K1 : 4; }
constexpr int kTileHead = 12;   // dwords per head: histogram (8) | la | level row | levels | first row
#ifndef NSM_TILE_TBL_SKEW
#define NSM_TILE_TBL_SKEW 0
#endif
constexpr int kTileTableSkew = NSM_TILE_TBL_SKEW;  // u64 words between the end of one mask table of a wave and the next
// 64-bit words per mask-table entry: K + 1, as in indel_wide.hpp -- at a stride of 2 (K + 1) dwords the 64-bit words of
// 32 symbols fall on 32 distinct bank pairs, so a ds_read_b64 of one word for 64 lanes is conflict-free.  (Tried: 16-byte
// aligned entries of 2 / 6 / 10 words, so that four limbs come with one ds_read_b128 -- only 16 symbols are then on
// distinct bank quads, and the term bench went from 75 to 94 ms: these loops live on LDS bandwidth.)
constexpr int tile_words(int K) { return K + 1; }
template <int K>
constexpr int kTileWords = tile_words(K);

// ---- The recurrence on 32-bit LIMBS.  The table entries hold 64-bit words, but every VALU op is 32 bits wide anyway, so a
// pattern of 65..96 code units runs on 3 limbs instead of 2 words (4 limbs): a quarter fewer and / add / xor / or per
// code unit for the commonest lengths of Term-like level strings (step 1: mean 59, step 2: mean 87).
template <int L>
__device__ __forceinline__ void limb_add(const uint32_t (&a)[L], const uint32_t (&b)[L], uint32_t (&d)[L]) {
  static_assert(L >= 2 && L <= 16, "2..16 limbs");
  if constexpr (L == 2) {
    asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
        : "=&v"(d[0]), "=&v"(d[1]) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]) : "vcc");
  } else if constexpr (L == 3) {
    asm("v_add_co_u32 %0, vcc, %3, %6\n\tv_addc_co_u32 %1, vcc, %4, %7, vcc\n\tv_addc_co_u32 %2, vcc, %5, %8, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]) : "vcc");
  } else if constexpr (L == 4) {
    asm("v_add_co_u32 %0, vcc, %4, %8\n\tv_addc_co_u32 %1, vcc, %5, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %6, %10, vcc\n\tv_addc_co_u32 %3, vcc, %7, %11, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
  } else if constexpr (L == 5) {
    asm("v_add_co_u32 %0, vcc, %5, %10\n\tv_addc_co_u32 %1, vcc, %6, %11, vcc\n\tv_addc_co_u32 %2, vcc, %7, %12, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %8, %13, vcc\n\tv_addc_co_u32 %4, vcc, %9, %14, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]) : "vcc");
  } else if constexpr (L == 6) {
    asm("v_add_co_u32 %0, vcc, %6, %12\n\tv_addc_co_u32 %1, vcc, %7, %13, vcc\n\tv_addc_co_u32 %2, vcc, %8, %14, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %9, %15, vcc\n\tv_addc_co_u32 %4, vcc, %10, %16, vcc\n\tv_addc_co_u32 %5, vcc, %11, %17, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
          "v"(b[4]), "v"(b[5]) : "vcc");
  } else if constexpr (L == 8) {
    asm("v_add_co_u32 %0, vcc, %8, %16\n\tv_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\tv_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\tv_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\tv_addc_co_u32 %7, vcc, %15, %23, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]),
          "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
  } else {  // 7, 9..16 limbs (strings beyond 192 code units at odd limb counts, beyond 256: rare): the compiler's chain
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const unsigned long long t = static_cast<unsigned long long>(a[k]) + b[k] + carry;
      d[k] = static_cast<uint32_t>(t);
      carry = static_cast<uint32_t>(t >> 32);
    }
  }
}

// one code unit: v' = (v + (v & m)) | (v ^ (v & m)) over L limbs
template <int L>
__device__ __forceinline__ void limb_step(uint32_t (&v)[L], const uint32_t (&m)[L]) {
  uint32_t u[L], t[L];
#pragma unroll
  for (int k = 0; k < L; ++k) u[k] = v[k] & m[k];
  limb_add<L>(v, u, t);
#pragma unroll
  for (int k = 0; k < L; ++k) v[k] = t[k] | (v[k] ^ u[k]);
}

template <int L>
__device__ __forceinline__ int limb_zeros(const uint32_t (&v)[L]) {
  int ones = 0;
#pragma unroll
  for (int k = 0; k < L; ++k) ones += __popc(v[k]);
  return 32 * L - ones;
}

// L limbs of a table entry.  The 64-bit words are read with VOLATILE loads so that the compiler keeps them as single
// ds_read_b64 (2 LDS cycles each for 64 lanes): left alone it fuses two of them into one ds_read2_b64, which the LDS
// serves at HALF the rate (8 cycles: MI355X_MICROARCH.md, LDS table) -- and these loops live on LDS bandwidth (§4.0).
template <int L>
__device__ __forceinline__ void load_limbs(const uint32_t* e, uint32_t (&m)[L]) {
#ifdef NSM_TILE_PLAIN_LOADS  // (A/B builds)
#pragma unroll
  for (int k = 0; k < L; ++k) m[k] = e[k];
#else
  // (mask tables always live in LDS: the explicit address space keeps the volatile loads ds_read, not flat_load)
  using lds_u64 = const volatile __attribute__((address_space(3))) unsigned long long;
  using lds_u32 = const volatile __attribute__((address_space(3))) uint32_t;
  lds_u64* e64 = (lds_u64*)(e);
#pragma unroll
  for (int k = 0; k + 1 < L; k += 2) {
    const unsigned long long w = e64[k >> 1];
    m[k] = static_cast<uint32_t>(w);
    m[k + 1] = static_cast<uint32_t>(w >> 32);
  }
  if constexpr (L & 1) m[L - 1] = ((lds_u32*)(e))[L - 1];
#endif
}

// LCS of ONE pattern (masks at `pm`: kTileWords<K> 64-bit words per symbol, the first L limbs live) and the lane's text:
// dword w of the text is text[w * ts].  `pm`, `text` and `ts` may differ per lane (dense pass: an image column, ts = 64;
// a row in global memory, ts = 1).
template <int K, int L>
__device__ __forceinline__ int tile_lcs1(const unsigned long long* pm, const uint32_t* text, int ts, int nchars) {
  uint32_t v[L];
#pragma unroll
  for (int k = 0; k < L; ++k) v[k] = ~0u;
  const int nw = (nchars + 3) >> 2;
  if (nw == 0) return 0;
  uint32_t w_next = text[0];
  for (int w = 0; w < nw; ++w) {
    const uint32_t word = w_next;
    w_next = text[min(w + 1, nw - 1) * ts];
    uint32_t m[4][L];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      load_limbs<L>(reinterpret_cast<const uint32_t*>(pm + ((word >> (8 * b)) & 0xffu) * kTileWords<K>), m[b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) limb_step<L>(v, m[b]);
  }
  return limb_zeros<L>(v);
}

template <int K>
__device__ __forceinline__ int tile_lcs1_any(const unsigned long long* pm, const uint32_t* text, int ts, int nchars, int la_max) {
  const int limbs = (la_max + 31) >> 5;  // wave-uniform
  if (limbs <= 2) return tile_lcs1<K, 2>(pm, text, ts, nchars);
  if (limbs == 3) return tile_lcs1<K, 3>(pm, text, ts, nchars);
  if (limbs == 4) return tile_lcs1<K, 4>(pm, text, ts, nchars);
  if (K >= 4 && limbs == 5) return tile_lcs1<K, (K >= 4 ? 5 : 2)>(pm, text, ts, nchars);
  if (K >= 4 && limbs == 6) return tile_lcs1<K, (K >= 4 ? 6 : 2)>(pm, text, ts, nchars);
  if (K >= 4 && limbs <= 8) return tile_lcs1<K, (K >= 4 ? 8 : 2)>(pm, text, ts, nchars);
  if (K >= 8 && limbs <= 12) return tile_lcs1<K, (K >= 8 ? 12 : 2)>(pm, text, ts, nchars);
  return tile_lcs1<K, 2 * K>(pm, text, ts, nchars);
}

// TWO wave-uniform patterns (tables pmA and pmA + tbl_entries) against the lane's image column in one pass: one text
// read and one table offset for both, two independent carry chains.
// EARLY (step 1): needA / needB = the smallest LCS that keeps the lane's pair alive (0xffff: dead), lb = the lane's text
// length.  Every 16 code units the wave checks whether any lane of either row can still reach its need (one code
// unit adds at most one to the LCS); if none can, the scan stops and both results are 0 (below every live need).
template <int K, int L, bool EARLY>
__device__ __forceinline__ void tile_lcs2(const unsigned long long* pmA, int tbl_entries, const uint32_t* text, int nchars,
                                          int& lcsA, int& lcsB, int needA = 0, int needB = 0, int lb = 0) {
  uint32_t va[L], vb[L];
#pragma unroll
  for (int k = 0; k < L; ++k) va[k] = vb[k] = ~0u;
  const int nw = (nchars + 3) >> 2;
  const unsigned long long* pmB = pmA + tbl_entries;
  uint32_t w_next = nw > 0 ? text[0] : 0u;
  for (int w = 0; w < nw; ++w) {
    const uint32_t word = w_next;
    w_next = text[min(w + 1, nw - 1) * kWave];
    uint32_t ma[4][L], mb[4][L];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int off = static_cast<int>((word >> (8 * b)) & 0xffu) * kTileWords<K>;
      load_limbs<L>(reinterpret_cast<const uint32_t*>(pmA + off), ma[b]);
      load_limbs<L>(reinterpret_cast<const uint32_t*>(pmB + off), mb[b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      limb_step<L>(va, ma[b]);
      limb_step<L>(vb, mb[b]);
    }
    if (EARLY && (w & 3) == 3) {
      const int left = max(0, lb - 4 * (w + 1));
      const bool can = limb_zeros<L>(va) + left >= needA || limb_zeros<L>(vb) + left >= needB;
      if (!__any(can)) {
        lcsA = lcsB = 0;
        return;
      }
    }
  }
  lcsA = limb_zeros<L>(va);
  lcsB = limb_zeros<L>(vb);
}

// la_max <= 128 code units
template <int K, bool EARLY>
__device__ __forceinline__ void tile_lcs2_any(const unsigned long long* pmA, int tbl_entries, const uint32_t* text, int nchars,
                                              int la_max, int& lcsA, int& lcsB, int needA = 0, int needB = 0, int lb = 0) {
  const int limbs = (la_max + 31) >> 5;  // wave-uniform
  if (limbs <= 2) tile_lcs2<K, 2, EARLY>(pmA, tbl_entries, text, nchars, lcsA, lcsB, needA, needB, lb);
  else if (limbs == 3) tile_lcs2<K, 3, EARLY>(pmA, tbl_entries, text, nchars, lcsA, lcsB, needA, needB, lb);
  else tile_lcs2<K, 4, EARLY>(pmA, tbl_entries, text, nchars, lcsA, lcsB, needA, needB, lb);
}

}  // namespace nsm
