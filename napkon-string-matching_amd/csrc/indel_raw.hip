// RAW Indel-ratio all-pairs grid:  fuzzy_match = rapidfuzz QRatio / 100 on one string per item
// (reference: napkon_string_matching/compare/score_functions.py:20-27; QRatio of rapidfuzz 2.1.x
// is the normalized Indel similarity, i.e. 1 - (|a|+|b|-2 LCS)/(|a|+|b|)).
//
// Mapping to CDNA4
//   * bit-parallel LCS (Hyyro / Allison-Dix):  V = ~0;  per text symbol c:  U = V & PM[c];
//     V = (V + U) | (V ^ U);  LCS = popcount(~V).  Patterns of <= 32 code units run in 32-bit
//     words (4 full-rate VALU ops per symbol), longer ones (<= 64) in 64-bit words;
//   * one LANE owns one right string (the "text") for the whole kernel.  Its 64 code units are kept
//     as 32 VGPRs of two 16-bit fields holding the LDS ADDRESS of the symbol's match mask
//     (wave's PM base + 8 * code), so one full-rate v_and / v_lshrrev per step yields the address;
//   * the left string (the "pattern") is wave-uniform.  Each wave builds its match-mask table
//     PM[alphabet] in LDS: one ds_write_b64 sweep to clear, ONE ds_or_b64 per wave to set (lane k
//     ORs bit k into PM[pattern[k]]).  PM[c] lookups are LDS reads indexed by the lane's symbol:
//     equal symbols broadcast, symbols c and c' only conflict when c == c' (mod 32);
//   * both tables are sorted by length (descending): a wave stops after its longest text; the
//     threshold test is an integer compare against lcsmin[la+lb], computed by the launcher with the
//     reference's double arithmetic; the double score is only computed for hits;
//   * exact prunes (NSM_FLAG_PRUNE): LCS <= min(la, lb), and LCS <= (la + lb - L1) / 2 where L1 is the
//     L1 distance of the two strings' 32-bucket symbol histograms (8 v_sad_u8 per pair): a wave only
//     runs the LCS for rows in which some lane can still reach its lcsmin.
#include "nsm_common.hpp"

namespace nsm {

// Table columns travel as __restrict__ kernel arguments so that the wave-uniform left side is
// fetched with scalar loads (see jaccard_raw_impl.hpp).
struct IndelRawParams {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t pm_stride;  // (alphabet + 1) rounded up to 64 entries
  int32_t zero_need;  // 0 when a 0.0 score reaches the threshold, else kNever
  unsigned long long cap;
  uint8_t lcsmin[132];  // indexed by la+lb (both >= 1)
};

// The exact double sequence of `QRatio(a, b) / 100` once LCS is known (oracle/score_functions.py).
__device__ __forceinline__ double indel_score(int la, int lb, int lcs) {
  if (la == 0 || lb == 0) return 0.0;
  const double maximum = static_cast<double>(la + lb);
  const double dist = static_cast<double>(la + lb - 2 * lcs);
  const double norm_sim = 1.0 - dist / maximum;
  return (norm_sim * 100.0) / 100.0;
}

static double indel_score_host(int la, int lb, int lcs) {
  if (la == 0 || lb == 0) return 0.0;
  const volatile double maximum = static_cast<double>(la + lb);
  const volatile double dist = static_cast<double>(la + lb - 2 * lcs);
  const volatile double q = dist / maximum;
  const volatile double norm_sim = 1.0 - q;
  const volatile double pct = norm_sim * 100.0;
  return pct / 100.0;
}

// Load from an LDS byte address held in a register (ds_read_b32 / ds_read_b64).
template <typename T>
__device__ __forceinline__ T lds_load(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T*>(addr);
}

// 64-bit add in ONE VALU op (5.3 cycles; v_add_co_u32 + v_addc_co_u32 take 9.3 on gfx950).
__device__ __forceinline__ unsigned long long add64(unsigned long long a, unsigned long long b) {
  unsigned long long d;
  asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// The wave's match-mask table of pattern row i: one ds_write_b64 sweep to clear, ONE ds_or_b64 to set.  A narrow pattern
// (<= 32 code units) stores its mask in BOTH halves of the entry (see the kernel's notes on LDS banks).
__device__ __forceinline__ void raw_build_masks(unsigned long long* pm, int pm_stride, const uint8_t* __restrict__ lcodes, int i,
                                                int la, bool wide, int lane) {
  for (int c = lane; c < pm_stride; c += kWave) pm[c] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane < la) {
    const unsigned c = lcodes[static_cast<size_t>(i) * 64 + lane];
    atomicOr(&pm[c], wide ? (1ull << lane) : ((1ull << lane) | (1ull << (32 + lane))));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// LCS of ONE pair (left row `row` of `la`, right row `j` of `lb` code units; all wave-uniform) on the SCALAR unit: lane k
// holds code unit k of both strings (two 64-byte loads), v_readlane hands the right string's units to the scalar unit one
// by one, and per unit c the match mask is a ballot -- M = lanes whose left unit equals c -- so the recurrence
// V' = (V + (V & M)) | (V & ~M) runs on one 64-bit scalar: a v_readlane, a v_cmp and five SALU ops per code unit, no mask
// table, no LDS, 50 VGPRs for the whole kernel.  (The wave-wide form scores 64 texts against one pattern; with one surviving
// pair per pattern -- 4.1e5 of them per configs[2] grid -- 63 of its 64 lanes computed nothing: 0.9 of 4.2 ms, and its 32
// address registers set the kernel's occupancy.)
__device__ __forceinline__ int raw_lcs_pair(const uint8_t* __restrict__ lcodes, int row, int la, const uint8_t* __restrict__ rcodes,
                                            int j, int lb, int lane) {
  // lanes past the pattern hold a value no code unit equals, so the right string's padding (code = alphabet, like the left
  // string's own) matches nothing and whole words can be processed without a per-unit length test
  const uint32_t code = lcodes[static_cast<size_t>(row) * 64 + lane];
  const uint32_t pat = lane < la ? code : 0x100u;
  const uint32_t text = rcodes[static_cast<size_t>(j) * 64 + lane];
  unsigned long long v = ~0ull;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (4 * w < lb) {  // (wave-uniform; no `break`: the loop must unroll)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(text), 4 * w + b));
        const unsigned long long m = __ballot(pat == c);
        const unsigned long long u = v & m;
        v = (v + u) | (v & ~m);
      }
    }
  }
  return __popcll(~v);
}

#ifndef NSM_C3_OCC
#define NSM_C3_OCC
#endif
// right tiles (of 64 strings) per wavefront of the pruning kernel: every histogram of a left row that the wave fetches
// with its scalar loads is tested against T x 64 right strings (section 4.3 of DESIGN.md: the histogram loop waits for
// its scalar loads, not for the VALU)
#ifndef NSM_C3_TILES
#define NSM_C3_TILES 2
#endif
template <bool PRUNE>
constexpr int c3_tiles() { return PRUNE ? NSM_C3_TILES : 1; }

template <bool PRUNE>
__global__ __launch_bounds__(kBlock) NSM_C3_OCC void indel_raw_kernel(
    const uint8_t* __restrict__ lcodes, const int32_t* __restrict__ llen, const int32_t* __restrict__ lstart,
    const int32_t* __restrict__ lorig, const uint32_t* __restrict__ lhist, const uint8_t* __restrict__ rcodes,
    const int32_t* __restrict__ rlen, const int32_t* __restrict__ rorig, const uint32_t* __restrict__ rhist,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const IndelRawParams p) {
  constexpr int T = c3_tiles<PRUNE>();
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  // layout: [wave][pm_stride] match masks, then the lcsmin bytes
  uint8_t* s_lcsmin = reinterpret_cast<uint8_t*>(s_mem + kWavesPerBlock * p.pm_stride);
  for (int t = threadIdx.x; t < 132; t += kBlock) s_lcsmin[t] = p.lcsmin[t];
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int tile0 = (blockIdx.x * kWavesPerBlock + wave) * T;  // the wave's tiles: tile0 .. tile0 + T - 1
  if (tile0 * kWave >= p.n_right) return;
  bool valid[T];
  int jc[T], lbj[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int j = (tile0 + t) * kWave + lane;
    valid[t] = j < p.n_right;
    jc[t] = valid[t] ? j : p.n_right - 1;
    lbj[t] = valid[t] ? rlen[jc[t]] : 0;
  }

  unsigned long long* pm = s_mem + wave * p.pm_stride;
  const uint32_t pm_base = static_cast<uint32_t>(wave * p.pm_stride * 8);  // s_mem starts at LDS offset 0

  // text as LDS addresses of the symbols' masks: two 16-bit fields per VGPR.
  // Patterns <= 32 read their masks with ds_read_b32: at the table's 8-byte stride those reads only touch
  // the even LDS banks, symbols c and c + 16 collide (37 symbols over 16 banks: half of the exhaustive kernel's
  // LDS cycles were bank conflicts, profiles/r02_before_c3_sq_pmc.txt).  The mask of a narrow pattern is
  // therefore stored in BOTH halves of its entry and symbol c is read from half (c >> 4) & 1: symbols 0..31
  // map to 32 different banks.  The half is baked into the text's addresses, which are rebuilt when the
  // pattern class changes between wide and narrow (rows are sorted by length: once per chunk).
  // PRUNE = false (every row runs the LCS): the text is unpacked once into 32 registers, below.  PRUNE = true: pairs that
  // reach the LCS are rare (1e-5 on C3) and are scored one by one on the scalar unit (raw_lcs_pair): the histogram loop is
  // what runs, and it runs at the occupancy its OWN registers allow (round 4: the 32 address registers of the rare LCS
  // path had put the whole kernel at 88 VGPRs = 5 waves per SIMD, with 65 % of the wave-cycles spent waiting for the scalar
  // loads of the histogram loop).
  uint32_t taddr[PRUNE ? 1 : 32];
  auto build_taddr = [&](bool narrow) {
    if constexpr (!PRUNE) {
      const uint4* tp = reinterpret_cast<const uint4*>(rcodes + static_cast<size_t>(jc[0]) * 64);
      const uint32_t hsel = narrow ? 4u : 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 v = tp[q];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t c0 = w[e] & 0xffu, c1 = (w[e] >> 8) & 0xffu, c2 = (w[e] >> 16) & 0xffu, c3 = w[e] >> 24;
          const uint32_t a0 = pm_base + 8 * c0 + (((c0 >> 4) & 1u) ? hsel : 0u), a1 = pm_base + 8 * c1 + (((c1 >> 4) & 1u) ? hsel : 0u);
          const uint32_t a2 = pm_base + 8 * c2 + (((c2 >> 4) & 1u) ? hsel : 0u), a3 = pm_base + 8 * c3 + (((c3 >> 4) & 1u) ? hsel : 0u);
          taddr[8 * q + 2 * e + 0] = a0 | (a1 << 16);
          taddr[8 * q + 2 * e + 1] = a2 | (a3 << 16);
        }
      }
    }
  };
  int taddr_narrow = -1;  // which variant the registers hold (exhaustive mode keeps them across rows)
  const int npairs = (wave_first(lbj[0]) + 1) >> 1;  // sorted descending: lane 0 of the first tile has the longest text
  uint32_t hr[T][8];
  if (PRUNE) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint4* hp = reinterpret_cast<const uint4*>(rhist + static_cast<size_t>(jc[t]) * 8);
      const uint4 h0 = hp[0], h1 = hp[1];
      hr[t][0] = h0.x; hr[t][1] = h0.y; hr[t][2] = h0.z; hr[t][3] = h0.w;
      hr[t][4] = h1.x; hr[t][5] = h1.y; hr[t][6] = h1.z; hr[t][7] = h1.w;
    }
  }
  uint32_t lowmask = 0xffffu;  // kept in a VGPR: v_and with a VGPR operand issues at full rate
  asm volatile("" : "+v"(lowmask));

  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);

  auto build_masks = [&](int i, int la, bool wide) { raw_build_masks(pm, p.pm_stride, lcodes, i, la, wide, lane); };

  // ---- exhaustive kernel: LCS of pattern row i (length la, match masks built here) against the lane's text
  auto lcs_row = [&](int i, int la, bool wide) -> int {
    if (taddr_narrow != static_cast<int>(!wide)) {
      build_taddr(!wide);
      taddr_narrow = static_cast<int>(!wide);
    }
    build_masks(i, la, wide);
    // 8 code units per group: their mask reads are issued together (one LDS round trip per group instead of one
    // per dependent step), the recurrence is spelled as e32 instructions (nsm_common.hpp: lcs_step32 / 64)
    const int nchars = 2 * npairs;
    if (!wide) {  // pattern <= 32: 32-bit words, and / add / xor / or all issue at full rate
      uint32_t v = ~0u;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g * 8 < nchars) {
          uint32_t m[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            m[2 * q] = lds_load<uint32_t>(taddr[(4 * g + q) % (PRUNE ? 1 : 32)] & lowmask);
            m[2 * q + 1] = lds_load<uint32_t>(taddr[(4 * g + q) % (PRUNE ? 1 : 32)] >> 16);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) v = lcs_step32(v, m[q]);
        }
      }
      return 32 - __popc(v);
    }
    unsigned long long v = ~0ull;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g * 8 < nchars) {
        unsigned long long m[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          m[2 * q] = lds_load<unsigned long long>(taddr[(4 * g + q) % (PRUNE ? 1 : 32)] & lowmask);
          m[2 * q + 1] = lds_load<unsigned long long>(taddr[(4 * g + q) % (PRUNE ? 1 : 32)] >> 16);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v = lcs_step64(v, m[q]);
      }
    }
    return 64 - __popcll(v);
  };

  // rows are sorted by length (descending): rows of length 64 - c are [lstart[c], lstart[c + 1])
  const int c_first = 64 - llen[i0];
  const int c_last = 64 - llen[i1 - 1];
  for (int c = c_first; c <= c_last; ++c) {
    const int a = max(i0, lstart[c]);
    const int b = min(i1, lstart[c + 1]);
    if (a >= b) continue;
    const int la = 64 - c;
    int need[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (la == 0 || lbj[t] == 0) need[t] = p.zero_need;
      else need[t] = s_lcsmin[la + lbj[t]];
      if (!valid[t]) need[t] = kNever;
    }
    const bool wide = la > 32;
    if constexpr (PRUNE) {
      // exact length filter, per class: LCS <= min(la, lb)
      bool fits[T], some = false;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        fits[t] = min(la, lbj[t]) >= need[t];
        some = some || fits[t];
      }
      if (!__any(some)) continue;
      // exact histogram filter, per row: LCS <= sum_c min(hA[c], hB[c]) = (la + lb - L1) / 2 over 32
      // symbol buckets, i.e. the row can only hit if L1 <= la + lb - 2 * lcsmin
      // `who`: bit t set = this lane's string of tile t passed the histogram filter for row i
      // (the pairs that pass are rare: each is scored on its own on the scalar unit, raw_lcs_pair -- no mask table, and no
      // register of the LCS outlives it)
      auto score_row = [&](int i, uint32_t who) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          for (unsigned long long todo = __ballot((who >> t) & 1u); todo; todo &= todo - 1ull) {
            const int leader = __builtin_ctzll(todo);
            const int j_s = __builtin_amdgcn_readlane(jc[t], leader);
            const int lb_s = __builtin_amdgcn_readlane(lbj[t], leader);
            const int need_s = __builtin_amdgcn_readlane(need[t], leader);
            const int lcs = raw_lcs_pair(lcodes, i, la, rcodes, j_s, lb_s, lane);
            if (lcs >= need_s && lane == 0) emit_hit(hits, p.cap, count, indel_score(la, lb_s, lcs), lorig[i], rorig[j_s]);
          }
        }
      };
      // 4 rows per iteration: their 32 histogram dwords arrive with two s_load_dwordx16 issued
      // back to back.  The SAD chain of a row is seeded with -(limit + 1), so its result is negative
      // exactly when L1 <= limit; one full-rate v_or per row folds the signs into the batch's verdict
      // (the scalar unit is shared by the CU's 4 SIMDs, keep it idle; v_cmp + v_addc per row cost
      // two half-rate ops).
      int limit[T];
      uint32_t seed[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        limit[t] = fits[t] ? la + lbj[t] - 2 * need[t] : -1;
        seed[t] = static_cast<uint32_t>(-(limit[t] + 1));
      }
      const uint32_t* __restrict__ hp = lhist + static_cast<size_t>(a) * 8;
      int i = a;
      // (Tried on top, same box, kernel ms against 6.52: a software pipeline over two scalar register sets of 2 rows each --
      // the compiler needs 78 VGPRs for it, 6 waves per SIMD: 8.36; the same capped at 7 waves with 480 B of scratch: 7.86;
      // 2 rows per batch, 73 SGPRs: 6.45.  At 8 waves per SIMD the other waves hide the scalar loads.)
#ifndef NSM_C3_BATCH
#define NSM_C3_BATCH 4
#endif
      constexpr int BATCH = NSM_C3_BATCH;
      for (; i + BATCH <= b; i += BATCH, hp += 8 * BATCH) {
        uint32_t h[8 * BATCH];
#pragma unroll
        for (int q = 0; q < 8 * BATCH; ++q) h[q] = hp[q];
        int margin[T][BATCH];
        int any_pass = 0;  // sign bit set when some row of the batch may reach the threshold
#pragma unroll
        for (int r = 0; r < BATCH; ++r) {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            uint32_t l1 = seed[t];
#pragma unroll
            for (int q = 0; q < 8; ++q) l1 = __builtin_amdgcn_sad_u8(h[8 * r + q], hr[t][q], l1);
            margin[t][r] = static_cast<int>(l1);
            any_pass |= margin[t][r];
          }
        }
        if (__any(any_pass < 0)) {  // rare
          uint32_t cand[T];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            cand[t] = 0;
#pragma unroll
            for (int r = 0; r < BATCH; ++r) cand[t] |= margin[t][r] < 0 ? (1u << r) : 0u;
          }
          for (int r = 0; r < BATCH; ++r) {
            uint32_t who = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) who |= ((cand[t] >> r) & 1u) << t;
            if (__any(who != 0u)) score_row(i + r, who);
          }
        }
      }
      for (; i < b; ++i, hp += 8) {
        uint32_t who = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          uint32_t l1 = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q) l1 = __builtin_amdgcn_sad_u8(hp[q], hr[t][q], l1);
          who |= (static_cast<int>(l1) <= limit[t] ? 1u : 0u) << t;
        }
        if (__any(who != 0u)) score_row(i, who);
      }
    } else {
      for (int i = a; i < b; ++i) {
        const int lcs = lcs_row(i, la, wide);
        const bool hit = lcs >= need[0];
        if (__any(hit)) {
          if (hit) emit_hit(hits, p.cap, count, indel_score(la, lbj[0], lcs), lorig[i], rorig[jc[0]]);
        }
      }
    }
  }
}

}  // namespace nsm
#include "indel_raw_coarse.hpp"
namespace nsm {

static int pick_rows_per_chunk(int n_left, int n_tiles) {
#ifndef NSM_C3_WANT_WAVES
#define NSM_C3_WANT_WAVES (16ll * 256 * 32)
#endif
  const long long want_waves = NSM_C3_WANT_WAVES;
  long long chunks = (want_waves + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
  if (chunks < 1) chunks = 1;
  long long rows = (n_left + chunks - 1) / chunks;
  if (rows < 64) rows = 64;
#ifndef NSM_C3_CHUNK_MAX
#define NSM_C3_CHUNK_MAX 4096
#endif
  if (rows > NSM_C3_CHUNK_MAX) rows = NSM_C3_CHUNK_MAX;
  return static_cast<int>(rows);
}

int indel_raw_wide(const nsm_str_table* left, const nsm_str_table* right, double threshold, uint32_t flags,
                   nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, hipStream_t stream);

}  // namespace nsm

extern "C" int nsm_indel_raw_grid(const nsm_str_table* left, const nsm_str_table* right,
                                  double threshold, uint32_t flags, nsm_hit* hits, uint64_t capacity,
                                  unsigned long long* hit_count, void* stream) {
  using namespace nsm;
  if (!left || !right || !hit_count || (!hits && capacity)) {
    set_error("nsm_indel_raw_grid: null argument");
    return NSM_E_BADARG;
  }
  if (left->stride != right->stride || (left->stride != 64 && left->stride != 128 && left->stride != 256 && left->stride != 512)) {
    set_error("nsm_indel_raw_grid: stride %d/%d unsupported (both sides 64, 128, 256 or 512 code units)",
              left->stride, right->stride);
    return NSM_E_UNSUPPORTED;
  }
  if (left->alphabet != right->alphabet || left->alphabet < 1 || left->alphabet > 255) {
    set_error("nsm_indel_raw_grid: alphabets differ or exceed 255 (%d, %d)", left->alphabet,
              right->alphabet);
    return NSM_E_BADARG;
  }
  if (left->n < 0 || right->n < 0) {
    set_error("nsm_indel_raw_grid: negative row count");
    return NSM_E_BADARG;
  }
  if (left->n == 0 || right->n == 0) return 0;
  if (!left->codes || !left->len || !left->orig || !left->len_start || !right->codes || !right->len ||
      !right->orig) {
    set_error("nsm_indel_raw_grid: table has a null column");
    return NSM_E_BADARG;
  }

  if (left->stride != 64)
    return indel_raw_wide(left, right, threshold, flags, hits, capacity, hit_count, static_cast<hipStream_t>(stream));

  IndelRawParams p;
  p.n_left = left->n; p.n_right = right->n;
  p.cap = capacity;
  p.pm_stride = ((left->alphabet + 1) + 63) / 64 * 64;
  p.zero_need = (0.0 >= threshold) ? 0 : kNever;
  for (int s = 0; s < 132; ++s) {
    p.lcsmin[s] = kNever;
    if (s < 2 || s > 128) continue;
    for (int lcs = 0; 2 * lcs <= s; ++lcs) {
      // any split la+lb = s gives the same score: it only depends on s and lcs
      if (indel_score_host(1, s - 1, lcs) >= threshold) {
        p.lcsmin[s] = static_cast<uint8_t>(lcs);
        break;
      }
    }
  }
  const bool prune = (flags & NSM_FLAG_PRUNE) && left->hist && right->hist;
  if (prune && left->hist16 && right->hist16 && !(flags & NSM_FLAG_ONE_STAGE)) {
    if ((reinterpret_cast<uintptr_t>(left->hist16) | reinterpret_cast<uintptr_t>(right->hist16)) & 15u) {
      set_error("nsm_indel_raw_grid: hist16 must be 16-byte aligned (rows are read as one 128-bit word)");
      return NSM_E_BADARG;
    }
    // both tables carry the 16-bucket histograms: the two-stage filter (indel_raw_coarse.hpp)
    constexpr int T = NSM_C3C_TILES;
    const int n_tiles = ((right->n + kWave - 1) / kWave + T - 1) / T;
    p.rows_per_chunk = pick_rows_per_chunk(left->n, n_tiles);
    dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
    if (grid.y > 65535) {
      p.rows_per_chunk = (left->n + 65534) / 65535;
      grid.y = (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
    }
    if (p.rows_per_chunk > (1 << 15)) {  // (a stack entry keeps its row relative to the chunk in 15 bits)
      set_error("nsm_indel_raw_grid: %d left rows per chunk", p.rows_per_chunk);
      return NSM_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL((indel_raw_coarse_kernel<T, NSM_C3C_ROWS>), grid, dim3(kBlock), c3c_lds_bytes(T),
                       static_cast<hipStream_t>(stream), left->codes, left->len, left->len_start, left->orig,
                       reinterpret_cast<const uint32_t*>(left->hist), reinterpret_cast<const uint32_t*>(left->hist16), right->codes,
                       right->len, right->orig, reinterpret_cast<const uint32_t*>(right->hist),
                       reinterpret_cast<const uint32_t*>(right->hist16), hits, hit_count, p);
    return hip_status(hipGetLastError(), "indel_raw_coarse_kernel launch");
  }
  // wavefronts along the right side: one per T tiles of 64 strings
  const int tiles_per_wave = prune ? c3_tiles<true>() : 1;
  const int n_tiles = ((right->n + kWave - 1) / kWave + tiles_per_wave - 1) / tiles_per_wave;
  p.rows_per_chunk = pick_rows_per_chunk(left->n, n_tiles);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock,
            (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (left->n + 65534) / 65535;
    grid.y = (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  const size_t lds = static_cast<size_t>(kWavesPerBlock) * p.pm_stride * 8 + 136;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t* lh = reinterpret_cast<const uint32_t*>(left->hist);
  const uint32_t* rh = reinterpret_cast<const uint32_t*>(right->hist);
  if (prune)
    hipLaunchKernelGGL((indel_raw_kernel<true>), grid, dim3(kBlock), lds, s, left->codes, left->len,
                       left->len_start, left->orig, lh, right->codes, right->len, right->orig, rh, hits, hit_count, p);
  else
    hipLaunchKernelGGL((indel_raw_kernel<false>), grid, dim3(kBlock), lds, s, left->codes, left->len,
                       left->len_start, left->orig, lh, right->codes, right->len, right->orig, rh, hits, hit_count, p);
  return hip_status(hipGetLastError(), "indel_raw_kernel launch");
}
