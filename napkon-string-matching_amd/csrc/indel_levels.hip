// Levels-mode Indel-ratio grid: the hot loop of gen_comparable with score_func = fuzzy_match
// (reference: napkon_string_matching/types/comparable_data.py:223-232 calling compare_terms :248-265
// and fuzzy_match, compare/score_functions.py:20-27).
//
//   score(i, j) = sum_{s=1..max(Ll,Lr)} 2^-s * ratio(left level min(s,Ll-1), right level min(s,Lr-1))
//
// Every level of every item is one pre-processed string (the host hoists join_sorted +
// default_process from per pair to per item) stored as a row of a string table; an item is
// (first row, number of levels).  Lane = right item, left item wave-uniform.  Per step the wave
// builds the left level's match-mask table in LDS exactly as the RAW kernel does, each lane loads
// its own right level row (64 B, re-loaded only when its level index changes) and runs the
// bit-parallel LCS; the double ratio and the power-of-two weighted sum follow the reference's
// operation order.
#include "indel_wide.hpp"

namespace nsm {

struct IndelLevParams {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t pm_stride;
  int32_t cat_mode;
  double threshold;
  unsigned long long cap;
};

__device__ __forceinline__ double indel_score_dev(int la, int lb, int lcs) {
  if (la == 0 || lb == 0) return 0.0;
  const double maximum = static_cast<double>(la + lb);
  const double dist = static_cast<double>(la + lb - 2 * lcs);
  const double norm_sim = 1.0 - dist / maximum;
  return (norm_sim * 100.0) / 100.0;
}

// One-word strings (la, lb <= 64): every ratio the kernel can produce, evaluated by the compiler with
// the same IEEE double operations (two divisions per step are ~25 double-rate VALU instructions; the
// lookup is one cached load).  Index (la + lb) * 65 + lcs.
struct RatioTable {
  double v[129 * 65];
  constexpr RatioTable() : v() {
    for (int s = 0; s <= 128; ++s)
      for (int lcs = 0; lcs <= 64; ++lcs) {
        double r = 0.0;
        if (s > 0 && 2 * lcs <= s) {
          const double maximum = static_cast<double>(s);
          const double dist = static_cast<double>(s - 2 * lcs);
          const double norm_sim = 1.0 - dist / maximum;
          r = (norm_sim * 100.0) / 100.0;
        }
        v[s * 65 + lcs] = r;
      }
  }
};
__device__ const RatioTable g_ratio64{};

__device__ __forceinline__ unsigned long long lev_add64(unsigned long long a, unsigned long long b) {
  unsigned long long d;  // one VALU op (see indel_raw.hip)
  asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

template <int K>
__global__ __launch_bounds__(kBlock) void indel_levels_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ lsegstart, const uint8_t* __restrict__ lcodes,
    const int32_t* __restrict__ llen, const int32_t* __restrict__ rfirst, const int32_t* __restrict__ rnlev,
    const int32_t* __restrict__ rorig, const uint64_t* __restrict__ rcat, const int32_t* __restrict__ rseg,
    const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const IndelLevParams p) {
  // K = 1: strings <= 64 code units, text in registers; K = 2 / 4: multi-word LCS of indel_wide.hpp,
  // text image in LDS.  LDS layout: [wave][pm_stride * K] masks | (K > 1) [wave][16 K][64] text dwords
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_pm[];
  constexpr int kRow = kWave * K;  // bytes per string row
  const int waves = blockDim.x >> 6;

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * waves + wave;
  if (tile * kWave >= p.n_right) return;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;

  const bool partitioned = rseg != nullptr;
  const int myseg = partitioned ? rseg[jc] : 0;
  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);
  if (partitioned) {  // most (tile, chunk) combinations hold no row of the tile's categories: leave early
    unsigned long long cats = wave_or_u64(valid ? (1ull << myseg) : 0ull);
    bool work = false;
    while (cats) {
      const int c = __builtin_ctzll(cats);
      cats &= cats - 1;
      work = work || (max(i0, lsegstart[c]) < min(i1, lsegstart[c + 1]));
    }
    if (!work) return;
  }

  const int lr = rnlev[jc];
  const int rrow0 = rfirst[jc];
  const int jorig = rorig[jc];
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  const int lr_max = wave_max_i32(valid ? lr : 0);
  unsigned long long* pm = s_pm + wave * p.pm_stride * K;
  uint32_t* wtext = reinterpret_cast<uint32_t*>(s_pm + waves * p.pm_stride * K) + wave * 16 * K * kWave;

  uint32_t text[16];
  int text_row = -1;
  int lb = 0;

  // ---- all steps of left row i against the lanes flagged `ok`
  auto score_row = [&](int i, bool ok) {
    const int ll = lnlev[i];
    const int lrow0 = lfirst[i];
    const int steps_w = max(ll, lr_max);
    const int steps_l = max(ll, lr);
    double score = 0.0;
    double factor = 1.0;
    int pm_row = -1;
    int la = 0;
    for (int s = 1; s <= steps_w; ++s) {
      const bool active = ok && s <= steps_l;
      // left level (wave-uniform): rebuild the match masks when the level changes
      const int lrow = lrow0 + max(0, min(s, ll - 1));
      if (lrow != pm_row) {
        pm_row = lrow;
        la = llen[lrow];
        wide_build_pm<K>(pm, p.pm_stride, lcodes + static_cast<size_t>(lrow) * kRow, la, lane);
      }
      // right level (per lane): reload the row only when its index changes
      const int rrow = rrow0 + max(0, min(s, lr - 1));
      if constexpr (K == 1) {
        if (rrow != text_row) {
          text_row = rrow;
          const uint4* tp = reinterpret_cast<const uint4*>(rcodes + static_cast<size_t>(rrow) * 64);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 v = tp[q];
            text[4 * q + 0] = v.x;
            text[4 * q + 1] = v.y;
            text[4 * q + 2] = v.z;
            text[4 * q + 3] = v.w;
          }
          lb = rlen[rrow];
        }
      } else {
        if (__any(rrow != text_row)) {  // the LDS image is rewritten by the whole wave
          text_row = rrow;
          wide_store_text<K>(wtext, rcodes + static_cast<size_t>(rrow) * kRow, lane);
          lb = rlen[rrow];
        }
      }
      const int nchars = wave_max_i32(active ? lb : 0);
      int lcs;
      if constexpr (K == 1) {
        const int nwords = (nchars + 3) >> 2;
        if (la <= 32) {  // wave-uniform: 32-bit words, and / add / xor / or all issue at full rate
          const uint32_t* pm32 = reinterpret_cast<const uint32_t*>(pm);
          uint32_t v = ~0u;
#pragma unroll
          for (int w = 0; w < 16; ++w) {
            if (w < nwords) {
#pragma unroll
              for (int b = 0; b < 4; ++b) {
                const unsigned c = (text[w] >> (8 * b)) & 0xffu;
                const uint32_t m = pm32[2 * c];
                const uint32_t u = v & m;
                v = (v + u) | (v ^ u);
              }
            }
          }
          lcs = 32 - __popc(v);
        } else {
          unsigned long long v = ~0ull;
#pragma unroll
          for (int w = 0; w < 16; ++w) {
            if (w < nwords) {
#pragma unroll
              for (int b = 0; b < 4; ++b) {
                const unsigned c = (text[w] >> (8 * b)) & 0xffu;
                const unsigned long long m = pm[c];
                const unsigned long long u = v & m;
                v = lev_add64(v, u) | (v ^ u);
              }
            }
          }
          lcs = 64 - __popcll(v);
        }
      } else {
        lcs = wide_lcs<K>(pm, wtext, nchars, lane, la);
      }
      factor *= 0.5;
      if (active) {
        double ratio;
        if constexpr (K == 1) ratio = (la == 0 || lb == 0) ? 0.0 : g_ratio64.v[(la + lb) * 65 + lcs];
        else ratio = indel_score_dev(la, lb, lcs);
        score += ratio * factor;
      }
      // exact early exit: the steps still to come add at most factor - 2^-steps < factor (ratios are
      // <= 1); when no lane can reach the threshold any more the rest of the row is skipped.  The
      // 1e-9 keeps the test safe under the rounding of the double sum.
      if (!__any(active && (score + factor + 1e-9 >= p.threshold))) break;
    }
    const bool hit = ok && score >= p.threshold;
    if (__any(hit)) {
      if (hit) emit_hit(hits, p.cap, count, score, lorig[i], jorig);
    }
  };

  if (partitioned) {
    // both sides are grouped by category: visit the left rows of the categories this wave's lanes
    // stand for, and report a pair in its lowest common category only
    unsigned long long cats = wave_or_u64(valid ? (1ull << myseg) : 0ull);
    while (cats) {
      const int c = __builtin_ctzll(cats);
      cats &= cats - 1;
      const int a = max(i0, lsegstart[c]);
      const int b = min(i1, lsegstart[c + 1]);
      const unsigned long long lower = (1ull << c) - 1ull;
      for (int i = a; i < b; ++i) {
        const bool ok = valid && myseg == c && ((lcat[i] & catr & lower) == 0ull);
        if (__any(ok)) score_row(i, ok);
      }
    }
  } else {
    for (int i = i0; i < i1; ++i) {
      bool ok = valid;
      if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(lcat[i], catr, p.cat_mode);
      if (__any(ok)) score_row(i, ok);
    }
  }
}

}  // namespace nsm

extern "C" int nsm_indel_levels_grid(const nsm_level_items* left, const nsm_str_table* left_strings,
                                     const nsm_level_items* right, const nsm_str_table* right_strings,
                                     double threshold, int32_t category_mode, uint32_t flags, nsm_hit* hits,
                                     uint64_t capacity, unsigned long long* hit_count, void* stream) {
  using namespace nsm;
  (void)flags;
  if (!left || !right || !left_strings || !right_strings || !hit_count || (!hits && capacity)) {
    set_error("nsm_indel_levels_grid: null argument");
    return NSM_E_BADARG;
  }
  const int stride = left_strings->stride;
  if (stride != right_strings->stride || (stride != 64 && stride != 128 && stride != 256)) {
    set_error("nsm_indel_levels_grid: stride %d/%d unsupported (both sides 64, 128 or 256 code units)",
              left_strings->stride, right_strings->stride);
    return NSM_E_UNSUPPORTED;
  }
  if (left_strings->alphabet != right_strings->alphabet || left_strings->alphabet < 1 ||
      left_strings->alphabet > 255) {
    set_error("nsm_indel_levels_grid: alphabets differ or exceed 255");
    return NSM_E_BADARG;
  }
  if (category_mode != NSM_CAT_NONE && category_mode != NSM_CAT_INTERSECT &&
      category_mode != NSM_CAT_INTERSECT_OR_BOTH_EMPTY) {
    set_error("nsm_indel_levels_grid: unknown category mode %d", category_mode);
    return NSM_E_BADARG;
  }
  if (left->n < 0 || right->n < 0) {
    set_error("nsm_indel_levels_grid: negative row count");
    return NSM_E_BADARG;
  }
  if (left->n == 0 || right->n == 0) return 0;
  if ((left->seg == nullptr) != (right->seg == nullptr) || (left->seg && (!left->seg_start || !left->cat ||
      !right->cat || category_mode != NSM_CAT_INTERSECT))) {
    set_error("nsm_indel_levels_grid: a category partition needs seg/seg_start/cat on both sides and "
              "NSM_CAT_INTERSECT");
    return NSM_E_BADARG;
  }
  if (!left->first || !left->nlev || !left->orig || !right->first || !right->nlev || !right->orig ||
      !left_strings->codes || !left_strings->len || !right_strings->codes || !right_strings->len ||
      (category_mode != NSM_CAT_NONE && (!left->cat || !right->cat))) {
    set_error("nsm_indel_levels_grid: table has a null column");
    return NSM_E_BADARG;
  }
  IndelLevParams p;
  p.n_left = left->n; p.n_right = right->n; p.cap = capacity;
  p.pm_stride = ((left_strings->alphabet + 1) + 63) / 64 * 64;
  p.cat_mode = category_mode;
  p.threshold = threshold;
  const int n_tiles = (right->n + kWave - 1) / kWave;
  const long long want_waves = 16ll * 256 * 32;
  long long chunks = (want_waves + n_tiles - 1) / n_tiles;
  long long rows = (left->n + chunks - 1) / chunks;
  if (rows < 32) rows = 32;
  if (rows > 4096) rows = 4096;
  // with a category partition a tile only works on the chunks that overlap its categories' row
  // ranges: small chunks, or a handful of long-running waves hold the whole launch
  if (left->seg && rows > 128) rows = 128;
  p.rows_per_chunk = static_cast<int>(rows);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (left->n + 65534) / 65535;
    grid.y = (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  const int K = stride / 64;
  const int waves = K == 4 ? 2 : 4;  // keeps the block under 64 KiB of LDS
  dim3 grid2((n_tiles + waves - 1) / waves, grid.y);
  const size_t lds = static_cast<size_t>(waves) * (p.pm_stride * K * 8 + (K > 1 ? 16 * K * kWave * 4 : 0));
#define NSM_LAUNCH_LEVELS(KK)                                                                                  \
  hipLaunchKernelGGL((indel_levels_kernel<KK>), grid2, dim3(waves * kWave), lds, static_cast<hipStream_t>(stream), \
                     left->first, left->nlev, left->orig, left->cat, left->seg_start, left_strings->codes,     \
                     left_strings->len, right->first, right->nlev, right->orig, right->cat, right->seg,         \
                     right_strings->codes, right_strings->len, hits, hit_count, p)
  if (K == 1) NSM_LAUNCH_LEVELS(1);
  else if (K == 2) NSM_LAUNCH_LEVELS(2);
  else NSM_LAUNCH_LEVELS(4);
#undef NSM_LAUNCH_LEVELS
  return hip_status(hipGetLastError(), "indel_levels_kernel launch");
}
