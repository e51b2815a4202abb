// Grids for operands the register-resident kernels cannot hold: token sets of more than 64 ids, strings of more than
// 512 code units, alphabets of more than 255 symbols.  The reference has no such limits (compare/score_functions.py:10-13
// builds Python sets of any size, :27 hands strings of any length to rapidfuzz), so a drop-in must not refuse them; they
// are rare in NAPKON data (a very long option list, a questionnaire in another script), so the host routes ONLY the
// items that need it through here (napkon_string_matching_amd/wide.py) and everything else stays on the fast kernels.
// Scores are accumulated in the reference's operation order.  One exact prune (round 4): LCS <= min(la, lb) bounds every
// step's ratio from the lengths alone, so a pair whose weighted bound stays below the threshold is dropped before any LCS
// runs, a lane dies as soon as its score so far plus the bound of the remaining steps cannot reach it, and a row none of
// whose lanes is alive is skipped.
//
//   nsm_indel_any_grid    compare_terms x fuzzy_match (or the RAW ratio) on CSR strings of 16-bit code units.
//       lane = right item, left item wave-uniform.  Bit-parallel LCS (Hyyro) with the PATTERN cut into chunks of 128
//       code units (4 limbs of 32 bits): chunk c is run over the lane's whole text with the carry of the multi-word add
//       handed from chunk c - 1 to chunk c through one bit per text position (carries only travel upwards, so the chunks
//       can be run one after the other).  Match masks of a chunk: one LDS table of 4 limbs per symbol.
//   nsm_jaccard_any_grid  (the same kind of bound from the set sizes: |A n B| <= min, |A u B| >= max)
//       compare_terms x intersection_vs_union (or the RAW quotient) on CSR id lists sorted by id, each id
//       with the first level that contains it: a two-pointer merge per pair finds the common ids, a per-lane histogram of
//       "first step that has it on both sides" gives every step's |A n B| as a prefix sum.
#include "nsm_common.hpp"

namespace nsm {

constexpr int kAnyChunk = 128;       // pattern code units per chunk
constexpr int kAnyLimbs = kAnyChunk / 32;
constexpr int kAnyMaxText = 4096;    // code units per string (carry bits: 128 dwords per lane)
constexpr int kAnyMaxAlphabet = 1023;
constexpr int kAnyMaxLevels = 64;

struct AnyParams {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t alphabet;  // pad symbol = alphabet (all-zero mask)
  int32_t cat_mode;
  int32_t raw;       // 1: score = ratio of the single level pair (RAW plugin call); 0: compare_terms
  double threshold;
  unsigned long long cap;
};

__device__ __forceinline__ double any_indel_score(int la, int lb, int lcs) {
  // rapidfuzz 2.x QRatio / 100 (compare/score_functions.py:27): 0 if either string is empty
  if (la == 0 || lb == 0) return 0.0;
  const double maximum = static_cast<double>(la + lb);
  const double dist = static_cast<double>(la + lb - 2 * lcs);
  const double norm_sim = 1.0 - dist / maximum;
  return (norm_sim * 100.0) / 100.0;
}

__global__ __launch_bounds__(kWave) void indel_any_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const uint16_t* __restrict__ lcodes, const long long* __restrict__ loff,
    const int32_t* __restrict__ rfirst, const int32_t* __restrict__ rnlev, const int32_t* __restrict__ rorig,
    const uint64_t* __restrict__ rcat, const uint16_t* __restrict__ rcodes, const long long* __restrict__ roff,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const AnyParams p) {
  // LDS: [alphabet + 1][4] u32 match masks of the current pattern chunk | [128][64] u32 carry bits ([dword][lane])
  extern __shared__ __attribute__((aligned(16))) uint32_t s_any[];
  const int lane = threadIdx.x;
  uint32_t* tbl = s_any;
  uint32_t* carry = s_any + static_cast<size_t>(p.alphabet + 1) * kAnyLimbs;
  const int j = blockIdx.x * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;
  const int lr = rnlev[jc], rr0 = rfirst[jc];
  const uint64_t catr = p.cat_mode != NSM_CAT_NONE ? rcat[jc] : 0ull;
  const int i0 = blockIdx.y * p.rows_per_chunk, i1 = min(p.n_left, i0 + p.rows_per_chunk);
  const int lr_max = wave_max_i32(valid ? lr : 0);

  for (int i = i0; i < i1; ++i) {
    const int ll = lnlev[i], lf = lfirst[i];
    bool ok = valid;
    if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(lcat[i], catr, p.cat_mode);
    if (!__any(ok)) continue;
    // items without levels (types/comparable_data.py:255-258): two of them score 0 (a hit when 0 >= threshold); one
    // against an item with levels is the reference's IndexError, raised by the host before the launch: no hit here
    if (ll == 0) {
      if (__any(ok && lr == 0 && 0.0 >= p.threshold))
        emit_hits_wave(hits, p.cap, count, ok && lr == 0 && 0.0 >= p.threshold, 0.0, lorig[i], rorig[jc]);
      continue;
    }
    ok = ok && lr > 0;
    const int S = p.raw ? 1 : max(ll, lr);
    const int s_hi = p.raw ? 1 : max(ll, lr_max);
    // ---- length bound of the whole pair: sum of 2^-s x ratio(la, lb, LCS = min(la, lb)) -- the same expression the score
    // is built from, with an LCS that is never smaller, accumulated in the same order (every operation is monotone)
    double ub_rest = 0.0;
    if (ok) {
      double f = 1.0;
      for (int s = p.raw ? 0 : 1; s <= (p.raw ? 0 : S); ++s) {
        f *= 0.5;
        const int a = max(0, min(s, ll - 1)), b = max(0, min(s, lr - 1));
        const int la = static_cast<int>(loff[lf + a + 1] - loff[lf + a]), lb = static_cast<int>(roff[rr0 + b + 1] - roff[rr0 + b]);
        const double ub = any_indel_score(la, lb, min(la, lb));
        ub_rest = p.raw ? ub : ub_rest + ub * f;
      }
    }
    constexpr double kSlack = 1e-9;  // (the bound is subtracted step by step below: rounding, far above 2^-52)
    ok = ok && ub_rest + kSlack >= p.threshold;
    if (!__any(ok)) continue;
    double score = 0.0, ratio = 0.0, factor = 1.0, ub_cur = 0.0;
    int prev_a = -1, prev_b = -1;
    for (int s = p.raw ? 0 : 1; s <= (p.raw ? 0 : s_hi); ++s) {
      factor *= 0.5;
      const int a = max(0, min(s, ll - 1));  // wave-uniform
      const int b = max(0, min(s, lr - 1));
      const bool run = ok && (p.raw || s <= S);
      const bool fresh = run && (a != prev_a || b != prev_b);
      if (__any(fresh)) {
        const long long la0 = loff[lf + a];
        const int la = static_cast<int>(loff[lf + a + 1] - la0);
        const long long lb0 = fresh ? roff[rr0 + b] : 0;
        const int lb = fresh ? static_cast<int>(roff[rr0 + b + 1] - lb0) : 0;
        const int nch = wave_max_i32(lb);
        int zeros = 0;
        for (int c0 = 0; c0 < la; c0 += kAnyChunk) {
          const int clen = min(kAnyChunk, la - c0);
          // match masks of pattern[c0 .. c0 + clen)
          for (int e = lane; e < (p.alphabet + 1) * kAnyLimbs; e += kWave) tbl[e] = 0u;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (int pos = lane; pos < clen; pos += kWave) {
            const unsigned sym = lcodes[la0 + c0 + pos];
            atomicOr(&tbl[sym * kAnyLimbs + (pos >> 5)], 1u << (pos & 31));
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          uint32_t v[kAnyLimbs];
#pragma unroll
          for (int k = 0; k < kAnyLimbs; ++k) v[k] = ~0u;
          uint32_t cin_word = 0, cout_word = 0;
          for (int t = 0; t < nch; ++t) {
            if ((t & 31) == 0) {
              cin_word = c0 > 0 ? carry[(t >> 5) * kWave + lane] : 0u;
              cout_word = 0u;
            }
            const unsigned sym = t < lb ? rcodes[lb0 + t] : static_cast<unsigned>(p.alphabet);
            unsigned long long carry_bit = (cin_word >> (t & 31)) & 1u;
            uint32_t nv[kAnyLimbs];
#pragma unroll
            for (int k = 0; k < kAnyLimbs; ++k) {
              const uint32_t m = tbl[sym * kAnyLimbs + k];
              const uint32_t u = v[k] & m;
              const unsigned long long sum = static_cast<unsigned long long>(v[k]) + u + carry_bit;
              carry_bit = sum >> 32;
              nv[k] = static_cast<uint32_t>(sum) | (v[k] ^ u);
            }
#pragma unroll
            for (int k = 0; k < kAnyLimbs; ++k) v[k] = nv[k];
            cout_word |= static_cast<uint32_t>(carry_bit) << (t & 31);
            if ((t & 31) == 31 || t == nch - 1) carry[(t >> 5) * kWave + lane] = cout_word;
          }
#pragma unroll
          for (int k = 0; k < kAnyLimbs; ++k) zeros += __popc(~v[k]);
        }
        if (fresh) {
          ratio = any_indel_score(la, lb, zeros);
          ub_cur = any_indel_score(la, lb, min(la, lb));
          prev_a = a;
          prev_b = b;
        }
      }
      if (run) {
        score = p.raw ? ratio : score + ratio * factor;
        // what the steps still to come can add at most; a lane that cannot reach the threshold any more stops (its score
        // is then below the threshold whatever follows: no hit is lost)
        ub_rest = p.raw ? 0.0 : ub_rest - ub_cur * factor;
        if (s < S && score + ub_rest + kSlack < p.threshold) ok = false;
      }
      if (!__any(ok)) break;
    }
    emit_hits_wave(hits, p.cap, count, ok && score >= p.threshold, score, lorig[i], rorig[jc]);
  }
}

// ---------------------------------------------------------------------------------------------------- Jaccard
__global__ __launch_bounds__(kWave) void jaccard_any_kernel(
    const int32_t* __restrict__ lids, const uint8_t* __restrict__ llv, const long long* __restrict__ loff,
    const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lplen /*[n][max_levels]*/, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ rids, const uint8_t* __restrict__ rlv,
    const long long* __restrict__ roff, const int32_t* __restrict__ rnlev, const int32_t* __restrict__ rplen,
    const int32_t* __restrict__ rorig, const uint64_t* __restrict__ rcat, int max_levels, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const AnyParams p, const int32_t* __restrict__ lfirst,
    const int32_t* __restrict__ rfirst) {
  // LDS: [65][64] u16: per lane, ids first common from step s on ([step][lane])
  __shared__ uint16_t s_first[(kAnyMaxLevels + 1) * kWave];
  const int lane = threadIdx.x;
  const int j = blockIdx.x * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;
  const int lr = rnlev[jc];
  const bool independent = lfirst != nullptr;  // (both sides alike: checked by the launcher)
  const long long rb0 = independent ? 0 : roff[jc];
  const int nb = independent ? 0 : static_cast<int>(roff[jc + 1] - rb0);
  const int rf = independent ? rfirst[jc] : 0;
  const uint64_t catr = p.cat_mode != NSM_CAT_NONE ? rcat[jc] : 0ull;
  const int i0 = blockIdx.y * p.rows_per_chunk, i1 = min(p.n_left, i0 + p.rows_per_chunk);
  for (int i = i0; i < i1; ++i) {
    const int ll = lnlev[i];
    bool ok = valid;
    if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(lcat[i], catr, p.cat_mode);
    if (!__any(ok)) continue;
    if (ll == 0) {  // (as in indel_any_kernel: (0, 0) pairs score 0, mixed pairs never hit)
      if (__any(ok && lr == 0 && 0.0 >= p.threshold))
        emit_hits_wave(hits, p.cap, count, ok && lr == 0 && 0.0 >= p.threshold, 0.0, lorig[i], rorig[jc]);
      continue;
    }
    ok = ok && lr > 0;
    if (independent) {
      // every step's two level rows are merged on their own (lane-local): nothing is assumed about how the levels of an
      // item relate, and an item may have any number of them
      double score = 0.0, factor = 1.0, q = 0.0;
      if (ok) {  // size bound of the whole pair: |A n B| <= min, |A u B| >= max at every step
        const int lf = lfirst[i];
        const int S = p.raw ? 0 : max(ll, lr);
        double ub = 0.0, f = 1.0;
        for (int s = p.raw ? 0 : 1; s <= S; ++s) {
          f *= 0.5;
          const int a = max(0, min(s, ll - 1)), b = max(0, min(s, lr - 1));
          const int na = static_cast<int>(loff[lf + a + 1] - loff[lf + a]), nb2 = static_cast<int>(roff[rf + b + 1] - roff[rf + b]);
          const double qb = max(na, nb2) > 0 ? static_cast<double>(min(na, nb2)) / static_cast<double>(max(na, nb2)) : 0.0;
          ub = p.raw ? qb : ub + qb * f;
        }
        ok = ub + 1e-9 >= p.threshold;
      }
      if (ok) {
        const int lf = lfirst[i];
        const int S = p.raw ? 0 : max(ll, lr);
        int prev_a = -1, prev_b = -1;
        for (int s = p.raw ? 0 : 1; s <= S; ++s) {
          factor *= 0.5;
          const int a = max(0, min(s, ll - 1)), b = max(0, min(s, lr - 1));
          if (a != prev_a || b != prev_b) {
            prev_a = a;
            prev_b = b;
            const long long a0 = loff[lf + a], b0 = roff[rf + b];
            const int na = static_cast<int>(loff[lf + a + 1] - a0), nb2 = static_cast<int>(roff[rf + b + 1] - b0);
            int x = 0, y = 0, inter = 0;
            int ida = na > 0 ? lids[a0] : 0, idb = nb2 > 0 ? rids[b0] : 0;
            while (x < na && y < nb2) {
              if (ida == idb) {
                ++inter; ++x; ++y;
                if (x < na) ida = lids[a0 + x];
                if (y < nb2) idb = rids[b0 + y];
              } else if (ida < idb) {
                ++x;
                if (x < na) ida = lids[a0 + x];
              } else {
                ++y;
                if (y < nb2) idb = rids[b0 + y];
              }
            }
            const int uni = na + nb2 - inter;
            q = uni > 0 ? static_cast<double>(inter) / static_cast<double>(uni) : 0.0;  // (0 / 0: the host raises beforehand)
          }
          score = p.raw ? q : score + q * factor;
        }
      }
      emit_hits_wave(hits, p.cap, count, ok && score >= p.threshold, score, lorig[i], rorig[jc]);
      continue;
    }
    const long long la0 = loff[i];
    const int na = static_cast<int>(loff[i + 1] - la0);
    const int S = p.raw ? 1 : max(ll, lr);
    if (ok) {  // size bound of the whole pair (|A n B| <= min, |A u B| >= max at every step): no merge for the rest
      double ub = 0.0, f = 1.0;
      if (p.raw) {
        ub = max(na, nb) > 0 ? static_cast<double>(min(na, nb)) / static_cast<double>(max(na, nb)) : 0.0;
      } else {
        for (int s = 1; s <= S; ++s) {
          f *= 0.5;
          const int ca = lplen[static_cast<size_t>(i) * max_levels + min(s, ll - 1)];
          const int cb = rplen[static_cast<size_t>(jc) * max_levels + min(s, lr - 1)];
          ub += (max(ca, cb) > 0 ? static_cast<double>(min(ca, cb)) / static_cast<double>(max(ca, cb)) : 0.0) * f;
        }
      }
      ok = ub + 1e-9 >= p.threshold;
    }
    if (!__any(ok)) continue;
    for (int s = 0; s <= kAnyMaxLevels; ++s) s_first[s * kWave + lane] = 0;
    if (ok) {  // two-pointer merge of the id lists (both sorted by id, unique per item)
      int x = 0, y = 0;
      int ida = na > 0 ? lids[la0] : 0, idb = nb > 0 ? rids[rb0] : 0;
      while (x < na && y < nb) {
        if (ida == idb) {
          // the id is in left level l for every l >= llv, i.e. at step s (level min(s, ll - 1)) from step max(1, llv)
          // on -- RAW: the single level 0, step "0"
          const int sa = p.raw ? 0 : max(1, static_cast<int>(llv[la0 + x]));
          const int sb = p.raw ? 0 : max(1, static_cast<int>(rlv[rb0 + y]));
          s_first[max(sa, sb) * kWave + lane] += 1;
          ++x;
          ++y;
          if (x < na) ida = lids[la0 + x];
          if (y < nb) idb = rids[rb0 + y];
        } else if (ida < idb) {
          ++x;
          if (x < na) ida = lids[la0 + x];
        } else {
          ++y;
          if (y < nb) idb = rids[rb0 + y];
        }
      }
    }
    double score = 0.0, factor = 1.0;
    int inter = 0;
    if (ok) {
      if (p.raw) {
        inter = s_first[lane];
        const int uni = na + nb - inter;
        score = uni > 0 ? static_cast<double>(inter) / static_cast<double>(uni) : 0.0;  // (0 / 0: the host raises beforehand)
      } else {
        for (int s = 1; s <= S; ++s) {
          factor *= 0.5;
          inter += s_first[min(s, kAnyMaxLevels) * kWave + lane];
          const int ca = lplen[static_cast<size_t>(i) * max_levels + min(s, ll - 1)];
          const int cb = rplen[static_cast<size_t>(jc) * max_levels + min(s, lr - 1)];
          const int uni = ca + cb - inter;
          const double q = uni > 0 ? static_cast<double>(inter) / static_cast<double>(uni) : 0.0;
          score += q * factor;
        }
      }
    }
    emit_hits_wave(hits, p.cap, count, ok && score >= p.threshold, score, lorig[i], rorig[jc]);
  }
}

}  // namespace nsm

extern "C" int nsm_indel_any_grid(const nsm_any_items* left, const nsm_any_strings* left_strings, const nsm_any_items* right,
                                  const nsm_any_strings* right_strings, double threshold, int32_t category_mode, uint32_t flags,
                                  nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, void* stream) {
  using namespace nsm;
  if (!left || !right || !left_strings || !right_strings || !hit_count || (!hits && capacity)) {
    set_error("nsm_indel_any_grid: null argument");
    return NSM_E_BADARG;
  }
  if (left_strings->alphabet != right_strings->alphabet || left_strings->alphabet < 1 || left_strings->alphabet > kAnyMaxAlphabet) {
    set_error("nsm_indel_any_grid: alphabets differ or exceed %d symbols", kAnyMaxAlphabet);
    return left_strings->alphabet > kAnyMaxAlphabet ? NSM_E_UNSUPPORTED : NSM_E_BADARG;
  }
  if (left_strings->max_len > kAnyMaxText || right_strings->max_len > kAnyMaxText) {
    set_error("nsm_indel_any_grid: a string has more than %d code units", kAnyMaxText);
    return NSM_E_UNSUPPORTED;
  }
  if (category_mode != NSM_CAT_NONE && (!left->cat || !right->cat)) {
    set_error("nsm_indel_any_grid: category masks missing");
    return NSM_E_BADARG;
  }
  if (left->n <= 0 || right->n <= 0) return left->n < 0 || right->n < 0 ? NSM_E_BADARG : 0;
  AnyParams p;
  p.n_left = left->n; p.n_right = right->n; p.cap = capacity; p.threshold = threshold;
  p.alphabet = left_strings->alphabet; p.cat_mode = category_mode; p.raw = (flags & NSM_FLAG_RAW_SCORE) ? 1 : 0;
  const int n_tiles = (right->n + kWave - 1) / kWave;
  long long chunks = (256ll * 16 + n_tiles - 1) / n_tiles;
  long long rows = (left->n + chunks - 1) / chunks;
  if (rows < 1) rows = 1;
  p.rows_per_chunk = static_cast<int>(rows);
  dim3 grid(n_tiles, static_cast<unsigned>((left->n + rows - 1) / rows));
  const size_t lds = (static_cast<size_t>(p.alphabet + 1) * kAnyLimbs + (kAnyMaxText / 32) * kWave) * 4;
  hipLaunchKernelGGL(indel_any_kernel, grid, dim3(kWave), lds, static_cast<hipStream_t>(stream), left->first, left->nlev,
                     left->orig, left->cat, left_strings->codes, reinterpret_cast<const long long*>(left_strings->offset),
                     right->first, right->nlev, right->orig, right->cat, right_strings->codes,
                     reinterpret_cast<const long long*>(right_strings->offset), hits, hit_count, p);
  return hip_status(hipGetLastError(), "indel_any_kernel launch");
}

extern "C" int nsm_jaccard_any_grid(const nsm_any_sets* left, const nsm_any_sets* right, double threshold, int32_t category_mode,
                                    uint32_t flags, nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count,
                                    void* stream) {
  using namespace nsm;
  if (!left || !right || !hit_count || (!hits && capacity)) {
    set_error("nsm_jaccard_any_grid: null argument");
    return NSM_E_BADARG;
  }
  if ((left->first == nullptr) != (right->first == nullptr)) {
    set_error("nsm_jaccard_any_grid: both sides in the same layout (nested, or independent levels with `first`)");
    return NSM_E_BADARG;
  }
  if (!left->first && (left->max_levels != right->max_levels || left->max_levels < 1 || left->max_levels > kAnyMaxLevels)) {
    set_error("nsm_jaccard_any_grid: max_levels %d / %d (both sides alike, 1..%d in the nested layout)", left->max_levels,
              right->max_levels, kAnyMaxLevels);
    return left->max_levels > kAnyMaxLevels || right->max_levels > kAnyMaxLevels ? NSM_E_UNSUPPORTED : NSM_E_BADARG;
  }
  if (left->max_ids > 65535 || right->max_ids > 65535) {
    set_error("nsm_jaccard_any_grid: an item has more than 65535 ids");
    return NSM_E_UNSUPPORTED;
  }
  if (category_mode != NSM_CAT_NONE && (!left->cat || !right->cat)) {
    set_error("nsm_jaccard_any_grid: category masks missing");
    return NSM_E_BADARG;
  }
  if (left->n <= 0 || right->n <= 0) return left->n < 0 || right->n < 0 ? NSM_E_BADARG : 0;
  AnyParams p;
  p.n_left = left->n; p.n_right = right->n; p.cap = capacity; p.threshold = threshold;
  p.alphabet = 0; p.cat_mode = category_mode; p.raw = (flags & NSM_FLAG_RAW_SCORE) ? 1 : 0;
  const int n_tiles = (right->n + kWave - 1) / kWave;
  long long chunks = (256ll * 16 + n_tiles - 1) / n_tiles;
  long long rows = (left->n + chunks - 1) / chunks;
  if (rows < 1) rows = 1;
  p.rows_per_chunk = static_cast<int>(rows);
  dim3 grid(n_tiles, static_cast<unsigned>((left->n + rows - 1) / rows));
  hipLaunchKernelGGL(jaccard_any_kernel, grid, dim3(kWave), 0, static_cast<hipStream_t>(stream), left->ids, left->lv,
                     reinterpret_cast<const long long*>(left->offset), left->nlev, left->plen, left->orig, left->cat, right->ids,
                     right->lv, reinterpret_cast<const long long*>(right->offset), right->nlev, right->plen, right->orig,
                     right->cat, left->max_levels, hits, hit_count, p, left->first, right->first);
  return hip_status(hipGetLastError(), "jaccard_any_kernel launch");
}
