// Levels-mode Jaccard grid: the hot loop of gen_comparable with score_func = intersection_vs_union
// (reference: napkon_string_matching/types/comparable_data.py:223-232 calling compare_terms :248-265
// and intersection_vs_union, compare/score_functions.py:6-13; category predicate :464-490).
//
//   score(i, j) = sum_{s=1..max(Ll,Lr)} 2^-s * |A_s n B_s| / |A_s u B_s|,
//   A_s = level min(s, Ll-1) of left item i,  B_s = level min(s, Lr-1) of right item j,
//   accumulated in double in that order (the weights are exact powers of two).
//
// Levels are suffix-nested (gen_comp_value, :283-285), so an item is ONE row of unique ids in
// first-appearance order and level l is its first plen[l] ids.  Per pair the kernel
//   1. finds, for every left position a, the right position pos[a] of the same id (or none) with the
//      xor/min3 matrix of the RAW kernel: ids are stored pre-shifted, (id << 6) | position on the
//      right and id << 6 on the left, so min_b (la ^ rb) is < 64 exactly when `a` occurs in the right
//      row and then IS its position -- 1.5 VALU ops per id pair;
//   2. for every step s counts the a < plenL[s] with pos[a] < plenR[s] (byte-parallel compare on
//      the packed pos words) -- that is |A_s n B_s|; the union follows from the two prefix lengths;
//   3. divides in double and accumulates; only lanes with at least one common id get here.
// Lane = right item, left item wave-uniform via scalar loads, size-class dispatch as in the RAW kernel.
#pragma once
#include "nsm_common.hpp"

namespace nsm {

template <int W>
struct JacLevScalars {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t lev_stride_l;  // row stride of plen (left / right)
  int32_t lev_stride_r;
  int32_t cat_mode;
  int32_t emit_all;      // threshold <= 0: every pair that passes the category predicate is a hit
  double threshold;
  unsigned long long cap;
};

// pos bytes of NL left positions against the lane's NB right ids; returns true if any matched.
template <int W, int NL, int NB>
__device__ __forceinline__ bool match_positions(const int32_t* __restrict__ lrow, const uint32_t (&r)[W],
                                                uint32_t (&posw)[W / 4]) {
  static_assert(NB >= 2 && NB % 2 == 0, "right class must be even");
  uint32_t best = 0xffffffffu;
#pragma unroll
  for (int q = 0; q < W / 4; ++q) posw[q] = 0xffffffffu;
#pragma unroll
  for (int a = 0; a < NL; ++a) {
    const uint32_t la = static_cast<uint32_t>(lrow[a]) << 6;  // SALU
    uint32_t m = min(la ^ r[0], la ^ r[1]);
#pragma unroll
    for (int b = 2; b < NB; b += 2) m = min(m, min(la ^ r[b], la ^ r[b + 1]));
    best = min(best, m);
    const uint32_t byte = min(m, 255u);
    // replace byte (a & 3) of word a >> 2 (it holds 0xff)
    posw[a >> 2] = (posw[a >> 2] & ~(0xffu << (8 * (a & 3)))) | (byte << (8 * (a & 3)));
  }
  return best < 64u;
}

template <int W, int NB>
__device__ __forceinline__ void levels_wave_rows(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const uint64_t* __restrict__ lsig,
    const int32_t* __restrict__ lorig, const int32_t* __restrict__ lnlev, const uint8_t* __restrict__ lplen,
    const uint64_t* __restrict__ lcat, const uint8_t* __restrict__ rplen_row, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const JacLevScalars<W>& p, const uint32_t (&r)[W], uint64_t sr,
    uint64_t catr, int lr, int jorig, bool valid, int i0, int i1) {
  constexpr int NLS = W / 4;
  for (int i = i0; i < i1; ++i) {
    bool ok = valid;
    if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(lcat[i], catr, p.cat_mode);
    bool want = ok;
    if (!p.emit_all) want = ok && ((lsig[i] & sr) != 0ull);  // no common hash bit -> score 0
    if (!__any(want)) continue;

    const int nl = lcnt[i];
    const int32_t* __restrict__ lrow = lids + static_cast<size_t>(i) * W;
    const int cls = (nl + NLS - 1) / NLS;
    uint32_t posw[W / 4];
    bool any;
    switch (cls) {
      case 0:
#pragma unroll
        for (int q = 0; q < W / 4; ++q) posw[q] = 0xffffffffu;
        any = false;
        break;
      case 1: any = match_positions<W, NLS, NB>(lrow, r, posw); break;
      case 2: any = match_positions<W, 2 * NLS, NB>(lrow, r, posw); break;
      case 3: any = match_positions<W, 3 * NLS, NB>(lrow, r, posw); break;
      default: any = match_positions<W, W, NB>(lrow, r, posw); break;
    }
    double score = 0.0;
    const bool work = ok && any;
    if (__any(work)) {
      if (work) {
        const int ll = lnlev[i];
        const uint8_t* __restrict__ lpl = lplen + static_cast<size_t>(i) * p.lev_stride_l;
        const int steps = max(ll, lr);
        double factor = 1.0;
        for (int s = 1; s <= steps; ++s) {
          const int pl = lpl[min(s, p.lev_stride_l - 1)];        // = plen[min(s, Ll-1)]: rows are padded
          const int pr = rplen_row[min(s, p.lev_stride_r - 1)];  //   with their last value
          const uint32_t prrep = static_cast<uint32_t>(pr) * 0x01010101u;
          int inter = 0;
#pragma unroll
          for (int q = 0; q < W / 4; ++q) {
            if (4 * q < pl) {
              uint32_t x = posw[q];
              const int keep = pl - 4 * q;  // bytes of this word that belong to the level
              if (keep < 4) x |= 0xffffffffu << (8 * keep);
              // per byte: x < pr  (x < 128 or x == 0xff; pr <= 64)
              const uint32_t y = (x | 0x80808080u) - prrep;
              inter += __popc(~(y | x) & 0x80808080u);
            }
          }
          const int uni = pl + pr - inter;
          const double part = uni ? static_cast<double>(inter) / static_cast<double>(uni) : 0.0;
          factor *= 0.5;
          score += part * factor;
        }
      }
    }
    const bool hit = ok && (p.emit_all ? (score >= p.threshold) : (work && score >= p.threshold));
    if (__any(hit)) {
      if (hit) emit_hit(hits, p.cap, count, score, lorig[i], jorig);
    }
  }
}

template <int W>
__global__ __launch_bounds__(kBlock) void jaccard_levels_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const uint64_t* __restrict__ lsig,
    const int32_t* __restrict__ lorig, const int32_t* __restrict__ lnlev, const uint8_t* __restrict__ lplen,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ rids, const int32_t* __restrict__ rcnt,
    const uint64_t* __restrict__ rsig, const int32_t* __restrict__ rorig, const int32_t* __restrict__ rnlev,
    const uint8_t* __restrict__ rplen, const uint64_t* __restrict__ rcat, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const JacLevScalars<W> p) {
  const int lane = threadIdx.x & (kWave - 1);
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (tile * kWave >= p.n_right) return;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;

  uint32_t r[W];
  const uint4* rp = reinterpret_cast<const uint4*>(rids + static_cast<size_t>(jc) * W);
#pragma unroll
  for (int q = 0; q < W / 4; ++q) {
    const uint4 v = rp[q];
    r[4 * q + 0] = (v.x << 6) | (4 * q + 0);
    r[4 * q + 1] = (v.y << 6) | (4 * q + 1);
    r[4 * q + 2] = (v.z << 6) | (4 * q + 2);
    r[4 * q + 3] = (v.w << 6) | (4 * q + 3);
  }
  const int nrj = valid ? rcnt[jc] : 0;
  const uint64_t sr = valid ? (rsig[jc] & ((1ull << 58) - 1)) : 0ull;  // hash bits only
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  const int lr = rnlev[jc];
  const int jorig = rorig[jc];
  const uint8_t* rplen_row = rplen + static_cast<size_t>(jc) * p.lev_stride_r;
  const int nbmax = wave_first(nrj);

  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);

  constexpr int NBS = W / 8;
  const int cls = (nbmax + NBS - 1) / NBS;
#define NSM_LEV_CASE(K)                                                                                \
  levels_wave_rows<W, (K) * NBS>(lids, lcnt, lsig, lorig, lnlev, lplen, lcat, rplen_row, hits, count, p, r, \
                                 sr, catr, lr, jorig, valid, i0, i1)
  switch (cls) {
    case 0:
    case 1: NSM_LEV_CASE(1); break;
    case 2: NSM_LEV_CASE(2); break;
    case 3: NSM_LEV_CASE(3); break;
    case 4: NSM_LEV_CASE(4); break;
    case 5: NSM_LEV_CASE(5); break;
    case 6: NSM_LEV_CASE(6); break;
    case 7: NSM_LEV_CASE(7); break;
    default: NSM_LEV_CASE(8); break;
  }
#undef NSM_LEV_CASE
}

inline int lev_rows_per_chunk(int n_left, int n_tiles) {
  const long long want_waves = 16ll * 256 * 32;
  long long chunks = (want_waves + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
  if (chunks < 1) chunks = 1;
  long long rows = (n_left + chunks - 1) / chunks;
  if (rows < 128) rows = 128;
  if (rows > 4096) rows = 4096;
  return static_cast<int>(rows);
}

template <int W>
int launch_levels(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode,
                  uint32_t flags, nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count,
                  hipStream_t stream) {
  (void)flags;  // the signature test is exact and always on (it is disabled by emit_all)
  JacLevScalars<W> p;
  p.n_left = l->n; p.n_right = r->n; p.cap = capacity;
  p.lev_stride_l = l->max_levels; p.lev_stride_r = r->max_levels;
  p.cat_mode = category_mode;
  p.threshold = threshold;
  p.emit_all = !(threshold > 0.0);  // also true for NaN: then nothing compares >= and nothing is emitted
  const int n_tiles = (r->n + kWave - 1) / kWave;
  p.rows_per_chunk = lev_rows_per_chunk(l->n, n_tiles);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (l->n + 65534) / 65535;
    grid.y = (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  hipLaunchKernelGGL((jaccard_levels_kernel<W>), grid, dim3(kBlock), 0, stream, l->ids, l->cnt, l->sig, l->orig,
                     l->nlev, l->plen, l->cat, r->ids, r->cnt, r->sig, r->orig, r->nlev, r->plen, r->cat, hits,
                     hit_count, p);
  return hip_status(hipGetLastError(), "jaccard_levels_kernel launch");
}

}  // namespace nsm
