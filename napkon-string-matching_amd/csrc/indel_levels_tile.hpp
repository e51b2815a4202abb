// Levels-mode Indel-ratio grid for MULTI-WORD level strings (65 .. 512 code units: what the reference's default
// configuration -- compare_column Term, score_func fuzzy_match, config.yml:13-14 -- produces), "shared tile" kernel.
// Included by indel_levels.hip (which owns the ratio helpers); K = 2, 4, 8 words of 64 bits per string row.
//
// Reference: types/comparable_data.py:223-232 -> compare_terms (:248-265) x fuzzy_match
// (compare/score_functions.py:20-27):  score = sum_{s=1..S} 2^-s * ratio(A[min(s,La-1)], B[min(s,Lb-1)]).
//
// Why this shape (round 3).  The park kernel (indel_levels_park.hpp) gives every wavefront its own right tile, so
// every wavefront needs its own LDS image of its 64 texts: 16 KB per wave at 256 code units, which caps a CU at
// six waves (1.5 per SIMD), and it re-stages texts and gathers level strings from global memory (95.7 GB of traffic
// per launch of the term bench).  Here ALL the waves of a block share ONE right tile of 64 items and divide the LEFT
// rows among themselves:
//
//   images   the level strings the tile's items use at steps 1 .. n_img (3 at K = 4), one LDS image per step, staged
//            once per block and read by every wave ([dword][lane] layout, conflict-free), with their lengths and
//            histograms.  One block per CU (all 160 KB): 11 waves at K = 4.  Because the images of ALL steps stay
//            resident, a row is carried through its steps one after the other (row-major) without re-staging texts;
//   scan     per batch of 4 left rows a wave stages their heads (lengths, histograms) and their step-1 / step-2
//            level strings in one go, then scores TWO rows per pass (two mask tables, one text read, two
//            independent carry chains) on 32-bit limbs (indel_tile_lcs.hpp).  After every step the histogram bound
//            of the next level pair decides which lanes are still alive; a row with more than park_max survivors
//            goes on wave-wide at the next step's image, fewer are parked;
//   dense    the wave's OWN park (no block barrier, no atomics), drained after every batch: lane = one pair, its
//            text is a COLUMN of the resident image (no gather from global memory), one mask table per batch row;
//   balance  batches come from a block-shared counter; blocks are dealt to the XCDs so that each XCD walks the tiles
//            x, x + 8, ... slice-major (same left slice in its L2, same mix of long and short strings everywhere).
//
// Steps whose right level is not resident (items deeper than n_img + 1 levels) read their text from global memory.
// Every test that drops a pair is an upper bound: hits are identical to the wave-wide kernel's and the oracle's.
// Measurements, and what was tried and dropped: DESIGN.md section 4.4.
#pragma once
#include "indel_tile_lcs.hpp"

namespace nsm {

struct TileParams {
  int32_t n_left;
  int32_t n_right;
  int32_t n_tiles;
  int32_t n_lstr;         // rows of the left / right string tables: a zero-level item's `first` may point one past the end
  int32_t n_rstr;
  int32_t y_slices;       // grid = n_tiles * y_slices blocks (linear: the kernel maps them XCD-aware)
  int32_t rows_per_slice; // unpartitioned tables: left rows per slice
  int32_t pm_stride;      // match-mask entries per table: alphabet + 1 rounded up to 8
  int32_t cat_mode;
  int32_t use_hist;       // both string tables carry histograms and NSM_FLAG_PRUNE is set
  int32_t use_h1;         // step-1 pre-filter: histogram bound of the step-1 pair itself (else lengths only)
  int32_t n_img;          // resident text images = steps whose right level string is in LDS (>= 2)
  int32_t park_max;       // park a row's survivors when at most this many of the 64 lanes are alive
  int32_t park_slots;     // capacity of a wave's park: batch rows x park_max (a row parks at most once, <= park_max pairs)
  double threshold;
  unsigned long long cap;
};

#ifndef NSM_TILE_OCC
#define NSM_TILE_OCC
#endif

// Work counters of the shared-tile kernel (variant builds only: -DNSM_TILE_STATS; read by tools/tile_stats.py through
// nsm_debug_tile_stats): 0 batches, 1 two-row passes of step 1, 2 their iterations, 3 one-row passes of step 1,
// 4 their iterations, 5 two-row passes of later steps, 6 their iterations, 7 one-row passes of later steps, 8 their
// iterations, 9 dense calls, 10 dense LCS passes, 11 their iterations, 12 parked pairs, 13 table builds
#ifdef NSM_TILE_STATS
__device__ unsigned long long g_tile_stats[16];
#define NSM_STAT(slot, v)                                                                        \
  do {                                                                                           \
    if (lane == 0) atomicAdd(&g_tile_stats[slot], static_cast<unsigned long long>(v));           \
  } while (0)
#else
#define NSM_STAT(slot, v) \
  do {                    \
  } while (0)
#endif

// waves per block the kernel is compiled for: 16 (128 VGPRs) for 128-unit strings, 12 (168 VGPRs) for 256, 8 for 512 --
// what the LDS of a CU holds beside the images anyway
constexpr int tile_max_waves(int K) { return K <= 2 ? 16 : K == 4 ? 12 : 8; }

template <int K>
__global__ __launch_bounds__(tile_max_waves(K) * kWave) NSM_TILE_OCC void indel_levels_tile_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ lsegstart, const uint8_t* __restrict__ lcodes,
    const int32_t* __restrict__ llen, const uint8_t* __restrict__ lhist, const int32_t* __restrict__ rfirst,
    const int32_t* __restrict__ rnlev, const int32_t* __restrict__ rorig, const uint64_t* __restrict__ rcat,
    const int32_t* __restrict__ rseg, const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen,
    const uint8_t* __restrict__ rhist, nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count,
    const TileParams p) {
  // LDS (dynamic):
  //   block: [n_img][16 K][64] u32 text images | [n_hist][8][64] u32 right histograms | [n_hist][64] i32 right lengths
  //          | [64] i32 levels | [64] i32 first level row | [66] float4 step weights by S | batch counter (16 B)
  //   wave:  [4] mask tables | [4][3][12] u32 heads | [2][4][64 K] u8 left level strings of steps 1, 2
  //          | [4][2] i32 (levels, first row) | park: [slots] f64 score, [slots] u32 meta
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  constexpr int kRow = kWave * K;   // code units per string row
  constexpr int kDw = 16 * K;       // dwords per string row
  constexpr int NB = 8;             // histogram dwords per level string
  constexpr int kBatch = tile_batch(K);
  constexpr int kBatchLog = kBatch == 8 ? 3 : 2;
  static_assert(kBatch == 4 || kBatch == 8, "4 or 8 left rows per batch");
  const int waves = blockDim.x >> 6;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int n_img = p.n_img;

  const int n_hist = max(n_img, 3);  // lengths and histograms of the steps' right level strings: steps 1..3 at least (H phase)
  uint32_t* img = reinterpret_cast<uint32_t*>(s_mem);
  uint32_t* s_rhist = img + static_cast<size_t>(n_img) * kDw * kWave;
  int32_t* s_rlen = reinterpret_cast<int32_t*>(s_rhist + n_hist * NB * kWave);
  int32_t* s_lr = s_rlen + n_hist * kWave;
  int32_t* s_rrow0 = s_lr + kWave;
  float4* wtab = reinterpret_cast<float4*>(s_rrow0 + kWave);
  int* s_next = reinterpret_cast<int*>(wtab + 66);  // next batch of the block (16 bytes reserved)
  unsigned char* wbase0 = reinterpret_cast<unsigned char*>(wtab + 67);
  const int tbl_entries = p.pm_stride * kTileWords<K>;
  // distance between a wave's tables: tbl_entries u64 plus a skew, so that the SAME symbol of different tables (the dense
  // pass reads one table per batch row) does not sit on the same LDS banks
  const int tbl_stride = tbl_entries + kTileTableSkew;
  const size_t wave_bytes = static_cast<size_t>(kBatch) * tbl_stride * 8 + kBatch * 3 * kTileHead * 4 +
                            2 * kBatch * kRow + kBatch * 2 * 4 + static_cast<size_t>(p.park_slots) * 12;
  unsigned char* wbase = wbase0 + wave * ((wave_bytes + 15) & ~static_cast<size_t>(15));
  unsigned long long* pm = reinterpret_cast<unsigned long long*>(wbase);
  double* pk_score = reinterpret_cast<double*>(pm + static_cast<size_t>(kBatch) * tbl_stride);
  uint32_t* pk_meta = reinterpret_cast<uint32_t*>(pk_score + p.park_slots);
  uint32_t* head = pk_meta + p.park_slots;
  int32_t* srow = reinterpret_cast<int32_t*>(head + kBatch * 3 * kTileHead);
  uint8_t* lstr = reinterpret_cast<uint8_t*>(srow + kBatch * 2);

  // ---- which (right tile, left slice) this block works on.  Blocks are dealt to the 8 XCDs round-robin (block b
  // runs on XCD b % 8) and every XCD has its own 4 MB L2: each XCD walks its tiles slice-major, so the blocks resident
  // on one XCD at the same time read the SAME left slice out of that XCD's L2.  XCD x takes the tiles x, x + 8, x + 16, ...:
  // the tables are sorted by depth and length, so a contiguous eighth of the tiles would give XCD 0 the deepest items
  // with the longest strings and XCD 7 the cheapest ones (measured: the chip 62 % occupied on average).
  int tile, yslice;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    const int mine = (p.n_tiles - xcd + 7) >> 3;    // tiles of this XCD
    // the linear grid has 8 * ceil(n_tiles / 8) * y_slices blocks: block k of this XCD = (slice k / mine, its tile k % mine)
    if (mine <= 0 || k >= mine * p.y_slices) return;  // (whole block: before any barrier)
    yslice = k / mine;
    tile = xcd + 8 * (k - yslice * mine);
  }
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;
  const bool partitioned = rseg != nullptr;
  const int myseg = partitioned ? rseg[jc] : 0;
  const bool use_hist = p.use_hist != 0;

  // ---- the lane's right item; per step t + 1 (t < 3) the level string's length and histogram in registers (H phase)
  const int lr = rnlev[jc];
  const int rrow0 = min(rfirst[jc], p.n_rstr - 1);  // (a trailing zero-level item points one past the table)
  const int jorig = rorig[jc];
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  // ---- block setup: images, right-side tables, weights
  if (wave == 0) {
    s_lr[lane] = lr;
    s_rrow0[lane] = rrow0;
    if (lane == 0) *s_next = 0;
  }
  for (int S = threadIdx.x; S < 66; S += blockDim.x) {
    // step weights of a pair with S steps: R = w2 * ub2 + w3 * ub3 + c bounds the steps >= 2 (rest_bound with both
    // histogram levels), stored as {w2, w3, threshold - (w2 + w3 + c)}  (as in the park kernel)
    const double pS = __builtin_ldexp(1.0, -S);
    const double w2 = S >= 2 ? 0.25 : 0.0;
    const double w3 = S >= 3 ? 0.125 + (S <= 4 ? 0.125 - pS : 0.0) : 0.0;
    const double c = S > 4 ? 0.125 - pS : 0.0;
    wtab[S] = make_float4(static_cast<float>(w2), static_cast<float>(w3), static_cast<float>(p.threshold - (w2 + w3 + c)), 0.0f);
  }
  for (int t = wave; t < n_hist; t += waves) {  // image t = the level strings of step t + 1
    const int row = rrow0 + max(0, min(t + 1, lr - 1));
    s_rlen[t * kWave + lane] = rlen[row];
    uint32_t h[NB];
    if (use_hist) load_hist<NB>(rhist, row, h);
    else
#pragma unroll
      for (int q = 0; q < NB; ++q) h[q] = 0u;
#pragma unroll
    for (int q = 0; q < NB; ++q) s_rhist[(t * NB + q) * kWave + lane] = h[q];
  }
  for (int u = wave; u < n_img * 4 * K; u += waves) {  // (image, 16-byte piece) pairs dealt to the waves
    const int t = u / (4 * K), q = u - t * 4 * K;
    const int row = rrow0 + max(0, min(t + 1, lr - 1));
    const uint4 v = reinterpret_cast<const uint4*>(rcodes + static_cast<size_t>(row) * kRow)[q];
    uint32_t* dst = img + (static_cast<size_t>(t) * kDw + 4 * q) * kWave + lane;
    dst[0] = v.x;
    dst[kWave] = v.y;
    dst[2 * kWave] = v.z;
    dst[3 * kWave] = v.w;
  }
  __syncthreads();  // the only block barrier

  int pk_cnt = 0;  // wave-uniform: parked pairs
  int cur_ib = 0, cur_nrows = 0;

  // ---- build the match-mask table `tb` from a code-unit string (LDS or global), la code units
  auto build_table = [&](unsigned long long* tb, const uint8_t* src, int la) __attribute__((always_inline)) {
    NSM_STAT(13, 1);
#ifdef NSM_X_NOBUILD  // (timing experiments: stale tables)
    return;
#endif
    for (int c = lane; c < tbl_entries; c += kWave) tb[c] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int pos = lane + kWave * k;
      if (pos < la) atomicOr(&tb[static_cast<unsigned>(src[pos]) * kTileWords<K> + k], 1ull << lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // head fields of batch row r (wave-uniform reads)
  auto head_ll = [&](int r) -> int { return wave_first(static_cast<int>(head[r * 3 * kTileHead + NB + 2])); };
  auto head_lf = [&](int r) -> int { return wave_first(static_cast<int>(head[r * 3 * kTileHead + NB + 3])); };
  // the left level string of batch row r at step s: its table row and length (heads hold steps 1..3)
  auto left_level = [&](int r, int s, int& lrow, int& la) __attribute__((always_inline)) {
    if (s <= 3) {
      la = wave_first(static_cast<int>(head[(r * 3 + s - 1) * kTileHead + NB]));
      lrow = wave_first(static_cast<int>(head[(r * 3 + s - 1) * kTileHead + NB + 1]));
    } else {
      lrow = head_lf(r) + max(0, min(s, head_ll(r) - 1));
      la = wave_first(llen[lrow]);
    }
  };
  // is the right level of step s resident for an item with `levels` levels?  image t holds level min(t + 1, levels - 1)
  auto resident = [&](int s, int levels) -> bool { return s <= n_img || levels - 1 <= n_img; };
  auto image_of = [&](int s) -> int { return min(s, n_img) - 1; };  // (only meaningful when resident)

  // per-lane histogram bound of step s + 1's level pair (left batch row r, right item column jl of the tile)
  auto next_ub = [&](int r, int ll, int lf, int s, int jl, int lrj, int rr0) -> float {
    if (!use_hist) return 1.0f;
    const int t = s + 1;
    uint32_t a[NB], b[NB];
    int la_n, lb_n;
    if (t <= 3) {
      const uint32_t* rec = head + (r * 3 + t - 1) * kTileHead;
      la_n = static_cast<int>(rec[NB]);
#pragma unroll
      for (int q = 0; q < NB; ++q) a[q] = rec[q];
    } else {
      const int lrow_n = lf + max(0, min(t, ll - 1));
      la_n = llen[lrow_n];
      load_hist<NB>(lhist, lrow_n, a);
    }
    if (t <= n_hist || lrj - 1 <= n_hist) {  // (lengths and histograms are kept for n_hist >= n_img steps)
      const int ti = min(t, n_hist) - 1;
      lb_n = s_rlen[ti * kWave + jl];
#pragma unroll
      for (int q = 0; q < NB; ++q) b[q] = s_rhist[(ti * NB + q) * kWave + jl];
    } else {
      const int rrow_n = rr0 + max(0, min(t, lrj - 1));
      lb_n = rlen[rrow_n];
      load_hist<NB>(rhist, rrow_n, b);
    }
    return hist_ratio_ub(hist_l1<NB>(a, b), la_n, lb_n);
  };

  // ---- remaining steps of up to 64 pairs, lane = one pair: left batch row r, right item = column jl of the tile,
  // next step s0, score so far.  One mask table per batch row; a lane's text is a COLUMN of the resident image of its
  // step or, for levels that are not resident (items deeper than the images), its row in global memory.
  // `staged3`: the step-3 level strings of the batch's rows have been staged over their step-1 copies.
  auto dense_steps = [&](bool active, int r, int jl, int s0, double score, bool staged3) __attribute__((always_inline)) {
    const uint32_t* rec0 = head + r * 3 * kTileHead;
    const int ll = static_cast<int>(rec0[NB + 2]), lf = static_cast<int>(rec0[NB + 3]);
    const int lrj = s_lr[jl], rr0 = s_rrow0[jl];
    const int S = max(ll, lrj);
    const int s_lo = 64 - wave_max_i32(active ? 64 - min(s0, 64) : 0);
    const int s_hi = wave_max_i32(active ? S : 0);
    int prev_a = -1, prev_b = -1;
    double ratio = 0.0;
    bool alive = active;
    double factor = __builtin_ldexp(1.0, 1 - s_lo);
    for (int s = s_lo; s <= s_hi; ++s) {
      factor *= 0.5;
      const bool run = alive && s >= s0 && s <= S;
      if (!__any(run)) continue;
      const int a = max(0, min(s, ll - 1)), b = max(0, min(s, lrj - 1));
      const bool fresh = run && (a != prev_a || b != prev_b);
      if (__any(fresh)) {
        const bool res = resident(s, lrj);
        const int ti = image_of(s);
        int la, lbj;
        if (s <= 3) la = static_cast<int>(head[(r * 3 + s - 1) * kTileHead + NB]);
        else la = llen[lf + a];
        if (res) lbj = s_rlen[ti * kWave + jl];
        else lbj = rlen[rr0 + b];
        // the mask tables of the rows that have a fresh lane, at THEIR level of step s
        const uint32_t rows_here = wave_reduce_u32(fresh ? (1u << r) : 0u, [](uint32_t x, uint32_t y) { return x | y; });
        int la_max = 0;
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
          if ((rows_here >> q) & 1u) {
            int lrow_q, la_q;
            left_level(q, s, lrow_q, la_q);
            const uint8_t* src = s == 2 ? lstr + (kBatch + q) * kRow
                                 : (s == 3 && staged3) ? lstr + q * kRow : lcodes + static_cast<size_t>(lrow_q) * kRow;
            build_table(pm + static_cast<size_t>(q) * tbl_stride, src, la_q);
            la_max = max(la_max, la_q);
          }
        }
        const unsigned long long* tbl = pm + static_cast<size_t>(fresh ? r : 0) * tbl_stride;
        const int nchars = wave_max_i32(fresh ? lbj : 0);
        NSM_STAT(10, 1);
        NSM_STAT(11, (nchars + 3) / 4);
        int lcs;
        if (!__any(fresh && !res)) {
          const uint32_t* tcol = img + static_cast<size_t>(ti) * kDw * kWave + jl;
          lcs = tile_lcs1_any<K>(tbl, tcol, kWave, nchars, la_max);
        } else {  // some texts come from global memory (deep items): one generic pass, every limb live
          const uint32_t* tptr = res ? img + static_cast<size_t>(ti) * kDw * kWave + jl
                                     : reinterpret_cast<const uint32_t*>(rcodes + static_cast<size_t>(fresh ? rr0 + b : rr0) * kRow);
          lcs = tile_lcs1<K, 2 * K>(tbl, tptr, res ? kWave : 1, nchars);
        }
        if (fresh) {
          ratio = indel_score_dev(la, lbj, lcs);
          prev_a = a;
          prev_b = b;
        }
      }
      if (run) {
        score += ratio * factor;
        float rest = 0.0f;
        if (s < S) rest = rest_bound(s, S, next_ub(r, ll, lf, s, jl, lrj, rr0));
        alive = score + static_cast<double>(rest) + 1e-6 >= p.threshold;
      }
    }
    const int jo = (active && alive && score >= p.threshold) ? rorig[min(tile * kWave + jl, p.n_right - 1)] : 0;
    emit_hits_wave(hits, p.cap, count, active && alive && score >= p.threshold, score, lorig[cur_ib + r], jo);
  };

  // ---- the wave's park, drained at the end of every batch (the heads and staged strings of the batch go away with
  // the next one): 64 pairs per pass
  auto drain = [&]() __attribute__((always_inline)) {
#ifdef NSM_X_NODENSE  // (timing experiments: the parked pairs are dropped)
    pk_cnt = 0;
    return;
#endif
    // step 3 is where most parked pairs start: its level strings are staged over the (finished) step-1 copies, one
    // round trip for the batch instead of byte gathers per table
    {
      constexpr int kIter = kBatch * kDw / kWave;
      uint32_t v[kIter];
#pragma unroll
      for (int g = 0; g < kIter; ++g) {
        const int f = g * kWave + lane;
        const int d = f % kDw, r = f / kDw;
        const int lrow = srow[2 * r + 1] + max(0, min(3, srow[2 * r] - 1));
        v[g] = reinterpret_cast<const uint32_t*>(lcodes + static_cast<size_t>(lrow) * kRow)[d];
      }
#pragma unroll
      for (int g = 0; g < kIter; ++g) reinterpret_cast<uint32_t*>(lstr)[g * kWave + lane] = v[g];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    NSM_STAT(12, pk_cnt);
    for (int base = 0; base < pk_cnt; base += kWave) {
      NSM_STAT(9, 1);
      const bool active = base + lane < pk_cnt;
      const int slot = active ? base + lane : base;
      const uint32_t meta = pk_meta[slot];
      dense_steps(active, static_cast<int>(meta & (kBatch - 1)), static_cast<int>((meta >> kBatchLog) & 63u),
                  static_cast<int>(meta >> (kBatchLog + 6)),
                  pk_score[slot], true);
    }
    pk_cnt = 0;
  };

  // ---- what follows the LCS of (row r, step s) in the wave-wide scan: score, bound on the rest, emit / park / go on.
  // Returns (wave-uniform) whether the row goes on wave-wide with step s + 1; `alive` then marks the lanes that run it.
  auto after_step = [&](int r, int s, int la, int lb, int lcs, bool run, double& score, bool& alive)
      __attribute__((always_inline)) -> bool {
    const int ll = head_ll(r), lf = head_lf(r);
    const int S = max(ll, lr);
    if (run) score += indel_score_dev(la, lb, lcs) * __builtin_ldexp(1.0, -s);
    float rest = 0.0f;
    if (s < S) rest = rest_bound(s, S, next_ub(r, ll, lf, s, lane, lr, rrow0));
    alive = run && (score + static_cast<double>(rest) + 1e-6 >= p.threshold);
    if (__any(alive && s >= S)) {  // pairs whose last step this was
      emit_hits_wave(hits, p.cap, count, alive && s >= S && score >= p.threshold, score, lorig[cur_ib + r], jorig);
    }
    alive = alive && s < S;
    const unsigned long long who = __ballot(alive);
    if (who == 0ull) return false;
    const int n = __popcll(who);
    // many survivors go on wave-wide (the scan picks the text source: resident image, or global memory for deep items);
    // few are parked: a row parks at most once, so batch rows x park_max slots always suffice
    if (n > p.park_max) return true;
    if (alive) {
      const int slot = pk_cnt + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                          __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
      pk_score[slot] = score;
      pk_meta[slot] = static_cast<uint32_t>(r) | (static_cast<uint32_t>(lane) << kBatchLog) |
                      (static_cast<uint32_t>(s + 1) << (kBatchLog + 6));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    pk_cnt += n;
    alive = false;
    return false;
  };

  // ---- this wave's tile against the batch rows [ib, ib + nrows); okbits bit r = the lane passes the category
  // predicate for row ib + r
  auto scan_batch = [&](int ib, int nrows, uint32_t okbits, uint32_t rows_ok) __attribute__((always_inline)) {
    const uint32_t okbits_all = okbits;  // category predicate of every lane (zero-level right items included)
    auto zero_ok = [&](int r) -> bool { return ((okbits_all >> r) & 1u) != 0u; };
    cur_ib = ib;
    cur_nrows = nrows;
    NSM_STAT(0, 1);
    // ---- stage: (levels, first row) of the batch's rows, then heads and the level strings of steps 1 and 2
    {
      const int i = min(ib + (lane & (kBatch - 1)), p.n_left - 1);
      const int ll = lnlev[i], lf = min(lfirst[i], p.n_lstr - 1);  // (a trailing zero-level item points one past the table)
      if (lane < kBatch) {
        srow[2 * lane] = ll;
        srow[2 * lane + 1] = lf;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      constexpr int kIter = 2 * kBatch * kDw / kWave;  // dwords per lane of the staged strings
      uint32_t v[kIter];
#pragma unroll
      for (int g = 0; g < kIter; ++g) {
        const int f = g * kWave + lane;
        const int d = f % kDw, r = (f / kDw) % kBatch, t = f / (kDw * kBatch);
        const int lrow = srow[2 * r + 1] + max(0, min(t + 1, srow[2 * r] - 1));
        v[g] = reinterpret_cast<const uint32_t*>(lcodes + static_cast<size_t>(lrow) * kRow)[d];
      }
      if (lane < 3 * kBatch) {
        const int r = lane & (kBatch - 1), t = lane >> kBatchLog;
        const int hl = srow[2 * r], hf = srow[2 * r + 1];
        const int lrow = hf + max(0, min(t + 1, hl - 1));
        uint32_t h[NB];
        if (use_hist) load_hist<NB>(lhist, lrow, h);
        else
#pragma unroll
          for (int q = 0; q < NB; ++q) h[q] = 0u;
        uint32_t* rec = head + (r * 3 + t) * kTileHead;
#pragma unroll
        for (int q = 0; q < NB; ++q) rec[q] = h[q];
        rec[NB] = static_cast<uint32_t>(llen[lrow]);
        rec[NB + 1] = static_cast<uint32_t>(lrow);
        rec[NB + 2] = static_cast<uint32_t>(hl);
        rec[NB + 3] = static_cast<uint32_t>(hf);
      }
#pragma unroll
      for (int g = 0; g < kIter; ++g) reinterpret_cast<uint32_t*>(lstr)[g * kWave + lane] = v[g];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    // ---- zero-level items (types/comparable_data.py:255-258: two items without levels score 0, one without levels
    // against one with levels is the reference's IndexError, which the host raises before the launch): no step runs
    // for them.  A (0, 0) pair is a hit exactly when 0 >= threshold.
    okbits = lr > 0 ? okbits : 0u;
    for (uint32_t rows = rows_ok; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      if (head_ll(r) > 0) continue;
      rows_ok &= ~(1u << r);
      const bool hit00 = valid && lr == 0 && 0.0 >= p.threshold && zero_ok(r);
      emit_hits_wave(hits, p.cap, count, hit00, 0.0, lorig[ib + r], jorig);
    }
    rows_ok = __any(okbits != 0u) ? rows_ok : 0u;

    // ---- H: the smallest step-1 LCS that keeps the pair (row r, lane) alive; 0xffff = cannot hit
    auto need_of = [&](int r) -> int {
#ifdef NSM_X_NONEED  // (timing experiments)
      return ((okbits >> r) & 1u) ? 0 : 0xffff;
#endif
      const uint32_t* rec = head + r * 3 * kTileHead;
      const int ll = wave_first(static_cast<int>(rec[NB + 2]));
      const int S = max(ll, lr);
      const float4 Wt = wtab[min(S, 65)];
      uint32_t l1[3];
      int la[3], lb_t[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        la[t] = static_cast<int>(rec[t * kTileHead + NB]);
        lb_t[t] = s_rlen[t * kWave + lane];
        l1[t] = 0u;
        if (use_hist && (t > 0 || p.use_h1)) {  // (t == 0: the histogram bound of the step-1 pair itself, >= |la - lb|)
          uint32_t hl[NB], hb[NB];
#pragma unroll
          for (int q = 0; q < NB; ++q) {
            hl[q] = rec[t * kTileHead + q];
            hb[q] = s_rhist[(t * NB + q) * kWave + lane];
          }
          l1[t] = hist_l1<NB>(hl, hb);
        } else if (t == 0) {
          l1[0] = static_cast<uint32_t>(abs(la[0] - lb_t[0]));  // LCS <= min(la, lb)
        }
      }
      // alive after step 1  <=>  lcs / n1 + R >= thr,  R = (w2 + w3 + c) - w2 l2 / n2 - w3 l3 / n3;
      // 2e-3 of an LCS unit covers the float rounding (la + lb <= 1024)
      const float i2 = __builtin_amdgcn_rcpf(static_cast<float>(max(la[1] + lb_t[1], 1)));
      const float i3 = __builtin_amdgcn_rcpf(static_cast<float>(max(la[2] + lb_t[2], 1)));
      const float x = Wt.x * static_cast<float>(l1[1]) * i2 + Wt.y * static_cast<float>(l1[2]) * i3;
      const int n1 = la[0] + lb_t[0];
      const float needf = static_cast<float>(n1) * (Wt.z + x) - 2e-3f;
      const int nd = max(0, static_cast<int>(__builtin_ceilf(needf)));
      const int m1 = (n1 - static_cast<int>(l1[0])) >> 1;  // LCS of step 1 <= m1 (<= min(la, lb))
      const bool can = ((okbits >> r) & 1u) && nd <= m1;
      return can ? nd : 0xffff;
    };

    const uint32_t* text1 = img + lane;  // image 0 = step 1
    const int lb1 = s_rlen[lane];
    for (uint32_t rows = rows_ok; rows;) {
      // the next two rows with a live lane (rB < 0: only one is left; rA < 0: none)
      int rA = -1, rB = -1, ndA = 0xffff, ndB = 0xffff;
      while (rows && rB < 0) {
        const int rc = __builtin_ctz(rows);
        rows &= rows - 1;
        const int nd = need_of(rc);
        if (!__any(nd != 0xffff)) continue;
        if (rA < 0) {
          rA = rc;
          ndA = nd;
        } else {
          rB = rc;
          ndB = nd;
        }
      }
      if (rA < 0) break;
      int lrowA, laA, lrowB = 0, laB = 0;
      left_level(rA, 1, lrowA, laA);
      if (rB >= 0) left_level(rB, 1, lrowB, laB);
      int lcsA = 0, lcsB = 0;
      build_table(pm, lstr + rA * kRow, laA);
      if (rB >= 0 && max(laA, laB) <= 2 * kWave) {
        build_table(pm + tbl_stride, lstr + rB * kRow, laB);
        // (the loops run to the longest text of a lane that can still hit)
        const int nch1 = wave_max_i32((ndA != 0xffff || ndB != 0xffff) ? lb1 : 0);
        NSM_STAT(1, 1);
        NSM_STAT(2, (nch1 + 3) / 4);
        tile_lcs2_any<K, true>(pm, tbl_stride, text1, nch1, max(laA, laB), lcsA, lcsB, ndA, ndB, lb1);
      } else {
        lcsA = tile_lcs1_any<K>(pm, text1, kWave, wave_max_i32(ndA != 0xffff ? lb1 : 0), laA);
        NSM_STAT(3, rB >= 0 ? 2 : 1);
        NSM_STAT(4, (rB >= 0 ? 2 : 1) * ((wave_max_i32(valid ? lb1 : 0) + 3) / 4));
        if (rB >= 0) {
          build_table(pm, lstr + rB * kRow, laB);
          lcsB = tile_lcs1_any<K>(pm, text1, kWave, wave_max_i32(ndB != 0xffff ? lb1 : 0), laB);
        }
      }
      // step 1 survives on the integer form of the bound
      double scA = 0.0, scB = 0.0;
      bool alA = false, alB = false;
#ifdef NSM_X_NOAFTER  // (timing experiments: step 1's LCS only)
      bool wideA = false, wideB = false;
      if (lcsA + lcsB == 0x7fffffff) scA = 1.0;
#else
      bool wideA = after_step(rA, 1, laA, lb1, lcsA, ndA != 0xffff && lcsA >= ndA, scA, alA);
      bool wideB = rB >= 0 ? after_step(rB, 1, laB, lb1, lcsB, ndB != 0xffff && lcsB >= ndB, scB, alB) : false;
#endif
#ifdef NSM_X_NOWIDE  // (timing experiments: nothing after step 1)
      wideA = wideB = false;
      pk_cnt = 0;
#endif
      // the steps that follow, while a row's survivors are too many to park: wave-wide, text = the resident image of
      // the step (for items deeper than the images: their row in global memory)
      for (int s = 2; wideA || wideB; ++s) {
        const int ti = image_of(s);
        const bool res = resident(s, lr);
        const bool all_res = !__any(((wideA && alA) || (wideB && alB)) && !res);
        const uint32_t* text = img + static_cast<size_t>(ti) * kDw * kWave + lane;
        int lb = s_rlen[ti * kWave + lane];
        if (!all_res && !res) {
          const int rrow = rrow0 + max(0, min(s, lr - 1));
          lb = rlen[rrow];
          text = reinterpret_cast<const uint32_t*>(rcodes + static_cast<size_t>(rrow) * kRow);
        }
        const int nchA = wave_max_i32((wideA && alA) ? lb : 0), nchB = wave_max_i32((wideB && alB) ? lb : 0);
        const int nch = max(nchA, nchB);
        int la2A = 0, la2B = 0, lrA = 0, lrB = 0;
        if (wideA) left_level(rA, s, lrA, la2A);
        if (wideB) left_level(rB, s, lrB, la2B);
        if (s > 2) {  // the level strings of steps >= 3 are staged on demand (over the rows' step-1 copies)
          constexpr int kG = (kDw + kWave - 1) / kWave;
          uint32_t v[2][kG];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int lrow = q == 0 ? lrA : lrB;  // (a row that is not running re-reads row 0 of the table: unused)
#pragma unroll
            for (int g = 0; g < kG; ++g) {
              const int d = min(g * kWave + lane, kDw - 1);
              v[q][g] = reinterpret_cast<const uint32_t*>(lcodes + static_cast<size_t>(lrow) * kRow)[d];
            }
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const bool on = q == 0 ? wideA : wideB;
            const int r = q == 0 ? rA : max(rB, 0);
#pragma unroll
            for (int g = 0; g < kG; ++g) {
              const int d = g * kWave + lane;
              if (on && d < kDw) reinterpret_cast<uint32_t*>(lstr + r * kRow)[d] = v[q][g];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const uint8_t* cA = lstr + ((s == 2 ? kBatch : 0) + rA) * kRow;
        const uint8_t* cB = lstr + ((s == 2 ? kBatch : 0) + max(rB, 0)) * kRow;
        int l2A = 0, l2B = 0;
        if (!all_res) {  // (rare) one generic pass per row, every limb live, per-lane text pointer and stride
          if (wideA) {
            build_table(pm, cA, la2A);
            l2A = tile_lcs1<K, 2 * K>(pm, text, res ? kWave : 1, nchA);
          }
          if (wideB) {
            build_table(pm, cB, la2B);
            l2B = tile_lcs1<K, 2 * K>(pm, text, res ? kWave : 1, nchB);
          }
        } else if (wideA && wideB && max(la2A, la2B) <= 2 * kWave) {
          build_table(pm, cA, la2A);
          build_table(pm + tbl_stride, cB, la2B);
          tile_lcs2_any<K, false>(pm, tbl_stride, text, nch, max(la2A, la2B), l2A, l2B);
          NSM_STAT(5, 1);
          NSM_STAT(6, (nch + 3) / 4);
        } else {
          NSM_STAT(7, (wideA ? 1 : 0) + (wideB ? 1 : 0));
          NSM_STAT(8, (wideA ? (nchA + 3) / 4 : 0) + (wideB ? (nchB + 3) / 4 : 0));
          if (wideA) {
            build_table(pm, cA, la2A);
            l2A = tile_lcs1_any<K>(pm, text, kWave, nchA, la2A);
          }
          if (wideB) {
            build_table(pm, cB, la2B);
            l2B = tile_lcs1_any<K>(pm, text, kWave, nchB, la2B);
          }
        }
        if (wideA) wideA = after_step(rA, s, la2A, lb, l2A, alA, scA, alA);
        if (wideB) wideB = after_step(rB, s, la2B, lb, l2B, alB, scB, alB);
      }
    }
    if (pk_cnt > 0) drain();
  };

  // ---- the left rows: the block's waves take batches from a block-shared counter (a static split leaves the CU half
  // empty while the block's slowest wave finishes: the LDS is only released when the whole block is done)
  const unsigned long long cats_tile = partitioned ? wave_or_u64(valid ? (1ull << myseg) : 0ull) : 1ull;
  int static_round = 0;
  for (;;) {
    int n = 0;
#ifdef NSM_TILE_STATIC  // (A/B builds: batches dealt round-robin instead of taken from the block's counter)
    n = wave + static_round * waves;
    ++static_round;
#else
    if (lane == 0) n = atomicAdd(s_next, 1);
    n = wave_first(n);
#endif
    // batch n of the block: the categories of the tile in order, each with its slice of left rows
    int c = -1, ib = 0, b = 0;
    for (unsigned long long cats = cats_tile; cats;) {
      const int cc = __builtin_ctzll(cats);
      cats &= cats - 1;
      int lo_c, hi_c;
      if (partitioned) {  // slice `yslice` of the category's rows, cut at batch boundaries
        const int lo = lsegstart[cc], len = lsegstart[cc + 1] - lo;
        const int per = (((len + p.y_slices - 1) / p.y_slices) + kBatch - 1) / kBatch * kBatch;
        lo_c = lo + min(len, yslice * per);
        hi_c = lo + min(len, (yslice + 1) * per);
      } else {
        lo_c = min(p.n_left, yslice * p.rows_per_slice);
        hi_c = min(p.n_left, lo_c + p.rows_per_slice);
      }
      const int nb = (hi_c - lo_c + kBatch - 1) / kBatch;
      if (n < nb) {
        c = cc;
        ib = lo_c + n * kBatch;
        b = hi_c;
        break;
      }
      n -= nb;
    }
    if (c < 0) break;  // every wave of the block gets there: the counter only grows
    const unsigned long long lower = (1ull << c) - 1ull;
    const int nrows = min(kBatch, b - ib);
    uint32_t okbits = 0, rows_ok = 0;
    for (int r = 0; r < nrows; ++r) {
      const uint64_t cl = (p.cat_mode != NSM_CAT_NONE) ? lcat[ib + r] : 0ull;
      bool ok = valid;
      if (partitioned) ok = ok && myseg == c && ((cl & catr & lower) == 0ull);
      else if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(cl, catr, p.cat_mode);
      okbits |= ok ? (1u << r) : 0u;
      rows_ok |= __any(ok) ? (1u << r) : 0u;
    }
    if (rows_ok) scan_batch(ib, nrows, okbits, rows_ok);
  }
}

}  // namespace nsm
