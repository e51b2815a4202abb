// Levels-mode Jaccard grid: the hot loop of gen_comparable with score_func = intersection_vs_union
// (reference: napkon_string_matching/types/comparable_data.py:223-232 calling compare_terms :248-265
// and intersection_vs_union, compare/score_functions.py:6-13; category predicate :464-490).
//
//   score(i, j) = sum_{s=1..max(Ll,Lr)} 2^-s * |A_s n B_s| / |A_s u B_s|,
//   A_s = level min(s, Ll-1) of left item i,  B_s = level min(s, Lr-1) of right item j,
//   accumulated in double in that order (the weights are exact powers of two).
//
// Levels are suffix-nested (gen_comp_value, :283-285), so an item is ONE row of unique ids in
// first-appearance order and level l is its first plen[l] ids.
//
// Structure: FILTER all pairs cheaply, VERIFY the survivors exactly -- per lane, not per wave.
//   * lane = right item (ids in registers); the left rows of the chunk stream by, wave-uniform;
//   * filter (per row, ~19 VALU ops): category predicate, and the necessary condition
//         score <= ( |A_1 n B_1| + min(|A n B|, m) ) / (2 m),   m = max(|A_1|, |B_1|) >= own |B_1|
//     (step 1 weighs 1/2 and compares the step-1 sets; every later step contains them, so its Jaccard
//     is at most min(1, |A n B| / m), and the later weights sum to less than 1/2), with
//         |X n Y| <= popcount(hashbits(X) & hashbits(Y)) + collisions(X)
//     for the whole rows and for the step-1 sets (signature words as in the RAW kernel).  A lane
//     that passes appends the row to ITS OWN candidate queue in LDS;
//   * verify: when some queue fills up (and at the end) the wave walks the queue slots; in slot k every
//     lane gathers ITS k-th candidate row from HBM/L2 and scores the pair exactly:
//       1. position matrix with the xor/min3 trick -- right ids are kept as (id << 6) | position, the
//          left id as id << 6, so min_b(la ^ rb) < 64 iff `a` occurs in the right row and then IS its
//          position;
//       2. per step s, |A_s n B_s| = #{a < plenL[s] : pos[a] < plenR[s]} by a byte-parallel compare;
//       3. the double quotient comes from an LDS table filled at block start with real IEEE divisions
//          (W <= 32; W = 64 divides in place) and is accumulated with the reference's weights.
//     A wave-wide "does ANY lane pass" test would almost always say yes for filters of this strength;
//     lane-private queues turn a filter with pass rate p into ~p of the exact work.
#pragma once
#include "nsm_common.hpp"

namespace nsm {

constexpr int kQueueSlots = 32;  // candidate rows per lane between two verify sweeps

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

template <int W>
constexpr bool kQuotTable = (W <= 32);

__device__ __forceinline__ uint32_t lev_min3u(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int W, int NB>
__device__ __forceinline__ void levels_wave(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const uint64_t* __restrict__ lsig,
    const int32_t* __restrict__ lorig, const int32_t* __restrict__ lnlev, const uint8_t* __restrict__ lplen,
    const uint64_t* __restrict__ lcat, const uint8_t* __restrict__ rplen_row, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const JacLevScalars<W>& p, const uint32_t (&r)[W], uint64_t sr,
    uint64_t sr1, uint64_t catr, int lr, int pr1, int jorig, bool valid, int i0, int i1, uint16_t* queue,
    const double* quot, const uint32_t* __restrict__ lfilt, const int32_t* __restrict__ lsegstart, bool partitioned,
    int myseg, int lane) {
  int qn = 0;  // candidates queued by this lane
  // least |A_1 n B_1| + min(|A n B|, |B_1|) that can reach the threshold; the factor (1 - 1e-9) keeps
  // the test necessary under the rounding of the double accumulation
  const int bneed_r = static_cast<int>(ceil(2.0 * p.threshold * static_cast<double>(pr1) * (1.0 - 1e-9)));

  // ---- exact score of (left row idx, this lane's right item); idx < 0: the lane idles
  auto verify = [&](int idx) {
    const bool active = idx >= 0;
    const int ii = active ? idx : i0;
    uint32_t l[W];
    const uint4* lp = reinterpret_cast<const uint4*>(lids + static_cast<size_t>(ii) * W);
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const uint4 v = lp[q];
      l[4 * q + 0] = v.x << 6;
      l[4 * q + 1] = v.y << 6;
      l[4 * q + 2] = v.z << 6;
      l[4 * q + 3] = v.w << 6;
    }
    const int nl = lcnt[ii];
    const int nl_max = wave_max_i32(active ? nl : 0);
    uint32_t posw[W / 4];
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      posw[q] = 0xffffffffu;
      if (4 * q < nl_max) {  // wave-uniform
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t la = l[4 * q + e];
          uint32_t m = lev_min3u(la ^ r[0], la ^ r[1], 255u);
#pragma unroll
          for (int b = 2; b < NB; b += 2) m = lev_min3u(m, la ^ r[b], la ^ r[b + 1]);
          word |= m << (8 * e);  // m <= 255: the position of the match, or >= 64
        }
        posw[q] = word;
      }
    }
    double score = 0.0;
    if (active) {
      const int ll = lnlev[ii];
      const uint8_t* __restrict__ lpl = lplen + static_cast<size_t>(ii) * p.lev_stride_l;
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
            // per byte: x < pr  (pr <= 64; bytes >= 128 never count)
            const uint32_t y = (x | 0x80808080u) - prrep;
            inter += __popc(~(y | x) & 0x80808080u);
          }
        }
        const int uni = pl + pr - inter;
        double part;
        if constexpr (kQuotTable<W>) part = quot[inter * (2 * W + 1) + uni];
        else part = uni ? static_cast<double>(inter) / static_cast<double>(uni) : 0.0;
        factor *= 0.5;
        score += part * factor;
      }
    }
    const bool hit = active && score >= p.threshold;
    if (__any(hit)) {
      if (hit) emit_hit(hits, p.cap, count, score, lorig[ii], jorig);
    }
  };

  auto flush = [&]() {
    const int deepest = wave_max_i32(qn);
    for (int k = 0; k < deepest; ++k) {
      const int idx = (k < qn) ? i0 + static_cast<int>(queue[k * kWave + lane]) : -1;
      verify(idx);
    }
    qn = 0;
  };

  // ---- filter: 4 left rows per iteration.  A row's filter record is 8 dwords (signature word,
  // category mask, sizes) so that 4 rows arrive with two s_load_dwordx16; the verdict is pushed
  // into the lane's queue without branches: every lane stores the row offset at its current queue
  // tail, only passing lanes advance the tail (v_addc).  Everything wave-uniform here costs scalar-
  // unit cycles, which the CU's 4 SIMDs share -- hence one pointer increment per batch, no per-row
  // address arithmetic, no exec-mask juggling.
  constexpr int BATCH = 4;
  const bool use_cat = p.cat_mode != NSM_CAT_NONE;
  const bool both_empty = p.cat_mode == NSM_CAT_INTERSECT_OR_BOTH_EMPTY;
  const int need_r = valid ? (p.emit_all ? -1 : bneed_r) : 1 << 20;  // invalid lanes never pass
  const uint32_t catr_lo = static_cast<uint32_t>(catr), catr_hi = static_cast<uint32_t>(catr >> 32);
  const uint32_t sr_lo = static_cast<uint32_t>(sr), sr_hi = static_cast<uint32_t>(sr >> 32);
  const uint32_t sr1_lo = static_cast<uint32_t>(sr1), sr1_hi = static_cast<uint32_t>(sr1 >> 32);
  uint32_t rowv = 0;                                  // row offset inside the chunk, in a VGPR
  const uint32_t qbase = static_cast<uint32_t>(lane * 2);  // byte offset of the lane's queue column

  // category partition: while the rows of category c stream by, only lanes standing for c take part
  // and a pair is dropped when the two items also share a LOWER category (it is reported there)
  bool seg_ok = true;
  uint32_t lower_lo = 0, lower_hi = 0;  // catr restricted to the categories below c
  auto filter_row = [&](uint32_t sig_lo, uint32_t sig_hi, uint32_t cat_lo, uint32_t cat_hi, uint32_t sig1_lo,
                        uint32_t sig1_hi) {
    int bound, bound1;
    // (the right words come with their top 6 bits set: the AND keeps the row's unary collision count)
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(bound) : "v"(sig_lo & sr_lo));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(bound) : "v"(sig_hi & sr_hi), "v"(bound));
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(bound1) : "v"(sig1_lo & sr1_lo));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(bound1) : "v"(sig1_hi & sr1_hi), "v"(bound1));
    bound = min(bound, pr1) + bound1;
    bool pass = bound >= need_r;
    if (partitioned) {
      pass = pass && seg_ok && (((cat_lo & lower_lo) | (cat_hi & lower_hi)) == 0u);
    } else if (use_cat) {
      const uint32_t common = (cat_lo & catr_lo) | (cat_hi & catr_hi);
      bool cat_ok = common != 0u;
      if (both_empty) cat_ok = cat_ok || ((cat_lo | cat_hi | catr_lo | catr_hi) == 0u);
      pass = pass && cat_ok;
    }
    // branch-free append
    queue[(static_cast<uint32_t>(qn) * (kWave * 2) + qbase) >> 1] = static_cast<uint16_t>(rowv);
    qn += pass ? 1 : 0;
    rowv += 1;
  };

  auto filter_range = [&](int a, int b) {  // rows [a, b) of the chunk
    rowv = static_cast<uint32_t>(a - i0);
    const uint32_t* __restrict__ fp = lfilt + static_cast<size_t>(a) * 8;
    int i = a;
    for (; i + BATCH <= b; i += BATCH, fp += 8 * BATCH) {
      uint32_t f[8 * BATCH];
#pragma unroll
      for (int q = 0; q < 8 * BATCH; ++q) f[q] = fp[q];
#pragma unroll
      for (int q = 0; q < BATCH; ++q) filter_row(f[8 * q + 0], f[8 * q + 1], f[8 * q + 2], f[8 * q + 3], f[8 * q + 5], f[8 * q + 6]);
      if (__any(qn > kQueueSlots - BATCH - 1)) flush();
    }
    for (; i < b; ++i, fp += 8) {
      filter_row(fp[0], fp[1], fp[2], fp[3], fp[5], fp[6]);
      if (__any(qn > kQueueSlots - BATCH - 1)) flush();
    }
  };

  if (partitioned) {
    unsigned long long cats = wave_or_u64(valid ? (1ull << myseg) : 0ull);
    while (cats) {
      const int c = __builtin_ctzll(cats);
      cats &= cats - 1;
      const unsigned long long lower = catr & ((1ull << c) - 1ull);
      lower_lo = static_cast<uint32_t>(lower);
      lower_hi = static_cast<uint32_t>(lower >> 32);
      seg_ok = myseg == c;
      const int a = max(i0, lsegstart[c]);
      const int b = min(i1, lsegstart[c + 1]);
      if (a < b) filter_range(a, b);
    }
  } else {
    filter_range(i0, i1);
  }
  flush();
}

template <int W>
__global__ __launch_bounds__(kBlock) void jaccard_levels_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const uint64_t* __restrict__ lsig,
    const int32_t* __restrict__ lorig, const int32_t* __restrict__ lnlev, const uint8_t* __restrict__ lplen,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ rids, const int32_t* __restrict__ rcnt,
    const uint64_t* __restrict__ rsig, const int32_t* __restrict__ rorig, const int32_t* __restrict__ rnlev,
    const uint8_t* __restrict__ rplen, const uint64_t* __restrict__ rcat, const uint32_t* __restrict__ lfilt,
    const uint32_t* __restrict__ rfilt, const int32_t* __restrict__ lsegstart, const int32_t* __restrict__ rseg, nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const JacLevScalars<W> p) {
  __shared__ uint16_t s_queue[kWavesPerBlock][kQueueSlots * kWave];
  __shared__ double s_quot[kQuotTable<W> ? (W + 1) * (2 * W + 1) : 1];
  if constexpr (kQuotTable<W>) {
    // k / u by real IEEE double division, exactly what the reference's Python `/` computes; 0/0
    // (an empty-vs-empty level the host has already cleared or blacklisted) is defined as 0
    for (int t = threadIdx.x; t < (W + 1) * (2 * W + 1); t += kBlock) {
      const int k = t / (2 * W + 1), u = t % (2 * W + 1);
      s_quot[t] = u ? static_cast<double>(k) / static_cast<double>(u) : 0.0;
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * kWavesPerBlock + wave;
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
  constexpr uint64_t kCollBits = ~((1ull << 58) - 1);  // top 6 bits: the OTHER side's unary collision count
  const uint64_t sr = valid ? (rsig[jc] | kCollBits) : 0ull;
  const uint32_t* rf = rfilt + static_cast<size_t>(jc) * 8;
  const uint64_t sr1 = valid ? (((static_cast<uint64_t>(rf[6]) << 32) | rf[5]) | kCollBits) : 0ull;
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  const int lr = rnlev[jc];
  const int jorig = rorig[jc];
  const uint8_t* rplen_row = rplen + static_cast<size_t>(jc) * p.lev_stride_r;
  const int pr1 = rplen_row[1];
  const int nbmax = wave_max_i32(nrj);  // (with a category partition lane 0 is not the largest)

  constexpr int NBS = W / 8;
  const int cls = (nbmax + NBS - 1) / NBS;
#define NSM_LEV_CASE(K)                                                                               \
  levels_wave<W, (K) * NBS>(lids, lcnt, lsig, lorig, lnlev, lplen, lcat, rplen_row, hits, count, p, r, sr, \
                            sr1, catr, lr, pr1, jorig, valid, i0, i1, s_queue[wave], s_quot, lfilt, lsegstart, partitioned, myseg, lane)
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
  // big chunks: the per-lane queues fill more evenly and the final flush is amortised; 4096 rows
  // keeps a row offset inside the queue's 16 bits
  const long long want_waves = 4ll * 256 * 32;
  long long chunks = (want_waves + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
  if (chunks < 1) chunks = 1;
  long long rows = (n_left + chunks - 1) / chunks;
  if (rows < 512) rows = 512;
  if (rows > 4096) rows = 4096;
  return static_cast<int>(rows);
}

template <int W>
int launch_levels(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode,
                  uint32_t flags, nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count,
                  hipStream_t stream) {
  (void)flags;  // the filter is exact and always on (emit_all turns its score bound off)
  JacLevScalars<W> p;
  p.n_left = l->n; p.n_right = r->n; p.cap = capacity;
  p.lev_stride_l = l->max_levels; p.lev_stride_r = r->max_levels;
  p.cat_mode = category_mode;
  p.threshold = threshold;
  p.emit_all = !(threshold > 0.0);  // also true for NaN: then nothing compares >= and nothing is emitted
  const int n_tiles = (r->n + kWave - 1) / kWave;
  p.rows_per_chunk = lev_rows_per_chunk(l->n, n_tiles);
  // with a category partition a tile only works on the chunks that overlap its categories' row
  // ranges: small chunks, or a handful of long-running waves hold the whole launch
  if (l->seg && p.rows_per_chunk > 512) p.rows_per_chunk = 512;
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    set_error("nsm_jaccard_levels_grid: more than 65535 * 4096 left rows");
    return NSM_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL((jaccard_levels_kernel<W>), grid, dim3(kBlock), 0, stream, l->ids, l->cnt, l->sig, l->orig,
                     l->nlev, l->plen, l->cat, r->ids, r->cnt, r->sig, r->orig, r->nlev, r->plen, r->cat, l->filt,
                     r->filt, l->seg_start, r->seg, hits, hit_count, p);
  return hip_status(hipGetLastError(), "jaccard_levels_kernel launch");
}

}  // namespace nsm
