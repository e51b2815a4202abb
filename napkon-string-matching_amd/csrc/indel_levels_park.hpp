// Levels-mode Indel-ratio grid, "scan + park + dense finish" kernel (the default path of
// nsm_indel_levels_grid; included by indel_levels.hip, which owns the ratio table and the helpers).
//
// Reference: types/comparable_data.py:223-232 -> compare_terms (:248-265) x fuzzy_match
// (compare/score_functions.py:20-27):  score = sum_{s=1..S} 2^-s * ratio(A[min(s,La-1)], B[min(s,Lb-1)]).
//
// Why this shape.  Step 1 has to be scored for every pair that passes the category predicate (no filter
// bounds an Indel ratio of 0.5), but after it only a small part of the pairs can still reach the
// threshold: the steps still to come are bounded by the symbol histograms of their level strings
// (LCS <= sum of bucket minima), which is weak for ONE ratio and strong once step 1 is known
// (configs[4], threshold 0.7: 49 % of the pairs survive step 1 on weights alone, 0.9 % with the bound).
// A wavefront that keeps scanning for those few lanes wastes the other ~63, so:
//
//   H     per (left row of the batch, lane = right item): R = upper bound of the steps >= 2 from the
//         histograms of their level strings -> the smallest step-1 LCS that keeps the pair alive, an
//         integer `need` (0xffff = the pair cannot hit or fails the category predicate); rows without a
//         live lane are never scored;
//   scan  step 1 wave-wide (lane = right item, left level wave-uniform, bit-parallel LCS); lanes with
//         lcs >= need survive.  Few survivors (<= park_max lanes): they are PARKED in block-shared LDS
//         as (score so far, right item row, batch row, next step).  Many survivors: the row goes on
//         wave-wide, step by step, with the same test (and the chance to park) after every step;
//   dense after a barrier the block's waves share the parked pairs of the batch, 64 per pass, lane =
//         one pair: match-mask tables of the batch's left rows side by side in the wave's LDS, the
//         lane's right level string gathered from L2, remaining steps in the reference's order with the
//         histogram bound after each; hits are emitted from here.
//
// Every test that drops a pair is an upper bound (exact: hits are identical to the wave-wide kernel's).
#pragma once

namespace nsm {

struct ParkParams {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t pm_stride;   // match-mask entries per table: alphabet + 1 rounded up to 8
  int32_t cat_mode;
  int32_t use_hist;    // both string tables carry histograms and NSM_FLAG_PRUNE is set
  int32_t fin_rows;    // mask tables per wave in the dense pass (1 .. batch)
  int32_t park_slots;  // capacity of the block's park
  int32_t park_max;    // park a row's survivors when at most this many of the 64 lanes are alive
  double threshold;
  unsigned long long cap;
};

constexpr int park_batch(int K) { return K >= 4 ? 4 : 8; }
constexpr uint16_t kDeadNeed = 0xffff;

// LCS <= (la + lb - L1) / 2 with L1 the distance of the bucketed symbol histograms, so
// ratio = 2 LCS / (la + lb) <= 1 - L1 / (la + lb); 0 when either string is empty (QRatio).  float with a
// relative error of ~1e-7: every user adds a margin.
__device__ __forceinline__ float hist_ratio_ub(uint32_t l1, int la, int lb) {
  const float n = static_cast<float>(la + lb);
  return (la == 0 || lb == 0) ? 0.0f : 1.0f - static_cast<float>(l1) * __builtin_amdgcn_rcpf(n);
}

template <int NB>
__device__ __forceinline__ uint32_t hist_l1(const uint32_t (&a)[NB], const uint32_t (&b)[NB]) {
  uint32_t acc = 0;
#pragma unroll
  for (int q = 0; q < NB; ++q) acc = __builtin_amdgcn_sad_u8(a[q], b[q], acc);
  return acc;
}

// 32-bucket histogram row -> NB dwords: NB = 8 as stored; NB = 4 folds bucket b + 16 onto bucket b (one-word
// strings: every count <= 64, the byte sums cannot carry) -- a coarser, still valid bound at half the v_sad_u8.
template <int NB>
__device__ __forceinline__ void load_hist(const uint8_t* __restrict__ hist, int row, uint32_t (&h)[NB]) {
  const uint4* hp = reinterpret_cast<const uint4*>(hist + static_cast<size_t>(row) * 32);
  const uint4 h0 = hp[0], h1 = hp[1];
  if constexpr (NB == 8) {
    h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w;
    h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;
  } else {
    h[0] = h0.x + h1.x; h[1] = h0.y + h1.y; h[2] = h0.z + h1.z; h[3] = h0.w + h1.w;
  }
}

// Upper bound of sum_{t > s} 2^-t * ratio_t for a pair with S steps, from the histogram bound `ub` of step
// s + 1: the steps after s + 1 repeat that level pair when both level indices are clamped (s + 1 >= S - 1),
// otherwise they are bounded by 1.
__device__ __forceinline__ float rest_bound(int s, int S, float ub) {
  if (s >= S) return 0.0f;
  const float wt = __builtin_ldexpf(1.0f, -(s + 1));
  const float tail = wt - __builtin_ldexpf(1.0f, -S);
  return wt * ub + tail * ((s + 1 >= S - 1) ? ub : 1.0f);
}

template <int K>
__global__ __launch_bounds__(kBlock) void indel_levels_park_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ lsegstart, const uint8_t* __restrict__ lcodes,
    const int32_t* __restrict__ llen, const uint8_t* __restrict__ lhist, const int32_t* __restrict__ rfirst,
    const int32_t* __restrict__ rnlev, const int32_t* __restrict__ rorig, const uint64_t* __restrict__ rcat,
    const int32_t* __restrict__ rseg, const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen,
    const uint8_t* __restrict__ rhist, nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count,
    const ParkParams p) {
  // LDS (dynamic, starts at offset 0 -- the one-word text images hold raw LDS addresses):
  //   per wave: [fin_rows][pm_stride * K] u64 mask tables (the scan uses table 0)
  //             (K > 1) [16 K][64] u32 text image | [batch][64] f64 running scores | [batch][64] u16 need
  //   per block: park score f64[P] | right item row i32[P] | batch row | next step << 8 i32[P] |
  //              cats u64 | count[2] valid[2] i32
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  constexpr int kRow = kWave * K;      // code units per string row
  constexpr int kBatch = park_batch(K);
  constexpr int NB = (K == 1) ? 4 : 8;  // histogram dwords per level string
  const int waves = blockDim.x >> 6;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  const int tbl_entries = p.pm_stride * K;
  const size_t wave_bytes = static_cast<size_t>(p.fin_rows) * tbl_entries * 8 + (K > 1 ? 16 * K * kWave * 4 : 0) +
                            kBatch * kWave * 8 + kBatch * kWave * 2;
  unsigned char* wbase = reinterpret_cast<unsigned char*>(s_mem) + wave * wave_bytes;
  unsigned long long* pm = reinterpret_cast<unsigned long long*>(wbase);
  uint32_t* wtext = reinterpret_cast<uint32_t*>(pm + static_cast<size_t>(p.fin_rows) * tbl_entries);
  double* sc = reinterpret_cast<double*>(wtext + (K > 1 ? 16 * K * kWave : 0));
  uint16_t* need = reinterpret_cast<uint16_t*>(sc + kBatch * kWave);
  unsigned char* bbase = reinterpret_cast<unsigned char*>(s_mem) + waves * wave_bytes;
  double* park_score = reinterpret_cast<double*>(bbase);
  int32_t* park_j = reinterpret_cast<int32_t*>(park_score + p.park_slots);
  int32_t* park_meta = park_j + p.park_slots;
  unsigned long long& s_cats = *reinterpret_cast<unsigned long long*>(park_meta + p.park_slots);
  int* s_cnt = reinterpret_cast<int*>(&s_cats + 1);  // [2], by batch parity
  int* s_valid = s_cnt + 2;                           // [2]: slots [0, valid) are written
  const uint32_t pm_base = static_cast<uint32_t>(wave * wave_bytes);

  const int tile = blockIdx.x * waves + wave;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;  // a whole wave may be beyond the table: it still takes part in the barriers
  const int jc = valid ? j : p.n_right - 1;
  const bool partitioned = rseg != nullptr;
  const int myseg = partitioned ? rseg[jc] : 0;
  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);
  const bool use_hist = p.use_hist != 0;
  const float thr_f = static_cast<float>(p.threshold);

  if (threadIdx.x == 0) s_cats = 0ull;
  if (threadIdx.x < 2) {
    s_cnt[threadIdx.x] = 0;
    s_valid[threadIdx.x] = p.park_slots;
  }
  __syncthreads();
  if (partitioned) {
    const unsigned long long mine = wave_or_u64(valid ? (1ull << myseg) : 0ull);
    if (lane == 0 && mine) atomicOr(&s_cats, mine);
  } else if (threadIdx.x == 0) {
    s_cats = 1ull;
  }
  __syncthreads();
  const unsigned long long cats_block = s_cats;  // block-uniform from here on
  if (partitioned) {  // most (tiles, chunk) combinations hold no row of the tiles' categories: leave early
    bool work = false;
    for (unsigned long long cats = cats_block; cats;) {
      const int c = __builtin_ctzll(cats);
      cats &= cats - 1;
      work = work || (max(i0, lsegstart[c]) < min(i1, lsegstart[c + 1]));
    }
    if (!work) return;  // the whole block
  }

  // ---- the lane's right item
  const int lr = rnlev[jc];
  const int rrow0 = rfirst[jc];
  const int jorig = rorig[jc];
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  const int lr_max = wave_max_i32(valid ? lr : 0);
  // level strings of steps 1..3: lengths and histograms stay in registers for the H phase
  int lb_t[3];
  uint32_t hb[3][NB];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int row = rrow0 + max(0, min(t + 1, lr - 1));
    lb_t[t] = rlen[row];
    if (use_hist) load_hist<NB>(rhist, row, hb[t]);
    else
#pragma unroll
      for (int q = 0; q < NB; ++q) hb[t][q] = 0u;
  }

  uint32_t lowmask = 0xffffu, sh16 = 16u;  // kept in VGPRs: e32 ops with VGPR operands issue at full rate

  auto ratio_of = [](int la_, int lb_, int lcs_) -> double {
    if constexpr (K == 1) return (la_ == 0 || lb_ == 0) ? 0.0 : g_ratio64.v[(la_ + lb_) * 65 + lcs_];
    else return indel_score_dev(la_, lb_, lcs_);
  };

  // per-lane histogram bound of one step's level pair, both rows gathered (continuation and dense pass)
  auto step_ub = [&](int lrow, int rrow, int la_, int lb_) -> float {
    if (!use_hist) return (la_ == 0 || lb_ == 0) ? 0.0f : 1.0f;
    uint32_t a[8], b[8];
    load_hist<8>(lhist, lrow, a);
    load_hist<8>(rhist, rrow, b);
    return hist_ratio_ub(hist_l1<8>(a, b), la_, lb_);
  };

  // reserve n park slots of parity pb for this wave; -1 when the park is full.  The counter is never rolled
  // back (a rollback races with the other waves' reservations): it stays inflated, every later reservation
  // fails too, and the written slots are exactly [0, first failing offset).
  auto reserve = [&](int pb, int n) -> int {
    int have = 0;
    if (lane == 0) have = atomicAdd(&s_cnt[pb], n);
    have = __builtin_amdgcn_readfirstlane(have);
    if (have + n > p.park_slots) {
      if (lane == 0) atomicMin(&s_valid[pb], have);
      return -1;
    }
    return have;
  };

  // ---- this wave's tile against the batch rows [ib, ib + nrows); okbits bit r = the lane passes the
  // category predicate for row ib + r
  auto scan_batch = [&](int ib, int nrows, uint32_t okbits, uint32_t rows_ok, int pb) __attribute__((always_inline)) {
    // ---- H: need[r][lane]
    uint32_t live = 0;
    for (uint32_t rows = rows_ok; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      const int i = ib + r;
      const int ll = lnlev[i];
      const int lf = lfirst[i];
      const int S = max(ll, lr);
      float ub[3];
      int la1 = 0, m1 = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int lrow = lf + max(0, min(t + 1, ll - 1));
        const int la = llen[lrow];
        uint32_t l1 = 0;
        if (use_hist) {
          uint32_t hl[NB];
          load_hist<NB>(lhist, lrow, hl);  // wave-uniform row: scalar loads
          l1 = hist_l1<NB>(hl, hb[t]);
          ub[t] = hist_ratio_ub(l1, la, lb_t[t]);
        } else {
          ub[t] = (la == 0 || lb_t[t] == 0) ? 0.0f : 1.0f;
          l1 = static_cast<uint32_t>(abs(la - lb_t[t]));
        }
        if (t == 0) {
          la1 = la;
          m1 = (la + lb_t[0] - static_cast<int>(l1)) >> 1;  // LCS of step 1 <= m1 (<= min(la, lb))
        }
      }
      const float R = (S >= 2 ? 0.25f * ub[1] : 0.0f) + rest_bound(2, S, ub[2]);
      const int n1 = la1 + lb_t[0];
      // alive after step 1  <=>  lcs / n1 + R >= thr; 2e-3 of an LCS unit covers the float rounding
      const float needf = (thr_f - R) * static_cast<float>(n1) - 2e-3f;
      const int nd = max(0, static_cast<int>(__builtin_ceilf(needf)));
      const bool ok = (okbits >> r) & 1u;
      const bool can = ok && nd <= m1;
      need[r * kWave + lane] = can ? static_cast<uint16_t>(nd) : kDeadNeed;
      live |= __any(can) ? (1u << r) : 0u;
    }
    if (!live) return;

    uint32_t taddr[K == 1 ? 32 : 1];
    int text_row = -1;
    int lb = 0;
    uint32_t cont = 0;  // rows that go on wave-wide after step 1 (running scores in sc, NaN = dropped)
    int ll_max = 0;
    for (uint32_t rows = live; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      ll_max = max(ll_max, lnlev[ib + r]);
    }
    const int steps_max = max(ll_max, lr_max);
    double factor = 1.0;
    uint32_t todo = live;
    for (int s = 1; s <= steps_max && todo; ++s) {
      factor *= 0.5;
      // right level of this step (per lane): rebuild the text image only when the row changes
      const int rrow = rrow0 + max(0, min(s, lr - 1));
      if constexpr (K == 1) {
        if (rrow != text_row) {
          text_row = rrow;
          const uint4* tp = reinterpret_cast<const uint4*>(rcodes + static_cast<size_t>(rrow) * 64);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 v = tp[q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t c0 = w[e] & 0xffu, c1 = (w[e] >> 8) & 0xffu, c2 = (w[e] >> 16) & 0xffu, c3 = w[e] >> 24;
              taddr[8 * q + 2 * e + 0] = (pm_base + 8 * c0) | ((pm_base + 8 * c1) << 16);
              taddr[8 * q + 2 * e + 1] = (pm_base + 8 * c2) | ((pm_base + 8 * c3) << 16);
            }
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
      const int nchars = wave_max_i32(valid ? lb : 0);
      for (uint32_t rows = todo; rows;) {
        const int r = __builtin_ctz(rows);
        rows &= rows - 1;
        const int i = ib + r;
        const int ll = lnlev[i];
        const int S = max(ll, lr);
        // lanes still in play for this row
        double score = 0.0;
        bool run;
        int nd = 0;
        if (s == 1) {
          nd = need[r * kWave + lane];
          run = nd != kDeadNeed;
        } else {
          score = sc[r * kWave + lane];
          run = (score == score) && s <= S;  // NaN = dropped
        }
        if (!__any(run)) {  // (s > 1) nothing left to score: finished lanes keep their scores in sc
          todo &= ~(1u << r);
          continue;
        }
        const int lrow = lfirst[i] + max(0, min(s, ll - 1));
        const int la = llen[lrow];
        wide_build_pm<K>(pm, p.pm_stride, lcodes + static_cast<size_t>(lrow) * kRow, la, lane);
        int lcs;
        if constexpr (K == 1) {
          const int npairs = (nchars + 1) >> 1;
          // opaque per row: otherwise the 64 unpacked addresses are hoisted out of the row loop into 64
          // more VGPRs
          asm volatile("" : "+v"(lowmask), "+v"(sh16));
          if (la <= 32) {  // wave-uniform: 32-bit words, and / add / xor / or all issue at full rate
            uint32_t v = ~0u;
#pragma unroll
            for (int w = 0; w < 32; ++w) {
              if (w < npairs) {
                const uint32_t m0 = lev_lds_load<uint32_t>(taddr[w] & lowmask);
                const uint32_t u0 = v & m0;
                v = (v + u0) | (v ^ u0);
                const uint32_t m1 = lev_lds_load<uint32_t>(taddr[w] >> sh16);
                const uint32_t u1 = v & m1;
                v = (v + u1) | (v ^ u1);
              }
            }
            lcs = 32 - __popc(v);
          } else {
            unsigned long long v = ~0ull;
#pragma unroll
            for (int w = 0; w < 32; ++w) {
              if (w < npairs) {
                const unsigned long long m0 = lev_lds_load<unsigned long long>(taddr[w] & lowmask);
                const unsigned long long u0 = v & m0;
                v = lev_add64(v, u0) | (v ^ u0);
                const unsigned long long m1 = lev_lds_load<unsigned long long>(taddr[w] >> sh16);
                const unsigned long long u1 = v & m1;
                v = lev_add64(v, u1) | (v ^ u1);
              }
            }
            lcs = 64 - __popcll(v);
          }
        } else {
          lcs = wide_lcs<K>(pm, wtext, nchars, lane, la);
        }
        bool alive;  // can still reach the threshold
        if (s == 1) {
          alive = run && lcs >= nd;
          if (!__any(alive)) {
            todo &= ~(1u << r);
            continue;
          }
          score = ratio_of(la, lb, lcs) * factor;
        } else {
          if (run) score += ratio_of(la, lb, lcs) * factor;
          // steps still to come: histogram bound of the next level pair (exact upper bound; 1e-6 covers
          // the float arithmetic of the bound and the rounding of the double sum)
          float rest = 0.0f;
          if (s < S) {
            const int t = s + 1;
            const int lrow_n = lfirst[i] + max(0, min(t, ll - 1));
            const int rrow_n = rrow0 + max(0, min(t, lr - 1));
            rest = rest_bound(s, S, step_ub(lrow_n, rrow_n, llen[lrow_n], rlen[rrow_n]));
          }
          alive = run && (score + static_cast<double>(rest) + 1e-6 >= p.threshold);
          if (run && !alive) score = __builtin_nan("");
        }
        // pairs that have seen their last step are final
        const bool more = alive && s < S;
        if (s == 1) {
          const bool hit = alive && !more && score >= p.threshold;
          if (__any(hit)) {
            if (hit) emit_hit(hits, p.cap, count, score, lorig[i], jorig);
          }
        }
        const unsigned long long who = __ballot(more);
        if (who == 0ull) {
          if (s > 1) sc[r * kWave + lane] = score;  // finished or dropped; hits of this row are emitted below
          todo &= ~(1u << r);
          continue;
        }
        const int n = __popcll(who);
        int have = -1;
        if (n <= p.park_max) have = reserve(pb, n);
        if (have >= 0) {
          if (more) {
            const int slot = have + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
            park_score[slot] = score;
            park_j[slot] = jc;
            park_meta[slot] = r | ((s + 1) << 8);
            score = __builtin_nan("");  // the dense pass owns the pair now
          }
          if (s > 1) sc[r * kWave + lane] = score;
          todo &= ~(1u << r);
          continue;
        }
        // dense enough (or the park is full): the row goes on wave-wide
        if (s == 1) {
          sc[r * kWave + lane] = more ? score : __builtin_nan("");
          cont |= 1u << r;
        } else {
          sc[r * kWave + lane] = score;
        }
      }
    }
    for (uint32_t rows = cont; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      const double score = sc[r * kWave + lane];
      const bool hit = score >= p.threshold;  // false for NaN (dropped or parked)
      if (__any(hit)) {
        if (hit) emit_hit(hits, p.cap, count, score, lorig[ib + r], jorig);
      }
    }
  };

  // ---- dense pass: parked pairs [base, base + 64) of the batch, lane = one pair
  auto finish_pass = [&](int ib, int nrows, int base, int n_p) __attribute__((always_inline)) {
    const bool active = base + lane < n_p;
    const int slot = active ? base + lane : base;
    const int meta = park_meta[slot];
    const int r = meta & 0xff, s0 = meta >> 8;
    const int jr = park_j[slot];
    double score = park_score[slot];
    const int i = ib + r;
    const int ll = lnlev[i], lf = lfirst[i];
    const int lrj = rnlev[jr], rr0 = rfirst[jr];
    const int S = max(ll, lrj);
    const int s_lo = -wave_max_i32(active ? 64 - s0 : 0) + 64;  // smallest next step (steps <= 64)
    const int s_hi = wave_max_i32(active ? S : 0);
    int prev_a = -1, prev_b = -1;
    double ratio = 0.0;
    bool alive = active;
    double factor = __builtin_ldexp(1.0, 1 - s_lo);  // the weight of step s_lo - 1
    for (int s = s_lo; s <= s_hi; ++s) {
      factor *= 0.5;
      const bool run = alive && s >= s0 && s <= S;
      if (!__any(run)) continue;
      const int a = max(0, min(s, ll - 1)), b = max(0, min(s, lrj - 1));
      const bool fresh = run && (a != prev_a || b != prev_b);
      if (__any(fresh)) {
        const int lrow = lf + a, rrow = rr0 + b;
        const int la = llen[lrow], lbj = rlen[rrow];
        int lcs = 0;
        for (int g0 = 0; g0 < nrows; g0 += p.fin_rows) {
          const bool mine = fresh && r >= g0 && r < g0 + p.fin_rows;
          uint32_t rows_here = wave_reduce_u32(mine ? (1u << r) : 0u, [](uint32_t x, uint32_t y) { return x | y; });
          if (!rows_here) continue;
          int la_max = 0;
          for (uint32_t rows = rows_here; rows;) {  // the group's mask tables, one per left row present
            const int rr = __builtin_ctz(rows);
            rows &= rows - 1;
            const int ii = ib + rr;
            const int lrow_u = lfirst[ii] + max(0, min(s, lnlev[ii] - 1));
            const int la_u = llen[lrow_u];
            la_max = max(la_max, la_u);
            wide_build_pm<K>(pm + static_cast<size_t>(rr - g0) * tbl_entries, p.pm_stride,
                             lcodes + static_cast<size_t>(lrow_u) * kRow, la_u, lane);
          }
          const unsigned long long* tbl = pm + static_cast<size_t>(mine ? r - g0 : 0) * tbl_entries;
          const int nchars = wave_max_i32(mine ? lbj : 0);
          const uint8_t* tptr = rcodes + static_cast<size_t>(mine ? rrow : rr0) * kRow;
          int got;
          if constexpr (K == 1) {
            uint32_t text[16];
            const uint4* tp = reinterpret_cast<const uint4*>(tptr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint4 v = tp[q];
              text[4 * q + 0] = v.x; text[4 * q + 1] = v.y; text[4 * q + 2] = v.z; text[4 * q + 3] = v.w;
            }
            const int nwords = (nchars + 3) >> 2;
            unsigned long long v = ~0ull;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
              if (w < nwords) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const unsigned c = (text[w] >> (8 * q)) & 0xffu;
                  const unsigned long long m = tbl[c];
                  const unsigned long long u = v & m;
                  v = lev_add64(v, u) | (v ^ u);
                }
              }
            }
            got = 64 - __popcll(v);
          } else {
            wide_store_text<K>(wtext, tptr, lane);
            got = wide_lcs<K>(tbl, wtext, nchars, lane, la_max);
          }
          if (mine) lcs = got;
        }
        if (fresh) {
          ratio = ratio_of(la, lbj, lcs);
          prev_a = a;
          prev_b = b;
        }
      }
      if (run) {
        score += ratio * factor;
        float rest = 0.0f;
        if (s < S) {
          const int t = s + 1;
          const int lrow_n = lf + max(0, min(t, ll - 1)), rrow_n = rr0 + max(0, min(t, lrj - 1));
          rest = rest_bound(s, S, step_ub(lrow_n, rrow_n, llen[lrow_n], rlen[rrow_n]));
        }
        alive = score + static_cast<double>(rest) + 1e-6 >= p.threshold;
      }
    }
    if (active && alive && score >= p.threshold) emit_hit(hits, p.cap, count, score, lorig[i], rorig[jr]);
  };

  int pb = 0;
  for (unsigned long long cats = cats_block; cats;) {  // the same sequence in every wave of the block
    const int c = __builtin_ctzll(cats);
    cats &= cats - 1;
    const int a = partitioned ? max(i0, lsegstart[c]) : i0;
    const int b = partitioned ? min(i1, lsegstart[c + 1]) : i1;
    const unsigned long long lower = (1ull << c) - 1ull;
    for (int ib = a; ib < b; ib += kBatch) {
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
      if (rows_ok) scan_batch(ib, nrows, okbits, rows_ok, pb);
      __syncthreads();  // every tile's survivors of this batch are parked
      const int n_p = min(s_cnt[pb], s_valid[pb]);
      if (threadIdx.x == 0) {  // the other parity's counters were consumed before the previous batch's last barrier
        s_cnt[pb ^ 1] = 0;
        s_valid[pb ^ 1] = p.park_slots;
      }
      for (int base = wave * kWave; base < n_p; base += waves * kWave) finish_pass(ib, nrows, base, n_p);
      __syncthreads();  // the park has been read
      pb ^= 1;
    }
  }
}

}  // namespace nsm
