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
//   stage per batch of left rows the wave copies what it needs of them into its LDS once: per row and step
//         1..3 the level string's length, row and histogram ("head"), and the step's level strings
//         themselves -- two memory latencies per batch instead of a dependent chain per row;
//   H     per (left row, lane = right item): R = upper bound of the steps >= 2 from the histograms ->
//         the smallest step-1 LCS that keeps the pair alive, an integer `need` (0xffff = the pair cannot
//         hit or fails the category predicate); rows without a live lane are never scored;
//   scan  step 1 wave-wide (lane = right item, left level wave-uniform, bit-parallel LCS); lanes with
//         lcs >= need survive.  Few survivors (<= park_max lanes): they are PARKED in block-shared LDS
//         as (score so far, right item row, batch row, next step).  Many survivors: the row goes on
//         wave-wide, step by step, with the same test (and the chance to park) after every step;
//   dense every kSub batches the block meets at a barrier; each wave takes the parked pairs of one
//         batch, 64 per pass, lane = one pair: match-mask tables of that batch's left rows side by side
//         in the wave's LDS, the lane's right level string gathered from L2, remaining steps in the
//         reference's order with the histogram bound after each; hits are emitted from here.
//
// Every test that drops a pair is an upper bound (exact: hits are identical to the wave-wide kernel's).
//
// SPLIT = true (round 3, one-word strings at thresholds where few pairs outlive step 1): the same stage / H / scan, but the
// survivors of step 1 go to a GLOBAL queue (collected per wave in LDS, one atomic per ~200 entries) and a second kernel
// finishes them lane-per-pair (indel_levels_finish.hpp).  No park, no dense pass, no block barrier, and without the dense
// pass's registers the scan fits 96 VGPRs: 5 waves per SIMD and no scratch traffic to speak of, where the fused kernel
// (4 waves, 256 B of scratch per lane) moved 234 GB per launch.  configs[4], three fuzzy grids: 409 -> 289 ms.
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
  int32_t park_slots;  // capacity of ONE park region (there are kSub)
  int32_t park_max;    // park a row's survivors when at most this many of the 64 lanes are alive
  int32_t xcd_slices;  // > 0 (partitioned tables): 1-D grid mapped XCD-aware, this many slices per category
  int32_t slice_base;  // first left slice (or row chunk) of this launch: the split path walks the slices in rounds
  int32_t slices_total;  // > 0: left slices per category over all rounds (the launch holds xcd_slices / gridDim.y of them)
  double threshold;
  unsigned long long cap;
  unsigned long long qcap;  // (SPLIT) capacity of the survivor queue, entries
};

// Survivor queue of the split path (SPLIT = true: scan kernel here, finish kernel in indel_levels_finish.hpp): one 64-bit
// entry per pair that is still alive after step 1 and has steps to go:  left row << 31 | right row << 7 | step-1 LCS.
// (SPLIT) FOUR left rows per pass: two mask tables at a fixed distance, so that one address register serves both
// (ds_read_b64 with an immediate offset) -- one address op and two reads per code unit for four rows, four independent chains
#ifndef NSM_SPLIT_QUAD
#define NSM_SPLIT_QUAD 0
#endif
#ifndef NSM_SPLIT_TABLE_BYTES
#define NSM_SPLIT_TABLE_BYTES 512
#endif
constexpr int kSplitTableBytes = NSM_SPLIT_TABLE_BYTES;  // 64 entries of 8 bytes (the split path requires pm_stride <= 64)
constexpr int kQueueBuf = 256;  // entries a wave collects in LDS before it reserves room in the global queue
constexpr int kQueueRowBits = 24;
__host__ __device__ constexpr unsigned long long queue_entry(int i, int j, int lcs) {
  return (static_cast<unsigned long long>(i) << 31) | (static_cast<unsigned long long>(j) << 7) | static_cast<unsigned long long>(lcs);
}

constexpr int park_batch(int K) { return K >= 4 ? 4 : 8; }
// scan batches between two block barriers, A/B-measured on C5-shaped cohorts (3 x 100k^2): 1 / 2 / 4 / 8 / 16 ->
// 41.9 / 24.0 / 22.0 / 20.8 / 23.2 ms (park regions of 64 slots; 128 slots: the same)
#ifndef NSM_KSUB
#define NSM_KSUB 8
#endif
constexpr int park_sub(int K) { return K == 1 ? NSM_KSUB : 4; }  // scan batches (= park regions) between two barriers
// (multi-word strings, Term-like 20k x 20k at 0.5: 4 -> 140 ms, 8 -> 158 ms)
constexpr uint16_t kDeadNeed = 0xffff;
#ifndef NSM_X_GROUPS
#define NSM_X_GROUPS 8
#endif
// 4 waves per SIMD (128 VGPRs): left alone hipcc settles at 169 VGPRs = 2 waves, and the kernel's latency-bound
// phases then run 1.4x slower (3 x 100k^2 levels bench: 38.4 vs 26.7 ms); the spills this forces sit in the dense pass.
// Multi-word strings are LDS-limited to <= 2 waves by their text images: no register cap there.
// step-1 pre-filter: 1 = histogram bound of the step-1 level pair (4-8 more v_sad_u8 per pair), 0 = lengths only
#ifndef NSM_PARK_H1
#define NSM_PARK_H1 0
#endif
#ifndef NSM_PARK_OCC
#ifndef NSM_SPLIT_WAVES
#define NSM_SPLIT_WAVES 5  // waves per SIMD of the split path's scan kernel: 4 (115 VGPRs, no scratch) / 5 (96, 72 B of scratch) / 6 -> configs[4] fuzzy grids 311 / 289 / 322 ms
#endif
#define NSM_PARK_OCC __attribute__((amdgpu_waves_per_eu(K == 1 ? (SPLIT ? NSM_SPLIT_WAVES : 4) : 1, K == 1 ? (SPLIT ? NSM_SPLIT_WAVES : 4) : 8)))
#endif
constexpr int kHeadDwords = 12;  // histogram (4 folded / 8 dwords) | la | row | levels | first row

// Work counters of the scan (variant builds, -DNSM_SCAN_STATS, read by tools/bench_levels.py --scan-stats through
// nsm_debug_scan_stats): 0 left rows a wave visited, 1 (row, lane) pairs that pass the category predicate, 2 of those
// alive after the H phase, 3 rows with a live lane (= rows whose step 1 is scored), 4 pairs alive after step 1,
// 5 two-row passes, 6 one-row passes
#ifdef NSM_SCAN_STATS
__device__ unsigned long long g_scan_stats[8];
#define NSM_SCAN_STAT(slot, v)                                                                     \
  do {                                                                                             \
    const unsigned long long stat_v = static_cast<unsigned long long>(v); /* (all lanes evaluate v) */ \
    if (lane == 0) atomicAdd(&g_scan_stats[slot], stat_v);                                         \
  } while (0)
#else
#define NSM_SCAN_STAT(slot, v) do {} while (0)
#endif

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

// LCS <= (la + lb - L1) / 2 with L1 the distance of the bucketed symbol histograms, so
// ratio = 2 LCS / (la + lb) <= 1 - L1 / (la + lb).  float with a relative error of ~1e-7: every user adds a
// margin.  (Both strings empty: the bound says 1 where the ratio is 0 -- weaker, still a bound.)
__device__ __forceinline__ float hist_ratio_ub(uint32_t l1, int la, int lb) {
  const float n = static_cast<float>(max(la + lb, 1));
  return 1.0f - static_cast<float>(l1) * __builtin_amdgcn_rcpf(n);
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


template <int K, bool SPLIT = false>
__global__ __launch_bounds__(kBlock) NSM_PARK_OCC void indel_levels_park_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ lsegstart, const uint8_t* __restrict__ lcodes,
    const int32_t* __restrict__ llen, const uint8_t* __restrict__ lhist, const int32_t* __restrict__ rfirst,
    const int32_t* __restrict__ rnlev, const int32_t* __restrict__ rorig, const uint64_t* __restrict__ rcat,
    const int32_t* __restrict__ rseg, const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen,
    const uint8_t* __restrict__ rhist, nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count,
    const ParkParams p, const int32_t* __restrict__ rsegstart, unsigned long long* __restrict__ queue,
    unsigned long long* __restrict__ qcount, int* __restrict__ qflag, const int* __restrict__ gate) {
  // `gate` (fused kernel only): the launch that follows the split path's rounds does nothing unless the queue overflowed
  if constexpr (!SPLIT) {
    if (gate != nullptr && *gate == 0) return;
  }
  // LDS (dynamic, starts at offset 0 -- the one-word text images hold raw LDS addresses):
  //   per wave: [fin_rows][pm_stride * kPmWords<K>] u64 mask tables (the scan uses table 0)
  //             (K > 1) [16 K][64] u32 text image | [batch][64] u16 need (then the step-1 LCS)
  //             | [batch][3][12] u32 heads | [batch][64 K] u8 the step's left level strings
  //   per block: kSub park regions of park_slots: score f64 | right item row i32 | batch row + next step << 8 i32
  //              | [66] float4 step weights by S | cats u64 | count[2][kSub] valid[2][kSub] i32
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  constexpr int kRow = kWave * K;      // code units per string row
  constexpr int kBatch = park_batch(K);
  constexpr int kSub = park_sub(K);
  constexpr int NB = (K == 1) ? 4 : 8;  // histogram dwords per level string
#ifndef NSM_STAGE_EARLY
#define NSM_STAGE_EARLY 1
#endif
  constexpr bool kStageEarly = NSM_STAGE_EARLY != 0;
  constexpr bool kPrefetchRows = false;
  // (SPLIT: four waves per block and the histogram bound are the launcher's conditions for the split path; as constants they
  // turn the wave's LDS layout into immediates -- 72 -> 48 B of scratch, configs[4]'s fuzzy grids 279 -> 264 ms)
  const int waves = SPLIT ? 4 : static_cast<int>(blockDim.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  const int tbl_entries = p.pm_stride * kPmWords<K>;
  const size_t tables_bytes = SPLIT ? 2 * kSplitTableBytes : static_cast<size_t>(p.fin_rows) * tbl_entries * 8;
  const size_t wave_bytes = tables_bytes + (K > 1 ? 16 * K * kWave * 4 : 0) +
                            kBatch * kWave * 2 + (K > 1 ? kBatch * kWave * 8 : 0) + kBatch * 3 * kHeadDwords * 4 + kBatch * kRow +
                            (SPLIT ? kQueueBuf * 8 : 0);
  unsigned char* wbase = reinterpret_cast<unsigned char*>(s_mem) + wave * wave_bytes;
  unsigned long long* pm = reinterpret_cast<unsigned long long*>(wbase);
  uint32_t* wtext = reinterpret_cast<uint32_t*>(pm + tables_bytes / 8);
  uint16_t* need = reinterpret_cast<uint16_t*>(wtext + (K > 1 ? 16 * K * kWave : 0));
  double* sc = reinterpret_cast<double*>(need + kBatch * kWave);  // (K > 1) running scores of the rows that go on wave-wide
  uint32_t* head = reinterpret_cast<uint32_t*>(sc + (K > 1 ? kBatch * kWave : 0));
  uint8_t* lstr = reinterpret_cast<uint8_t*>(head + kBatch * 3 * kHeadDwords);
  unsigned long long* qbuf = reinterpret_cast<unsigned long long*>(lstr + kBatch * kRow);  // (SPLIT) [kQueueBuf]
  int qn = 0;  // (SPLIT, wave-uniform) entries in qbuf
  unsigned char* bbase = reinterpret_cast<unsigned char*>(s_mem) + waves * wave_bytes;
  const int park_total = p.park_slots * kSub;
  double* park_score = reinterpret_cast<double*>(bbase);
  int32_t* park_j = reinterpret_cast<int32_t*>(park_score + park_total);
  int32_t* park_meta = park_j + park_total;
  float4* wtab = reinterpret_cast<float4*>(park_meta + park_total);
  unsigned long long& s_cats = *reinterpret_cast<unsigned long long*>(wtab + 66);
  int* s_cnt = reinterpret_cast<int*>(&s_cats + 1);  // [2][kSub], by super-batch parity
  int* s_valid = s_cnt + 2 * kSub;                    // [2][kSub]: slots [0, valid) are written
  const uint32_t pm_base = static_cast<uint32_t>(wave * wave_bytes);

  // Which (tile block, left slice) this block works on.  With a category partition and p.xcd_slices > 0 the grid is 1-D
  // and mapped XCD-aware (blocks go to the 8 XCDs round-robin, each with its own 4 MB L2): a UNIT = (category c, slice y
  // of its left rows) is walked by the tile blocks that hold rows of c; units are dealt to the XCDs round-robin and an XCD
  // takes its units one after the other, so the ~128 blocks resident on an XCD read one or two left slices out of its
  // L2 instead of a dozen (x-fastest order: the resident blocks span ~11 categories, 18 MB of left rows per XCD).
  int bx = blockIdx.x, by = blockIdx.y, ny = gridDim.y, only_cat = -1;
  if (p.xcd_slices > 0) {
    const int xcd = blockIdx.x & 7;
    int k = blockIdx.x >> 3;
    const int rows_per_tb = kWave * waves;
    ny = p.xcd_slices;
    bool found = false;
    for (int c = 0; c < 64 && !found; ++c) {
      const int r0 = rsegstart[c], r1 = rsegstart[c + 1];
      if (r0 >= r1) continue;
      const int first = r0 / rows_per_tb, ntb = (r1 - 1) / rows_per_tb - first + 1;
      const int y0 = ((xcd - c * ny) % 8 + 8) % 8;  // units u = c * ny + y with u % 8 == xcd
      if (y0 >= ny) continue;
      const int cnt_y = (ny - y0 + 7) / 8;
      if (k < cnt_y * ntb) {
        const int yi = k / ntb;
        by = y0 + 8 * yi;
        bx = first + (k - yi * ntb);
        only_cat = c;
        found = true;
      } else {
        k -= cnt_y * ntb;
      }
    }
    if (!found) return;  // (the whole block, before any barrier)
  }
  const int tile = bx * waves + wave;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;  // a whole wave may be beyond the table: it still takes part in the barriers
  const int jc = valid ? j : p.n_right - 1;
  const bool partitioned = rseg != nullptr;
  const int myseg = partitioned ? rseg[jc] : 0;
  // Left rows of this block.  Without a partition: chunk blockIdx.y of rows_per_chunk rows.  With one: slice
  // blockIdx.y of gridDim.y of EACH of the tiles' categories' row ranges -- every block has work (a grid of
  // (tile block, row chunk) pairs is 97 % blocks that find no row of their categories and leave: 7 % of the
  // kernel's time at configs[4]'s shape).
  const int i0 = partitioned ? 0 : (by + p.slice_base) * p.rows_per_chunk;
  const int i1 = partitioned ? p.n_left : min(p.n_left, i0 + p.rows_per_chunk);
  const bool use_hist = SPLIT ? true : p.use_hist != 0;

  if (threadIdx.x == 0) s_cats = 0ull;
  if (threadIdx.x < 2 * kSub) {
    s_cnt[threadIdx.x] = 0;
    s_valid[threadIdx.x] = p.park_slots;
  }
  // step weights of a pair with S steps: R = w2 * ub2 + w3 * ub3 + c bounds the steps >= 2 (rest_bound with
  // both histogram levels), stored as {w2, w3, threshold - (w2 + w3 + c)}
  for (int S = threadIdx.x; S < 66; S += blockDim.x) {
    const double pS = __builtin_ldexp(1.0, -S);
    const double w2 = S >= 2 ? 0.25 : 0.0;
    const double w3 = S >= 3 ? 0.125 + (S <= 4 ? 0.125 - pS : 0.0) : 0.0;
    const double c = S > 4 ? 0.125 - pS : 0.0;
    wtab[S] = make_float4(static_cast<float>(w2), static_cast<float>(w3), static_cast<float>(p.threshold - (w2 + w3 + c)), 0.0f);
  }
  __syncthreads();
  if (partitioned) {
    const unsigned long long mine = wave_or_u64(valid ? (1ull << myseg) : 0ull);
    if (lane == 0 && mine) atomicOr(&s_cats, mine);
  } else if (threadIdx.x == 0) {
    s_cats = 1ull;
  }
  __syncthreads();
  // block-uniform from here on (XCD-aware mapping: this visit of the tile block only serves its unit's category)
  const unsigned long long cats_block = only_cat >= 0 ? (s_cats & (1ull << only_cat)) : s_cats;
  // ---- the lane's right item
  const int lr = rnlev[jc];
  const int rrow0 = rfirst[jc];
  const int jorig = rorig[jc];
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  const int lr_max = wave_max_i32(valid ? lr : 0);
  const unsigned long long lr_deep_m = __builtin_amdgcn_ballot_w64(lr > 1);  // (SPLIT) lanes whose item has more than one level
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
  // the lane's text of the current step; step 1 always reads the same level, so the image survives from
  // batch to batch unless a row went on wave-wide
  uint32_t taddr[K == 1 ? 32 : 1];
  int text_row = -1;
  int lb = 0;
  int pre_ib = -1, pre_ll = 0, pre_lf = 0;

  auto ratio_of = [](int la_, int lb_, int lcs_) -> double {
    // (the index is clamped: an idle lane's operands must never turn into a wild read)
    if constexpr (K == 1) return (la_ == 0 || lb_ == 0) ? 0.0 : g_ratio64.v[min((la_ + lb_) * 65 + lcs_, 129 * 65 - 1)];
    else return indel_score_dev(la_, lb_, lcs_);
  };

  // per-lane histogram bound of one step's level pair, both rows gathered (continuation and dense pass)
  auto step_ub = [&](int lrow, int rrow, int la_, int lb_) -> float {
    if (!use_hist) return 1.0f;
    uint32_t a[8], b[8];
    load_hist<8>(lhist, lrow, a);
    load_hist<8>(rhist, rrow, b);
    return hist_ratio_ub(hist_l1<8>(a, b), la_, lb_);
  };

  // reserve n slots of park region `reg` for this wave; -1 when the region is full.  The counter is never
  // rolled back (a rollback races with the other waves' reservations): it stays inflated, every later
  // reservation fails too, and the written slots are exactly [0, first failing offset).
  auto reserve = [&](int reg, int n) -> int {
    int have = 0;
    if (lane == 0) have = atomicAdd(&s_cnt[reg], n);
    have = __builtin_amdgcn_readfirstlane(have);
    if (have + n > p.park_slots) {
      if (lane == 0) atomicMin(&s_valid[reg], have);
      return -1;
    }
    return have;
  };

  // "some lane": the ballot builtin on the bool (SPLIT only: the fused kernel's code is left as measured).  (Spelling the
  // per-row selects of the H phase and of the category predicate as integer arithmetic, to get rid of the v_cndmask + v_cmp
  // pairs hipcc builds around them, changed nothing: 251.3 vs 250.7 ms.)
  auto any_lane = [](bool pred) -> bool {
    if constexpr (SPLIT) return __builtin_amdgcn_ballot_w64(pred) != 0ull;
    else return __any(pred) != 0;
  };

  // (SPLIT) the wave's collected survivors go to the global queue: one atomic per ~200 entries.  A full queue raises the
  // flag and drops the entries: the host side then runs the fused kernel over the whole grid (gate).
  auto flush_queue = [&]() {
    if (qn == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(qcount, static_cast<unsigned long long>(qn));
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32));
    base = (static_cast<unsigned long long>(hi) << 32) | lo;
    if (base + static_cast<unsigned long long>(qn) > p.qcap) {
      if (lane == 0) atomicOr(qflag, 1);
    } else {
      for (int t = lane; t < qn; t += kWave) queue[base + t] = qbuf[t];
    }
    __builtin_amdgcn_wave_barrier();
    qn = 0;
  };

  // copy the level strings of step s of the rows in `rows` into the wave's LDS (lane = one dword of one row).  All
  // rows' dwords are REQUESTED first (unconditionally; the lanes of a row that is not wanted read a wanted row's string)
  // and stored afterwards: with a load and its store inside a per-row branch the rows' round trips came one after
  // the other -- four per step and batch for 256-byte strings, at 1.5 waves per SIMD.  The level's row comes from the
  // batch's heads (first row and depth are kept there), for every step.
  auto stage_strings = [&](int ib, uint32_t rows, int s) __attribute__((always_inline)) {
    constexpr int kDw = 16 * K;              // dwords per string row
    constexpr int kPer = kWave / kDw > 0 ? kWave / kDw : 1;  // rows per pass (K = 1: 4, K = 2: 2, K >= 4: 1)
    constexpr int kPass = kDw > kWave ? kDw / kWave : 1;     // passes per row (K = 8: 2)
    constexpr int kGroups = (kBatch + kPer - 1) / kPer;
    if (!rows) return;
    const int r_any = __builtin_ctz(rows);  // a row whose head is valid: what the lanes of unwanted rows read
    uint32_t v[kGroups][kPass];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const int r_mine = g * kPer + (kPer > 1 ? lane / kDw : 0);
      const int r = (r_mine < kBatch && ((rows >> r_mine) & 1u)) ? r_mine : r_any;
      const uint32_t* rec = head + r * 3 * kHeadDwords;
      const int lrow = static_cast<int>(rec[NB + 3]) + max(0, min(s, static_cast<int>(rec[NB + 2]) - 1));
#pragma unroll
      for (int q = 0; q < kPass; ++q) {
        const int dw = (kPer > 1 ? lane % kDw : lane) + q * kWave;
        v[g][q] = reinterpret_cast<const uint32_t*>(lcodes + static_cast<size_t>(lrow) * kRow)[dw];
      }
    }
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const int r = g * kPer + (kPer > 1 ? lane / kDw : 0);
      if (r < kBatch && ((rows >> r) & 1u)) {
#pragma unroll
        for (int q = 0; q < kPass; ++q) {
          const int dw = (kPer > 1 ? lane % kDw : lane) + q * kWave;
          reinterpret_cast<uint32_t*>(lstr + r * kRow)[dw] = v[g][q];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // match masks of the staged string of row r (la code units) into table 0
  auto build_pm_table = [&](unsigned long long* tb, int r, int la) __attribute__((always_inline)) {
    for (int c = lane; c < tbl_entries; c += kWave) tb[c] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int pos = lane + kWave * k;
      if (pos < la) {
        const unsigned c = lstr[r * kRow + pos];
        atomicOr(&tb[c * kPmWords<K> + k], 1ull << lane);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto build_pm_staged = [&](int r, int la) __attribute__((always_inline)) { build_pm_table(pm, r, la); };

  // ---- remaining steps of up to 64 pairs, lane = one pair (left row ib + r of the batch, right item row jr,
  // next step s0, score so far -- or, `packed`, the step-1 LCS in the score's bits).  Used for the parked
  // pairs (reg < 0) and for a row whose step-1 survivors were too many to park (reg >= 0: the lanes are the
  // wave's own tile; leftovers may still be parked once they are few).
  auto dense_steps = [&](int ib, int nrows, bool active, int r, int jr, int s0, double score, bool packed,
                         int reg) __attribute__((always_inline)) {
    // (Requesting everything a step needs at its top -- text, next step's histograms -- was tried: the 32 more
    // live registers spilled and the dense pass went from 5.5 to 22 ms per 3 x 100k^2 grids.)
    const int i = ib + r;
    const int ll = lnlev[i], lf = lfirst[i];
    const int lrj = rnlev[jr], rr0 = rfirst[jr];
    // lane q < nrows also looks after batch row q's mask table
    const int q_lf = lane < nrows ? lfirst[ib + lane] : 0;
    const int q_ll = lane < nrows ? lnlev[ib + lane] : 1;
    if (packed && active) {  // the score of step 1 is 2^-1 * ratio (idle lanes carry no LCS)
      const int lcs1 = static_cast<int>(__double_as_longlong(score));
      score = indel_score_dev(llen[lf + max(0, min(1, ll - 1))], rlen[rr0 + max(0, min(1, lrj - 1))], lcs1) * 0.5;
    }
    const int S = max(ll, lrj);
    const int s_lo = 64 - wave_max_i32(active ? 64 - s0 : 0);  // smallest next step (steps <= 64)
    const int s_hi = wave_max_i32(active ? S : 0);
    int prev_a = -1, prev_b = -1;
    double ratio = 0.0;
    bool alive = active;  // still a candidate: running, or finished with its final score
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
        const int my_lrow = q_lf + max(0, min(s, q_ll - 1));
        const int my_la = llen[my_lrow];
        int lcs = 0;
        for (int g0 = 0; g0 < nrows; g0 += p.fin_rows) {
          const bool mine = fresh && r >= g0 && r < g0 + p.fin_rows;
          const uint32_t rows_here = wave_reduce_u32(mine ? (1u << r) : 0u, [](uint32_t x, uint32_t y) { return x | y; });
          if (!rows_here) continue;
          const uint8_t* tptr = rcodes + static_cast<size_t>(mine ? rrow : rr0) * kRow;
          // the group's mask tables, one per left row present: every row's code units are requested before
          // the first is used, then the tables are zeroed and the bits set
          unsigned cu[kBatch][K];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) {
            const int rr = g0 + q;
            if (q < p.fin_rows && ((rows_here >> rr) & 1u)) {
              const int lrow_u = __builtin_amdgcn_readlane(my_lrow, rr);
#pragma unroll
              for (int k = 0; k < K; ++k) cu[q][k] = lcodes[static_cast<size_t>(lrow_u) * kRow + lane + kWave * k];
            }
          }
          for (int c = lane; c < p.fin_rows * tbl_entries; c += kWave) pm[c] = 0ull;
          if constexpr (K > 1) wide_store_text<K>(wtext, tptr, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          int la_max = 0;
#pragma unroll
          for (int q = 0; q < kBatch; ++q) {
            const int rr = g0 + q;
            if (q < p.fin_rows && ((rows_here >> rr) & 1u)) {
              const int la_u = __builtin_amdgcn_readlane(my_la, rr);
              la_max = max(la_max, la_u);
              unsigned long long* tb = pm + static_cast<size_t>(q) * tbl_entries;
#pragma unroll
              for (int k = 0; k < K; ++k)
                if (lane + kWave * k < la_u) atomicOr(&tb[cu[q][k] * kPmWords<K> + k], 1ull << lane);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const unsigned long long* tbl = pm + static_cast<size_t>(mine ? r - g0 : 0) * tbl_entries;
          const int nchars = wave_max_i32(mine ? lbj : 0);
          int got;
          if constexpr (K == 1) {
            uint32_t text[16];
            const uint4* tp = reinterpret_cast<const uint4*>(tptr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint4 v = tp[q];
              text[4 * q + 0] = v.x; text[4 * q + 1] = v.y; text[4 * q + 2] = v.z; text[4 * q + 3] = v.w;
            }
            unsigned long long v = ~0ull;
#pragma unroll
            for (int g = 0; g < 8; ++g) {  // 8 code units per group, mask reads issued together (see the scan)
              if (g * 8 < nchars) {
                unsigned long long m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = tbl[(text[2 * g + (q >> 2)] >> (8 * (q & 3))) & 0xffu];
#pragma unroll
                for (int q = 0; q < 8; ++q) v = lcs_step64(v, m[q]);
              }
            }
            got = 64 - __popcll(v);
          } else {
            got = wide_lcs<K>(tbl, wtext, nchars, lane, la_max);
          }
          if (mine) lcs = got;
        }
        if (fresh) {
          ratio = indel_score_dev(la, lbj, lcs);  // arithmetic, not the table: no gather in the chain
          prev_a = a;
          prev_b = b;
        }
      }
      if (run) {
        score += ratio * factor;
        // steps still to come: histogram bound of the next level pair (exact upper bound; 1e-6 covers the
        // float arithmetic of the bound and the rounding of the double sum)
        float rest = 0.0f;
        if (s < S) {
          const int lrow_n = lf + max(0, min(s + 1, ll - 1)), rrow_n = rr0 + max(0, min(s + 1, lrj - 1));
          rest = rest_bound(s, S, step_ub(lrow_n, rrow_n, llen[lrow_n], rlen[rrow_n]));
        }
        alive = score + static_cast<double>(rest) + 1e-6 >= p.threshold;
      }
      if (reg >= 0) {  // few pairs left with steps to go: hand them to the block's dense pass
        const bool pending = alive && s >= s0 && s < S;
        const unsigned long long who = __ballot(pending);
        const int n = __popcll(who);
        if (n > 0 && n <= p.park_max) {
          const int have = reserve(reg, n);
          if (have >= 0) {
            if (pending) {
              const int slot = (reg % kSub) * p.park_slots + have +
                               __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
              park_score[slot] = score;
              park_j[slot] = jr;
              park_meta[slot] = r | ((s + 1) << 8);
              alive = false;
            }
          }
        }
      }
    }
    emit_hits_wave(hits, p.cap, count, active && alive && score >= p.threshold, score, lorig[i], rorig[jr]);
  };

  // ---- this wave's tile against the batch rows [ib, ib + nrows); okbits bit r = the lane passes the
  // category predicate for row ib + r; survivors go to park region `reg`
  auto scan_batch = [&](int ib, int nrows, uint32_t okbits, uint32_t rows_ok, int reg) __attribute__((always_inline)) {
#ifdef NSM_X_EMPTY
    return;
#endif
    // ---- stage: heads of the batch's rows (lane = (row, step 1..3))
    // One-word strings: the step-1 level strings of ALL the batch's rows ride along (lanes 24..55 = 16 bytes of a row
    // each), one dependent load behind the row's (first, depth) like the heads -- staging them after the H phase, for
    // the rows with a live lane only, was a third dependent round trip per batch, and at configs[4]'s survival rate
    // every row has a live lane (3 x 100k^2: 20.95 -> 20.6 ms).  The (first, depth) loads are UNCONDITIONAL, rows past
    // the batch clamped: inside the lane-dependent branches each branch waited for its own round trips, one branch
    // after the other (20.6 -> 19.1 ms).  Tried on top and dropped: the second-level loads unconditional too (every
    // lane loads a histogram, a length and 16 bytes of string: 20.2 ms); (first, depth) requested one batch ahead
    // and kept in two registers across the scan (19.7 ms).
    // (Spelled exactly like this on purpose: at the 128 registers of 4 waves per SIMD the one-word kernel spills a
    // little, and WHERE it spills moves with the spelling -- the same statements with the row index reused in the
    // branches measured 20.4 ms.)
    const int my_r = lane < 3 * kBatch ? lane / 3 : (lane - 3 * kBatch) >> 2;
    int cur_ll, cur_lf;
    if (kPrefetchRows && pre_ib == ib) {
      cur_ll = pre_ll;
      cur_lf = pre_lf;
    } else {
      const int i = min(ib + my_r, p.n_left - 1);
      cur_ll = lnlev[i];
      cur_lf = lfirst[i];
    }
    if (kPrefetchRows) {  // (first, depth) of the next batch's rows one batch ahead: measured slower, compiled out
      const int i = min(ib + kBatch + my_r, p.n_left - 1);
      pre_ll = lnlev[i];
      pre_lf = lfirst[i];
      pre_ib = ib + kBatch;
    }
    if (lane < nrows * 3) {
      const int r = lane / 3, t = lane - 3 * r;
      const int ll = cur_ll, lf = cur_lf;
      const int lrow = lf + max(0, min(t + 1, ll - 1));
      uint32_t h[NB];
      if (use_hist) load_hist<NB>(lhist, lrow, h);
      else
#pragma unroll
        for (int q = 0; q < NB; ++q) h[q] = 0u;
      uint32_t* rec = head + lane * kHeadDwords;
#pragma unroll
      for (int q = 0; q < NB; ++q) rec[q] = h[q];
      rec[NB] = static_cast<uint32_t>(llen[lrow]);
      rec[NB + 1] = static_cast<uint32_t>(lrow);
      rec[NB + 2] = static_cast<uint32_t>(ll);
      rec[NB + 3] = static_cast<uint32_t>(lf);
    } else if (K == 1 && kStageEarly && lane >= 3 * kBatch && lane < 3 * kBatch + 4 * nrows) {
      const int q = lane - 3 * kBatch, r = q >> 2, part = q & 3;
      const int lrow = cur_lf + max(0, min(1, cur_ll - 1));
      reinterpret_cast<uint4*>(lstr + r * kRow)[part] = reinterpret_cast<const uint4*>(lcodes + static_cast<size_t>(lrow) * kRow)[part];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- H: need[r][lane]
    uint32_t live = 0;
    // (SPLIT) the rows' step-1 lengths and depths as bytes of two scalars: the row loop and after_lcs read them with scalar
    // shifts instead of an LDS round trip + v_readfirstlane per row
    unsigned long long la_pack = 0ull, ll_pack = 0ull;
    for (uint32_t rows = rows_ok; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      const uint32_t* rec = head + r * 3 * kHeadDwords;
      const int ll = wave_first(static_cast<int>(rec[NB + 2]));
      const int S = max(ll, lr);
      const float4 W = wtab[min(S, 65)];
      uint32_t l1[3];
      int la[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        la[t] = static_cast<int>(rec[t * kHeadDwords + NB]);
        l1[t] = 0u;
        if (use_hist && (t > 0 || NSM_PARK_H1)) {
          uint32_t hl[NB];
#pragma unroll
          for (int q = 0; q < NB; ++q) hl[q] = rec[t * kHeadDwords + q];
          l1[t] = hist_l1<NB>(hl, hb[t]);
        } else if (t == 0) {
          l1[0] = static_cast<uint32_t>(abs(la[0] - lb_t[0]));  // LCS <= min(la, lb)
        }
      }
      // alive after step 1  <=>  lcs / n1 + R >= thr,  R = (w2 + w3 + c) - w2 l2 / n2 - w3 l3 / n3;
      // 2e-3 of an LCS unit covers the float rounding
      const float i2 = __builtin_amdgcn_rcpf(static_cast<float>(max(la[1] + lb_t[1], 1)));
      const float i3 = __builtin_amdgcn_rcpf(static_cast<float>(max(la[2] + lb_t[2], 1)));
      const float x = W.x * static_cast<float>(l1[1]) * i2 + W.y * static_cast<float>(l1[2]) * i3;
      const int n1 = la[0] + lb_t[0];
      const float needf = static_cast<float>(n1) * (W.z + x) - 2e-3f;
      const int nd = max(0, static_cast<int>(__builtin_ceilf(needf)));
      const int m1 = (n1 - static_cast<int>(l1[0])) >> 1;  // LCS of step 1 <= m1 (<= min(la, lb))
      const bool can = ((okbits >> r) & 1u) && nd <= m1;
      need[r * kWave + lane] = can ? static_cast<uint16_t>(nd) : kDeadNeed;
      live |= any_lane(can) ? (1u << r) : 0u;
      if constexpr (SPLIT) {
        la_pack |= static_cast<unsigned long long>(wave_first(la[0]) & 0xff) << (8 * r);
        ll_pack |= static_cast<unsigned long long>(ll & 0xff) << (8 * r);
      }
      NSM_SCAN_STAT(0, 1);
      NSM_SCAN_STAT(1, __popcll(__ballot((okbits >> r) & 1u)));
      NSM_SCAN_STAT(2, __popcll(__ballot(can)));
      NSM_SCAN_STAT(3, __any(can) ? 1 : 0);
    }
    if (!live) return;
#ifdef NSM_X_HONLY
    return;
#endif

    // ---- step 1, wave-wide
    if (!(K == 1 && kStageEarly)) stage_strings(ib, live, 1);
    // (SPLIT) the code unit at position `lane` of every row's staged string, fetched once per batch (eight reads in flight
    // together): per pass the read -> address -> ds_or chain was an LDS round trip per row before the LCS could start
    unsigned long long my_codes = 0ull;
    if constexpr (SPLIT && K == 1) {
      uint32_t c8[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) c8[q] = lstr[q * kRow + lane];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) my_codes |= static_cast<unsigned long long>(c8[q]) << (8 * q);
    }
    const int rrow = rrow0 + max(0, min(1, lr - 1));
    if constexpr (K == 1) {
      if (rrow != text_row) {  // step 1 reads the same right level for every batch
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
    uint32_t over = 0;  // rows whose survivors were too many to park: their remaining steps follow below
    // what follows a row's step-1 LCS: the survivors are parked, or the row is marked for the dense steps below
    auto after_lcs = [&](int r, int la, int lcs, int nd_early = -1) __attribute__((always_inline)) {
      const int ll = SPLIT ? static_cast<int>((ll_pack >> (8 * r)) & 0xff)
                           : wave_first(static_cast<int>(head[r * 3 * kHeadDwords + NB + 2]));
      const int S = max(ll, lr);
      const int nd = nd_early >= 0 ? nd_early : need[r * kWave + lane];
      if constexpr (SPLIT) {
        // (masks combined as scalars)
        const bool alive = lcs >= nd;  // (kDeadNeed > any LCS)
        const unsigned long long alive_m = __builtin_amdgcn_ballot_w64(alive);
        if (alive_m == 0ull) return;
        const unsigned long long deep_m = ll > 1 ? ~0ull : lr_deep_m;  // lanes whose pair has more than one step
        if (alive_m & ~deep_m) {  // single-step pairs (both items have one level): final here
          const double score = ratio_of(la, lb, lcs) * 0.5;
          const bool hit = alive && S <= 1 && score >= p.threshold;
          emit_hits_wave(hits, p.cap, count, hit, score, lorig[ib + r], jorig);
        }
        const unsigned long long who = alive_m & deep_m;
        if (who == 0ull) return;
        NSM_SCAN_STAT(4, __popcll(who));
        if (alive && S > 1)
          qbuf[qn + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u))] = queue_entry(ib + r, jc, lcs);
        qn += __popcll(who);
        if (qn > kQueueBuf - kWave) flush_queue();
        return;
      }
      const bool alive = nd != kDeadNeed && lcs >= nd;  // can still reach the threshold
      if (!any_lane(alive)) return;
      const bool more = alive && S > 1;
      if (any_lane(alive && !more)) {  // single-step pairs (both items have one level): final here
        const double score = ratio_of(la, lb, lcs) * 0.5;
        const bool hit = alive && !more && score >= p.threshold;
        emit_hits_wave(hits, p.cap, count, hit, score, lorig[ib + r], jorig);
      }
      const unsigned long long who = __ballot(more);
      if (who == 0ull) return;
      const int n = __popcll(who);
      NSM_SCAN_STAT(4, n);
      int have = -1;
      if (n <= p.park_max) have = reserve(reg, n);
      if (have >= 0) {
        // parked with the LCS itself (bit 16 of meta): the dense pass turns it into the score -- the ratio
        // table is a gather from L2 whose latency would be paid per row here
        if (more) {
          const int slot = (reg % kSub) * p.park_slots + have +
                           __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
          park_score[slot] = __longlong_as_double(static_cast<long long>(lcs));
          park_j[slot] = jc;
          park_meta[slot] = r | (2 << 8) | 0x10000;
        }
      } else {  // dense enough (or the park is full): the row's remaining steps are scored by this wave itself
        need[r * kWave + lane] = more ? static_cast<uint16_t>(lcs) : kDeadNeed;
        over |= 1u << r;
      }
    };
    for (uint32_t rows = live; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      const int la = SPLIT ? static_cast<int>((la_pack >> (8 * r)) & 0xff)
                           : wave_first(static_cast<int>(head[r * 3 * kHeadDwords + NB]));
      if constexpr (K == 1) {
        // Two left rows per pass when both fit 32-bit words: row A's masks in the low, row B's in the high half
        // of ONE table entry, so a code unit of the text costs one address op and one ds_read_b64 for both
        // rows, and the two recurrences are independent chains the SIMD can interleave.  (Per row: 4.5 VALU ops
        // per code unit instead of 5, half the LDS reads, half the table builds and loop overhead.)
        const int r2 = rows ? __builtin_ctz(rows) : -1;
        const int la2 = r2 < 0 ? 64
                        : SPLIT ? static_cast<int>((la_pack >> (8 * r2)) & 0xff)
                                : wave_first(static_cast<int>(head[r2 * 3 * kHeadDwords + NB]));
        if constexpr (SPLIT && NSM_SPLIT_QUAD) {
          if (la <= 32 && la2 <= 32 && __builtin_popcount(rows) >= 3) {
            const uint32_t rest = rows & (rows - 1);
            const int r3 = __builtin_ctz(rest), r4 = __builtin_ctz(rest & (rest - 1));
            const int la3 = wave_first(static_cast<int>(head[r3 * 3 * kHeadDwords + NB]));
            const int la4 = wave_first(static_cast<int>(head[r4 * 3 * kHeadDwords + NB]));
            if (la3 <= 32 && la4 <= 32) {
              rows = rest & (rest - 1);
              rows &= rows - 1;
              if (lane < tbl_entries) {
                pm[lane] = 0ull;
                pm[kSplitTableBytes / 8 + lane] = 0ull;
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              uint32_t* pm32 = reinterpret_cast<uint32_t*>(pm);
              if (lane < la) atomicOr(&pm32[2 * lstr[r * kRow + lane]], 1u << lane);
              if (lane < la2) atomicOr(&pm32[2 * lstr[r2 * kRow + lane] + 1], 1u << lane);
              if (lane < la3) atomicOr(&pm32[kSplitTableBytes / 4 + 2 * lstr[r3 * kRow + lane]], 1u << lane);
              if (lane < la4) atomicOr(&pm32[kSplitTableBytes / 4 + 2 * lstr[r4 * kRow + lane] + 1], 1u << lane);
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
              asm volatile("" : "+v"(lowmask), "+v"(sh16));
              uint32_t va = ~0u, vb = ~0u, vc = ~0u, vd = ~0u;
#pragma unroll
              for (int g = 0; g < 8; ++g) {
                if (g * 8 < nchars) {
                  unsigned long long m[8], n[8];
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    // (volatile: left alone hipcc fuses the two reads into ds_read2st64_b64, which the LDS serves at a
                    // quarter of the rate of two ds_read_b64 -- indel_tile_lcs.hpp)
                    using lds_u64 = const volatile __attribute__((address_space(3))) unsigned long long;
                    const uint32_t a0 = taddr[4 * g + q] & lowmask, a1 = taddr[4 * g + q] >> sh16;
                    m[2 * q] = reinterpret_cast<lds_u64*>(a0)[0];
                    n[2 * q] = reinterpret_cast<lds_u64*>(a0)[kSplitTableBytes / 8];
                    m[2 * q + 1] = reinterpret_cast<lds_u64*>(a1)[0];
                    n[2 * q + 1] = reinterpret_cast<lds_u64*>(a1)[kSplitTableBytes / 8];
                  }
#pragma unroll
                  for (int q = 0; q < 8; ++q) {
                    va = lcs_step32(va, static_cast<uint32_t>(m[q]));
                    vb = lcs_step32(vb, static_cast<uint32_t>(m[q] >> 32));
                    vc = lcs_step32(vc, static_cast<uint32_t>(n[q]));
                    vd = lcs_step32(vd, static_cast<uint32_t>(n[q] >> 32));
                  }
                }
              }
              NSM_SCAN_STAT(5, 2);
              after_lcs(r, la, 32 - __popc(va));
              after_lcs(r2, la2, 32 - __popc(vb));
              after_lcs(r3, la3, 32 - __popc(vc));
              after_lcs(r4, la4, 32 - __popc(vd));
              continue;
            }
          }
        }
        if (la <= 32 && la2 <= 32) {
          rows &= rows - 1;
          if constexpr (SPLIT) pm[lane] = 0ull;  // (the table region holds 64 entries: no bound to test)
          else
            for (int c = lane; c < tbl_entries; c += kWave) pm[c] = 0ull;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          uint32_t* pm32 = reinterpret_cast<uint32_t*>(pm);
          int nd_a = -1, nd_b = -1;
          if constexpr (SPLIT) {
            // codes from the batch's prefetch; the rows' `need` is requested here too, so that it has arrived when the LCS
            // ends.  (The ORs stay under `lane < la`: unconditional, with a zero bit for the lanes past the string's end, they
            // all hit the pad symbol's entry -- 30-odd same-address atomics per pass, 254 -> 303 ms.)
            if (lane < la) atomicOr(&pm32[2 * static_cast<uint32_t>((my_codes >> (8 * r)) & 0xffu)], 1u << lane);
            if (lane < la2) atomicOr(&pm32[2 * static_cast<uint32_t>((my_codes >> (8 * r2)) & 0xffu) + 1], 1u << lane);
#ifndef NSM_SPLIT_ND_EARLY
#define NSM_SPLIT_ND_EARLY 1
#endif
            if (NSM_SPLIT_ND_EARLY) {
              nd_a = need[r * kWave + lane];
              nd_b = need[r2 * kWave + lane];
            }
          } else {
            if (lane < la) atomicOr(&pm32[2 * lstr[r * kRow + lane]], 1u << lane);
            if (lane < la2) atomicOr(&pm32[2 * lstr[r2 * kRow + lane] + 1], 1u << lane);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          asm volatile("" : "+v"(lowmask), "+v"(sh16));
          uint32_t va = ~0u, vb = ~0u;
#ifdef NSM_PARK_GROUP4  // (A/B builds: 4 code units per group -- less padding past the longest text, more branches)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            if (g * 4 < nchars) {
              unsigned long long m[4];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                m[2 * q] = lev_lds_load<unsigned long long>(taddr[2 * g + q] & lowmask);
                m[2 * q + 1] = lev_lds_load<unsigned long long>(taddr[2 * g + q] >> sh16);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                va = lcs_step32(va, static_cast<uint32_t>(m[q]));
                vb = lcs_step32(vb, static_cast<uint32_t>(m[q] >> 32));
              }
            }
          }
#else
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            if (g * 8 < nchars) {
              unsigned long long m[8];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                m[2 * q] = lev_lds_load<unsigned long long>(taddr[4 * g + q] & lowmask);
                m[2 * q + 1] = lev_lds_load<unsigned long long>(taddr[4 * g + q] >> sh16);
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                va = lcs_step32(va, static_cast<uint32_t>(m[q]));
                vb = lcs_step32(vb, static_cast<uint32_t>(m[q] >> 32));
              }
            }
          }
#endif
          NSM_SCAN_STAT(5, 1);
          after_lcs(r, la, 32 - __popc(va), nd_a);
          after_lcs(r2, la2, 32 - __popc(vb), nd_b);
          continue;
        }
      }
      if constexpr (K > 1) {
        // two left rows per pass (multi-word strings): tables 0 and 1 of the wave, one text read for both
        const int r2 = rows ? __builtin_ctz(rows) : -1;
        const int la2 = r2 >= 0 ? wave_first(static_cast<int>(head[r2 * 3 * kHeadDwords + NB])) : 0;
        if (r2 >= 0 && p.fin_rows >= 2 && max(la, la2) <= 2 * kWave) {
          rows &= rows - 1;
          build_pm_table(pm, r, la);
          build_pm_table(pm + tbl_entries, r2, la2);
          int lcs_a, lcs_b;
          wide_lcs2<K>(pm, pm + tbl_entries, wtext, nchars, lane, max(la, la2), lcs_a, lcs_b);
          after_lcs(r, la, lcs_a);
          after_lcs(r2, la2, lcs_b);
          continue;
        }
      }
      NSM_SCAN_STAT(6, 1);
#ifndef NSM_X_NOPM
      build_pm_staged(r, la);
#endif
      int lcs;
#ifdef NSM_X_NOLCS
      lcs = 0;
      if (false)
#endif
      if constexpr (K == 1) {
        // opaque per row: otherwise the 64 unpacked addresses are hoisted out of the row loop into 64
        // more VGPRs
        asm volatile("" : "+v"(lowmask), "+v"(sh16));
        // 8 code units per group: the 8 mask reads are issued together and their LDS latency overlaps the
        // recurrence of the group before; positions past the text's end read the all-zero pad mask
        if (la <= 32) {  // wave-uniform: 32-bit words, and / add / xor / or all issue at full rate
          uint32_t v = ~0u;
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            if (g * 8 < nchars) {
              uint32_t m[8];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                m[2 * q] = lev_lds_load<uint32_t>(taddr[4 * g + q] & lowmask);
                m[2 * q + 1] = lev_lds_load<uint32_t>(taddr[4 * g + q] >> sh16);
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) v = lcs_step32(v, m[q]);
            }
          }
          lcs = 32 - __popc(v);
        } else {
          unsigned long long v = ~0ull;
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            if (g * 8 < nchars) {
              unsigned long long m[8];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                m[2 * q] = lev_lds_load<unsigned long long>(taddr[4 * g + q] & lowmask);
                m[2 * q + 1] = lev_lds_load<unsigned long long>(taddr[4 * g + q] >> sh16);
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) v = lcs_step64(v, m[q]);
            }
          }
          lcs = 64 - __popcll(v);
        }
      } else {
        lcs = wide_lcs<K>(pm, wtext, nchars, lane, la);
      }
      after_lcs(r, la, lcs);
    }
    if constexpr (SPLIT) return;
    if (over) {
      if constexpr (K == 1) {
        text_row = -1;  // the text image's registers are free for the dense steps
        for (uint32_t rows = over; rows;) {
          const int r = __builtin_ctz(rows);
          rows &= rows - 1;
          const int lcs1 = need[r * kWave + lane];
          dense_steps(ib, nrows, lcs1 != kDeadNeed, r, jc, 2, __longlong_as_double(static_cast<long long>(lcs1)), true, reg);
        }
      } else {
        // Multi-word strings at low thresholds keep most lanes alive after step 1 (Term-like items at 0.5: 61 %):
        // those rows go on wave-wide and STEP-MAJOR -- per step the lane's right level string is staged once for
        // all of them, the left level strings come from one staging pass, the left histograms of steps <= 3 from
        // the heads -- until few enough lanes are left to park.  (Row-major dense steps re-staged the 256-byte
        // text and chased the rows' metadata through global memory per row and step: 163 vs 140 ms.)
#ifdef NSM_X_NOCONT
        over = 0;
#endif
        const uint32_t over0 = over;
        int ll_max = 0;
        for (uint32_t rows = over; rows;) {
          const int r = __builtin_ctz(rows);
          rows &= rows - 1;
          const int lcs1 = need[r * kWave + lane];
          const int la1 = wave_first(static_cast<int>(head[r * 3 * kHeadDwords + NB]));
          sc[r * kWave + lane] = lcs1 != kDeadNeed ? ratio_of(la1, lb, lcs1) * 0.5 : __builtin_nan("");
          ll_max = max(ll_max, wave_first(static_cast<int>(head[r * 3 * kHeadDwords + NB + 2])));
        }
        const int steps_max = max(ll_max, lr_max);
        double factor = 0.5;
        for (int s = 2; s <= steps_max && over; ++s) {
          factor *= 0.5;
          stage_strings(ib, over, s);
          const int rrow_s = rrow0 + max(0, min(s, lr - 1));
          if (__any(rrow_s != text_row)) {
            text_row = rrow_s;
            wide_store_text<K>(wtext, rcodes + static_cast<size_t>(rrow_s) * kRow, lane);
            lb = rlen[rrow_s];
          }
          const int nch = wave_max_i32(valid ? lb : 0);
          // the lane's histogram of the NEXT step's level bounds what is still to come: once per step
          const int rrow_n = rrow0 + max(0, min(s + 1, lr - 1));
          const int lb_n = rlen[rrow_n];
          uint32_t hbn[8];
          if (use_hist) load_hist<8>(rhist, rrow_n, hbn);
          // the left level of step s of row r: its length (heads for s <= 3)
          auto level_len = [&](int r) -> int {
            const uint32_t* rec = head + r * 3 * kHeadDwords;
            const int ll = wave_first(static_cast<int>(rec[NB + 2]));
            const int lf = wave_first(static_cast<int>(rec[NB + 3]));
            const int la = s <= 3 ? static_cast<int>(rec[(s - 1) * kHeadDwords + NB]) : llen[lf + max(0, min(s, ll - 1))];
            return wave_first(la);
          };
          auto still_running = [&](int r) -> bool {  // wave-uniform: some lane of row r still scores step s
            const int ll = wave_first(static_cast<int>(head[r * 3 * kHeadDwords + NB + 2]));
            const double score = sc[r * kWave + lane];
            return __any((score == score) && s <= max(ll, lr));
          };
          // what follows the LCS of (row r, step s): score, bound on the rest, park or go on
          auto after_step = [&](int r, int la, int lcs) __attribute__((always_inline)) {
            const uint32_t* rec = head + r * 3 * kHeadDwords;
            const int ll = wave_first(static_cast<int>(rec[NB + 2]));
            const int lf = wave_first(static_cast<int>(rec[NB + 3]));
            const int S = max(ll, lr);
            double score = sc[r * kWave + lane];
            const bool run = (score == score) && s <= S;  // NaN = dropped or parked
            if (run) score += ratio_of(la, lb, lcs) * factor;
            float rest = 0.0f;
            if (s < S) {
              const int t = s + 1;
              float ub = 1.0f;
              if (use_hist) {
                uint32_t hl[8];
                int la_n;
                if (t <= 3) {
                  la_n = static_cast<int>(rec[(t - 1) * kHeadDwords + NB]);
#pragma unroll
                  for (int q = 0; q < 8; ++q) hl[q] = rec[(t - 1) * kHeadDwords + q];
                } else {
                  const int lrow_n = lf + max(0, min(t, ll - 1));
                  la_n = llen[lrow_n];
                  load_hist<8>(lhist, lrow_n, hl);
                }
                ub = hist_ratio_ub(hist_l1<8>(hl, hbn), la_n, lb_n);
              }
              rest = rest_bound(s, S, ub);
            }
            const bool alive = run && (score + static_cast<double>(rest) + 1e-6 >= p.threshold);
            if (run && !alive) score = __builtin_nan("");
            const bool pending = alive && s < S;
            const unsigned long long who = __ballot(pending);
            const int n = __popcll(who);
            if (n > 0 && n <= p.park_max) {
              const int have = reserve(reg, n);
              if (have >= 0) {
                if (pending) {
                  const int slot = (reg % kSub) * p.park_slots + have +
                                   __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
                  park_score[slot] = score;
                  park_j[slot] = jc;
                  park_meta[slot] = r | ((s + 1) << 8);
                  score = __builtin_nan("");  // the dense pass owns the pair now
                }
                sc[r * kWave + lane] = score;
                over &= ~(1u << r);
                return;
              }
            }
            sc[r * kWave + lane] = score;
            if (n == 0) over &= ~(1u << r);
          };
          for (uint32_t rows = over; rows;) {
            const int r = __builtin_ctz(rows);
            rows &= rows - 1;
            if (!still_running(r)) {  // finished lanes keep their final scores in sc
              over &= ~(1u << r);
              continue;
            }
            const int la = level_len(r);
            // two rows per pass when the next row is still running too (tables 0 and 1, one text read for both)
            const int r2 = rows ? __builtin_ctz(rows) : -1;
            if (r2 >= 0 && p.fin_rows >= 2 && still_running(r2)) {
              const int la2 = level_len(r2);
              if (max(la, la2) <= 2 * kWave) {
                rows &= rows - 1;
                build_pm_table(pm, r, la);
                build_pm_table(pm + tbl_entries, r2, la2);
                int lcs_a, lcs_b;
                wide_lcs2<K>(pm, pm + tbl_entries, wtext, nch, lane, max(la, la2), lcs_a, lcs_b);
                after_step(r, la, lcs_a);
                after_step(r2, la2, lcs_b);
                continue;
              }
            }
            build_pm_staged(r, la);
            after_step(r, la, wide_lcs<K>(pm, wtext, nch, lane, la));
          }
        }
        for (uint32_t rows = over0; rows;) {  // pairs that finished in the wave-wide steps
          const int r = __builtin_ctz(rows);
          rows &= rows - 1;
          const double score = sc[r * kWave + lane];
          emit_hits_wave(hits, p.cap, count, score >= p.threshold, score, lorig[ib + r], jorig);
        }
        text_row = -1;  // step 1 of the next batch reads another level
      }
    }
  };

  int pb = 0;  // parity of the super-batch: which half of the park counters is in use
  for (unsigned long long cats = cats_block; cats;) {  // the same sequence in every wave of the block
    const int c = __builtin_ctzll(cats);
    cats &= cats - 1;
    int a = i0, b = i1;
    if (partitioned) {  // slice blockIdx.y of the category's rows, cut at batch boundaries
      const int lo = lsegstart[c], len = lsegstart[c + 1] - lo;
      const int tot = p.slices_total > 0 ? p.slices_total : ny;
      const int per = (((len + tot - 1) / tot) + kBatch - 1) / kBatch * kBatch;
      a = lo + min(len, (by + p.slice_base) * per);
      b = lo + min(len, (by + p.slice_base + 1) * per);
    }
    const unsigned long long lower = (1ull << c) - 1ull;
    for (int sb = a; sb < b; sb += kBatch * kSub) {
      for (int g = 0; g < kSub; ++g) {
        const int ib = sb + g * kBatch;
        if (ib >= b) break;
        const int nrows = min(kBatch, b - ib);
        // (the rows' category masks as scalar loads: one vector load + v_readlane per row, issued together with the
        // staging lanes' (depth, first) loads, was slower: 20.5 vs 19.1 ms)
        uint32_t okbits = 0, rows_ok = 0;
        for (int r = 0; r < nrows; ++r) {
          const uint64_t cl = (p.cat_mode != NSM_CAT_NONE) ? lcat[ib + r] : 0ull;
          bool ok = valid;
          if (partitioned) ok = ok && myseg == c && ((cl & catr & lower) == 0ull);
          else if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(cl, catr, p.cat_mode);
          okbits |= ok ? (1u << r) : 0u;
          rows_ok |= any_lane(ok) ? (1u << r) : 0u;
        }
        if (rows_ok) scan_batch(ib, nrows, okbits, rows_ok, pb * kSub + g);
      }
      if constexpr (SPLIT) continue;  // no park, no dense pass, no barrier: the finish kernel takes the survivors
      __syncthreads();  // every tile's survivors of this super-batch are parked
      text_row = -1;    // the text image does not outlive the super-batch: its registers are free in the dense pass
      if (threadIdx.x < kSub) {  // the other parity's counters were consumed before the previous super-batch's last barrier
        s_cnt[(pb ^ 1) * kSub + threadIdx.x] = 0;
        s_valid[(pb ^ 1) * kSub + threadIdx.x] = p.park_slots;
      }
      for (int g = wave; g < kSub; g += waves) {
        const int ib = sb + g * kBatch;
        if (ib >= b) break;
        const int n_p = min(s_cnt[pb * kSub + g], s_valid[pb * kSub + g]);
#ifndef NSM_NO_FINISH
        for (int base = 0; base < n_p; base += kWave) {
          const bool active = base + lane < n_p;
          const int slot = g * p.park_slots + (active ? base + lane : base);
          const int meta = park_meta[slot];
          dense_steps(ib, min(kBatch, b - ib), active, meta & 0xff, park_j[slot], (meta >> 8) & 0xff, park_score[slot],
                      (meta & 0x10000) != 0, -1);
        }
#endif
      }
      __syncthreads();  // the park has been read
      pb ^= 1;
    }
  }
  if constexpr (SPLIT) flush_queue();
}

}  // namespace nsm
