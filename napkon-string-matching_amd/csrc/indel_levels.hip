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

// Load from an LDS byte address held in a register (ds_read_b32 / ds_read_b64).
template <typename T>
__device__ __forceinline__ T lev_lds_load(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T*>(addr);
}

// tunables, A/B-measured on C5-shaped cohorts (3 x 100k^2): batch 4 / 8 / 16 -> 59.3 / 53.7 / 63.1 ms,
// chunk 64 / 128 / 256 -> 53.2 / 53.7 / 54.8 ms
#ifndef NSM_LEV_BATCH
#define NSM_LEV_BATCH 8
#endif
#ifndef NSM_LEV_CHUNK
#define NSM_LEV_CHUNK 128
#endif
// left rows scored together, step by step; the 4-word kernel already spends 18 KB of LDS per wave on
// masks and text image, 4 rows (2 KB of scores) keep 8 waves per CU
constexpr int lev_batch(int K) { return K >= 4 ? NSM_LEV_BATCH / 2 : NSM_LEV_BATCH; }

template <int K>
__global__ __launch_bounds__(kBlock) void indel_levels_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint64_t* __restrict__ lcat, const int32_t* __restrict__ lsegstart, const uint8_t* __restrict__ lcodes,
    const int32_t* __restrict__ llen, const int32_t* __restrict__ rfirst, const int32_t* __restrict__ rnlev,
    const int32_t* __restrict__ rorig, const uint64_t* __restrict__ rcat, const int32_t* __restrict__ rseg,
    const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const IndelLevParams p) {
  // K = 1: strings <= 64 code units, text in registers; K = 2 / 4: multi-word LCS of indel_wide.hpp,
  // text image in LDS.  LDS layout: [wave][pm_stride * K] masks | (K > 1) [wave][16 K][64] text dwords |
  // [wave][lev_batch(K)][64] running scores
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_pm[];
  constexpr int kRow = kWave * K;  // bytes per string row
  constexpr int kBatch = lev_batch(K);
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
  unsigned long long* pm = s_pm + wave * p.pm_stride * kPmWords<K>;
  uint32_t* wtext = reinterpret_cast<uint32_t*>(s_pm + waves * p.pm_stride * kPmWords<K>) + wave * 16 * K * kWave;

  // K = 1: the lane's text as LDS addresses of its symbols' match masks (wave's PM base + 8 * code),
  // two 16-bit fields per VGPR, so one full-rate v_and / v_lshrrev per symbol yields the address; built
  // once per step and used for every row of the batch
  uint32_t taddr[32];
  const uint32_t pm_base = static_cast<uint32_t>(wave * p.pm_stride * kPmWords<K> * 8);  // s_pm starts at LDS offset 0
  uint32_t lowmask = 0xffffu, sh16 = 16u;  // kept in VGPRs: e32 ops with VGPR operands issue at full rate
  int text_row = -1;
  int lb = 0;
  // running scores of the batch's rows: [row][lane] doubles behind the masks (and the text image)
  double* sc = reinterpret_cast<double*>(reinterpret_cast<uint32_t*>(s_pm + waves * p.pm_stride * kPmWords<K>) +
                                         (K > 1 ? waves * 16 * K * kWave : 0)) +
               wave * kBatch * kWave;

  auto step_ratio = [](int la_, int lb_, int lcs_) -> double {
    if constexpr (K == 1) return (la_ == 0 || lb_ == 0) ? 0.0 : g_ratio64.v[(la_ + lb_) * 65 + lcs_];
    else return indel_score_dev(la_, lb_, lcs_);
  };

  // ---- rows [ib, ib + nrows) against the lanes flagged in okbits (bit r = row ib + r), STEP-MAJOR: the
  // lane's right level string of a step is fetched once for the whole batch (row-major order fetched it
  // per row and step: 3.4 TB of 64-byte reloads per C5 pass, no longer L2-resident at 500k items), and
  // the wave maximum of the text lengths is taken once per step.  `live` (scalar) = rows some lane can
  // still bring over the threshold.
  auto score_batch = [&](int ib, int nrows, uint32_t okbits, uint32_t live) __attribute__((always_inline)) {
    for (int r = 0; r < nrows; ++r) sc[r * kWave + lane] = 0.0;
    int steps_max = 0;
    for (uint32_t rows = live; rows;) {
      const int r = __builtin_ctz(rows);
      rows &= rows - 1;
      steps_max = max(steps_max, max(lnlev[ib + r], lr_max));
    }
    double factor = 1.0;
    for (int s = 1; s <= steps_max && live; ++s) {
      factor *= 0.5;
      // right level of this step (per lane): reload the row only when its index changes
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
      const int nchars = wave_max_i32(okbits ? lb : 0);
      for (uint32_t rows = live; rows;) {
        const int r = __builtin_ctz(rows);
        rows &= rows - 1;
        const int i = ib + r;
        const int ll = lnlev[i];
        if (s > max(ll, lr_max)) {  // every lane has seen all its steps of this row
          live &= ~(1u << r);
          continue;
        }
        const bool active = ((okbits >> r) & 1u) && s <= max(ll, lr);
        // left level (wave-uniform): its match masks
        const int lrow = lfirst[i] + max(0, min(s, ll - 1));
        const int la = llen[lrow];
        wide_build_pm<K>(pm, p.pm_stride, lcodes + static_cast<size_t>(lrow) * kRow, la, lane);
        int lcs;
        if constexpr (K == 1) {
          const int npairs = (nchars + 1) >> 1;
          // opaque per row: otherwise the 64 unpacked addresses are hoisted out of the row loop into 64
          // more VGPRs (3 waves/SIMD instead of 6)
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
        double score = sc[r * kWave + lane];
        if (active) {
          score += step_ratio(la, lb, lcs) * factor;
          sc[r * kWave + lane] = score;
        }
        // exact early exit: the steps still to come add at most factor - 2^-steps < factor (ratios are
        // <= 1); when no lane can reach the threshold any more the row is dropped.  The 1e-9 keeps the
        // test safe under the rounding of the double sum.
        if (!__any(active && (score + factor + 1e-9 >= p.threshold))) live &= ~(1u << r);
      }
    }
    for (int r = 0; r < nrows; ++r) {
      const double score = sc[r * kWave + lane];
      const bool hit = ((okbits >> r) & 1u) && score >= p.threshold;
      emit_hits_wave(hits, p.cap, count, hit, score, lorig[ib + r], jorig);
    }
  };

  // With a category partition both sides are grouped by category: visit the left rows of the categories
  // this wave's lanes stand for, and report a pair in its lowest common category only.  Without one:
  // a single pass over the chunk with the per-lane predicate.
  unsigned long long cats = partitioned ? wave_or_u64(valid ? (1ull << myseg) : 0ull) : 1ull;
  while (cats) {
    const int c = __builtin_ctzll(cats);
    cats &= cats - 1;
    const int a = partitioned ? max(i0, lsegstart[c]) : i0;
    const int b = partitioned ? min(i1, lsegstart[c + 1]) : i1;
    const unsigned long long lower = (1ull << c) - 1ull;
    for (int ib = a; ib < b; ib += kBatch) {
      const int nrows = min(kBatch, b - ib);
      uint32_t okbits = 0, live = 0;
      for (int r = 0; r < nrows; ++r) {
        const uint64_t cl = (p.cat_mode != NSM_CAT_NONE) ? lcat[ib + r] : 0ull;
        bool ok = valid;
        if (partitioned) ok = ok && myseg == c && ((cl & catr & lower) == 0ull);
        else if (p.cat_mode != NSM_CAT_NONE) ok = ok && category_match(cl, catr, p.cat_mode);
        okbits |= ok ? (1u << r) : 0u;
        live |= __any(ok) ? (1u << r) : 0u;
      }
      if (live) score_batch(ib, nrows, okbits, live);
    }
  }
}


}  // namespace nsm

#include "indel_levels_park.hpp"
#include "indel_levels_finish.hpp"
#include "indel_levels_tile.hpp"

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

namespace nsm {

// Split path (one-word strings, histogram bound on, thresholds where few pairs outlive step 1): scan kernel -> survivor queue
// -> finish kernel.  The queue and its control words live in the CALLER's workspace (include/nsm_hip.h); what the library
// owns per (device, caller stream) is a side stream and four events, no device memory: the finish kernel of round k runs on
// the side stream beside the scan of round k + 1 (two queue halves) -- the scan leaves ~40 % of the VALU issue slots and
// nearly all of the HBM bandwidth unused, the finish kernel waits on gathers.  Events order the two streams; the caller's
// stream waits for the last finish before the call's last launches, so the call behaves like any other sequence of launches
// on `stream`.  nsm_release() destroys them.
#ifndef NSM_SPLIT_OVERLAP
#define NSM_SPLIT_OVERLAP 1
#endif
struct SplitSide {
  hipStream_t side = nullptr;
  hipEvent_t scanned[2] = {nullptr, nullptr}, finished[2] = {nullptr, nullptr};
};
constexpr int kSplitMaxRounds = 62;
constexpr int kSplitCtlWords = 2 + kSplitMaxRounds;  // [0] hit counter at the start | [1] overflow flag (int) | [2 ..] queue counter per round
constexpr unsigned long long kSplitCtlBytes = kSplitCtlWords * 8ull;
constexpr unsigned long long kSplitMinWorkspace = 1024;

#ifndef NSM_SPLIT_MIN_THRESHOLD
// (3 x 100k^2 C5-shaped grids, split vs fused, ms: 0.65 112.9 vs 81.9, 0.675 43.1 vs 32.5, 0.7 11.9 vs 18.1, 0.8 11.7 vs 14.3,
// 0.9 5.4 vs 7.4 -- below 0.7 the survivors multiply (5x the hits per 0.025) and lane-per-pair finishing costs more than the
// dense passes it replaces)
#define NSM_SPLIT_MIN_THRESHOLD 0.7
#endif
#ifndef NSM_SPLIT_QUEUE_MAX
#define NSM_SPLIT_QUEUE_MAX (128ull << 20)  // entries per half (1 GB)
#endif

static std::mutex g_side_mu;
static std::map<std::pair<int, void*>, SplitSide> g_sides;

static void destroy_side(SplitSide& w) {
  for (int k = 0; k < 2; ++k) {
    if (w.scanned[k]) (void)hipEventDestroy(w.scanned[k]);
    if (w.finished[k]) (void)hipEventDestroy(w.finished[k]);
    w.scanned[k] = w.finished[k] = nullptr;
  }
  if (w.side) (void)hipStreamDestroy(w.side);
  w.side = nullptr;
}

// The side stream of (current device, stream); created at the first call.  On any failure *out stays empty and the caller
// runs its rounds on `stream` alone (slower by a few per cent, same result): never an error.
static void split_side(void* stream, SplitSide* out) {
  if (!NSM_SPLIT_OVERLAP) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_side_mu);
  auto it = g_sides.find(std::make_pair(dev, stream));
  if (it != g_sides.end()) {
    *out = it->second;
    return;
  }
  SplitSide w;
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);  // (hi = the numerically lowest = greatest priority)
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&w.side, hipStreamNonBlocking, hi);
  for (int k = 0; k < 2 && e == hipSuccess; ++k) {
    e = hipEventCreateWithFlags(&w.scanned[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w.finished[k], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    destroy_side(w);
    return;
  }
  g_sides[std::make_pair(dev, stream)] = w;
  *out = w;
}

// Does this grid take the split path when it is given a workspace, and how many survivors does it expect?  (2 % of the pairs
// visited; a partition visits ~1/16 of the grid or less.  configs[4] measures 0.9 %.)
static bool split_eligible(const nsm_level_items* left, const nsm_str_table* left_strings, const nsm_level_items* right,
                           const nsm_str_table* right_strings, double threshold, uint32_t flags, double* expect) {
  if (!left || !right || !left_strings || !right_strings) return false;
  if (left_strings->stride != 64 || right_strings->stride != 64) return false;
  if (left_strings->alphabet != right_strings->alphabet || left_strings->alphabet < 1) return false;
  if (flags & (NSM_FLAG_PARK | NSM_FLAG_WAVE_WIDE)) return false;
  if (!(flags & NSM_FLAG_PRUNE) || !left_strings->hist || !right_strings->hist) return false;
  if (!(threshold >= NSM_SPLIT_MIN_THRESHOLD) && !(flags & NSM_FLAG_SPLIT)) return false;
  if (flags & NSM_FLAG_TILE) return false;
  if ((left_strings->alphabet + 1 + 7) / 8 * 8 > 64) return false;  // the scan's two 64-entry tables
  if (left->n <= 0 || right->n <= 0 || left->n >= (1 << kQueueRowBits) || right->n >= (1 << kQueueRowBits)) return false;
  *expect = static_cast<double>(left->n) * static_cast<double>(right->n) * 0.02 * (left->seg ? 1.0 / 16 : 1.0);
  return true;
}

}  // namespace nsm

extern "C" uint64_t nsm_indel_levels_workspace_bytes(const nsm_level_items* left, const nsm_str_table* left_strings,
                                                     const nsm_level_items* right, const nsm_str_table* right_strings,
                                                     double threshold, uint32_t flags, double expected_survivors) {
  using namespace nsm;
  double expect = 0.0;
  if (!split_eligible(left, left_strings, right, right_strings, threshold, flags, &expect)) return 0;
  if (expected_survivors > 0.0) expect = expected_survivors;
  const unsigned long long qmax = NSM_SPLIT_QUEUE_MAX;
  const unsigned long long rounds = static_cast<unsigned long long>(expect / static_cast<double>(qmax)) + 1;
  unsigned long long entries = static_cast<unsigned long long>(expect / static_cast<double>(rounds)) + (1ull << 16);
  if (entries > qmax) entries = qmax;
  return kSplitCtlBytes + 2 * entries * 8;
}

extern "C" int nsm_release(void* stream) {
  using namespace nsm;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(g_side_mu);
  auto it = g_sides.find(std::make_pair(dev, stream));
  if (it != g_sides.end()) {
    destroy_side(it->second);
    g_sides.erase(it);
  }
  return 0;
}

extern "C" int nsm_release_all(void) {
  using namespace nsm;
  std::lock_guard<std::mutex> lock(g_side_mu);
  for (auto& kv : g_sides) destroy_side(kv.second);  // (streams and events of another device are destroyed from here just as well)
  g_sides.clear();
  return 0;
}

#ifdef NSM_SCAN_STATS
// variant builds only: copy the one-word scan's work counters to `out[8]` and reset them (synchronises the device)
extern "C" int nsm_debug_scan_stats(unsigned long long* out) {
  unsigned long long zero[8] = {0};
  hipDeviceSynchronize();
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(nsm::g_scan_stats), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(nsm::g_scan_stats), zero, sizeof(zero));
  return static_cast<int>(e);
}
#endif

#ifdef NSM_TILE_STATS
// variant builds only: copy the tile kernel's work counters to `out[16]` and reset them (synchronises the device)
extern "C" int nsm_debug_tile_stats(unsigned long long* out) {
  unsigned long long zero[16] = {0};
  hipDeviceSynchronize();
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(nsm::g_tile_stats), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(nsm::g_tile_stats), zero, sizeof(zero));
  return static_cast<int>(e);
}
#endif

extern "C" int nsm_indel_levels_grid(const nsm_level_items* left, const nsm_str_table* left_strings,
                                     const nsm_level_items* right, const nsm_str_table* right_strings,
                                     double threshold, int32_t category_mode, uint32_t flags, nsm_hit* hits,
                                     uint64_t capacity, unsigned long long* hit_count, void* workspace,
                                     uint64_t workspace_bytes, double expected_survivors, void* stream) {
  using namespace nsm;
  if (!left || !right || !left_strings || !right_strings || !hit_count || (!hits && capacity)) {
    set_error("nsm_indel_levels_grid: null argument");
    return NSM_E_BADARG;
  }
#ifdef NSM_DEFAULT_PARK  // A/B builds (tools/build_variant.sh): the round-2 kernel for multi-word strings
  flags |= NSM_FLAG_PARK;
#endif
  const int stride = left_strings->stride;
  if (stride != right_strings->stride || (stride != 64 && stride != 128 && stride != 256 && stride != 512)) {
    set_error("nsm_indel_levels_grid: stride %d/%d unsupported (both sides 64, 128, 256 or 512 code units)",
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
  if (workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 7u)) {
    set_error("nsm_indel_levels_grid: workspace must be 8-byte aligned");
    return NSM_E_BADARG;
  }
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
  p.pm_stride = ((left_strings->alphabet + 1) + 7) / 8 * 8;  // entries per mask table (the pad symbol included)
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
  if (left->seg && rows > NSM_LEV_CHUNK) rows = NSM_LEV_CHUNK;
  p.rows_per_chunk = static_cast<int>(rows);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (left->n + 65534) / 65535;
    grid.y = (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  const int K = stride / 64;
  const int pm_words = K == 1 ? 1 : K + 1;  // kPmWords<K>
  const size_t lds_wave = p.pm_stride * pm_words * 8 + (K > 1 ? 16 * K * kWave * 4 : 0) + lev_batch(K) * kWave * 8;
  int waves = 4;
  while (waves > 1 && waves * lds_wave > 60 * 1024) waves >>= 1;  // keeps the block under 64 KiB of LDS
  dim3 grid2((n_tiles + waves - 1) / waves, grid.y);
  const size_t lds = static_cast<size_t>(waves) * lds_wave;
#define NSM_LAUNCH_LEVELS(KK)                                                                                  \
  hipLaunchKernelGGL((indel_levels_kernel<KK>), grid2, dim3(waves * kWave), lds, static_cast<hipStream_t>(stream), \
                     left->first, left->nlev, left->orig, left->cat, left->seg_start, left_strings->codes,     \
                     left_strings->len, right->first, right->nlev, right->orig, right->cat, right->seg,         \
                     right_strings->codes, right_strings->len, hits, hit_count, p)
  // (One-word strings stay on the park kernel: its scan is built around texts held in registers as packed LDS
  // addresses and 16-dword folded histograms; the shared-tile kernel instantiated for K = 1 -- with the same packed
  // two-row pass -- ran configs[4]'s fuzzy grids in 660 ms against 418 ms: per left row its H phase and double-precision
  // bookkeeping cost more than the 27-code-unit LCS they guard.)
  // ... at the thresholds it was tuned for.  Where MOST pairs outlive step 1 its finishing passes gather every survivor's
  // level strings again and again (c5w -- configs[4]'s shape on word-like text -- at the cache threshold 0.5: a quarter of
  // the same-category pairs survive step 1, 4.1 TB of HBM traffic per 500k x 500k grid, 651 ms); the shared-tile kernel
  // keeps the tile's level strings of steps 1..3 resident in LDS and carries such rows on wave-wide: 297 ms per grid
  // (fuzzy grids of a step 1953 -> 892 ms).  At 0.6 the picture is the reverse (572 vs 351 ms, split path 218).
#ifndef NSM_TILE_K1_BELOW
#define NSM_TILE_K1_BELOW 0.55
#endif
  // (NSM_FLAG_TILE / NSM_FLAG_SPLIT: a host that has measured the survival rate of step 1 -- NSM_FLAG_PROBE on a sample of
  // the left rows -- routes by it: measured on configs[4]-shaped cohorts, 3 x 100k^2, ms: word-like text at 0.55 / 0.6
  // (2.8 % / 0.09 % survive) split 13.9 / 10.5, fused 20.0 / 16.4, tile 30.9 / 25.3; digit strings at 0.6 / 0.65 / 0.675
  // (most / 60 % / 21 % survive) tile 72.7 / 43.5 / 38.4, fused 123.9 / 81.2 / 33.0, split - / 112.9 / 43.1)
  const bool want_split = (flags & NSM_FLAG_SPLIT) && workspace != nullptr && !(flags & NSM_FLAG_TILE);
  const bool tile_ok = K > 1 || (flags & NSM_FLAG_TILE) || (threshold < NSM_TILE_K1_BELOW && !want_split);
  if (!(flags & NSM_FLAG_WAVE_WIDE) && tile_ok && !(flags & NSM_FLAG_PARK)) {
    // multi-word strings: shared-tile kernel (indel_levels_tile.hpp) -- the waves of a block share one right tile
    // whose level strings stay resident in LDS, and divide the left rows
    TileParams q;
    q.n_left = left->n; q.n_right = right->n; q.cap = capacity;
    q.n_tiles = n_tiles;
    q.n_lstr = left_strings->n > 0 ? left_strings->n : 1;
    q.n_rstr = right_strings->n > 0 ? right_strings->n : 1;
    q.cat_mode = category_mode;
    q.threshold = threshold;
    q.use_hist = ((flags & NSM_FLAG_PRUNE) && left_strings->hist && right_strings->hist) ? 1 : 0;
    // the histogram bound of the step-1 pair (8 more v_sad_u8 per pair) only kills whole rows at high thresholds
    // (term, 20k x 20k: 10.50 -> 9.90 ms at 0.7, 21.2 -> 21.3 at 0.6, 41.6 -> 41.9 at 0.5)
    q.use_h1 = (q.use_hist && threshold >= 0.65) ? 1 : 0;
    q.pm_stride = (left_strings->alphabet + 1 + 7) / 8 * 8;
#ifndef NSM_TILE_PARK_MAX
#define NSM_TILE_PARK_MAX 24
#endif
    q.park_max = NSM_TILE_PARK_MAX;
    const int tile_rows = tile_batch(K);  // left rows per batch = mask tables per wave
    q.park_slots = tile_rows * q.park_max;  // a row parks at most once, <= park_max pairs; drained after every batch
    const size_t tbl_bytes = (static_cast<size_t>(q.pm_stride) * tile_words(K) + kTileTableSkew) * 8;
    const size_t wave_bytes = (tile_rows * tbl_bytes + tile_rows * 3 * kTileHead * 4 + 2 * tile_rows * kWave * K +
                               tile_rows * 2 * 4 + static_cast<size_t>(q.park_slots) * 12 + 15) & ~static_cast<size_t>(15);
    auto block_bytes = [&](int n_img) -> size_t {
      const int n_hist = n_img > 3 ? n_img : 3;
      return static_cast<size_t>(n_img) * 16 * K * kWave * 4 + static_cast<size_t>(n_hist) * (8 * kWave * 4 + kWave * 4) +
             2 * kWave * 4 + 67 * 16;
    };
    // One block per CU (all 160 KB of its LDS): as many waves as fit beside the images, 16 at most.  Three resident
    // images (the level strings of steps 1..3) when at least 12 waves fit with them, else two.
#ifndef NSM_TILE_IMG
#define NSM_TILE_IMG 0
#endif
#ifndef NSM_TILE_WAVES
#define NSM_TILE_WAVES 0
#endif
    constexpr size_t kLdsCu = 160 * 1024;
    auto waves_for = [&](int n_img) -> int {
      const size_t b = block_bytes(n_img);
      if (b + wave_bytes > kLdsCu) return 0;
      const size_t w = (kLdsCu - b) / wave_bytes;
      const size_t most = static_cast<size_t>(tile_max_waves(K));  // (the kernel's launch bound)
      return static_cast<int>(w > most ? most : w);
    };
    // (term, 20k x 20k at 0.5: 12 waves with two images 98 ms, with three 90 ms -- items of 4+ levels read their step-3
    // texts from global memory when only two steps are resident)
    int n_img = NSM_TILE_IMG ? NSM_TILE_IMG
                             : (waves_for(4) >= tile_max_waves(K) ? 4 : waves_for(3) >= (3 * tile_max_waves(K)) / 4 ? 3 : 2);
    int tw = waves_for(n_img);
    if (NSM_TILE_WAVES && tw > NSM_TILE_WAVES) tw = NSM_TILE_WAVES;
    if (tw < 1) {
      set_error("nsm_indel_levels_grid: alphabet %d at stride %d needs %zu bytes of LDS", left_strings->alphabet, stride,
                block_bytes(n_img) + wave_bytes);
      return NSM_E_UNSUPPORTED;
    }
    q.n_img = n_img;
    // left slices: enough blocks to fill the chip a few times over, every wave of a block with a few batches of work
    const long long rows_cat = left->seg ? (left->n + 31) / 32 : left->n;  // (rows a tile visits, roughly)
#ifndef NSM_TILE_ROUNDS
#define NSM_TILE_ROUNDS 10
#endif
    long long slices = (256ll * NSM_TILE_ROUNDS + n_tiles - 1) / n_tiles;
    const long long max_slices = rows_cat / (static_cast<long long>(tw) * tile_rows * 4) + 1;
    if (slices > max_slices) slices = max_slices;
    // ... and left slices that stay in an XCD's 4 MB L2 while its blocks (neighbouring tiles) walk them: <= 4096 rows
    // (heads, histograms and two level strings per row: ~0.2 KB + 2 x 64 K bytes)
#ifndef NSM_TILE_SLICE_ROWS
#define NSM_TILE_SLICE_ROWS 4096
#endif
    if (slices < rows_cat / NSM_TILE_SLICE_ROWS) slices = rows_cat / NSM_TILE_SLICE_ROWS;
    if (slices < 1) slices = 1;
    if (slices > 4096) slices = 4096;
    q.y_slices = static_cast<int>(slices);
    q.rows_per_slice = static_cast<int>(((left->n + slices - 1) / slices + tile_rows - 1) / tile_rows * tile_rows);
    const size_t lds_tile = block_bytes(n_img) + static_cast<size_t>(tw) * wave_bytes;
    const unsigned blocks = static_cast<unsigned>(8ll * ((n_tiles + 7) / 8) * slices);
#define NSM_LAUNCH_TILE(KK)                                                                                        \
  do {                                                                                                             \
    static std::atomic<unsigned long long> attr_devs{0}; /* one bit per device ordinal: the attribute is per device */ \
    int dev_ = 0;                                                                                                  \
    (void)hipGetDevice(&dev_);                                                                                     \
    const unsigned long long bit_ = dev_ >= 0 && dev_ < 64 ? 1ull << dev_ : 0ull;                                  \
    if (!(attr_devs.load(std::memory_order_acquire) & bit_) || !bit_) {                                            \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&indel_levels_tile_kernel<KK>),       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
      if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(indel_levels_tile_kernel)");                  \
      attr_devs.fetch_or(bit_, std::memory_order_release);                                                         \
    }                                                                                                              \
    hipLaunchKernelGGL((indel_levels_tile_kernel<KK>), dim3(blocks), dim3(tw * kWave), lds_tile,                   \
                       static_cast<hipStream_t>(stream), left->first, left->nlev, left->orig, left->cat,           \
                       left->seg_start, left_strings->codes, left_strings->len, left_strings->hist, right->first,  \
                       right->nlev, right->orig, right->cat, right->seg, right_strings->codes, right_strings->len, \
                       right_strings->hist, hits, hit_count, q);                                                   \
  } while (0)
    if (K == 1) NSM_LAUNCH_TILE(1);
    else if (K == 2) NSM_LAUNCH_TILE(2);
    else if (K == 4) NSM_LAUNCH_TILE(4);
    else NSM_LAUNCH_TILE(8);
#undef NSM_LAUNCH_TILE
  } else if (!(flags & NSM_FLAG_WAVE_WIDE)) {
    // scan + park + dense finish (indel_levels_park.hpp)
    ParkParams q;
    q.n_left = left->n; q.n_right = right->n; q.cap = capacity;
    q.cat_mode = category_mode;
    q.threshold = threshold;
    q.use_hist = ((flags & NSM_FLAG_PRUNE) && left_strings->hist && right_strings->hist) ? 1 : 0;
    q.pm_stride = (left_strings->alphabet + 1 + 7) / 8 * 8;
    const int batch = park_batch(K);
#ifndef NSM_PARK_MAX
#define NSM_PARK_MAX 24
#endif
#ifndef NSM_PARK_SLOTS_WIDE
#define NSM_PARK_SLOTS_WIDE 96
#endif
    q.park_slots = K >= 4 ? NSM_PARK_SLOTS_WIDE : 128;  // per region (park_sub(K) regions); 64 overflow on Term-like strings (140 -> 157 ms)
    q.park_max = NSM_PARK_MAX;
    q.xcd_slices = 0;
    q.rows_per_chunk = p.rows_per_chunk;
    const size_t tbl_bytes = static_cast<size_t>(q.pm_stride) * pm_words * 8;
    q.slice_base = 0;
    q.slices_total = 0;
    q.qcap = 0;
    // Split path (one-word strings, bound on, thresholds where few pairs outlive step 1): scan kernel -> survivor queue in
    // the caller's workspace -> finish kernel (indel_levels_finish.hpp), the left slices in rounds sized to the queue.  A
    // queue that overflows anyway (the survival rate is a guess) raises a flag: the hit counter is put back and the fused
    // kernel, launched behind the rounds and gated on that flag, redoes the grid.  No workspace, NSM_FLAG_PARK = the fused
    // kernel alone.
    double expect = 0.0;
    const bool split = workspace != nullptr && workspace_bytes >= kSplitMinWorkspace &&
                       split_eligible(left, left_strings, right, right_strings, threshold, flags, &expect);
    if (expected_survivors > 0.0) expect = expected_survivors;
    const bool probe = (flags & NSM_FLAG_PROBE) != 0;
    if (probe && !split) {
      set_error("nsm_indel_levels_grid: NSM_FLAG_PROBE needs NSM_FLAG_SPLIT, a workspace and a grid the split path takes "
                "(strings up to 64 code units with histograms, NSM_FLAG_PRUNE)");
      return NSM_E_BADARG;
    }
    const size_t fixed_wave = (K > 1 ? 16 * K * kWave * 4 + batch * kWave * 8 : 0) + batch * kWave * 2 +
                              batch * 3 * kHeadDwords * 4 + batch * kWave * K;
    const int sub = park_sub(K);
    const size_t park_bytes = static_cast<size_t>(q.park_slots) * sub * 16 + 66 * 16 + 8 + 4 * sub * 4;
    // one-word text images hold 16-bit LDS addresses: the block stays under 64 KiB (and so do the others)
#ifndef NSM_PARK_LDS_BUDGET_WIDE
#define NSM_PARK_LDS_BUDGET_WIDE (53 * 1024)  // three blocks of two waves per CU (60 KB: two blocks; term 113 -> 93 ms)
#endif
    const size_t budget = K >= 4 ? NSM_PARK_LDS_BUDGET_WIDE : 60 * 1024;
    int pw = 4;  // waves (= right tiles) per block: as many as fit with one mask table each ...
    while (pw > 1 && pw * (tbl_bytes + fixed_wave) + park_bytes > budget) pw >>= 1;
    q.fin_rows = batch;  // ... then as many tables for the dense pass as fit
    while (q.fin_rows > 1 && pw * (q.fin_rows * tbl_bytes + fixed_wave) + park_bytes > budget) --q.fin_rows;
    const size_t park_lds = pw * (q.fin_rows * tbl_bytes + fixed_wave) + park_bytes;
    if (park_lds > 64 * 1024) {
      set_error("nsm_indel_levels_grid: alphabet %d at stride %d needs %zu bytes of LDS", left_strings->alphabet,
                stride, park_lds);
      return NSM_E_UNSUPPORTED;
    }
    dim3 pgrid((n_tiles + pw - 1) / pw, grid.y);
    if (left->seg) {
      // partitioned: y = slices of every category's row range (all blocks have work); ~256 rows per slice when
      // the rows spread over ~32 categories, enough blocks to fill the chip when they do not
#ifndef NSM_PARK_SLICE_ROWS
#define NSM_PARK_SLICE_ROWS 16384  // (configs[4], split path: 4096 / 8192 / 16384 / 32768 rows -> fuzzy grids 294 / 283 / 279 / 282 ms)
#endif
      long long slices = (left->n + NSM_PARK_SLICE_ROWS - 1) / NSM_PARK_SLICE_ROWS;
      if (slices < 1) slices = 1;
      if (slices > 1024) slices = 1024;
      while (slices < 64 && static_cast<long long>(pgrid.x) * slices < 4096) slices *= 2;
      pgrid.y = static_cast<unsigned>(slices);
#ifndef NSM_PARK_XCD
#define NSM_PARK_XCD 1  // (configs[4]: fuzzy grids 419.6 -> 409.1 ms; what it does to the traffic: DESIGN.md section 4.4)
#endif
      if (NSM_PARK_XCD) {  // 1-D grid mapped XCD-aware in the kernel (indel_levels_park.hpp); spare blocks leave at once
        q.xcd_slices = static_cast<int>(slices);
        const long long per_xcd = ((slices + 7) / 8) * (static_cast<long long>(pgrid.x) + 64);
        pgrid.x = static_cast<unsigned>(8 * per_xcd);
        pgrid.y = 1;
      }
    }
    const int* gate = nullptr;
    if (split && pw == 4) {  // (the scan kernel's LDS layout is compiled for four waves; always the case at pm_stride <= 64)
      // rounds: the expected number of survivors against the queue the caller gave (two halves when the finish kernels run
      // on the side stream)
      hipStream_t hs = static_cast<hipStream_t>(stream);
      hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(hs, &capture) != hipSuccess) capture = hipStreamCaptureStatusNone;
      SplitSide ws;
      if (capture == hipStreamCaptureStatusNone) split_side(stream, &ws);  // (a capturing stream stays on its own)
      unsigned long long* ctl = static_cast<unsigned long long*>(workspace);
      unsigned long long* queue = ctl + kSplitCtlWords;
      unsigned long long entries = (workspace_bytes - kSplitCtlBytes) / 16;  // per half (>= 32)
      if (entries > NSM_SPLIT_QUEUE_MAX) entries = NSM_SPLIT_QUEUE_MAX;
      const long long slices_all = left->seg ? static_cast<long long>(q.xcd_slices ? q.xcd_slices : pgrid.y) : pgrid.y;
      long long rounds = static_cast<long long>(expect / static_cast<double>(entries)) + 1;
      if (rounds > kSplitMaxRounds) rounds = kSplitMaxRounds;
      if (rounds > slices_all) rounds = slices_all;
      {
      hipLaunchKernelGGL(split_begin_kernel, dim3(1), dim3(kWave), 0, hs, ctl, kSplitCtlWords, hit_count);
      ParkParams sq = q;
      sq.park_slots = 0;
      sq.fin_rows = 1;
      sq.qcap = entries;
      sq.slices_total = static_cast<int>(slices_all);
      const size_t scan_lds = pw * (2 * kSplitTableBytes + fixed_wave + kQueueBuf * 8) + 66 * 16 + 8 + 4 * sub * 4;
      FinishParams fp;
      fp.pm_stride = q.pm_stride;
      fp.pad_code = left_strings->alphabet;
      fp.use_hist = q.use_hist;
      fp.threshold = threshold;
      fp.cap = capacity;
      fp.qcap = entries;
      const long long per_round = (slices_all + rounds - 1) / rounds;
      int* qflag = reinterpret_cast<int*>(ctl + 1);
      long long n_rounds = 0;
      for (long long rd = 0; rd * per_round < slices_all; ++rd, ++n_rounds) {
        const long long s0 = rd * per_round;
        const long long ns = slices_all - s0 < per_round ? slices_all - s0 : per_round;
        sq.slice_base = static_cast<int>(s0);
        dim3 sgrid = pgrid;
        if (left->seg && q.xcd_slices) {
          sq.xcd_slices = static_cast<int>(ns);
          const long long per_xcd = ((ns + 7) / 8) * (static_cast<long long>((n_tiles + pw - 1) / pw) + 64);
          sgrid = dim3(static_cast<unsigned>(8 * per_xcd), 1);
        } else {
          sgrid.y = static_cast<unsigned>(ns);
        }
        const int half = ws.side ? static_cast<int>(rd & 1) : 0;
        unsigned long long* qhalf = queue + static_cast<size_t>(half) * entries;
        hipStream_t fs = hs;
        if (ws.side) {
          fs = ws.side;
          if (rd >= 2) {  // the finish kernel of round rd - 2 has read this half
            const hipError_t e = hipStreamWaitEvent(hs, ws.finished[half], 0);
            if (e != hipSuccess) return hip_status(e, "hipStreamWaitEvent(finished)");
          }
        }
        hipLaunchKernelGGL((indel_levels_park_kernel<1, true>), sgrid, dim3(pw * kWave), scan_lds, hs, left->first,
                           left->nlev, left->orig, left->cat, left->seg_start, left_strings->codes, left_strings->len,
                           left_strings->hist, right->first, right->nlev, right->orig, right->cat, right->seg,
                           right_strings->codes, right_strings->len, right_strings->hist, hits, hit_count, sq,
                           right->seg_start, qhalf, ctl + 2 + rd, qflag, static_cast<const int*>(nullptr));
        if (ws.side) {
          hipError_t e = hipEventRecord(ws.scanned[half], hs);
          if (e == hipSuccess) e = hipStreamWaitEvent(fs, ws.scanned[half], 0);
          if (e != hipSuccess) return hip_status(e, "split path: scan -> finish ordering");
        }
        if (!probe) {  // (NSM_FLAG_PROBE: the scan's queue counters are all the caller wants)
          hipLaunchKernelGGL(indel_levels_finish_kernel, dim3(kFinishBlocks), dim3(kWave),
                             static_cast<size_t>(fp.pm_stride) * 2 * kWave * 4, fs, left->first, left->nlev, left->orig,
                             left_strings->codes, left_strings->len, left_strings->hist, right->first, right->nlev,
                             right->orig, right_strings->codes, right_strings->len, right_strings->hist, hits, hit_count,
                             qhalf, ctl + 2 + rd, qflag, fp);
        }
        if (ws.side) {
          const hipError_t e = hipEventRecord(ws.finished[half], fs);
          if (e != hipSuccess) return hip_status(e, "hipEventRecord(finished)");
        }
      }
      if (ws.side) {  // join: the last (two) finish kernels before the counter is looked at
        for (int k = 0; k < 2 && k < n_rounds; ++k) {
          const hipError_t e = hipStreamWaitEvent(hs, ws.finished[k], 0);
          if (e != hipSuccess) return hip_status(e, "hipStreamWaitEvent(join)");
        }
      }
      if (probe) return hip_status(hipGetLastError(), "nsm_indel_levels_grid (probe)");  // the counters are the result
      hipLaunchKernelGGL(split_end_kernel, dim3(1), dim3(kWave), 0, hs, ctl, hit_count);
      gate = qflag;
      }
    }
    unsigned long long* no_queue = nullptr;
    int* no_flag = nullptr;
#define NSM_LAUNCH_PARK(KK)                                                                                       \
  hipLaunchKernelGGL((indel_levels_park_kernel<KK>), pgrid, dim3(pw * kWave), park_lds,                          \
                     static_cast<hipStream_t>(stream), left->first, left->nlev, left->orig, left->cat,           \
                     left->seg_start, left_strings->codes, left_strings->len, left_strings->hist, right->first,  \
                     right->nlev, right->orig, right->cat, right->seg, right_strings->codes, right_strings->len, \
                     right_strings->hist, hits, hit_count, q, right->seg_start, no_queue, no_queue, no_flag, gate)
    if (K == 1) NSM_LAUNCH_PARK(1);
    else if (K == 2) NSM_LAUNCH_PARK(2);
    else if (K == 4) NSM_LAUNCH_PARK(4);
    else NSM_LAUNCH_PARK(8);
#undef NSM_LAUNCH_PARK
  } else if (K == 1) NSM_LAUNCH_LEVELS(1);
  else if (K == 2) NSM_LAUNCH_LEVELS(2);
  else if (K == 4) NSM_LAUNCH_LEVELS(4);
  else NSM_LAUNCH_LEVELS(8);
#undef NSM_LAUNCH_LEVELS
  return hip_status(hipGetLastError(), "indel_levels_kernel launch");
}
