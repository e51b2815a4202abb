// RAW Jaccard through the right table's GLOBAL inverted index: candidate pairs instead of all N x M
// (reference: napkon_string_matching/compare/score_functions.py:6-13; the loop around it visits every pair).
//
// Prefix filter (the all-pairs similarity-join principle).  Every row keeps its ids in one global order (ascending id;
// the builder sorts).  A pair of sets of a and b ids is a hit iff |A n B| >= kmin[a + b] (the launcher's table, computed
// with the reference's double division).  Let o(a) = the smallest kmin[a + b] over all sizes b that can reach it at all:
// a hit shares at least o(a) ids with A, so its SMALLEST common id w sits among the first P(a) = a - o(a) + 1 ids of A,
// and likewise among the first P(b) ids of B.  So: probe the posting lists of the first P(a) ids of every left row,
// restricted to the entries whose position in the right row is < P(b), and every hit is found -- exactly once, at its
// smallest common id, which is how duplicates are dropped (a candidate whose first common id is not the probed one is
// somebody else's).  C4 (1M x 1M sets of ~8 of 2^17 ids, threshold 0.8): P = 2, ~60 entries per probed list, of which the
// positional bound keeps a handful -- 10^7 candidates instead of 10^12 pairs.
//
// Mapping.  A wavefront takes batches of 64 / S left rows (S = the longest prefix, a power of two): lane = (row, prefix
// slot) looks up its id's posting range; a wave-wide scan of the range lengths turns the batch into ONE flat list of
// candidates that the 64 lanes walk together, 64 per pass, whatever rows they belong to (lane = one posting; the row is
// found by a 6-step binary search over the batch's offsets in LDS).  Per candidate, cheapest test first:
//   entry alone (position, size): right prefix, size filter kmin <= min(a, b), positional bound 1 + min(a - pa, b - pb) - 1
//   -> gather the right row's signature word: |A n B| <= popcount(sigA & sigB) + cA  (second word likewise)
//   -> merge the two sorted rows (lane-local): the count, and the first common position.
// Hits leave through emit_hits_wave (one atomic per wavefront).
#include "jaccard_raw_impl.hpp"

namespace nsm {

template <int W>
struct JacGlobalParams {
  int32_t n_left;
  int32_t n_right;
  int32_t vocab;
  int32_t slot_shift;      // log2 S
  int32_t rows_per_batch;  // 64 >> slot_shift
  int32_t n_batches;
  int32_t cls_end;         // which of a list's five boundaries ends the useful entries (1..5)
  int32_t row_bits;        // 0: 64-bit posting entries; else 32-bit entries with this many row bits (nsm_hip.h: post)
  unsigned long long cap;
  uint8_t kmin[2 * W + 4];  // indexed by |A| + |B|
  uint8_t prefix[W + 4];    // P(size)
};

// 27-bit fold of a signature word's 58 hash bits (nsm_hip.h: sig): bit j = OR of the hash bits j, j + 27 and j + 54, i.e. the
// signature of the same ids under (hash mod 27).  With c' = size - popcount(fold) (ids that share a bit inside their row),
// |A n B| <= popcount(foldA & foldB) + min(c'A, c'B).  A fold that is all ones says nothing: its c' counts as the width.
__host__ __device__ inline uint32_t sig_fold27(uint64_t sig) {
  constexpr uint32_t kM = (1u << 27) - 1u;
  return (static_cast<uint32_t>(sig) & kM) | (static_cast<uint32_t>(sig >> 27) & kM) | (static_cast<uint32_t>(sig >> 54) & 0xFu);
}

// FORMAT of the right table's posting entries (nsm_hip.h: post_format): 0 = 64 bits, row | position << 32 | size << 40;
// 1 = 32 bits, row | position << row_bits | (size - 1) << (row_bits + log2 W); 2 = 64 bits, the 32-bit entry with the fold of
// the row's signature word above it -- the entry alone then decides most candidates, and the 8-byte signature gather
// (a 64-byte sector each: two thirds of configs[3]'s HBM traffic) is left to the few that pass
#ifndef NSM_GLOBAL_PREFETCH
#define NSM_GLOBAL_PREFETCH 1
#endif
template <int W, int FORMAT>
__global__ __launch_bounds__(kBlock) void jaccard_raw_global_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const uint64_t* __restrict__ lsig,
    const uint64_t* __restrict__ lsig2, const int32_t* __restrict__ lorig, const int32_t* __restrict__ rids,
    const uint64_t* __restrict__ rsig, const uint64_t* __restrict__ rsig2, const int32_t* __restrict__ rorig,
    const unsigned long long* __restrict__ post, const int32_t* __restrict__ post_start, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const JacGlobalParams<W> p) {
  __shared__ uint8_t s_kmin[2 * W + 4];
  __shared__ uint8_t s_prefix[W + 4];
  __shared__ int s_off[kWavesPerBlock][kWave];
  __shared__ int s_start[kWavesPerBlock][kWave];
  __shared__ int s_row[kWavesPerBlock][kWave];
  __shared__ int s_meta[kWavesPerBlock][kWave];  // probed position | size << 8
  __shared__ unsigned long long s_sig[kWavesPerBlock][kWave];
  __shared__ unsigned long long s_sig2[kWavesPerBlock][kWave];
  for (int t = threadIdx.x; t < 2 * W + 4; t += kBlock) s_kmin[t] = p.kmin[t];
  for (int t = threadIdx.x; t < W + 4; t += kBlock) s_prefix[t] = p.prefix[t];
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int first_wave = blockIdx.x * kWavesPerBlock + wave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  constexpr uint64_t kCollBits = ~((1ull << 58) - 1);  // the top 6 bits of a signature word hold cA (unary)
  const int sub = lane >> p.slot_shift;                // row of the batch
  const int slot = lane & ((1 << p.slot_shift) - 1);   // prefix position probed by this lane

  for (int batch = first_wave; batch < p.n_batches; batch += n_waves) {
    // ---- lane = (row, prefix slot): the id's posting range
    const int row = batch * p.rows_per_batch + sub;
    int len = 0, start = 0, a = 0;
    unsigned long long sg = 0ull, sg2 = ~0ull;
    if (sub < p.rows_per_batch && row < p.n_left) {
      a = lcnt[row];
      if (slot < s_prefix[a]) {
        const int tok = lids[static_cast<size_t>(row) * W + slot];
        if (tok < p.vocab) {  // (an id the right side's vocabulary does not reach has no postings)
          const long long at = 5ll * tok;
          start = post_start[at];
          len = post_start[at + p.cls_end] - start;
        }
        if (len > 0) {
          sg = lsig[row];
          if (lsig2 != nullptr) sg2 = lsig2[row];
        }
      }
    }
    int incl = len;  // wave-wide inclusive scan
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    const int total = __builtin_amdgcn_readlane(incl, kWave - 1);
    if (total == 0) continue;
    s_off[wave][lane] = incl - len;
    s_start[wave][lane] = start;
    s_row[wave][lane] = row;
    s_meta[wave][lane] = slot | (a << 8);
    s_sig[wave][lane] = sg;
    s_sig2[wave][lane] = sg2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- lane = one candidate of the batch's flat list.  The kernel lives on the latency of the posting loads (one dependent
    // round trip per 64 candidates, 8 waves per SIMD to hide it): the NEXT 64 entries are requested before this chunk's are
    // looked at.
    auto fetch = [&](int base, int& seg, bool& live) -> unsigned long long {
      const int idx = base + lane;
      live = idx < total;
      seg = 0;  // the last segment that starts at or before idx (empty segments share their successor's offset)
#pragma unroll
      for (int step = kWave / 2; step > 0; step >>= 1) {
        const int mid = seg + step;
        if (s_off[wave][mid] <= idx) seg = mid;
      }
      const size_t at = static_cast<size_t>(s_start[wave][seg]) + (idx - s_off[wave][seg]);
      unsigned long long raw = 0ull;
      if (live) {
        if constexpr (FORMAT == 1) raw = reinterpret_cast<const uint32_t*>(post)[at];
        else raw = post[at];
      }
      return raw;
    };
    int seg_n = 0;
    bool live_n = false;
    unsigned long long raw_n = fetch(0, seg_n, live_n);
    for (int base = 0; base < total; base += kWave) {
#if !NSM_GLOBAL_PREFETCH  // (A/B builds)
      if (base > 0) raw_n = fetch(base, seg_n, live_n);
#endif
      const int seg = seg_n;
      const bool live = live_n;
      const unsigned long long raw = raw_n;
#if NSM_GLOBAL_PREFETCH
      if (base + kWave < total) raw_n = fetch(base + kWave, seg_n, live_n);
#endif
      // the posting entry: right row, position of the probed id in it, its size
      int rrow, pb, b;
      uint32_t fold_b = 0u;
      if constexpr (FORMAT != 0) {
        constexpr int kLogW = W == 16 ? 4 : W == 32 ? 5 : 6;
        const uint32_t e = static_cast<uint32_t>(raw);
        fold_b = static_cast<uint32_t>(raw >> 32);
        rrow = static_cast<int>(e & ((1u << p.row_bits) - 1u));
        pb = static_cast<int>((e >> p.row_bits) & (W - 1));
        b = live ? static_cast<int>((e >> (p.row_bits + kLogW)) & (W - 1)) + 1 : 0;
      } else {
        rrow = static_cast<int>(static_cast<uint32_t>(raw));
        pb = static_cast<int>((raw >> 32) & 0xffu);
        b = static_cast<int>((raw >> 40) & 0xffu);
      }
      const int meta = s_meta[wave][seg];
      const int pa = meta & 0xff, la = meta >> 8;
      const int need = s_kmin[la + b];
      // entry alone: right prefix, size filter, positional bound (the probed id as the FIRST common one: at most
      // 1 + what follows it on either side can be common)
      bool ok = live && pb < s_prefix[b] && need <= min(la, b) && 1 + min(la - pa - 1, b - pb - 1) >= need;
      if constexpr (FORMAT == 2) {  // the folded signatures of both rows: no memory access
        constexpr uint32_t kM = (1u << 27) - 1u;
        const uint32_t fold_a = sig_fold27(s_sig[wave][seg]);
        const int ca = fold_a == kM ? W : la - __popc(fold_a), cb = fold_b == kM ? W : b - __popc(fold_b);
        ok = ok && __popc(fold_a & fold_b) + min(ca, cb) >= need;
      }
      if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      if (ok) {
        const uint64_t sr = rsig[rrow] | kCollBits;
        ok = __popcll(s_sig[wave][seg] & sr) >= need;
      }
      if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      if (rsig2 != nullptr && lsig2 != nullptr) {
        if (ok) {
          const uint64_t sr2 = rsig2[rrow] | kCollBits;
          ok = __popcll(s_sig2[wave][seg] & sr2) >= need;
        }
        if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      }
      // exact: merge the two ascending rows
      int k = 0, first = -1;
      const int lrow = s_row[wave][seg];
      if (ok) {
        const int32_t* __restrict__ lp = lids + static_cast<size_t>(lrow) * W;
        const int32_t* __restrict__ rp = rids + static_cast<size_t>(rrow) * W;
        int x = 0, y = 0;
        int32_t lv = lp[0], rv = rp[0];
        while (x < la && y < b) {
          if (lv == rv) {
            if (first < 0) first = x;
            ++k; ++x; ++y;
            lv = lp[min(x, W - 1)];
            rv = rp[min(y, W - 1)];
          } else if (lv < rv) {
            ++x;
            lv = lp[min(x, W - 1)];
          } else {
            ++y;
            rv = rp[min(y, W - 1)];
          }
        }
      }
      const bool hit = ok && k >= need && first == pa;
      const double score = hit ? static_cast<double>(k) / static_cast<double>(la + b - k) : 0.0;
      emit_hits_wave(hits, p.cap, count, hit, score, hit ? lorig[lrow] : 0, hit ? rorig[rrow] : 0);
    }
    __builtin_amdgcn_wave_barrier();  // (the next batch overwrites the wave's LDS rows)
  }
}

// P(size) = size - o(size) + 1, o(size) = the least kmin[size + b] over the partner sizes b that can reach it
template <int W>
static int fill_prefix(const uint8_t* kmin, uint8_t* prefix) {
  int longest = 0;
  for (int a = 0; a < W + 4; ++a) {
    prefix[a] = 0;
    if (a < 1 || a > W) continue;
    int o = 255;
    for (int b = 1; b <= W; ++b) {
      const int need = kmin[a + b];
      if (need == kNever || need < 1 || need > (a < b ? a : b)) continue;
      if (need < o) o = need;
    }
    if (o == 255) continue;  // no partner size reaches the threshold with a set of this size
    prefix[a] = static_cast<uint8_t>(a - o + 1);
    if (prefix[a] > longest) longest = prefix[a];
  }
  return longest;
}

// Estimated candidates (entries the probes visit) and the launch.  `probe_only`: return the estimate, launch nothing.
template <int W>
int launch_raw_global(const nsm_set_table* l, const nsm_set_table* r, double threshold, nsm_hit* hits, uint64_t capacity,
                      unsigned long long* hit_count, hipStream_t stream, bool probe_only, double* estimate) {
  JacGlobalParams<W> p;
  p.n_left = l->n; p.n_right = r->n; p.vocab = r->vocab; p.cap = capacity;
  if (int rc = check_post_format(r, "nsm_jaccard_raw_grid")) return rc;
  p.row_bits = r->post_row_bits;
  fill_kmin<W>(p.kmin, threshold);
  for (int s = 0; s < 2 * W + 4; ++s)
    if (p.kmin[s] == 0) {  // a threshold <= 0: every pair hits, no index can help
      if (estimate) *estimate = static_cast<double>(l->n) * static_cast<double>(r->n);
      return probe_only ? 0 : NSM_E_UNSUPPORTED;
    }
  const int longest = fill_prefix<W>(p.kmin, p.prefix);
  p.cls_end = longest <= 1 ? 1 : longest <= 2 ? 2 : longest <= 4 ? 3 : longest <= 8 ? 4 : 5;
  int shift = 0;
  while ((1 << shift) < longest) ++shift;
  p.slot_shift = shift;
  p.rows_per_batch = kWave >> shift;
  // ... but a small left table in few batches leaves most of the chip idle (50k rows at 32 rows per batch: 1563 batches for
  // 8192 wave slots -- vocabulary 500 at threshold 0.9: 0.32 ms); fewer rows per batch then, the surplus lanes only idle
  // in the lookup phase
  constexpr int kWaveSlots = 256 * 8 * kWavesPerBlock;
  while (p.rows_per_batch > 1 && (l->n + p.rows_per_batch - 1) / p.rows_per_batch < kWaveSlots) p.rows_per_batch >>= 1;
  p.n_batches = (l->n + p.rows_per_batch - 1) / p.rows_per_batch;
  if (estimate) {
    // entries visited ~ sum over ids of (left probes of the id) x (right entries of the id in the useful classes); the left
    // table carries no statistics of its own, so its probes are taken to spread like the right side's entries
    const double sq = static_cast<double>(r->post_sq[p.cls_end - 1]);
    *estimate = r->n > 0 ? sq * static_cast<double>(l->n) / static_cast<double>(r->n) : 0.0;
  }
  if (probe_only || longest == 0) return 0;  // (longest == 0: no pair of sizes can reach the threshold)
  long long blocks = (p.n_batches + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 256 * 8) blocks = 256 * 8;
#define NSM_LAUNCH_GLOBAL(F)                                                                                                  \
  hipLaunchKernelGGL((jaccard_raw_global_kernel<W, F>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, l->ids, \
                     l->cnt, l->sig, (l->sig2 && r->sig2) ? l->sig2 : nullptr, l->orig, r->ids, r->sig, r->sig2, r->orig,     \
                     reinterpret_cast<const unsigned long long*>(r->post), r->post_start, hits, hit_count, p)
  if (r->post_format == 2) NSM_LAUNCH_GLOBAL(2);
  else if (r->post_format == 1) NSM_LAUNCH_GLOBAL(1);
  else NSM_LAUNCH_GLOBAL(0);
#undef NSM_LAUNCH_GLOBAL
  return hip_status(hipGetLastError(), "jaccard_raw_global_kernel launch");
}

template int launch_raw_global<16>(const nsm_set_table*, const nsm_set_table*, double, nsm_hit*, uint64_t, unsigned long long*,
                                   hipStream_t, bool, double*);
template int launch_raw_global<32>(const nsm_set_table*, const nsm_set_table*, double, nsm_hit*, uint64_t, unsigned long long*,
                                   hipStream_t, bool, double*);
template int launch_raw_global<64>(const nsm_set_table*, const nsm_set_table*, double, nsm_hit*, uint64_t, unsigned long long*,
                                   hipStream_t, bool, double*);

}  // namespace nsm
