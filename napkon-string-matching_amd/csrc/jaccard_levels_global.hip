// Levels-mode Jaccard grid through the right table's GLOBAL inverted index (reference: types/comparable_data.py:223-232 ->
// compare_terms :248-265 x intersection_vs_union, compare/score_functions.py:6-13; category predicate :464-490).
//
// Every step's quotient is |A_s n B_s| / |A_s u B_s|, so a pair without a common id scores exactly 0: with a positive
// threshold only pairs that SHARE an id can hit.  The per-tile index kernel (jaccard_levels_index.hip) finds them by
// rebuilding a hash table per right tile and streaming every left row past it -- N x M / 64 probes whatever the answer.
// Here the index is built ONCE with the table (nsm_build_set_table: postings sorted by (category segment, id), so a probe
// only sees right rows of the left row's category), and the grid walks posting lists: the cost follows the number of
// (left row, right row, common id) triples, not N x M.  configs[4]'s Jaccard grids (500k x 500k, 20 000 words, 32
// categories): ~50 candidates per left row instead of 31 000 same-category rows.
//
// Mapping (as jaccard_raw_global.hip): a wavefront takes batches of 64 / W left rows; lane = (row, id slot) looks up the
// posting range of (the row's category segment, its id); a wave-wide scan makes the batch ONE flat candidate list, walked
// 64 per pass (lane = one posting; a 6-step binary search over the batch's offsets in LDS finds its row and slot).  Per
// candidate: gather the right row's 32-byte filter record -> category predicate (partition: no LOWER common category) and
// the joint step-1 bound of jaccard_levels_impl.hpp with m = max(|A_1|, |B_1|) known exactly -> gather both rows, position
// matrix (xor / min3), per-level intersections by byte-parallel compare, double quotients from an LDS table of real IEEE
// divisions, the reference's weights and order.  A pair that shares several ids is scored once: at the FIRST left slot
// that has a match.  Hits are identical to jaccard_levels_kernel's and the oracle's.
#include "jaccard_levels_impl.hpp"

namespace nsm {

template <int W>
struct JacLevGlobalParams {
  int32_t n_left;
  int32_t n_right;
  int32_t vocab;
  int32_t partitioned;
  int32_t n_batches;
  int32_t lev_stride_l;
  int32_t lev_stride_r;
  int32_t cat_mode;
  int32_t row_bits;  // 0: 64-bit posting entries; else 32-bit entries with this many row bits (nsm_hip.h: post)
  double threshold;
  unsigned long long cap;
};

// FORMAT of the right table's posting entries (nsm_hip.h: post_format): 0 = 64 bits (the row in the low word), 1 = 32 bits and
// 2 = 64 bits with the row in the low row_bits (this kernel reads nothing else of an entry)
template <int W, int FORMAT>
__global__ __launch_bounds__(kBlock) void jaccard_levels_global_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const uint8_t* __restrict__ lplen,
    const uint32_t* __restrict__ lfilt, const int32_t* __restrict__ lorig, const int32_t* __restrict__ lseg,
    const int32_t* __restrict__ rids, const uint8_t* __restrict__ rplen, const uint32_t* __restrict__ rfilt,
    const int32_t* __restrict__ rorig, const unsigned long long* __restrict__ post, const int32_t* __restrict__ post_start,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const JacLevGlobalParams<W> p) {
  constexpr int kRows = kWave / W;  // left rows per batch
  constexpr int kSlotShift = W == 16 ? 4 : W == 32 ? 5 : 6;
  __shared__ double s_quot[kQuotTable<W> ? (W + 1) * (2 * W + 1) : 1];
  __shared__ int s_off[kWavesPerBlock][kWave];
  __shared__ int s_start[kWavesPerBlock][kWave];
  __shared__ uint32_t s_lf[kWavesPerBlock][kRows][8];  // the batch rows' filter records; [7] = the row's category segment
  __shared__ int s_lrow[kWavesPerBlock][kRows];
  if constexpr (kQuotTable<W>) {
    for (int t = threadIdx.x; t < (W + 1) * (2 * W + 1); t += kBlock) {
      const int k = t / (2 * W + 1), u = t % (2 * W + 1);
      s_quot[t] = u ? static_cast<double>(k) / static_cast<double>(u) : 0.0;  // real IEEE divisions, as the reference's `/`
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int first_wave = blockIdx.x * kWavesPerBlock + wave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  constexpr uint64_t kCollBits = ~((1ull << 58) - 1);
  const int sub = lane >> kSlotShift, slot = lane & (W - 1);
  const bool use_cat = p.cat_mode != NSM_CAT_NONE;

  for (int batch = first_wave; batch < p.n_batches; batch += n_waves) {
    const int row = batch * kRows + sub;
    int len = 0, start = 0;
    if (row < p.n_left) {
      const int a = lcnt[row];
      const int seg = p.partitioned ? lseg[row] : 0;
      if (slot < a) {
        const int tok = lids[static_cast<size_t>(row) * W + slot];
        if (tok < p.vocab) {
          const long long at = 5ll * (static_cast<long long>(seg) * p.vocab + tok);
          start = post_start[at];
          len = post_start[at + 5] - start;
        }
      }
      if (slot < 7) s_lf[wave][sub][slot] = lfilt[static_cast<size_t>(row) * 8 + slot];
      if (slot == 7) {
        s_lf[wave][sub][7] = static_cast<uint32_t>(seg);
        s_lrow[wave][sub] = row;
      }
    }
    int incl = len;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    const int total = __builtin_amdgcn_readlane(incl, kWave - 1);
    if (total == 0) continue;
    s_off[wave][lane] = incl - len;
    s_start[wave][lane] = start;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int base = 0; base < total; base += kWave) {
      const int idx = base + lane;
      const bool live = idx < total;
      int seg = 0;
#pragma unroll
      for (int step = kWave / 2; step > 0; step >>= 1) {
        const int mid = seg + step;
        if (s_off[wave][mid] <= idx) seg = mid;
      }
      int rrow = 0;  // the posting entry's right row (position and size are not used here)
      if constexpr (FORMAT == 1) {
        uint32_t e = 0u;
        if (live) e = reinterpret_cast<const uint32_t*>(post)[static_cast<size_t>(s_start[wave][seg]) + (idx - s_off[wave][seg])];
        rrow = static_cast<int>(e & ((1u << p.row_bits) - 1u));
      } else if constexpr (FORMAT == 2) {
        unsigned long long entry = 0ull;
        if (live) entry = post[static_cast<size_t>(s_start[wave][seg]) + (idx - s_off[wave][seg])];
        rrow = static_cast<int>(static_cast<uint32_t>(entry) & ((1u << p.row_bits) - 1u));
      } else {
        unsigned long long entry = 0ull;
        if (live) entry = post[static_cast<size_t>(s_start[wave][seg]) + (idx - s_off[wave][seg])];
        rrow = static_cast<int>(static_cast<uint32_t>(entry));
      }
      const int lsub = seg >> kSlotShift, pa = seg & (W - 1);
      const uint32_t* lf = s_lf[wave][lsub];
      // ---- filter on the two 32-byte records
      uint32_t rf[8];
      {
        const uint4* rp = reinterpret_cast<const uint4*>(rfilt + static_cast<size_t>(live ? rrow : 0) * 8);
        const uint4 v0 = rp[0], v1 = rp[1];
        rf[0] = v0.x; rf[1] = v0.y; rf[2] = v0.z; rf[3] = v0.w;
        rf[4] = v1.x; rf[5] = v1.y; rf[6] = v1.z; rf[7] = v1.w;
      }
      const uint64_t sl = (static_cast<uint64_t>(lf[1]) << 32) | lf[0], cl = (static_cast<uint64_t>(lf[3]) << 32) | lf[2];
      const uint64_t sl1 = (static_cast<uint64_t>(lf[6]) << 32) | lf[5];
      const uint64_t sr = ((static_cast<uint64_t>(rf[1]) << 32) | rf[0]) | kCollBits, cr = (static_cast<uint64_t>(rf[3]) << 32) | rf[2];
      const uint64_t sr1 = ((static_cast<uint64_t>(rf[6]) << 32) | rf[5]) | kCollBits;
      const int pl1 = static_cast<int>(lf[4] & 0xffu), nl = static_cast<int>((lf[4] >> 8) & 0xffu), ll = static_cast<int>(lf[4] >> 16);
      const int pr1 = static_cast<int>(rf[4] & 0xffu), nr = static_cast<int>((rf[4] >> 8) & 0xffu), lr = static_cast<int>(rf[4] >> 16);
      // score <= (|A_1 n B_1| + min(|A n B|, m)) / (2 m) with m = max(|A_1|, |B_1|)  (jaccard_levels_impl.hpp); the factor
      // (1 - 1e-9) keeps the test necessary under the rounding of the double accumulation
      const int m = max(pl1, pr1);
      const int need = static_cast<int>(ceil(2.0 * p.threshold * static_cast<double>(m) * (1.0 - 1e-9)));
      bool ok = live && min(__popcll(sl & sr), m) + __popcll(sl1 & sr1) >= need;
      if (p.partitioned) {
        const int c = static_cast<int>(lf[7]);  // both rows stand for category c: report the pair in its LOWEST common category
        ok = ok && ((cl & cr & ((1ull << c) - 1ull)) == 0ull);
      } else if (use_cat) {
        ok = ok && category_match(cl, cr, p.cat_mode);
      }
      if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      // ---- exact score
      const int lrow = s_lrow[wave][lsub];
      const size_t lbase = static_cast<size_t>(ok ? lrow : 0) * W, rbase = static_cast<size_t>(ok ? rrow : 0) * W;
      uint32_t l[W], r[W];
#pragma unroll
      for (int q = 0; q < W / 4; ++q) {
        const uint4 a4 = reinterpret_cast<const uint4*>(lids + lbase)[q];
        const uint4 b4 = reinterpret_cast<const uint4*>(rids + rbase)[q];
        l[4 * q + 0] = a4.x << 6; l[4 * q + 1] = a4.y << 6; l[4 * q + 2] = a4.z << 6; l[4 * q + 3] = a4.w << 6;
        r[4 * q + 0] = (b4.x << 6) | (4 * q + 0); r[4 * q + 1] = (b4.y << 6) | (4 * q + 1);
        r[4 * q + 2] = (b4.z << 6) | (4 * q + 2); r[4 * q + 3] = (b4.w << 6) | (4 * q + 3);
      }
      // (left padding is -1, right padding -2: (-1 << 6) ^ ((-2 << 6) | pos) >= 64, padding never matches)
      const int nl_max = wave_max_i32(ok ? nl : 0), nr_max = wave_max_i32(ok ? nr : 0);
      uint32_t posw[W / 4];
      int first = W;  // first left slot with a match
#pragma unroll
      for (int q = 0; q < W / 4; ++q) {
        posw[q] = 0xffffffffu;
        if (4 * q < nl_max) {  // wave-uniform
          uint32_t word = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t la = l[4 * q + e];
            uint32_t mm = lev_min3u(la ^ r[0], la ^ r[1], 255u);
#pragma unroll
            for (int b = 2; b < W; b += 2)
              if (b < nr_max) mm = lev_min3u(mm, la ^ r[b], la ^ r[b + 1]);  // wave-uniform
            word |= mm << (8 * e);
            if (mm < 64u && first == W) first = 4 * q + e;
          }
          posw[q] = word;
        }
      }
      double score = 0.0;
      if (ok && first == pa) {
        const uint8_t* __restrict__ lpl = lplen + static_cast<size_t>(lrow) * p.lev_stride_l;
        const uint8_t* __restrict__ rpl = rplen + static_cast<size_t>(rrow) * p.lev_stride_r;
        const int steps = max(ll, lr);
        double factor = 1.0;
        for (int s = 1; s <= steps; ++s) {
          const int pl = lpl[min(s, p.lev_stride_l - 1)];  // = plen[min(s, L - 1)]: rows are padded with their last value
          const int pr = rpl[min(s, p.lev_stride_r - 1)];
          const uint32_t prrep = static_cast<uint32_t>(pr) * 0x01010101u;
          int inter = 0;
#pragma unroll
          for (int q = 0; q < W / 4; ++q) {
            if (4 * q < pl) {
              uint32_t x = posw[q];
              const int keep = pl - 4 * q;
              if (keep < 4) x |= 0xffffffffu << (8 * keep);
              const uint32_t y = (x | 0x80808080u) - prrep;  // per byte: x < pr
              inter += __popc(~(y | x) & 0x80808080u);
            }
          }
          const int uni = pl + pr - inter;
          double part;
          if constexpr (kQuotTable<W>) part = s_quot[inter * (2 * W + 1) + uni];
          else part = uni ? static_cast<double>(inter) / static_cast<double>(uni) : 0.0;
          factor *= 0.5;
          score += part * factor;
        }
      }
      const bool hit = ok && first == pa && score >= p.threshold;
      emit_hits_wave(hits, p.cap, count, hit, score, hit ? lorig[lrow] : 0, hit ? rorig[rrow] : 0);
    }
    __builtin_amdgcn_wave_barrier();  // (the next batch overwrites the wave's LDS rows)
  }
}

// Posting entries the probes visit (estimate, from the right table's statistics) and the launch.
template <int W>
int launch_levels_global(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode, nsm_hit* hits,
                         uint64_t capacity, unsigned long long* hit_count, hipStream_t stream, bool probe_only, double* estimate) {
  if (estimate) *estimate = r->n > 0 ? static_cast<double>(r->post_sq[4]) * static_cast<double>(l->n) / static_cast<double>(r->n) : 0.0;
  if (probe_only) return 0;
  JacLevGlobalParams<W> p;
  p.n_left = l->n; p.n_right = r->n; p.vocab = r->vocab; p.partitioned = l->seg != nullptr ? 1 : 0;
  p.lev_stride_l = l->max_levels; p.lev_stride_r = r->max_levels; p.cat_mode = category_mode;
  p.threshold = threshold; p.cap = capacity;
  if (int rc = check_post_format(r, "nsm_jaccard_levels_grid")) return rc;
  p.row_bits = r->post_row_bits;
  constexpr int kRows = kWave / W;
  p.n_batches = (l->n + kRows - 1) / kRows;
  long long blocks = (p.n_batches + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 256 * 8) blocks = 256 * 8;
#define NSM_LAUNCH_GLOBAL(F)                                                                                                     \
  hipLaunchKernelGGL((jaccard_levels_global_kernel<W, F>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, l->ids, \
                     l->cnt, l->plen, l->filt, l->orig, l->seg, r->ids, r->plen, r->filt, r->orig,                               \
                     reinterpret_cast<const unsigned long long*>(r->post), r->post_start, hits, hit_count, p)
  if (r->post_format == 2) NSM_LAUNCH_GLOBAL(2);
  else if (r->post_format == 1) NSM_LAUNCH_GLOBAL(1);
  else NSM_LAUNCH_GLOBAL(0);
#undef NSM_LAUNCH_GLOBAL
  return hip_status(hipGetLastError(), "jaccard_levels_global_kernel launch");
}

template int launch_levels_global<16>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                      unsigned long long*, hipStream_t, bool, double*);
template int launch_levels_global<32>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                      unsigned long long*, hipStream_t, bool, double*);
template int launch_levels_global<64>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                      unsigned long long*, hipStream_t, bool, double*);

}  // namespace nsm
