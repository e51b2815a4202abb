// RAW Indel-ratio grid for strings of 65..512 code units (stride 128 / 256 / 512).  Same structure as
// indel_raw.hip (length classes, exact length + histogram prunes, integer threshold test against a
// launcher-computed table), with the multi-word LCS of indel_wide.hpp.
#include "indel_wide.hpp"

namespace nsm {

struct IndelWideParams {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t pm_stride;
  int32_t zero_need;  // 0 when a 0.0 score reaches the threshold, else kNeverWide
  unsigned long long cap;
  uint16_t lcsmin[1032];  // indexed by la + lb (both >= 1), up to 1024
};

__device__ __forceinline__ double indel_score_wide(int la, int lb, int lcs) {
  if (la == 0 || lb == 0) return 0.0;
  const double maximum = static_cast<double>(la + lb);
  const double dist = static_cast<double>(la + lb - 2 * lcs);
  const double norm_sim = 1.0 - dist / maximum;
  return (norm_sim * 100.0) / 100.0;
}

static double indel_score_wide_host(int s, int lcs) {
  const volatile double maximum = static_cast<double>(s);
  const volatile double dist = static_cast<double>(s - 2 * lcs);
  const volatile double q = dist / maximum;
  const volatile double norm_sim = 1.0 - q;
  const volatile double pct = norm_sim * 100.0;
  return pct / 100.0;
}

template <bool PRUNE, int K>
__global__ __launch_bounds__(kBlock) void indel_raw_wide_kernel(
    const uint8_t* __restrict__ lcodes, const int32_t* __restrict__ llen, const int32_t* __restrict__ lstart,
    const int32_t* __restrict__ lorig, const uint32_t* __restrict__ lhist, const uint8_t* __restrict__ rcodes,
    const int32_t* __restrict__ rlen, const int32_t* __restrict__ rorig, const uint32_t* __restrict__ rhist,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const IndelWideParams p) {
  constexpr int kMaxLen = kWave * K;
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  // layout: [wave][pm_stride * K] masks | [wave][16 K][64] text dwords | lcsmin
  const int waves = blockDim.x >> 6;  // 4 (K = 2), 2 (K = 4) or 1 (K = 8): keeps the block under 64 KiB of LDS
  unsigned long long* pm_all = s_mem;
  uint32_t* text_all = reinterpret_cast<uint32_t*>(pm_all + waves * p.pm_stride * kPmWords<K>);
  uint16_t* s_lcsmin = reinterpret_cast<uint16_t*>(text_all + waves * 16 * K * kWave);
  for (int t = threadIdx.x; t < 2 * kMaxLen + 4; t += blockDim.x) s_lcsmin[t] = p.lcsmin[t];
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * waves + wave;
  if (tile * kWave >= p.n_right) return;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;

  unsigned long long* pm = pm_all + wave * p.pm_stride * kPmWords<K>;
  uint32_t* text = text_all + wave * 16 * K * kWave;
  wide_store_text<K>(text, rcodes + static_cast<size_t>(jc) * kMaxLen, lane);
  const int lbj = valid ? rlen[jc] : 0;
  const int jorig = rorig[jc];
  const int nchars = wave_first(lbj);  // sorted descending: lane 0 has the longest text
  uint32_t hr[8];
  if (PRUNE) {
    const uint4* hp = reinterpret_cast<const uint4*>(rhist + static_cast<size_t>(jc) * 8);
    const uint4 h0 = hp[0], h1 = hp[1];
    hr[0] = h0.x; hr[1] = h0.y; hr[2] = h0.z; hr[3] = h0.w;
    hr[4] = h1.x; hr[5] = h1.y; hr[6] = h1.z; hr[7] = h1.w;
  }

  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);

  // rows are sorted by length (descending): rows of length kMaxLen - c are [lstart[c], lstart[c + 1])
  const int c_first = kMaxLen - llen[i0];
  const int c_last = kMaxLen - llen[i1 - 1];
  for (int c = c_first; c <= c_last; ++c) {
    const int a = max(i0, lstart[c]);
    const int b = min(i1, lstart[c + 1]);
    if (a >= b) continue;
    const int la = kMaxLen - c;
    int need;
    if (la == 0 || lbj == 0) need = p.zero_need;
    else need = s_lcsmin[la + lbj];
    if (!valid) need = kNeverWide;
    const bool fits = min(la, lbj) >= need;  // LCS <= min(la, lb)
    if (PRUNE && !__any(fits)) continue;
    const int limit = fits ? la + lbj - 2 * need : -1;
    for (int i = a; i < b; ++i) {
      if (PRUNE) {
        // LCS <= (la + lb - L1(histograms)) / 2 (counts saturate at 255: only weakens the bound)
        const uint32_t* __restrict__ hl = lhist + static_cast<size_t>(i) * 8;
        uint32_t l1 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) l1 = __builtin_amdgcn_sad_u8(hl[q], hr[q], l1);
        if (!__any(static_cast<int>(l1) <= limit)) continue;
      }
      wide_build_pm<K>(pm, p.pm_stride, lcodes + static_cast<size_t>(i) * kMaxLen, la, lane);
      const int lcs = wide_lcs<K, true>(pm, text, nchars, lane, la, lbj, need);
      const bool hit = lcs >= need;
      if (__any(hit)) {
        if (hit) emit_hit(hits, p.cap, count, indel_score_wide(la, lbj, lcs), lorig[i], jorig);
      }
    }
  }
}

template <int K>
static int launch_wide(const nsm_str_table* left, const nsm_str_table* right, double threshold, uint32_t flags,
                       nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, hipStream_t stream) {
  constexpr int kMaxLen = kWave * K;
  IndelWideParams p;
  p.n_left = left->n; p.n_right = right->n; p.cap = capacity;
  p.pm_stride = ((left->alphabet + 1) + 7) / 8 * 8;  // entries per mask table (the pad symbol included)
  p.zero_need = (0.0 >= threshold) ? 0 : kNeverWide;
  for (int s = 0; s < 1032; ++s) {
    p.lcsmin[s] = kNeverWide;
    if (s < 2 || s > 2 * kMaxLen) continue;
    for (int lcs = 0; 2 * lcs <= s; ++lcs) {
      if (indel_score_wide_host(s, lcs) >= threshold) {
        p.lcsmin[s] = static_cast<uint16_t>(lcs);
        break;
      }
    }
  }
  const int n_tiles = (right->n + kWave - 1) / kWave;
  const long long want_waves = 16ll * 256 * 32;
  long long chunks = (want_waves + n_tiles - 1) / n_tiles;
  long long rows = (left->n + chunks - 1) / chunks;
  if (rows < 64) rows = 64;
  if (rows > 4096) rows = 4096;
  p.rows_per_chunk = static_cast<int>(rows);
  constexpr int kWaves = K == 2 ? 4 : (K == 4 ? 2 : 1);
  dim3 grid((n_tiles + kWaves - 1) / kWaves, (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (left->n + 65534) / 65535;
    grid.y = (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  const size_t lds = static_cast<size_t>(kWaves) * (p.pm_stride * kPmWords<K> * 8 + 16 * K * kWave * 4) +
                     (2 * kMaxLen + 4) * 2 + 16;
  const uint32_t* lh = reinterpret_cast<const uint32_t*>(left->hist);
  const uint32_t* rh = reinterpret_cast<const uint32_t*>(right->hist);
  if ((flags & NSM_FLAG_PRUNE) && lh && rh)
    hipLaunchKernelGGL((indel_raw_wide_kernel<true, K>), grid, dim3(kWaves * kWave), lds, stream, left->codes, left->len,
                       left->len_start, left->orig, lh, right->codes, right->len, right->orig, rh, hits, hit_count, p);
  else
    hipLaunchKernelGGL((indel_raw_wide_kernel<false, K>), grid, dim3(kWaves * kWave), lds, stream, left->codes, left->len,
                       left->len_start, left->orig, lh, right->codes, right->len, right->orig, rh, hits, hit_count, p);
  return hip_status(hipGetLastError(), "indel_raw_wide_kernel launch");
}

// called by nsm_indel_raw_grid for stride 128 / 256 / 512
int indel_raw_wide(const nsm_str_table* left, const nsm_str_table* right, double threshold, uint32_t flags,
                   nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, hipStream_t stream) {
  if (left->stride == 128)
    return launch_wide<2>(left, right, threshold, flags, hits, capacity, hit_count, stream);
  if (left->stride == 256)
    return launch_wide<4>(left, right, threshold, flags, hits, capacity, hit_count, stream);
  return launch_wide<8>(left, right, threshold, flags, hits, capacity, hit_count, stream);
}

}  // namespace nsm
