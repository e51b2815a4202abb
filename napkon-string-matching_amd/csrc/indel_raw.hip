// RAW Indel-ratio all-pairs grid:  fuzzy_match = rapidfuzz QRatio / 100 on one string per item
// (reference: napkon_string_matching/compare/score_functions.py:20-27; QRatio of rapidfuzz 2.1.x
// is the normalized Indel similarity, i.e. 1 - (|a|+|b|-2 LCS)/(|a|+|b|)).
//
// Mapping to CDNA4
//   * bit-parallel LCS (Hyyro / Allison-Dix):  V = ~0;  per text symbol c:  U = V & PM[c];
//     V = (V + U) | (V - U);  LCS = popcount(~V).  Strings of <= 64 code units -> one 64-bit word;
//   * one LANE owns one right string (the "text") for the whole kernel: 64 code units in 16 VGPRs,
//     consumed with statically indexed v_bfe -- no memory traffic for the text in the loop;
//   * the left string (the "pattern") is wave-uniform.  Each wave builds its match-mask table
//     PM[alphabet] in LDS (one ds_write_b64 sweep to clear, ONE ds_or_b64 per wave to set: lane k
//     ORs bit k into PM[pattern[k]]), double buffered so the next pattern's build overlaps;
//   * PM[c] lookups are 8-byte LDS reads indexed by the lane's symbol: lanes with the same symbol
//     broadcast, symbols c and c' only conflict when c == c' (mod 32);
//   * both tables are sorted by length (descending): a wave stops after its longest text, and the
//     per-row threshold test is an integer compare against lcsmin[la+lb], computed by the launcher
//     with the reference's double arithmetic; the double score is only computed for hits.
#include "nsm_common.hpp"

namespace nsm {

// Table columns travel as __restrict__ kernel arguments so that the wave-uniform left side is
// fetched with scalar loads (see jaccard_raw_impl.hpp).
struct IndelRawParams {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  int32_t pm_stride;  // (alphabet + 1) rounded up to 64 entries
  int32_t zero_need;  // 0 when a 0.0 score reaches the threshold, else kNever
  unsigned long long cap;
  uint8_t lcsmin[132];  // indexed by la+lb (both >= 1)
};

// The exact double sequence of `QRatio(a, b) / 100` once LCS is known (oracle/score_functions.py).
__device__ __forceinline__ double indel_score(int la, int lb, int lcs) {
  if (la == 0 || lb == 0) return 0.0;
  const double maximum = static_cast<double>(la + lb);
  const double dist = static_cast<double>(la + lb - 2 * lcs);
  const double norm_sim = 1.0 - dist / maximum;
  return (norm_sim * 100.0) / 100.0;
}

static double indel_score_host(int la, int lb, int lcs) {
  if (la == 0 || lb == 0) return 0.0;
  const volatile double maximum = static_cast<double>(la + lb);
  const volatile double dist = static_cast<double>(la + lb - 2 * lcs);
  const volatile double q = dist / maximum;
  const volatile double norm_sim = 1.0 - q;
  const volatile double pct = norm_sim * 100.0;
  return pct / 100.0;
}

template <bool PRUNE>
__global__ __launch_bounds__(kBlock) void indel_raw_kernel(
    const uint8_t* __restrict__ lcodes, const int32_t* __restrict__ llen, const int32_t* __restrict__ lorig,
    const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen, const int32_t* __restrict__ rorig,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const IndelRawParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  // layout: [wave][buffer][pm_stride] match masks, then the lcsmin bytes
  uint8_t* s_lcsmin = reinterpret_cast<uint8_t*>(s_mem + kWavesPerBlock * 2 * p.pm_stride);
  for (int t = threadIdx.x; t < 132; t += kBlock) s_lcsmin[t] = p.lcsmin[t];
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * kWavesPerBlock + wave;
  if (tile * kWave >= p.n_right) return;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;

  uint32_t text[16];
  const uint4* tp = reinterpret_cast<const uint4*>(rcodes + static_cast<size_t>(jc) * 64);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = tp[q];
    text[4 * q + 0] = v.x;
    text[4 * q + 1] = v.y;
    text[4 * q + 2] = v.z;
    text[4 * q + 3] = v.w;
  }
  const int lbj = valid ? rlen[jc] : 0;
  const int jorig = rorig[jc];
  const int nwords = (wave_first(lbj) + 3) >> 2;  // sorted descending: lane 0 has the longest text

  unsigned long long* pm_base = s_mem + wave * 2 * p.pm_stride;
  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);

  for (int i = i0; i < i1; ++i) {
    const int la = llen[i];  // wave-uniform
    int need;
    if (la == 0 || lbj == 0) need = p.zero_need;
    else need = s_lcsmin[la + lbj];
    if (!valid) need = kNever;
    if (PRUNE) {
      // LCS <= min(la, lb): skip the row when no lane can reach its bound
      if (!__any(min(la, lbj) >= need)) continue;
    }
    // ---- build PM for pattern i (this wave's private buffer, alternating)
    unsigned long long* pm = pm_base + (i & 1) * p.pm_stride;
    for (int c = lane; c < p.pm_stride; c += kWave) pm[c] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < la) {
      const unsigned c = lcodes[static_cast<size_t>(i) * 64 + lane];
      atomicOr(&pm[c], 1ull << lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- Hyyro LCS over the text, one symbol per step
    unsigned long long v = ~0ull;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < nwords) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned c = (text[w] >> (8 * b)) & 0xffu;
          const unsigned long long m = pm[c];
          const unsigned long long u = v & m;
          v = (v + u) | (v - u);
        }
      }
    }
    const int lcs = 64 - __popcll(v);
    const bool hit = lcs >= need;
    if (__any(hit)) {
      if (hit) emit_hit(hits, p.cap, count, indel_score(la, lbj, lcs), lorig[i], jorig);
    }
  }
}

static int pick_rows_per_chunk(int n_left, int n_tiles) {
  const long long want_waves = 16ll * 256 * 32;
  long long chunks = (want_waves + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
  if (chunks < 1) chunks = 1;
  long long rows = (n_left + chunks - 1) / chunks;
  if (rows < 64) rows = 64;
  if (rows > 4096) rows = 4096;
  return static_cast<int>(rows);
}

}  // namespace nsm

extern "C" int nsm_indel_raw_grid(const nsm_str_table* left, const nsm_str_table* right,
                                  double threshold, uint32_t flags, nsm_hit* hits, uint64_t capacity,
                                  unsigned long long* hit_count, void* stream) {
  using namespace nsm;
  if (!left || !right || !hit_count || (!hits && capacity)) {
    set_error("nsm_indel_raw_grid: null argument");
    return NSM_E_BADARG;
  }
  if (left->stride != 64 || right->stride != 64) {
    set_error("nsm_indel_raw_grid: stride %d/%d unsupported (strings longer than 64 code units)",
              left->stride, right->stride);
    return NSM_E_UNSUPPORTED;
  }
  if (left->alphabet != right->alphabet || left->alphabet < 1 || left->alphabet > 255) {
    set_error("nsm_indel_raw_grid: alphabets differ or exceed 255 (%d, %d)", left->alphabet,
              right->alphabet);
    return NSM_E_BADARG;
  }
  if (left->n < 0 || right->n < 0) {
    set_error("nsm_indel_raw_grid: negative row count");
    return NSM_E_BADARG;
  }
  if (left->n == 0 || right->n == 0) return 0;
  if (!left->codes || !left->len || !left->orig || !right->codes || !right->len || !right->orig) {
    set_error("nsm_indel_raw_grid: table has a null column");
    return NSM_E_BADARG;
  }

  IndelRawParams p;
  p.n_left = left->n; p.n_right = right->n;
  p.cap = capacity;
  p.pm_stride = ((left->alphabet + 1) + 63) / 64 * 64;
  p.zero_need = (0.0 >= threshold) ? 0 : kNever;
  for (int s = 0; s < 132; ++s) {
    p.lcsmin[s] = kNever;
    if (s < 2 || s > 128) continue;
    for (int lcs = 0; 2 * lcs <= s; ++lcs) {
      // any split la+lb = s gives the same score: it only depends on s and lcs
      if (indel_score_host(1, s - 1, lcs) >= threshold) {
        p.lcsmin[s] = static_cast<uint8_t>(lcs);
        break;
      }
    }
  }
  const int n_tiles = (right->n + kWave - 1) / kWave;
  p.rows_per_chunk = pick_rows_per_chunk(left->n, n_tiles);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock,
            (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (left->n + 65534) / 65535;
    grid.y = (left->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  const size_t lds = static_cast<size_t>(kWavesPerBlock) * 2 * p.pm_stride * 8 + 136;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (flags & NSM_FLAG_PRUNE)
    hipLaunchKernelGGL((indel_raw_kernel<true>), grid, dim3(kBlock), lds, s, left->codes, left->len, left->orig,
                       right->codes, right->len, right->orig, hits, hit_count, p);
  else
    hipLaunchKernelGGL((indel_raw_kernel<false>), grid, dim3(kBlock), lds, s, left->codes, left->len, left->orig,
                       right->codes, right->len, right->orig, hits, hit_count, p);
  return hip_status(hipGetLastError(), "indel_raw_kernel launch");
}
