// RAW Jaccard all-pairs grid:  intersection_vs_union on one id-set per item
// (reference: napkon_string_matching/compare/score_functions.py:6-13).
//
// Mapping to CDNA4
//   * one LANE owns one right row for the whole kernel: its W ids live in W VGPRs;
//   * the left row is wave-uniform: it is fetched with scalar loads (s_load_dwordx4..16) and its
//     ids are SGPR operands of the VALU ops -- no LDS and no vector memory traffic in the loop;
//   * equality matrix without compares:  m = min3(m, a^b0, a^b1)  is zero iff `a` occurs in the
//     right row (ids are unique per row), 1.5 VALU ops per id pair, no SGPR write hazards;
//   * both tables are sorted by set size (descending), so a wave's 64 right rows have (nearly) the
//     same size NB and a run of left rows the same size class NL: the matrix is NL x NB, not W x W.
//     The dispatch on (NL, NB) is wave-uniform (scalar branches only);
//   * threshold test on integers: hit  <=>  matches >= kmin[|A|+|B|], with kmin computed by the
//     launcher with the very double division/compare the reference performs; the double score is
//     only computed for hits;
//   * optional exact prune: |A n B| <= popcount(sigA & sigB) + min(cA, cB) (64-bit id signatures,
//     c = in-row signature collisions); a wave skips the matrix when no lane can reach kmin.
#pragma once
#include "nsm_common.hpp"

namespace nsm {

// The table columns are passed as __restrict__ kernel arguments (not inside a struct): only then
// does hipcc prove them read-only and fetch the wave-uniform left row with s_load_dwordxN.
template <int W>
struct JacRawScalars {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  unsigned long long cap;
  uint8_t kmin[2 * W + 4];  // indexed by |A|+|B|
};

// Number of left ids (out of NL, padding included) that do NOT occur in the lane's right row.
template <int W, int NL, int NB>
__device__ __forceinline__ int nonmatches(const int32_t* __restrict__ lrow, const uint32_t (&r)[W]) {
  static_assert(NB >= 2 && NB % 2 == 0, "right class must be even");
  int nm = 0;
#pragma unroll
  for (int a = 0; a < NL; ++a) {
    const uint32_t la = static_cast<uint32_t>(lrow[a]);  // SGPR
    uint32_t m = min(la ^ r[0], la ^ r[1]);
#pragma unroll
    for (int b = 2; b < NB; b += 2) m = min(m, min(la ^ r[b], la ^ r[b + 1]));  // v_min3_u32
    nm += static_cast<int>(min(m, 1u));
  }
  return nm;
}

template <int W, int NB, bool PRUNE>
__device__ __forceinline__ void wave_rows(const int32_t* __restrict__ lids,
                                          const int32_t* __restrict__ lcnt,
                                          const uint64_t* __restrict__ lsig,
                                          const int32_t* __restrict__ lorig,
                                          nsm_hit* __restrict__ hits, unsigned long long cap,
                                          unsigned long long* __restrict__ count,
                                          const uint32_t (&r)[W], int nrj, uint64_t sr, int jorig,
                                          bool valid, int i0, int i1, const uint8_t* s_kmin) {
  constexpr int NLS = W / 4;  // left size classes: NLS, 2 NLS, 3 NLS, W
  int prev_nl = -1;
  int need = kNever;
  const int extra_r = nrj - __popcll(sr);
  for (int i = i0; i < i1; ++i) {
    const int nl = lcnt[i];  // wave-uniform -> scalar load
    if (nl != prev_nl) {       // rows are sorted by size: at most W+1 changes per chunk
      prev_nl = nl;
      need = valid ? s_kmin[nl + nrj] : kNever;
    }
    if (PRUNE) {
      // |A n B| <= popcount(sigA & sigB) + min(cA, cB), c = ids of the row that share a signature
      // bit with an earlier id of the same row (|row| - popcount(sig)): common ids that collide
      // inside both rows are the only ones the AND can miss.
      const uint64_t sl = lsig[i];
      const int bound = __popcll(sl & sr) + min(nl - __popcll(sl), extra_r);
      if (!__any(bound >= need)) continue;
    }
    const int32_t* __restrict__ lrow = lids + static_cast<size_t>(i) * W;
    const int cls = (nl + NLS - 1) / NLS;
    int k;
    switch (cls) {
      case 0: k = 0; break;
      case 1: k = NLS - nonmatches<W, NLS, NB>(lrow, r); break;
      case 2: k = 2 * NLS - nonmatches<W, 2 * NLS, NB>(lrow, r); break;
      case 3: k = 3 * NLS - nonmatches<W, 3 * NLS, NB>(lrow, r); break;
      default: k = W - nonmatches<W, W, NB>(lrow, r); break;
    }
    const bool hit = k >= need;
    if (__any(hit)) {
      if (hit) {
        const double score = static_cast<double>(k) / static_cast<double>(nl + nrj - k);
        emit_hit(hits, cap, count, score, lorig[i], jorig);
      }
    }
  }
}

template <int W, bool PRUNE>
__global__ __launch_bounds__(kBlock) void jaccard_raw_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt,
    const uint64_t* __restrict__ lsig, const int32_t* __restrict__ lorig,
    const int32_t* __restrict__ rids, const int32_t* __restrict__ rcnt,
    const uint64_t* __restrict__ rsig, const int32_t* __restrict__ rorig,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const JacRawScalars<W> p) {
  __shared__ uint8_t s_kmin[2 * W + 4];
  for (int t = threadIdx.x; t < 2 * W + 4; t += kBlock) s_kmin[t] = p.kmin[t];
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (tile * kWave >= p.n_right) return;  // whole wave
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;

  uint32_t r[W];
  const uint4* rp = reinterpret_cast<const uint4*>(rids + static_cast<size_t>(jc) * W);
#pragma unroll
  for (int q = 0; q < W / 4; ++q) {
    const uint4 v = rp[q];
    r[4 * q + 0] = v.x;
    r[4 * q + 1] = v.y;
    r[4 * q + 2] = v.z;
    r[4 * q + 3] = v.w;
  }
  const int nrj = valid ? rcnt[jc] : 0;
  const uint64_t sr = (PRUNE && valid) ? rsig[jc] : 0ull;
  const int jorig = rorig[jc];
  const int nbmax = wave_first(nrj);  // sorted descending: lane 0 holds the tile's largest set

  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);

  constexpr int NBS = W / 8;  // right size classes: NBS, 2 NBS, ..., W
  const int cls = (nbmax + NBS - 1) / NBS;
  switch (cls) {
    case 0:
    case 1: wave_rows<W, 1 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    case 2: wave_rows<W, 2 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    case 3: wave_rows<W, 3 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    case 4: wave_rows<W, 4 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    case 5: wave_rows<W, 5 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    case 6: wave_rows<W, 6 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    case 7: wave_rows<W, 7 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
    default: wave_rows<W, 8 * NBS, PRUNE>(lids, lcnt, lsig, lorig, hits, p.cap, count, r, nrj, sr, jorig, valid, i0, i1, s_kmin); break;
  }
}

// kmin[s] = least k with double(k)/double(s-k) >= threshold (k <= s/2), kNever if none.
template <int W>
inline void fill_kmin(uint8_t* kmin, double threshold) {
  for (int s = 0; s < 2 * W + 4; ++s) {
    kmin[s] = kNever;
    if (s == 0 || s > 2 * W) continue;  // 0/0 is the reference's ZeroDivisionError: host raises
    for (int k = 0; 2 * k <= s; ++k) {
      const volatile double q = static_cast<double>(k) / static_cast<double>(s - k);
      if (q >= threshold) {
        kmin[s] = static_cast<uint8_t>(k);
        break;
      }
    }
  }
}

inline int jac_rows_per_chunk(int n_left, int n_tiles) {
  // aim for >= 16 waves per wave slot of the chip (256 CUs x 32) while keeping chunks >= 128 rows
  const long long want_waves = 16ll * 256 * 32;
  long long chunks = (want_waves + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
  if (chunks < 1) chunks = 1;
  long long rows = (n_left + chunks - 1) / chunks;
  if (rows < 128) rows = 128;
  if (rows > 4096) rows = 4096;
  return static_cast<int>(rows);
}

template <int W>
int launch_raw(const nsm_set_table* l, const nsm_set_table* r, double threshold, uint32_t flags,
                      nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count,
                      hipStream_t stream) {
  JacRawScalars<W> p;
  p.n_left = l->n; p.n_right = r->n; p.cap = capacity;
  fill_kmin<W>(p.kmin, threshold);
  const int n_tiles = (r->n + kWave - 1) / kWave;
  p.rows_per_chunk = jac_rows_per_chunk(l->n, n_tiles);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock,
            (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (l->n + 65534) / 65535;
    grid.y = (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  // a zero threshold (kmin == 0 everywhere) makes the bound useless; PRUNE only changes speed
  const bool prune = (flags & NSM_FLAG_PRUNE) && l->sig && r->sig;
  if (prune)
    hipLaunchKernelGGL((jaccard_raw_kernel<W, true>), grid, dim3(kBlock), 0, stream, l->ids, l->cnt,
                       l->sig, l->orig, r->ids, r->cnt, r->sig, r->orig, hits, hit_count, p);
  else
    hipLaunchKernelGGL((jaccard_raw_kernel<W, false>), grid, dim3(kBlock), 0, stream, l->ids, l->cnt,
                       l->sig, l->orig, r->ids, r->cnt, r->sig, r->orig, hits, hit_count, p);
  return hip_status(hipGetLastError(), "jaccard_raw_kernel launch");
}

}  // namespace nsm
