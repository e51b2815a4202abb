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
//   * optional exact prune, two stages: |A n B| <= popcount(hashbits(A) & hashbits(B)) + cA with 58
//     hash bits per signature word (cA = in-row collisions, kept in the word's top 6 bits); a wave
//     skips the matrix unless some lane can reach kmin under two independent signatures;
//   * exact size filter: a whole size class is skipped when kmin > min(|A|, |B|) for every lane.
#pragma once
#include <cmath>

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
  unsigned long long weak[W + 1];  // weak[|A|] bit |B|: the signature bound rarely fails for these sizes
};

// Inverted-index kernel (jaccard_raw_index.hip): slots of a block's open-addressing hash table.  A tile holding more
// than 3/4 of that many ids is "dense" and is indexed in TWO passes of 32 lanes each (n_pass in the kernel): half a
// tile holds at most 32 W ids, which never fills the table -- the insertion loop cannot wrap around a full table.
template <int W>
constexpr int index_slots() { return W == 16 ? 1024 : 2048; }
static_assert(32 * 16 <= index_slots<16>() && 32 * 32 <= index_slots<32>(), "half a tile must fit the hash table");

template <int W>
__device__ __forceinline__ bool tile_is_dense(int nrj) {  // wave-uniform; all 64 lanes enabled
  // wave sum of the set sizes on DPP (row sums) + v_readlane: a scalar, no ds_bpermute round trips
  const uint32_t total = wave_reduce_u32(static_cast<uint32_t>(nrj), [](uint32_t x, uint32_t y) { return x + y; });
  return static_cast<int>(total) > index_slots<W>() * 3 / 4;
}

template <int W>
int launch_raw_index(const nsm_set_table* l, const nsm_set_table* r, double threshold, nsm_hit* hits, uint64_t capacity,
                     unsigned long long* hit_count, hipStream_t stream);

// Candidate generation from the right table's GLOBAL inverted index (jaccard_raw_global.hip).  `probe_only`: nothing is
// launched, *estimate receives the number of posting entries the probes would visit.
template <int W>
int launch_raw_global(const nsm_set_table* l, const nsm_set_table* r, double threshold, nsm_hit* hits, uint64_t capacity,
                      unsigned long long* hit_count, hipStream_t stream, bool probe_only, double* estimate);

// Wave-uniform value -> VGPR.  On gfx950 a VALU op with an SGPR source issues at half rate
// (4.5 vs 2.4 cycles per wave64 v_xor_b32, profiles/r01_valu_issue_rates_gfx950.txt), so an id that
// is XORed against NB registers is first broadcast with ONE v_mov_b32.
__device__ __forceinline__ uint32_t to_vgpr(uint32_t uniform) {
  uint32_t v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
  return v;
}

// hipcc re-associates nested umin() into v_min_u32 pairs and turns umin(x, 1) into cmp + cndmask;
// the matrix wants exactly one v_min3_u32 per two id pairs, so it is spelled out.
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t min3u_one(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_min3_u32 %0, %1, %2, 1" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// Number of left ids (out of NL, padding included) that do NOT occur in the lane's right row.
// Per id: 1 v_mov + NB v_xor (full rate) + NB/2 v_min3 (half rate) + 1 v_add; the constant 1 rides
// in the first min3, so m ends as 0 (found) or 1 (not found) without a separate clamp.
template <int W, int NL, int NB>
__device__ __forceinline__ int nonmatches(const int32_t* __restrict__ lrow, const uint32_t (&r)[W]) {
  static_assert(NB >= 2 && NB % 2 == 0, "right class must be even");
  uint32_t l[NL];
#pragma unroll
  for (int a = 0; a < NL; ++a) l[a] = static_cast<uint32_t>(lrow[a]);  // one s_load_dwordxN
  uint32_t nm = 0;
#pragma unroll
  for (int a = 0; a < NL; ++a) {
    const uint32_t la = to_vgpr(l[a]);
    uint32_t m = min3u_one(la ^ r[0], la ^ r[1]);
#pragma unroll
    for (int b = 2; b < NB; b += 2) m = min3u(m, la ^ r[b], la ^ r[b + 1]);
    nm += m;
  }
  return static_cast<int>(nm);
}

// Score the left rows [a, b) -- all of size `nl`, size class NL -- against the lane's right row.
template <int W, int NL, int NB, bool PRUNE>
__device__ __forceinline__ void class_rows(const int32_t* __restrict__ lids,
                                           const uint64_t* __restrict__ lsig,
                                           const uint64_t* __restrict__ lsig2,
                                           const int32_t* __restrict__ lorig,
                                           nsm_hit* __restrict__ hits, unsigned long long cap,
                                           unsigned long long* __restrict__ count,
                                           const uint32_t (&r)[W], int nrj, uint64_t sr, uint64_t sr2,
                                           int jorig, int a, int b, int nl, int need, bool use_prune) {
  auto exact_row = [&](int i) {
    const int k = NL - nonmatches<W, NL, NB>(lids + static_cast<size_t>(i) * W, r);
    const bool hit = k >= need;
    if (__any(hit)) {
      if (hit) {
        const double score = static_cast<double>(k) / static_cast<double>(nl + nrj - k);
        emit_hit(hits, cap, count, score, lorig[i], jorig);
      }
    }
  };
  if (PRUNE && use_prune) {
    // |A n B| <= popcount(hashbits(A) & hashbits(B)) + cA, cA = the ids of the row that share a hash
    // bit with an earlier id of the same row: common ids that collide inside the row are the only
    // ones the AND can miss.  A signature word holds 58 hash bits and cA in unary in its top 6 bits;
    // sr / sr2 come with their top 6 bits SET, so the bound is just popcount(sl & sr).
    // Per row: 2 v_and + 2 v_bcnt (chained, seeded with the lane's -need) + 1 full-rate v_and that
    // folds the sign (bound - need < 0 = cannot reach the threshold) into the batch's verdict = 5
    // VALU; the scalar unit is shared by the CU's 4 SIMDs, so verdicts are NOT collected with
    // s_cselect/s_or.  (Before: cA as a number in the top bits, v_cmp + v_addc per row: 6 VALU, the
    // last two at half rate.  Tried earlier: reduce 8 bounds with v_max3 -- slower, see DESIGN.md.)
    constexpr int BATCH = 8;
    const int neg_need = -need;
    auto margin_of = [&](uint64_t sl, uint64_t srm) {  // bound - need
      const uint32_t lo = static_cast<uint32_t>(sl) & static_cast<uint32_t>(srm);
      const uint32_t hi = static_cast<uint32_t>(sl >> 32) & static_cast<uint32_t>(srm >> 32);
      int margin;
      asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(margin) : "v"(lo), "v"(neg_need));
      asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(margin) : "v"(hi), "v"(margin));
      return margin;
    };
    auto second_stage = [&](int i) {
      return lsig2 == nullptr || __any(margin_of(lsig2[i], sr2) >= 0);
    };
    // one batch: 8 rows' signature words (already in SGPRs) against the lane's
    auto test_batch = [&](const uint64_t (&sg)[BATCH], int i) {
      int margin[BATCH];
      int all_fail = -1;  // sign bit stays set while every row of the batch fails
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        margin[q] = margin_of(sg[q], sr);
        all_fail &= margin[q];
      }
      if (__any(all_fail >= 0)) {  // rare
        uint32_t cand = 0;  // per lane: bit q = row q may reach the threshold
#pragma unroll
        for (int q = 0; q < BATCH; ++q) cand |= margin[q] >= 0 ? (1u << q) : 0u;
        for (int q = 0; q < BATCH; ++q) {  // ONE copy of the matrix body (I-cache footprint)
          if (__any((cand >> q) & 1u)) {
            // second stage: an independent signature must agree before the matrix is paid for
            if (second_stage(i + q)) exact_row(i + q);
          }
        }
      }
    };
    auto load_batch = [&](uint64_t (&sg)[BATCH], int i) {
#pragma unroll
      for (int q = 0; q < BATCH; ++q) sg[q] = lsig[i + q];  // one s_load_dwordx16
    };
    int i = a;
    if (i + BATCH <= b) {
      // software pipeline over two register sets: the signature words of the next batch are in flight
      // while this one is tested, and no set is copied (8 s_mov_b64 per batch on the shared scalar unit)
      uint64_t s0[BATCH], s1[BATCH];
      load_batch(s0, i);
      while (true) {
        const bool more1 = i + 2 * BATCH <= b;
        load_batch(s1, more1 ? i + BATCH : i);  // the last batch re-reads itself
        test_batch(s0, i);
        i += BATCH;
        if (!more1) break;
        const bool more0 = i + 2 * BATCH <= b;
        load_batch(s0, more0 ? i + BATCH : i);
        test_batch(s1, i);
        i += BATCH;
        if (!more0) break;
      }
    }
    for (; i < b; ++i) {
      if (__any(margin_of(lsig[i], sr) >= 0) && second_stage(i)) exact_row(i);
    }
  } else {
    for (int i = a; i < b; ++i) exact_row(i);
  }
}

template <int W, int NB, bool PRUNE>
__device__ __forceinline__ void wave_rows(const int32_t* __restrict__ lids,
                                          const int32_t* __restrict__ lcnt,
                                          const int32_t* __restrict__ lstart,
                                          const uint64_t* __restrict__ lsig,
                                          const uint64_t* __restrict__ lsig2,
                                          const int32_t* __restrict__ lorig,
                                          nsm_hit* __restrict__ hits, unsigned long long cap,
                                          unsigned long long* __restrict__ count,
                                          const uint32_t (&r)[W], int nrj, uint64_t sr, uint64_t sr2,
                                          int jorig, bool valid, int i0, int i1, const uint8_t* s_kmin,
                                          const unsigned long long* p_weak) {
  constexpr int NLS = W / 4;  // left size classes: NLS, 2 NLS, 3 NLS, W
  // rows are sorted by size (descending): rows of size W - c are [lstart[c], lstart[c + 1]); the
  // chunk [i0, i1) only touches the sizes between its first and its last row
  const int c_first = W - lcnt[i0];
  const int c_last = W - lcnt[i1 - 1];
  for (int c = c_first; c <= c_last; ++c) {
    const int a = max(i0, lstart[c]);
    const int b = min(i1, lstart[c + 1]);
    if (a >= b) continue;
    const int nl = W - c;
    const int need = valid ? s_kmin[nl + nrj] : kNever;
    // exact size filter: a pair can only reach the threshold if kmin <= min(|A|, |B|)
    if (!__any(need <= min(nl, nrj))) continue;
    // The signature bound only pays when it fails for (nearly) every lane of the wave; the launcher
    // marks the size pairs for which a random pair passes it too often (low thresholds) and such a
    // class runs without the prune.
    const bool weak = valid && nrj < 64 && ((p_weak[nl] >> nrj) & 1ull);
    const bool use_prune = !__any(weak);
#define NSM_ROWS(NL)                                                                                        \
  class_rows<W, NL, NB, PRUNE>(lids, lsig, lsig2, lorig, hits, cap, count, r, nrj, sr, sr2, jorig, a, b, nl, need, \
                               use_prune)
    switch ((nl + NLS - 1) / NLS) {
      case 0: {  // empty left sets: no common id, hit only when kmin == 0 (threshold <= 0)
        const bool hit = need == 0;
        if (__any(hit)) {
          for (int i = a; i < b; ++i)
            if (hit) emit_hit(hits, cap, count, 0.0 / static_cast<double>(nrj), lorig[i], jorig);
        }
        break;
      }
      case 1: NSM_ROWS(NLS); break;
      case 2: NSM_ROWS(2 * NLS); break;
      case 3: NSM_ROWS(3 * NLS); break;
      default: NSM_ROWS(W); break;
    }
#undef NSM_ROWS
  }
}

#ifndef NSM_JAC_OCC
#define NSM_JAC_OCC
#endif
template <int W, bool PRUNE>
__global__ __launch_bounds__(kBlock) NSM_JAC_OCC void jaccard_raw_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt,
    const int32_t* __restrict__ lstart, const uint64_t* __restrict__ lsig,
    const uint64_t* __restrict__ lsig2,
    const int32_t* __restrict__ lorig, const int32_t* __restrict__ rids,
    const int32_t* __restrict__ rcnt, const uint64_t* __restrict__ rsig,
    const uint64_t* __restrict__ rsig2, const int32_t* __restrict__ rorig,
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
  constexpr uint64_t kCollBits = ~((1ull << 58) - 1);  // the top 6 bits of a signature word hold cA (unary)
  const uint64_t sr = (PRUNE && valid) ? (rsig[jc] | kCollBits) : 0ull;
  const uint64_t sr2 = (PRUNE && valid && rsig2 != nullptr) ? (rsig2[jc] | kCollBits) : 0ull;
  if (rsig2 == nullptr) lsig2 = nullptr;
  const int jorig = rorig[jc];
  const int nbmax = wave_first(nrj);  // sorted descending: lane 0 holds the tile's largest set

  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);

  constexpr int NBS = W / 8;  // right size classes: NBS, 2 NBS, ..., W
  const int cls = (nbmax + NBS - 1) / NBS;
#define NSM_WAVE(K) \
  wave_rows<W, (K) * NBS, PRUNE>(lids, lcnt, lstart, lsig, lsig2, lorig, hits, p.cap, count, r, nrj, sr, sr2, jorig, valid, i0, i1, s_kmin, p.weak)
  switch (cls) {
    case 0:
    case 1: NSM_WAVE(1); break;
    case 2: NSM_WAVE(2); break;
    case 3: NSM_WAVE(3); break;
    case 4: NSM_WAVE(4); break;
    case 5: NSM_WAVE(5); break;
    case 6: NSM_WAVE(6); break;
    case 7: NSM_WAVE(7); break;
    default: NSM_WAVE(8); break;
  }
#undef NSM_WAVE
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
  // (A/B on C2, kernel ms: 2 / 4 / 8 / 16 / 32 waves per slot -> 0.72 / 0.54 / 0.465 / 0.436 / 0.455)
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
  // weak[nl] bit nr: two random sets of nl and nr ids share ~Poisson(nl*nr/58) hash bits; when the
  // chance of reaching kmin that way exceeds 1/128 a wave of 64 lanes passes more often than not
  for (int a = 0; a <= W; ++a) {
    p.weak[a] = 0;
    for (int b = 0; b <= W && b < 64; ++b) {
      const int need = p.kmin[a + b];
      if (need == kNever || need > (a < b ? a : b)) continue;
      const double mu = static_cast<double>(a) * b / 58.0;
      double term = std::exp(-mu), below = 0.0;  // P(X < need)
      for (int k = 0; k < need; ++k) {
        below += term;
        term *= mu / (k + 1);
      }
      if (1.0 - below > 1.0 / 128.0) p.weak[a] |= 1ull << b;
    }
  }
  const int n_tiles = (r->n + kWave - 1) / kWave;
  p.rows_per_chunk = jac_rows_per_chunk(l->n, n_tiles);
  dim3 grid((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock,
            (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk);
  if (grid.y > 65535) {
    p.rows_per_chunk = (l->n + 65534) / 65535;
    grid.y = (l->n + p.rows_per_chunk - 1) / p.rows_per_chunk;
  }
  // a zero threshold (kmin == 0 everywhere) makes the bound useless; PRUNE only changes speed
  // W = 64: the pruned instantiation needs > 256 VGPRs (416 B of scratch, -Rpass-analysis=kernel-resource-usage) and a
  // 58-bit signature of 33..64 ids is saturated anyway (every size class is "weak": the prune switches itself off).
  // Measured on 20k x 20k sets of ~44 ids at 0.5: pruned 39.1 ms, exhaustive 38.2 ms (tools/bench_w64.py) -- the
  // spill-free exhaustive instantiation (142 VGPRs) runs wide sets.
  const bool prune = (flags & NSM_FLAG_PRUNE) && l->sig && r->sig && W < 64;
  // The right table carries a global inverted index: generate candidates from it (prefix filter) instead of visiting every
  // pair, whenever the posting statistics say that is cheaper.  The index kernel's time follows the posting entries its
  // probes visit (measured ~2e-9 ms per entry on one MI355X, lists L2-resident, plus ~0.01 ms); the signature kernel's
  // follows N x M at a rate that falls with the threshold (C2-shaped, 2.5e9 pairs: 0.43 / 0.25 / 0.15 / 0.10 ms at 0.5 /
  // 0.6 / 0.8 / 0.9), and where the signature bound is weak (thresholds below ~0.45 at W = 16) every pair pays the
  // position matrix, ~3e-9 ms per pair.  tools/sweep_global.py, profiles/r04_global_index_sweep.txt: vocabularies of 2^17,
  // 4096, 500 and 60 ids x thresholds 0.1 .. 0.9; this rule picks the faster kernel in 26 of the 28 cases (the two misses
  // within 25 %).
  if (r->post && r->post_start && r->vocab > 0 && l->sig && r->sig && threshold > 0.0 && !(flags & NSM_FLAG_NO_INDEX) &&
      !(flags & NSM_FLAG_TILE_INDEX) && (flags & (NSM_FLAG_PRUNE | NSM_FLAG_INDEX))) {
    double visited = 0.0;
    if (int rc = launch_raw_global<W>(l, r, threshold, hits, capacity, hit_count, stream, true, &visited)) return rc;
    const double pairs = static_cast<double>(l->n) * static_cast<double>(r->n);
    constexpr int kMid = 3 * W / 4;
    const bool weak = W >= 64 || ((p.weak[kMid] >> kMid) & 1ull);
    const double ms_global = 0.01 + visited * 2.0e-9;
    const double t = threshold < 1.0 ? threshold : 1.0;
    const double ms_pairs = pairs * (weak ? 3.0e-9 : 3.4e-10 * (1.0 - t) + 1.0e-11);
    if ((flags & NSM_FLAG_INDEX) || ms_global < ms_pairs)
      return launch_raw_global<W>(l, r, threshold, hits, capacity, hit_count, stream, false, nullptr);
  }
  if constexpr (W <= 32) {
    // Low thresholds: the signature bound passes too often for typical set sizes (the mid-size class is "weak"),
    // every pair would pay the position matrix.  Candidate generation by inverted index instead.
    // Which class decides (tools/sweep_index.py, 50k x 50k Poisson(8) sets, ids of 2^17 / 4096 / 500 / 60 values, kernel
    // + ordering ms): for W = 16 the class of two 12-id sets -- it turns weak below ~0.45, where the matrix kernel
    // takes 3.5 ms at 0.4 and 5.9 ms at 0.35 against the index kernel's flat 0.7, and small vocabularies lose at most
    // 7 % (500 ids at 0.4: 1.04 vs 0.97 ms); from 0.45 up the signature kernel wins on small vocabularies (0.50 vs
    // 0.98 ms) and from 0.55 up everywhere (C4 at 0.8: 38 vs 127 ms).  W = 32 (Poisson(16) ids): the class of two
    // 24-id sets is weak up to ~0.65; there the index wins 6 - 30x on 2^17 ids (0.5: 1.1 vs 15.2 ms, 0.6: 1.1 vs 6.5)
    // and loses at most 39 % on 500 ids (0.6: 1.75 vs 1.26 ms).
    constexpr int kDecider = 3 * W / 4;
    const bool weak_mid = (p.weak[kDecider] >> kDecider) & 1ull;
    // (the index kernel addresses the left ids with 32-bit byte offsets)
    const bool use_index = threshold > 0.0 && !(flags & NSM_FLAG_NO_INDEX) && static_cast<long long>(l->n) * W * 4 < (1ll << 32) &&
                           ((flags & NSM_FLAG_INDEX) || ((flags & NSM_FLAG_PRUNE) && weak_mid));
    if (use_index) {
      return launch_raw_index<W>(l, r, threshold, hits, capacity, hit_count, stream);
    }
  }
  if (prune)
    hipLaunchKernelGGL((jaccard_raw_kernel<W, true>), grid, dim3(kBlock), 0, stream, l->ids, l->cnt,
                       l->size_start, l->sig, l->sig2, l->orig, r->ids, r->cnt, r->sig, r->sig2, r->orig, hits,
                       hit_count, p);
  else
    hipLaunchKernelGGL((jaccard_raw_kernel<W, false>), grid, dim3(kBlock), 0, stream, l->ids, l->cnt,
                       l->size_start, l->sig, l->sig2, l->orig, r->ids, r->cnt, r->sig, r->sig2, r->orig, hits,
                       hit_count, p);
  return hip_status(hipGetLastError(), "jaccard_raw_kernel launch");
}

}  // namespace nsm
