// Levels-mode Jaccard grid at LOW thresholds: candidates from a per-tile inverted index, then the exact per-level
// scores of jaccard_levels_impl.hpp (reference: types/comparable_data.py:223-232 -> compare_terms :248-265 x
// intersection_vs_union, compare/score_functions.py:6-13; the API's default score_threshold is 0.1, comparable_data.py:75,
// and BASELINE configs[0] runs this mode at 0.1).
//
// jaccard_levels_kernel filters EVERY pair with a signature bound (~19 VALU ops per pair); at low thresholds one common
// id is enough to pass it and 58 hash bits say "maybe" for most pairs.  But a pair can only score above a positive
// threshold if it SHARES an id -- every step's quotient is |A_s n B_s| / |A_s u B_s| -- and that is rare when the
// vocabulary is large.  So, as in jaccard_raw_index.hip: a block = ONE right tile whose ids go into an open-addressing
// hash table in LDS (id -> mask of the lanes that hold it) behind a two-bit presence bitmap; the block's four waves walk
// disjoint quarters of the left rows, 64 ids per probe (lane = (row of the group, id slot)).  A row that shares an id
// with the tile is appended to the private queue of exactly the lanes that hold that id (and pass the category
// predicate); queues are verified as in the matrix kernel: position matrix, per-level intersections by byte-parallel
// compare, double quotients in the reference's order.  Hits are identical to jaccard_levels_kernel's and the oracle's.
#include "jaccard_levels_impl.hpp"
#include "jaccard_raw_impl.hpp"

namespace nsm {

constexpr int kLevIdxBloomLog = 15;
constexpr int32_t kLevIdxEmpty = -3;

template <int W>
__global__ __launch_bounds__(kBlock) void jaccard_levels_index_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const int32_t* __restrict__ lorig,
    const int32_t* __restrict__ lnlev, const uint8_t* __restrict__ lplen, const uint64_t* __restrict__ lcat,
    const int32_t* __restrict__ rids, const int32_t* __restrict__ rcnt, const int32_t* __restrict__ rorig,
    const int32_t* __restrict__ rnlev, const uint8_t* __restrict__ rplen, const uint64_t* __restrict__ rcat,
    const int32_t* __restrict__ lsegstart, const int32_t* __restrict__ rseg, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const JacLevScalars<W> p, int y_slices) {
  constexpr int T = index_slots<W>();
  constexpr int kLog = W == 16 ? 10 : 11;
  constexpr int kRowsPerGroup = kWave / W;
  constexpr int kBloomWords = (1 << kLevIdxBloomLog) / 32;
  // LDS: [T] u64 masks | [T] i32 keys | bitmap | per wave [kQueueSlots][64] u16 queues | quotient table
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_lix[];
  unsigned long long* tmask = s_lix;
  int32_t* tkey = reinterpret_cast<int32_t*>(tmask + T);
  uint32_t* bloom = reinterpret_cast<uint32_t*>(tkey + T);
  uint16_t* queues = reinterpret_cast<uint16_t*>(bloom + kBloomWords);
  double* quot = reinterpret_cast<double*>(queues + kWavesPerBlock * kQueueSlots * kWave);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  uint16_t* queue = queues + wave * kQueueSlots * kWave;
  for (int t = threadIdx.x; t < (W + 1) * (2 * W + 1); t += kBlock) {
    const int k = t / (2 * W + 1), u = t % (2 * W + 1);
    quot[t] = u ? static_cast<double>(k) / static_cast<double>(u) : 0.0;  // real IEEE divisions, as the reference's `/`
  }

  const int tile = blockIdx.x;
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;
  const bool partitioned = rseg != nullptr;
  const int myseg = partitioned ? rseg[jc] : 0;
  const int nrj = valid ? rcnt[jc] : 0;
  const uint64_t catr = (p.cat_mode != NSM_CAT_NONE) ? rcat[jc] : 0ull;
  const int lr = rnlev[jc];
  const int jorig = rorig[jc];
  const uint8_t* rplen_row = rplen + static_cast<size_t>(jc) * p.lev_stride_r;
  uint32_t r[W];
  {
    const uint4* rp = reinterpret_cast<const uint4*>(rids + static_cast<size_t>(jc) * W);
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const uint4 v = rp[q];
      r[4 * q + 0] = (v.x << 6) | (4 * q + 0);
      r[4 * q + 1] = (v.y << 6) | (4 * q + 1);
      r[4 * q + 2] = (v.z << 6) | (4 * q + 2);
      r[4 * q + 3] = (v.w << 6) | (4 * q + 3);
    }
  }
  const int n_pass = tile_is_dense<W>(nrj) ? 2 : 1;  // (a tile with more ids than 3/4 of the table: two halves of 32 lanes)
  const unsigned long long cats_tile = partitioned ? wave_or_u64(valid ? (1ull << myseg) : 0ull) : 1ull;

  int qn = 0;
  int qbase = 0;  // left row the queue's offsets are relative to

  // ---- exact score of (left row idx, this lane's right item); idx < 0: the lane idles (jaccard_levels_impl.hpp's verify
  // with the full W x W position matrix)
  auto verify = [&](int idx) {
    const bool active = idx >= 0;
    const int ii = active ? idx : 0;
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
    const int nl_max = wave_max_i32(active ? lcnt[ii] : 0);
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
          for (int b = 2; b < W; b += 2) m = lev_min3u(m, la ^ r[b], la ^ r[b + 1]);
          word |= m << (8 * e);
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
        const int pl = lpl[min(s, p.lev_stride_l - 1)];
        const int pr = rplen_row[min(s, p.lev_stride_r - 1)];
        const uint32_t prrep = static_cast<uint32_t>(pr) * 0x01010101u;
        int inter = 0;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
          if (4 * q < pl) {
            uint32_t x = posw[q];
            const int keep = pl - 4 * q;
            if (keep < 4) x |= 0xffffffffu << (8 * keep);
            const uint32_t y = (x | 0x80808080u) - prrep;
            inter += __popc(~(y | x) & 0x80808080u);
          }
        }
        const int uni = pl + pr - inter;
        factor *= 0.5;
        score += quot[inter * (2 * W + 1) + uni] * factor;
      }
    }
    const bool hit = active && score >= p.threshold;
    if (__any(hit)) emit_hits_wave(hits, p.cap, count, hit, score, lorig[ii], jorig);
  };
  auto flush = [&]() {
    const int deepest = wave_max_i32(qn);
    for (int k = 0; k < deepest; ++k) verify(k < qn ? qbase + static_cast<int>(queue[k * kWave + lane]) : -1);
    qn = 0;
  };

  for (int pass = 0; pass < n_pass; ++pass) {
    const bool in_pass = n_pass == 1 || (lane >> 5) == pass;
    // ---- build the index of the tile
    for (int c = threadIdx.x; c < T; c += blockDim.x) {
      tmask[c] = 0ull;
      tkey[c] = kLevIdxEmpty;
    }
    for (int c = threadIdx.x; c < kBloomWords; c += blockDim.x) bloom[c] = 0u;
    __syncthreads();
    {
      const int32_t* rrow = rids + static_cast<size_t>(jc) * W;
      for (int q = wave; q < (in_pass ? nrj : 0); q += kWavesPerBlock) {
        const int32_t id = rrow[q];
        const uint32_t hh = static_cast<uint32_t>(id) * 0x9E3779B1u;
        uint32_t h = hh >> (32 - kLog);
        for (int tries = 0; tries < T; ++tries) {
          const int32_t prev = atomicCAS(&tkey[h], kLevIdxEmpty, id);
          if (prev == kLevIdxEmpty || prev == id) break;
          h = (h + 1) & (T - 1);
        }
        atomicOr(&tmask[h], 1ull << lane);
        const uint32_t bit = hh >> (32 - kLevIdxBloomLog);
        atomicOr(&bloom[bit >> 5], 1u << (bit & 31u));
        const uint32_t bit2 = (hh >> 1) & ((1u << kLevIdxBloomLog) - 1u);
        atomicOr(&bloom[bit2 >> 5], 1u << (bit2 & 31u));
      }
    }
    __syncthreads();

    // ---- the left rows: per category of the tile (one range without a partition) slice blockIdx.y, a quarter per wave
    for (unsigned long long cats = cats_tile; cats;) {
      const int c = __builtin_ctzll(cats);
      cats &= cats - 1;
      int a, b;
      if (partitioned) {
        const int lo = lsegstart[c], len = lsegstart[c + 1] - lo;
        const int per = (len + y_slices - 1) / y_slices;
        a = lo + min(len, static_cast<int>(blockIdx.y) * per);
        b = lo + min(len, (static_cast<int>(blockIdx.y) + 1) * per);
      } else {
        const int per = (p.n_left + y_slices - 1) / y_slices;
        a = min(p.n_left, static_cast<int>(blockIdx.y) * per);
        b = min(p.n_left, a + per);
      }
      const int quarter = (((b - a) + kWavesPerBlock - 1) / kWavesPerBlock + kRowsPerGroup - 1) / kRowsPerGroup * kRowsPerGroup;
      const int i0 = min(b, a + wave * quarter), i1 = min(b, i0 + quarter);
      if (i0 >= i1) continue;
      const unsigned long long lower = partitioned ? (catr & ((1ull << c) - 1ull)) : 0ull;
      const bool lane_in = valid && in_pass && (!partitioned || myseg == c);
      qbase = i0;
      const int sub = lane / W;
      const uint32_t slot_off = static_cast<uint32_t>(lane & (W - 1)) * 4u;
      auto load_group = [&](int ig) -> int32_t {
        const uint32_t off = static_cast<uint32_t>(min(ig + sub, i1 - 1)) * (W * 4u) + slot_off;
        return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(lids) + off);
      };
      constexpr int kDepth = 4;
      int32_t id_q[kDepth];
#pragma unroll
      for (int d = 0; d < kDepth; ++d) id_q[d] = load_group(i0 + d * kRowsPerGroup);
      for (int ig0 = i0; ig0 < i1; ig0 += kDepth * kRowsPerGroup) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
          const int ig = ig0 + d * kRowsPerGroup;
          const int32_t id = ig + sub < i1 ? id_q[d] : -1;
          id_q[d] = load_group(ig + kDepth * kRowsPerGroup);
          const uint32_t hh = static_cast<uint32_t>(id) * 0x9E3779B1u;
          const uint32_t bit = hh >> (32 - kLevIdxBloomLog);
          const uint32_t bit2 = (hh >> 1) & ((1u << kLevIdxBloomLog) - 1u);
          const uint32_t seen = (bloom[bit >> 5] >> (bit & 31u)) & (bloom[bit2 >> 5] >> (bit2 & 31u)) & 1u;
          if (__ballot(id >= 0 && seen) == 0ull) continue;  // no id of the group occurs in the tile
          bool found = false;
          unsigned long long m = 0ull;
          if (id >= 0 && seen) {
            uint32_t h = hh >> (32 - kLog);
            for (int tries = 0; tries < T; ++tries) {
              const int32_t k = tkey[h];
              if (k == id) {
                found = true;
                m = tmask[h];
                break;
              }
              if (k == kLevIdxEmpty) break;
              h = (h + 1) & (T - 1);
            }
          }
          const unsigned long long who = __ballot(found);
          if (who == 0ull) continue;
          const uint32_t m_lo = static_cast<uint32_t>(m), m_hi = static_cast<uint32_t>(m >> 32);
          for (int rr = 0; rr < kRowsPerGroup; ++rr) {
            unsigned long long mine = (who >> (rr * W)) & ((1ull << (W % 64)) - 1ull);
            if (mine == 0ull) continue;
            // the lanes that hold one of the row's ids
            unsigned long long cand = 0ull;
            while (mine) {
              const int q = __builtin_ctzll(mine) + rr * W;
              mine &= mine - 1;
              cand |= (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m_hi), q))) << 32) |
                      static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m_lo), q)));
            }
            const int i = ig + rr;
            bool pass_lane = lane_in && ((cand >> lane) & 1ull);
            if (p.cat_mode != NSM_CAT_NONE) {
              const uint64_t cl = lcat[i];
              if (partitioned) pass_lane = pass_lane && ((cl & lower) == 0ull);
              else pass_lane = pass_lane && category_match(cl, catr, p.cat_mode);
            }
            if (!__any(pass_lane)) continue;
            queue[qn * kWave + lane] = static_cast<uint16_t>(i - qbase);  // (every lane stores, only passing lanes advance)
            qn += pass_lane ? 1 : 0;
            if (__any(qn > kQueueSlots - 2)) flush();
          }
        }
      }
      flush();
    }
    __syncthreads();  // the table is rebuilt by the next pass
  }
}

template <int W>
int launch_levels_index(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode, nsm_hit* hits,
                        uint64_t capacity, unsigned long long* hit_count, hipStream_t stream) {
  JacLevScalars<W> p;
  p.n_left = l->n; p.n_right = r->n; p.cap = capacity;
  p.lev_stride_l = l->max_levels; p.lev_stride_r = r->max_levels;
  p.cat_mode = category_mode;
  p.threshold = threshold;
  p.emit_all = 0;
  p.rows_per_chunk = 0;
  constexpr int T = index_slots<W>();
  const int n_tiles = (r->n + kWave - 1) / kWave;
  // slices of <= 16 K rows per block (queue offsets are 16 bits, a wave walks a quarter), enough blocks to fill the chip
  const long long rows_cat = l->seg ? (l->n + 31) / 32 : l->n;  // rows a tile visits, roughly
  long long slices = (rows_cat + 16383) / 16384;
  while (slices < 64 && static_cast<long long>(n_tiles) * slices < 4096 && rows_cat / (slices * 2) >= 256) slices *= 2;
  if (l->seg) {  // a category may hold (nearly) all rows: the 16-bit offsets need quarter <= 65535 rows whatever the split
    while ((static_cast<long long>(l->n) + slices - 1) / slices > 4 * 60000ll) slices *= 2;
  } else {
    while ((static_cast<long long>(l->n) + slices - 1) / slices > 4 * 60000ll) slices *= 2;
  }
  if (slices > 65535) {
    set_error("nsm_jaccard_levels_grid: more than 65535 * 240000 left rows");
    return NSM_E_UNSUPPORTED;
  }
  if (static_cast<unsigned long long>(l->n) * W * 4ull >= (1ull << 32)) {
    set_error("nsm_jaccard_levels_grid (index): left table beyond 4 GB of ids");
    return NSM_E_UNSUPPORTED;
  }
  dim3 grid(n_tiles, static_cast<unsigned>(slices));
  const size_t lds = static_cast<size_t>(T) * 12 + (1u << kLevIdxBloomLog) / 8 + kWavesPerBlock * kQueueSlots * kWave * 2 +
                     static_cast<size_t>(W + 1) * (2 * W + 1) * 8;
  hipLaunchKernelGGL((jaccard_levels_index_kernel<W>), grid, dim3(kBlock), lds, stream, l->ids, l->cnt, l->orig, l->nlev, l->plen,
                     l->cat, r->ids, r->cnt, r->orig, r->nlev, r->plen, r->cat, l->seg_start, r->seg, hits, hit_count, p,
                     static_cast<int>(slices));
  return hip_status(hipGetLastError(), "jaccard_levels_index_kernel launch");
}

template int launch_levels_index<16>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                     unsigned long long*, hipStream_t);
template int launch_levels_index<32>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                     unsigned long long*, hipStream_t);

}  // namespace nsm
