// RAW Jaccard grid at LOW thresholds: candidate generation by a per-tile inverted index
// (reference: napkon_string_matching/compare/score_functions.py:6-13; the API default score_threshold is 0.1,
// types/comparable_data.py:74).
//
// The signature prune of jaccard_raw_kernel bounds |A n B| from above; at low thresholds one or two common ids
// already reach kmin and the bound passes for most wavefronts, so the kernel falls back to the full W x W
// position matrix for every pair (C2 at 0.1: 17x slower than at 0.5).  But a pair can only score above a
// positive threshold if it SHARES an id, and that is rare when the vocabulary is large.  Here every wavefront
// builds, once per chunk of left rows, an open-addressing hash table in LDS over the ids of its 64 right rows:
//   id -> 64-bit mask of the lanes (right rows) that hold it.
// Left rows are then probed 64 ids at a time -- lane = (row of the group, id slot): one coalesced load of
// 64 / W left rows, one hash, one or two LDS reads per lane.  A row none of whose ids is in the table (the common
// case: C2 draws 8 of 2^17 ids per row, a tile holds ~500) is finished with that; for the others the masks of
// the found ids are added up per lane (|A n B| exactly: ids are unique per row on both sides) and compared with
// kmin[|A| + |B|].  No signature words, no position matrix.  Data dependent by nature: with a small vocabulary
// every id is found and the cost is one mask accumulation per left id.
#include "jaccard_raw_impl.hpp"

namespace nsm {

template <int W>
struct JacIndexScalars {
  int32_t n_left;
  int32_t n_right;
  int32_t rows_per_chunk;
  unsigned long long cap;
  uint8_t kmin[2 * W + 4];  // indexed by |A|+|B|
  int32_t kfloor[2 * W + 4];  // min of kmin[s'] over s' >= s (kmin itself is not monotone: odd sums cannot reach 1.0); dwords: scalar loads
  int32_t one_id_from;        // smallest s with kfloor[s] > 1
};

#ifndef NSM_IDX_BLOOM_LOG
#define NSM_IDX_BLOOM_LOG 15
#endif
#ifndef NSM_IDX_DEPTH
#define NSM_IDX_DEPTH 4
#endif
#ifndef NSM_IDX_WAVES
#define NSM_IDX_WAVES 4
#endif
// A/B on C2 at threshold 0.1 (kernel ms).  First version (one bit per id, probe on every bitmap pass, one group of ids in
// flight): bitmap 2^13 / 14 / 15 / 16 / 17 bits at 4 (2) waves per block -> 4.47 / 4.11 / 2.96 / 3.48 (2.38) / (3.06); waves per
// block 4 / 2 / 1 at 2^16 -> 3.48 / 2.38 / 2.05.  Then, at 1 wave per block: unconditional 4-deep id queue 1.94; two bits per
// id + scalar reject of rows that found one id 1.48; reject from the bitmap's pass count before probing 1.19; bitmap
// 2^13 / 14 / 15 / 16 / 17 -> 1.22 / 1.10 / 1.08 / 1.19 / 1.43 (LDS per wave decides the occupancy); both bits from one
// product + 32-bit load offsets 1.03; queue depth 2 / 4 / 8 -> 1.21 / 1.19 / 1.18; waves per block 1 / 2 / 4 -> 1.19 / 1.33 / 1.81.
// One index per BLOCK (4 waves, 4 chunks; 16 KB per block instead of per wave): 0.83; bitmap 2^15 / 16 / 17 then 0.83 / 0.83 /
// 0.85, 2 waves per block 0.89.
// Second bit of the presence bitmap: another field of the same product (a 32-bit multiply is a quarter-rate op).  Two
// bits per id: 512 ids set 3 % of 2^15 bits, 0.1 % of absent ids pass.
constexpr int kBloom2Shift = 1;
constexpr int32_t kEmptyKey = -3;  // ids are >= 0, padding is -1 (left) / -2 (right)

template <int W>
__global__ __launch_bounds__(kBlock) void jaccard_raw_index_kernel(
    const int32_t* __restrict__ lids, const int32_t* __restrict__ lcnt, const int32_t* __restrict__ lorig,
    const int32_t* __restrict__ rids, const int32_t* __restrict__ rcnt, const int32_t* __restrict__ rorig,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const JacIndexScalars<W> p) {
  constexpr int T = index_slots<W>();
  constexpr int kLog = W == 16 ? 10 : 11;
  constexpr int kRowsPerGroup = kWave / W;  // left rows probed at once (W = 16: 4, W = 32: 2)
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_idx[];
  // per BLOCK (its waves share one right tile and take different chunks of left rows, so the index is built once
  // for all of them and a CU holds 8 waves per SIMD instead of 2.5): [T] u64 masks | [T] i32 keys | [kBloomWords] u32
  // presence bitmap; then kmin
  constexpr int kBloomLog = NSM_IDX_BLOOM_LOG;  // presence bitmap: 2^15 bits = 4 KB
  constexpr int kBloomWords = (1 << kBloomLog) / 32;
  constexpr int kTableWords = T + T / 2 + kBloomWords / 2;  // u64 units
  const int waves = blockDim.x >> 6;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  unsigned long long* tmask = s_idx;
  int32_t* tkey = reinterpret_cast<int32_t*>(tmask + T);
  uint32_t* bloom = reinterpret_cast<uint32_t*>(tkey + T);
  uint8_t* s_kmin = reinterpret_cast<uint8_t*>(s_idx + kTableWords);
  for (int t = threadIdx.x; t < 2 * W + 4; t += blockDim.x) s_kmin[t] = p.kmin[t];

  const int tile = blockIdx.x;  // (block-uniform: every wave reaches every barrier below)
  const int j = tile * kWave + lane;
  const bool valid = j < p.n_right;
  const int jc = valid ? j : p.n_right - 1;
  const int nrj = valid ? rcnt[jc] : 0;
  // a tile with more ids than 3/4 of the table is indexed in two halves (lanes 0..31, then 32..63): the masks of a
  // pass only carry its lanes, so the other lanes count 0 common ids and cannot hit
  const int n_pass = tile_is_dense<W>(nrj) ? 2 : 1;
  const int jorig = rorig[jc];
  const int nr_min = W - wave_max_i32(valid ? W - nrj : 0);  // smallest |B| of the tile (wave-uniform)
  for (int pass = 0; pass < n_pass; ++pass) {
  const bool in_pass = n_pass == 1 || (lane >> 5) == pass;

  // ---- build: every lane inserts its row's ids (hash table + presence bitmap), the block's waves take turns on the slots
  for (int c = threadIdx.x; c < T; c += blockDim.x) {
    tmask[c] = 0ull;
    tkey[c] = kEmptyKey;
  }
  for (int c = threadIdx.x; c < kBloomWords; c += blockDim.x) bloom[c] = 0u;
  __syncthreads();
  {
    const int32_t* rrow = rids + static_cast<size_t>(jc) * W;
    for (int q = wave; q < (in_pass ? nrj : 0); q += waves) {
      const int32_t id = rrow[q];
      const uint32_t hh = static_cast<uint32_t>(id) * 0x9E3779B1u;
      uint32_t h = hh >> (32 - kLog);
      for (int tries = 0; tries < T; ++tries) {
        const int32_t prev = atomicCAS(&tkey[h], kEmptyKey, id);
        if (prev == kEmptyKey || prev == id) break;
        h = (h + 1) & (T - 1);
      }
      atomicOr(&tmask[h], 1ull << lane);
      const uint32_t bit = hh >> (32 - kBloomLog);
      atomicOr(&bloom[bit >> 5], 1u << (bit & 31u));
      const uint32_t bit2 = (hh >> kBloom2Shift) & ((1u << kBloomLog) - 1u);
      atomicOr(&bloom[bit2 >> 5], 1u << (bit2 & 31u));
    }
  }
  __syncthreads();

  const int i0 = (blockIdx.y * waves + wave) * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);
  if (i0 < i1) {  // (the last block's spare waves have no chunk)
  const int sub = lane / W;  // which row of the group this lane probes for
  // Software pipeline, kDepth groups deep: a group's ids are requested kDepth iterations before they are probed
  // (one iteration is ~300 cycles of work, a load ~1000).  The loads are UNCONDITIONAL -- rows past the chunk's end
  // are clamped to its last row and masked when consumed: with a predicated load the paths through an iteration
  // hold different numbers of outstanding loads and the compiler can only wait for all of them (s_waitcnt vmcnt(0)
  // at every group, which is why deeper queues used to be slower).  One load per group: |A| of a row is the number
  // of its slots that hold an id (padding is -1), counted from a ballot where a row is scored.
  const uint32_t slot_off = static_cast<uint32_t>(lane & (W - 1)) * 4u;
  auto load_group = [&](int ig) -> int32_t {
    // 32-bit byte offset from the uniform base: one address op per load (the launcher checks n_left * W * 4 < 2^32)
    const uint32_t off = static_cast<uint32_t>(min(ig + sub, i1 - 1)) * (W * 4u) + slot_off;
    return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(lids) + off);
  };
  constexpr int kDepth = NSM_IDX_DEPTH;
  int32_t id_q[kDepth];
#pragma unroll
  for (int d = 0; d < kDepth; ++d) id_q[d] = load_group(i0 + d * kRowsPerGroup);
  for (int ig0 = i0; ig0 < i1; ig0 += kDepth * kRowsPerGroup) {
#pragma unroll
   for (int d = 0; d < kDepth; ++d) {
    const int ig = ig0 + d * kRowsPerGroup;
    const int32_t id = ig + sub < i1 ? id_q[d] : -1;  // (groups past the end find nothing: no early exit, see above)
    id_q[d] = load_group(ig + kDepth * kRowsPerGroup);
    // ---- presence test: lane = (row ig + sub, id slot lane % W); the bitmap answers "not in this tile" for ~99.6 %
    const uint32_t hh = static_cast<uint32_t>(id) * 0x9E3779B1u;
    const uint32_t bit = hh >> (32 - kBloomLog);
    const uint32_t bit2 = (hh >> kBloom2Shift) & ((1u << kBloomLog) - 1u);
    const uint32_t seen = (bloom[bit >> 5] >> (bit & 31u)) & (bloom[bit2 >> 5] >> (bit2 & 31u)) & 1u;  // both reads in flight together
    const unsigned long long maybe = __ballot(id >= 0 && seen);
    if (maybe == 0ull) continue;  // none of the group's ids occurs in the tile: no pair shares an id
    // ---- |A n B| <= the number of the row's ids that pass the bitmap.  When that is below the smallest kmin any lane can
    // have (kfloor[s] = min of kmin over sums >= s; nr_min = the tile's smallest row) no lane hits: at low thresholds the
    // usual fate of a row that shares ONE id with the tile, decided on the scalar unit -- no hash probe, no mask
    // accumulation, no memory access (p.one_id_from = the smallest |A| + |B| from which a single common id is too few)
    const unsigned long long has_id = __ballot(id >= 0);
    uint32_t rows_go = 0;
    for (int rr = 0; rr < kRowsPerGroup; ++rr) {
      const unsigned long long seg = (maybe >> (rr * W)) & ((1ull << (W % 64)) - 1ull);
      if (seg == 0ull) continue;
      const int nl = __popcll((has_id >> (rr * W)) & ((1ull << (W % 64)) - 1ull));
      const int upto = __popcll(seg);
      const bool too_few = upto == 1 ? nl + nr_min >= p.one_id_from : upto < p.kfloor[nl + nr_min];
      if (!too_few) rows_go |= 1u << rr;
    }
    if (rows_go == 0u) continue;
    // ---- probe the hash table: the lane masks of the ids that are really there
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
        if (k == kEmptyKey) break;
        h = (h + 1) & (T - 1);
      }
    }
    const unsigned long long who = __ballot(found);
    const uint32_t m_lo = static_cast<uint32_t>(m), m_hi = static_cast<uint32_t>(m >> 32);
    // ---- the rows that may hit: |A n B| per lane = sum of the found ids' masks
    for (int rr = 0; rr < kRowsPerGroup; ++rr) {
      unsigned long long mine = (who >> (rr * W)) & ((1ull << (W % 64)) - 1ull);
      if (mine == 0ull || !((rows_go >> rr) & 1u)) continue;
      const int nl = __popcll((has_id >> (rr * W)) & ((1ull << (W % 64)) - 1ull));
      const int i = ig + rr;
      int k = 0;
      while (mine) {
        const int q = __builtin_ctzll(mine) + rr * W;
        mine &= mine - 1;
        // (readlane returns a signed int: without the casts the low half sign-extends into the high one)
        const unsigned long long mq =
            (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m_hi), q))) << 32) |
            static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m_lo), q)));
        k += static_cast<int>((mq >> lane) & 1ull);
      }
      const int need = valid ? s_kmin[nl + nrj] : kNever;
      const bool hit = k >= need;
      if (__any(hit))
        emit_hits_wave(hits, p.cap, count, hit, static_cast<double>(k) / static_cast<double>(nl + nrj - k), lorig[i], jorig);
    }
   }
  }
  }
  __syncthreads();  // the table is rebuilt by the next pass
  }
}

// kmin as in jaccard_raw_impl.hpp (fill_kmin): least k with double(k)/double(s-k) >= threshold
template <int W>
static void index_fill_kmin(uint8_t* kmin, double threshold) {
  for (int s = 0; s < 2 * W + 4; ++s) {
    kmin[s] = kNever;
    if (s == 0 || s > 2 * W) continue;
    for (int k = 0; 2 * k <= s; ++k) {
      const volatile double q = static_cast<double>(k) / static_cast<double>(s - k);
      if (q >= threshold) {
        kmin[s] = static_cast<uint8_t>(k);
        break;
      }
    }
  }
}

template <int W>
int launch_raw_index(const nsm_set_table* l, const nsm_set_table* r, double threshold, nsm_hit* hits, uint64_t capacity,
                     unsigned long long* hit_count, hipStream_t stream) {
  JacIndexScalars<W> p;
  p.n_left = l->n; p.n_right = r->n; p.cap = capacity;
  index_fill_kmin<W>(p.kmin, threshold);
  for (int s = 2 * W + 3, lo = kNever; s >= 0; --s) {
    lo = p.kmin[s] < lo ? p.kmin[s] : lo;
    p.kfloor[s] = lo;
  }
  p.one_id_from = 2 * W + 4;
  for (int s = 2 * W + 3; s >= 0 && p.kfloor[s] > 1; --s) p.one_id_from = s;
  constexpr int T = index_slots<W>();
  const int waves = NSM_IDX_WAVES;
  const int n_tiles = (r->n + kWave - 1) / kWave;
  // the table is rebuilt per (tile, waves chunks): long chunks, but enough blocks to fill the chip
#ifndef NSM_IDX_ROWS
#define NSM_IDX_ROWS 1536  // 1024 ... 3072: 0.81 ms, 2048 and 4096: 0.83 (how the blocks fill the last round)
#endif
  long long rows = NSM_IDX_ROWS;
  auto blocks = [&](long long rows_per_chunk) {
    const long long chunks = (l->n + rows_per_chunk - 1) / rows_per_chunk;
    return static_cast<long long>(n_tiles) * ((chunks + waves - 1) / waves);
  };
  while (rows > 256 && blocks(rows) < 4096) rows /= 2;
  while ((l->n + rows - 1) / rows > 65535ll * waves) rows *= 2;
  p.rows_per_chunk = static_cast<int>(rows);
  const long long chunks = (l->n + rows - 1) / rows;
  dim3 grid(n_tiles, static_cast<unsigned>((chunks + waves - 1) / waves));
  const size_t lds = static_cast<size_t>(T + T / 2 + (1 << NSM_IDX_BLOOM_LOG) / 64) * 8 + 2 * W + 4;
  hipLaunchKernelGGL((jaccard_raw_index_kernel<W>), grid, dim3(waves * kWave), lds, stream, l->ids, l->cnt, l->orig, r->ids,
                     r->cnt, r->orig, hits, hit_count, p);
  return hip_status(hipGetLastError(), "jaccard_raw_index_kernel launch");
}

template int launch_raw_index<16>(const nsm_set_table*, const nsm_set_table*, double, nsm_hit*, uint64_t, unsigned long long*,
                                  hipStream_t);
template int launch_raw_index<32>(const nsm_set_table*, const nsm_set_table*, double, nsm_hit*, uint64_t, unsigned long long*,
                                  hipStream_t);

}  // namespace nsm
